"""Batch-sharded data parallelism for the CFM step: one process per GPU, gradients summed with RCCL over xGMI
(reference: DDP(model.net) + DistributedSampler, experiments/base_experiment.py:161-167, calochallenge/experiment.py:94-112).

The gradient lives in ONE flat f32 buffer laid out in state_dict() order, so each backward stage of the HIP runtime
(final layer, block depth-1 ... block 0, embedders) finishes a CONTIGUOUS slice.  ``BucketReducer`` all-reduces each
slice as soon as its stage's gradients are final, on a communication stream, so the collective of stage s overlaps the kernels of
the later stages.  On the GPU the whole backward is ONE library call that records an event per stage (v4h_vit_backward_events); the
collectives are enqueued behind those events.  (CPU / gloo: one call per stage, reduce after each.)  Buckets are therefore 0.1 / 17.3 x depth / 2.0 MB (f32): few, large messages as xGMI's point-to-point rings like.
The 1/world factor of DDP's gradient averaging is folded into the loss gradient, so the collective is a plain SUM.
Works on CPU tensors with the gloo backend too (tests/test_dp_gloo.py).
"""

from __future__ import annotations

import torch
import torch.distributed as dist


import os


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def collectives_enabled():
    """True when gradient collectives must run: more than one rank, or V4H_FORCE_COLLECTIVES=1 with an initialised
    process group (exercises the RCCL / stream plumbing on a single GPU; a 1-rank all-reduce is the identity)."""
    if world() > 1:
        return True
    return os.environ.get("V4H_FORCE_COLLECTIVES") == "1" and dist.is_available() and dist.is_initialized()


class BucketReducer:
    """``compress="bf16"`` (off by default; SURVEY section 5, condition (iii) for the 8-GPU target): a bucket is cast to bf16 on the communication
    stream, all-reduced as bf16 - 52 MB instead of 104 MB per step over xGMI's per-link-bound rings - and written back into the f32 gradient buffer, which
    stays the master copy the optimizer reads (f32 moments, f32 parameters).  What is lost is the rounding of each rank's contribution and of the
    running sum to 8 significant bits; the 2-rank tests hold the result to bf16 rounding of the f32 path."""

    def __init__(self, flat: torch.Tensor, group=None, compress=None):
        if compress not in (None, "none", "f32", "bf16"):
            raise ValueError(f"BucketReducer: compress must be None or 'bf16', got {compress!r}")
        self.flat = flat
        self.group = group
        self.pending = []
        self.cuda = flat.is_cuda
        self.compress = "bf16" if compress == "bf16" else None
        self.buf16 = torch.empty(flat.numel(), dtype=torch.bfloat16, device=flat.device) if self.compress else None
        self._done = None  # (compressed path, device tensors) event behind the last bucket's write-back
        # A HIGH-PRIORITY stream: HIP multiplexes ordinary streams onto a few hardware queues, and a communication stream that lands on the queue of the
        # compute stream runs its collectives only after everything enqueued there before it - i.e. after the whole backward pass, which is enqueued as one
        # library call (measured: every bucket ran after the last backward kernel, profiles/r03_comm_interference.md).  Priority streams have hardware
        # queues of their own, and a collective's few workgroups should win CUs ahead of the next contraction kernel anyway.
        self.comm_stream = torch.cuda.Stream(device=flat.device, priority=-1) if self.cuda else None

    def _all_reduce(self, lo: int, hi: int):
        """Enqueue the collective of flat[lo:hi] (the caller has set the stream and made it wait for the producer)."""
        view = self.flat[lo:hi]
        if not self.compress:
            self.pending.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            return
        half = self.buf16[lo:hi]
        half.copy_(view)  # f32 -> bf16 (round to nearest even) on the communication stream
        w = dist.all_reduce(half, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        w.wait()          # device tensors: the communication stream waits (no host wait); CPU / gloo: the host does
        view.copy_(half)  # back into the f32 master gradient
        if self.cuda:
            if self._done is None:
                self._done = torch.cuda.Event()
            self._done.record(self.comm_stream)

    def reduce_slice(self, lo: int, hi: int):
        """Call right after the kernels producing flat[lo:hi] were enqueued on the current stream."""
        if not collectives_enabled() or hi <= lo:
            return
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.flat.device))
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                self._all_reduce(lo, hi)
        else:
            self._all_reduce(lo, hi)

    def make_stage_events(self, n: int):
        """Events the HIP runtime records when a backward stage's gradients are final (v4h_vit_backward_events)."""
        evs = [torch.cuda.Event() for _ in range(n)]
        for ev in evs:  # torch creates the underlying hipEvent_t at the first record
            ev.record(torch.cuda.current_stream(self.flat.device))
        return evs

    def reduce_slice_after(self, lo: int, hi: int, event):
        """All-reduce flat[lo:hi] on the communication stream once ``event`` (recorded by the library inside the backward call) is reached."""
        if not collectives_enabled() or hi <= lo:
            return
        with torch.cuda.stream(self.comm_stream):
            self.comm_stream.wait_event(event)
            self._all_reduce(lo, hi)

    def finish(self):
        """Make the current stream (or the host, on CPU) wait for every outstanding bucket.  Over RCCL the buckets of one communicator complete in
        issue order on one stream, so the current stream waits for the LAST one only: each wait is a barrier packet in the compute queue, and eight of them
        in a row at the end of the backward cost 2 % of the step (profiles/r03_comm_interference.md).  Any other backend (gloo with device tensors stages
        every bucket through the host on a stream of its own) is waited for bucket by bucket."""
        if self.compress:  # every bucket's collective and write-back sit in order on the communication stream: wait for the last write-back
            if self.cuda and self._done is not None:
                torch.cuda.current_stream(self.flat.device).wait_event(self._done)
            return
        last_only = self.cuda and self.pending and dist.get_backend(self.group) == "nccl"
        for w in (self.pending[-1:] if last_only else self.pending):
            w.wait()
        self.pending = []


def shard_rows(n_rows: int, rank: int, world_size: int):
    """Contiguous row range of ``rank`` when n_rows independent units (e.g. condition rows to sample) are split."""
    base, rem = divmod(n_rows, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
