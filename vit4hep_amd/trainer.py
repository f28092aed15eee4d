"""Native CFM update step: the semantics of BaseExperiment._step (reference experiments/base_experiment.py:555-602
with configs/training/default.yaml) on flat HBM buffers, without autograd and without per-step host syncs.

  loss  = CFM._batch_loss                         (models/base_model.py:203-218)   -> HIP forward + fused MSE
  grads = loss.backward()                         (:560)                           -> HIP backward, one call (stage events for the buckets)
  DDP gradient averaging                          (:161-167)                       -> bucketed RCCL all-reduce, overlapped
  clip_grad_norm_(..., clip, error_if_nonfinite)  (:573-585)                       -> one norm kernel; clip folded into AdamW
  AdamW(lr, betas, eps, wd) + CosineAnnealingLR   (:592-597, :329-431)             -> one fused kernel over all parameters
  all_reduce(loss, AVG)                           (:600-601)

Loss and gradient norm come back as 0-dim device tensors; call ``check_finite`` (or .item()) when the host needs them.
"""

from __future__ import annotations

import math
import os

import torch
import torch.distributed as dist

from . import _lib
from .autograd import run_backward, run_backward_events, run_forward
from .parallel import BucketReducer, collectives_enabled, world


def _aligned(n, a=64):
    return (n + a - 1) // a * a


class CFMTrainer:
    NONFINITE_MSG = "The total norm for gradients is non-finite, so it cannot be clipped."  # torch.nn.utils.clip_grad_norm_'s own text

    def __init__(self, model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.1, clip_grad_norm=1000.0, iterations=50000, group=None,
                 nonfinite_check_every=50):
        from .experiments.calochallenge.calochallenge_cfm.model import _unwrap

        self.model = model
        self.net = model._core() if hasattr(model, "_core") else _unwrap(model.net)  # _core() (re)binds the wrapper's geometry to the net
        if self.net.x_embed_in() or self.net.c_embed_in():
            raise NotImplementedError("CFMTrainer: networks with a fine-tuning embedding mapper train through the autograd node and a torch optimizer "
                                      "(per-module learning rates, experiment_finetuning.py:173-205)")
        self.lr, self.betas, self.eps, self.wd = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        self.clip = float(clip_grad_norm) if clip_grad_norm is not None else None
        self.iterations = int(iterations)
        self.step_count = 0
        self.group = group
        # The reference raises on a non-finite gradient norm BEFORE optimizer.step(), also without clipping (max_norm = inf; base_experiment.py:573-585).
        # Here the update kernel skips such a step on the device and bumps a sticky counter - and while that counter is non-zero it skips (and counts)
        # EVERY later update too, so no update is ever applied with a shifted Adam / LR-schedule index.  The host looks at the counter every
        # `nonfinite_check_every` steps (one 4-byte read) and in raise_if_nonfinite(), raises the same error and rewinds step_count by the skipped
        # updates: parameters, moments and step index are then exactly those of the last finite step, as after the reference's raise.  (All ranks see
        # the same all-reduced gradient norm, hence the same counter.)
        self.nonfinite_check_every = int(nonfinite_check_every)
        self._flatten()

    # ---------------------------------------------------------------------------------------------- flat buffers
    def _flatten(self):
        params = self.net.parameter_list()
        dev = params[0].device
        if dev.type != "cuda":
            raise RuntimeError("CFMTrainer: move the model to the MI355X first (model.to('cuda')); there is no CPU path")
        self.offsets, off = [], 0
        for p in params:
            self.offsets.append(off)
            off += _aligned(p.numel())
        self.total = off
        self.flat_p = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(off, dtype=torch.float32, device=dev)
        self.p_views, self.g_views = [], []
        with torch.no_grad():
            for p, o in zip(params, self.offsets):
                view = self.flat_p[o : o + p.numel()].view_as(p)
                view.copy_(p.data)
                p.data = view
                self.p_views.append(view)
                self.g_views.append(self.flat_g[o : o + p.numel()].view_as(p))
        self.params = params
        # contiguous gradient slice of each backward stage (include/vit4hep_hip.h: stage 0 final layer, 1+j block depth-1-j, last embedders)
        depth = int(self.net.depth)
        blk0 = 11
        bounds = lambda i0, i1: (self.offsets[i0], self.offsets[i1] if i1 < len(params) else self.total)
        self.stage_slices = [bounds(blk0 + 10 * depth, len(params))]
        for j in range(depth):
            i = depth - 1 - j
            self.stage_slices.append(bounds(blk0 + 10 * i, blk0 + 10 * (i + 1)))
        self.stage_slices.append(bounds(0, blk0))
        self.reducer = BucketReducer(self.flat_g, self.group)
        self.comm_reserve_cus = int(os.environ.get("V4H_COMM_RESERVE_CUS", "0"))  # multiple of 8 in [0, 64]
        self.stage_events = None
        self.gnorm_sq = torch.zeros((), dtype=torch.float32, device=dev)
        self.loss = torch.zeros((), dtype=torch.float32, device=dev)
        self.nonfinite = torch.zeros((), dtype=torch.int32, device=dev)

    def _check_alias(self):
        if any(p.data_ptr() != v.data_ptr() for p, v in ((self.params[0], self.p_views[0]), (self.params[-1], self.p_views[-1]))):
            m, v, n = self.flat_m, self.flat_v, self.step_count
            self._flatten()  # parameters were re-allocated (e.g. model.to(...)): adopt the new storage, keep the optimizer state
            if m.numel() == self.flat_m.numel() and m.device == self.flat_m.device:
                self.flat_m, self.flat_v, self.step_count = m, v, n

    def lr_at(self, k):
        """CosineAnnealingLR(T_max=iterations, eta_min=0) after k scheduler steps (closed form)."""
        return self.lr * 0.5 * (1.0 + math.cos(math.pi * k / self.iterations))

    # ---------------------------------------------------------------------------------------------- one update
    def loss_and_grads(self, x, c, t=None, x0=None):
        """Forward + backward into the flat gradient buffer (all-reduced over the data-parallel group)."""
        self._check_alias()
        with _lib.on_device(self.flat_p):
            return self._loss_and_grads(x, c, t, x0)

    def _loss_and_grads(self, x, c, t, x0):
        lib = _lib.load()
        dev = self.flat_p.device
        x = _lib.require_cuda(x, "x")
        c = _lib.require_cuda(c, "c")
        B = x.shape[0]
        if t is None:  # reference: CPU generator for t, device generator for x_0 (models/base_model.py:209-212)
            t = self.model.time_distribution.sample([B] + [1] * (x.dim() - 1)).to(dev, torch.float32, non_blocking=True)
        if x0 is None:
            x0 = torch.randn_like(x)
        t = _lib.require_cuda(t, "t").reshape(-1)
        x0 = _lib.require_cuda(x0, "x0")
        s = _lib.stream_ptr(dev)
        xt, target = torch.empty_like(x), torch.empty_like(x)
        _lib.check(lib.v4h_cfm_prepare(_lib.ptr(x), _lib.ptr(x0), _lib.ptr(t), _lib.ptr(xt), _lib.ptr(target), B, x[0].numel(), s), "v4h_cfm_prepare")
        v, ws = run_forward(self.net, self.p_views, xt, t, c, True)
        dv = torch.empty_like(v)
        _lib.check(lib.v4h_mse_loss(_lib.ptr(v), _lib.ptr(target), _lib.ptr(self.loss), _lib.ptr(dv), v.numel(), s), "v4h_mse_loss")
        W = world()
        if W > 1:  # DDP averages gradients: fold 1/world into the seed, then SUM
            _lib.check(lib.v4h_axpby(_lib.ptr(dv), _lib.ptr(dv), _lib.ptr(dv), 1.0 / W, 0.0, dv.numel(), s), "v4h_axpby")
        self.flat_g.zero_()
        if collectives_enabled() and os.environ.get("V4H_STAGED_CALLS") != "1":
            # one call; the library records an event when a stage's gradient slice is final and the bucket is reduced behind it
            if self.stage_events is None:
                self.stage_events = self.reducer.make_stage_events(len(self.stage_slices))
            # While buckets are in flight the persistent kernels can leave CUs to the collective's workgroups (read at enqueue time, so it brackets
            # exactly this pass; results do not depend on it).  Rehearsed on one GPU it is neutral (-0.9 ... +1.0 % over two runs of the table in
            # profiles/r03_comm_interference.md), so the default is 0; the switch is for runs beside the real RCCL kernels.
            _lib.check(lib.v4h_reserve_compute_units(self.comm_reserve_cus), "v4h_reserve_compute_units")
            try:
                run_backward_events(self.net, self.p_views, self.g_views, dv, ws, self.stage_events)
            finally:
                lib.v4h_reserve_compute_units(0)
            for (lo, hi), ev in zip(self.stage_slices, self.stage_events):
                self.reducer.reduce_slice_after(lo, hi, ev)
            self.reducer.finish()
        elif collectives_enabled():  # A/B hook: one call per stage, each ending with a join of the library's two streams
            for st, (lo, hi) in enumerate(self.stage_slices):
                run_backward(self.net, self.p_views, self.g_views, dv, ws, st, st)
                self.reducer.reduce_slice(lo, hi)
            self.reducer.finish()
        else:  # single rank: one call, so the weight-gradient stream is joined only once at the end
            run_backward(self.net, self.p_views, self.g_views, dv, ws, 0, len(self.stage_slices) - 1)
        return self.loss

    def step(self, x, c, t=None, x0=None):
        """One BaseExperiment._step.  Returns (loss, grad_norm) as 0-dim device tensors (pre-clip norm, like clip_grad_norm_)."""
        with _lib.on_device(self.flat_p):
            return self._step(x, c, t, x0)

    def _step(self, x, c, t, x0):
        lib = _lib.load()
        loss = self.loss_and_grads(x, c, t, x0)
        s = _lib.stream_ptr(self.flat_p.device)
        self.gnorm_sq.zero_()
        _lib.check(lib.v4h_sq_norm_accum(_lib.ptr(self.flat_g), self.total, _lib.ptr(self.gnorm_sq), s), "v4h_sq_norm_accum")
        self.step_count += 1
        lr = self.lr_at(self.step_count - 1)
        _lib.check(
            lib.v4h_adamw_step(_lib.ptr(self.flat_p), _lib.ptr(self.flat_g), _lib.ptr(self.flat_m), _lib.ptr(self.flat_v), self.total,
                               _lib.ptr(self.gnorm_sq), self.clip if self.clip is not None else float("inf"), lr, self.betas[0], self.betas[1], self.eps, self.wd,
                               self.step_count, s, _lib.ptr(self.nonfinite)),
            "v4h_adamw_step",
        )
        self.net.weights_epoch += 1  # parameters rewritten through raw pointers: invalidate cached operand copies (ViT.operands_current)
        out_loss = loss.clone()
        if collectives_enabled():
            dist.all_reduce(out_loss, op=dist.ReduceOp.SUM, group=self.group)
            out_loss /= world()
        if self.nonfinite_check_every > 0 and self.step_count % self.nonfinite_check_every == 0:
            self.raise_if_nonfinite()
        return out_loss, self.gnorm_sq.sqrt()

    def raise_if_nonfinite(self):
        """Host look at the sticky device counter of skipped (non-finite) updates: raise like the reference, with step_count and the LR schedule
        rewound to the last update that was applied."""
        skipped = int(self.nonfinite.item())
        if skipped:
            self.nonfinite.zero_()
            self.step_count -= skipped
            raise RuntimeError(f"{self.NONFINITE_MSG} ({skipped} update(s) skipped on the device; step_count rewound to {self.step_count})")

    @staticmethod
    def check_finite(grad_norm):
        """error_if_nonfinite=True of the reference's clip_grad_norm_ call (base_experiment.py:581); needs a host sync."""
        g = float(grad_norm)
        if not math.isfinite(g):
            raise RuntimeError(CFMTrainer.NONFINITE_MSG)
        return g
