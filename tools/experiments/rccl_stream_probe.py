"""On which hardware queue does a collective issued from a side stream run?  (One GPU, 1-rank RCCL group: an out-of-place all-gather is a device copy.)

The main stream is given ~40 ms of kernels; then a collective is issued from a communication stream that depends on nothing.  If the collective completes
in microseconds it has a hardware queue of its own; if it completes after the main stream's work it was multiplexed onto the main stream's queue (HIP maps
ordinary streams onto a handful of hardware queues) - and a gradient all-reduce issued that way overlaps nothing.
usage: python tools/experiments/rccl_stream_probe.py"""
import os
import time

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="env://", device_id=torch.device("cuda:0"))
opts = dist.ProcessGroupNCCL.Options()
opts.is_high_priority_stream = True
hp_group = dist.new_group(ranks=[0], backend="nccl", pg_options=opts)

a = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
inp = torch.randn(1 << 20, device="cuda")
out = torch.empty(1 << 20, device="cuda")
side_streams = [torch.cuda.Stream() for _ in range(6)]  # occupy a few ordinary streams like a real process does
streams = {"ordinary": torch.cuda.Stream(), "high priority": torch.cuda.Stream(priority=-1)}


def busy():
    for _ in range(40):
        a @ a


def probe(name, stream, group, async_op, op):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    busy()  # main stream: tens of ms, enqueued in well under a ms... (each launch ~10 us)
    t_enq = time.perf_counter() - t0
    done = torch.cuda.Event()
    with torch.cuda.stream(stream):
        if op == "copy":
            out.copy_(inp, non_blocking=True)
        else:
            w = dist.all_gather_into_tensor(out, inp, group=group, async_op=async_op)
            if async_op:
                w.wait()  # (stream-level wait)
        done.record(stream)
    done.synchronize()
    t_coll = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"{name:70s} main enqueued {t_enq * 1e3:6.2f} ms, collective done at {t_coll * 1e3:7.2f} ms, main done at {t_all * 1e3:7.2f} ms  ->  "
          f"{'OWN queue' if t_coll < 0.5 * t_all else 'BEHIND the main stream'}")


busy()
for sname, st in streams.items():
    probe(f"plain device copy on a(n) {sname} stream", st, None, False, "copy")
    for gname, g in (("default group", None), ("group with high-priority streams", hp_group)):
        for async_op in (True, False):
            probe(f"all_gather, {sname} stream, {gname}, async_op={async_op}", st, g, async_op, "ag")
dist.destroy_process_group()
