"""Diagnostic: is the one-time 40-70 ms host stall inside the timed region a CPU-quota throttle of the container (cgroup cpu.stat) ?"""
import os, sys, time, glob
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

def cpu_stat():
    out = {}
    for p in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat", "/sys/fs/cgroup/cpu,cpuacct/cpu.stat"):
        if os.path.exists(p):
            for ln in open(p):
                k, v = ln.split()
                out[k] = int(v)
            out["_path"] = p
            break
    return out

def quota():
    for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        if os.path.exists(p):
            return p, open(p).read().strip()
    return None, None

print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "quota", quota())
import torch
print("torch threads", torch.get_num_threads(), "interop", torch.get_num_interop_threads())
import bench
from vit4hep_amd.trainer import CFMTrainer
w = bench.WORKLOADS["ds2"]
s0 = cpu_stat()
model = bench.build_model(w, "bf16", "cuda:0")
tr = CFMTrainer(model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.1, clip_grad_norm=1000.0, iterations=50000)
x, c = bench.synthetic(w["shape"], w["B"], seed=0, device="cuda:0", cond=w["cond"])
for _ in range(5):
    tr.step(x, c)
torch.cuda.synchronize()
s1 = cpu_stat()
host = []
t0 = time.perf_counter()
for _ in range(20):
    tr.step(x, c)
    host.append(time.perf_counter())
torch.cuda.synchronize()
t1 = time.perf_counter()
s2 = cpu_stat()
d = lambda a, b: {k: b[k] - a[k] for k in a if k != "_path" and b.get(k) != a.get(k)}
print("setup + warmup: cpu.stat delta", d(s0, s1))
print("timed region  : cpu.stat delta", d(s1, s2), "wall ms", round((t1 - t0) * 1e3, 2), "steps/s", round(20 / (t1 - t0), 1))
print("host ms per call", [round((b - a) * 1e3, 2) for a, b in zip([t0] + host[:-1], host)])
