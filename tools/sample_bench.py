"""Sampling throughput of a shape model (BASELINE config 5 = ds2): showers/s for RK4 (reference default, 80 NFE) and Heun (40 NFE).
usage (GPU box): python tools/sample_bench.py [bf16|f32] [batches] [workload]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import WORKLOADS, build_model, fwd_flops_per_sample, synthetic, tokens_and_patch_dim

mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = "cuda:0"
w = WORKLOADS[sys.argv[3] if len(sys.argv) > 3 else "ds2"]
model = build_model(w, mode, dev).eval()
_, c = synthetic(w["shape"], 256, 0, dev, cond=w["cond"])
T, P = tokens_and_patch_dim(w)
flops = fwd_flops_per_sample(T, P, w["depth"], K=w["cond"])
for method, step, nfe in (("rk4", 0.05, 80), ("heun2", 0.05, 40)):
    model.odeint_kwargs = {"method": method, "options": {"step_size": step}}
    model.sample_batch(c)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(nb):
        s = model.sample_batch(c)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / nb
    assert torch.isfinite(s).all()
    print(f"{mode} {method} step {step}: {dt*1e3:.1f} ms per batch of 256 ({nfe} NFE) = {256/dt:.0f} showers/s; 100k showers in {1e5/256*dt:.1f} s; "
          f"{nfe*256*flops/dt/1e12:.0f} TFLOP/s", flush=True)
