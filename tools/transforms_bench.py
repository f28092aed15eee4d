"""Pre-/post-processing chain throughput: the fused kernels vs the same chain as PyTorch-ROCm eager ops on the same
GPU (the oracle's restatement of the reference's nine transforms, run on device tensors) and vs the CPU oracle.
usage (GPU box): python tools/transforms_bench.py [n_showers]      prints one JSON line per direction"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import transforms_oracle as TO
from vit4hep_amd.transforms import ShapeChain

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
dev = "cuda:0"
s = TO.ds2_spec(mean=-1.7, std=2.9)
ch = ShapeChain(s.layer_boundaries, s.shape, s.eps, s.norm_cut, s.factor, s.cut, s.delta, s.mean, s.std, s.alpha, s.e_min, s.e_max)
g = torch.Generator().manual_seed(0)
dep = (torch.exp(torch.randn((N, 6480), generator=g) * 2.0) * (torch.rand((N, 6480), generator=g) < 0.3)).to(dev)
energy = torch.exp(torch.rand((N, 1), generator=g) * (s.e_max - s.e_min) + s.e_min).to(dev)
dep = dep / dep.sum(1, keepdim=True) * energy * 0.8


def ev(fn, reps):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): out = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out


ms_pre, (x, c) = ev(lambda: ch.preprocess(dep, energy), 10)
ms_post, (back, e) = ev(lambda: ch.postprocess(x, c), 10)
ms_pre_t, _ = ev(lambda: TO.preprocess(dep, energy, s), 2)
ms_post_t, (back_t, _) = ev(lambda: TO.postprocess(x, c, s), 2)
torch.set_num_threads(min(len(os.sched_getaffinity(0)), 16))
ns = 2000
xc, cc = x[:ns].cpu(), c[:ns].cpu()
t0 = time.perf_counter(); TO.postprocess(xc, cc, s); cpu_s = time.perf_counter() - t0
bytes_pre = N * (6480 * 4 * 2 + 4 + 46 * 4)   # algorithmic: shower read once + x written once (+ energy, conditions)
bytes_post = N * (6480 * 4 * 2 + 46 * 4 + 4)
for name, ms, ms_t, byt in (("preprocess (forward chain)", ms_pre, ms_pre_t, bytes_pre), ("postprocess (reverse chain)", ms_post, ms_post_t, bytes_post)):
    rec = {"metric": f"shape-model {name}, ds2", "value": round(N / ms * 1e3, 0), "unit": "showers/s", "n_showers": N, "ms": round(ms, 3),
           "roofline": {"bound": "hbm", "achieved": round(byt / ms / 1e6, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(byt / ms / 1e6 / 8000.0, 3),
                        "traffic": None, "algorithmic_bytes_per_shower": byt // N},
           "torch_rocm_eager_same_gpu_ms": round(ms_t, 2), "speedup_vs_torch_eager": round(ms_t / ms, 1)}
    if "post" in name:
        rec["cpu_baseline"] = {"value": round(ns / cpu_s, 0), "unit": "showers/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"{ns} showers through the PyTorch-CPU oracle's reverse chain ({cpu_s:.2f} s)"}
        rec["max_rel_diff_vs_torch"] = float(((back - back_t).abs() / (back_t.abs() + 1e-6 * back_t.abs().max())).max())
    print(json.dumps(rec), flush=True)
