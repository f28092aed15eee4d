"""Time v4h_sq_norm_accum on a gradient-sized buffer (26 M floats).  usage: V4H_SQNORM_BLOCKS=<cap> python tools/experiments/sqnorm_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vit4hep_amd import _lib
lib = _lib.load()
n = 26042528
g = torch.randn(n, device="cuda:0")
out = torch.zeros((), device="cuda:0")
s = _lib.stream_ptr(g.device)
for _ in range(5):
    lib.v4h_sq_norm_accum(_lib.ptr(g), n, _lib.ptr(out), s)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    lib.v4h_sq_norm_accum(_lib.ptr(g), n, _lib.ptr(out), s)
e1.record()
torch.cuda.synchronize()
print(os.environ.get("V4H_SQNORM_BLOCKS", "2048"), "blocks:", round(e0.elapsed_time(e1) * 1e3 / 50, 1), "us per call;", round(n * 4 / (e0.elapsed_time(e1) * 1e-3 / 50) / 1e12, 2), "TB/s")
