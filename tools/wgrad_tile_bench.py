"""Tile-shape A/B of the split-K weight-gradient contraction (two-workgroup kernel, v4h_gemm.h) through v4h_op_gemm_wgrad_slab.
usage (GPU box): python tools/wgrad_tile_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit4hep_amd import _lib

lib = _lib.load()
from _abl import require_ablation_lib
require_ablation_lib(lib)

dev = "cuda:0"
D, M, BT = 480, 1920, 17280
MODE = _lib.MODES["bf16"]

def timeit(fn, reps=20):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps

for nm, I, J in (("wgrad qkv", 3 * D, D), ("wgrad proj", D, D), ("wgrad fc1", M, D), ("wgrad fc2", D, M)):
    P = torch.randn((BT, I), device=dev).bfloat16()
    Q = torch.randn((BT, J), device=dev).bfloat16()
    ref = P.float().t() @ Q.float()
    for sk in (8, 16):
        out = torch.zeros((I, J), device=dev)
        slab = torch.empty((sk, I, J), device=dev)
        cs = torch.zeros(I, device=dev)
        s = _lib.stream_ptr(dev)
        fn = lambda: _lib.check(lib.v4h_op_gemm_wgrad_slab(MODE, _lib.ptr(P), I, _lib.ptr(Q), J, _lib.ptr(slab), _lib.ptr(out), I, J, BT, sk, _lib.ptr(cs), s))
        res = []
        for cfg, name in ((12, "160x96"), (0, "96x160")):
            lib.v4h_debug_set_gemm_cfg(0, 1000 + cfg)
            out.zero_(); cs.zero_(); fn(); torch.cuda.synchronize()
            err = ((out - ref).abs().max() / ref.abs().max()).item()
            ts = []
            for _ in range(3):
                ts.append(timeit(fn))
            res.append(f"{name} {sorted(ts)[1]:6.1f} us ({2.0*I*J*BT/sorted(ts)[1]/1e6:5.0f} TF, err {err:.0e})")
        print(f"{nm:11s} split {sk:2d} | " + " | ".join(res), flush=True)
lib.v4h_debug_set_gemm_cfg(0, -1)
