"""Time the contraction kernel at the real shapes of one DiT block through the C ABI (HIP events).
usage (GPU box): python tools/gemm_bench.py [bf16|f32] [BT]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit4hep_amd import _lib

mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
BT = int(sys.argv[2]) if len(sys.argv) > 2 else 17280
dt = torch.bfloat16 if mode == "bf16" else torch.float32
lib = _lib.load()
from _abl import require_ablation_lib
require_ablation_lib(lib)

dev = "cuda:0"
D, M = 480, 1920

def timeit(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps

def run(name, I, J, K, pks, qks, splitk=1):
    P = torch.randn((K, I) if pks else (I, K), device=dev).to(dt)
    Q = torch.randn((K, J) if qks else (J, K), device=dev).to(dt)
    f32out = bool(pks and qks)
    out = torch.zeros((I, J), device=dev, dtype=torch.float32 if f32out else dt)
    s = _lib.stream_ptr(dev)
    def fn():
        _lib.check(lib.v4h_op_gemm(_lib.MODES[mode], _lib.ptr(P), P.stride(0), int(pks), _lib.ptr(Q), Q.stride(0), int(qks), None, _lib.ptr(out), out.stride(0),
                                   int(f32out), I, J, K, splitk, None, s))
    us = timeit(fn)
    print(f"{name:34s} I={I:6d} J={J:5d} K={K:6d} split={splitk:2d}  {us:8.1f} us  {2.0*I*J*K/us/1e6:7.1f} TFLOP/s", flush=True)

cfgs = [int(c) for c in os.environ.get("CFGS", "0,20,21").split(",") if c]
wcfgs = [int(c) for c in os.environ.get("WCFGS", "0,8,7").split(",") if c]
for cfg in cfgs:
    lib.v4h_debug_set_gemm_cfg(cfg, 0)
    print(f"--- fwd/dgrad cfg {cfg}")
    for nm, J, K in (("fwd qkv", 3*D, D), ("fwd proj", D, D), ("fwd fc1", M, D), ("fwd fc2", D, M)):
        run(nm, BT, J, K, 0, 0)
    for nm, J, K in (("dgrad qkv", D, 3*D), ("dgrad proj", D, D), ("dgrad fc1", D, M), ("dgrad fc2", M, D)):
        run(nm, BT, J, K, 0, 1)
for cfg in wcfgs:
    lib.v4h_debug_set_gemm_cfg(0, cfg)
    print(f"--- wgrad cfg {cfg}")
    for nm, I, J in (("wgrad qkv", 3*D, D), ("wgrad proj", D, D), ("wgrad fc1", M, D), ("wgrad fc2", D, M)):
        for sk in (8, 16):
            run(nm, I, J, BT, 1, 1, sk)
