"""A/B of the two contraction kernels (v4h_gemm.h 128x160 / two workgroups per CU  vs  v4h_gemm2.h 256x160 / 8 waves / 3-stage ring)
at the shapes of one DiT block, interleaved rounds in ONE process, random operands, each result checked against an f32 torch
matmul of the same bf16 operands.
usage (GPU box): python tools/gemm2_bench.py [BT] [rounds]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vit4hep_amd import _lib

BT = int(sys.argv[1]) if len(sys.argv) > 1 else 17280
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 5
lib = _lib.load()
from _abl import require_ablation_lib
require_ablation_lib(lib)

dev = "cuda:0"
D, M = 480, 1920
dt = torch.bfloat16
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


def make(name, I, J, K, pks, qks, splitk=1):
    g = torch.Generator(device=dev).manual_seed(hash(name) % 1000)
    P = torch.randn((K, I) if pks else (I, K), device=dev, generator=g).to(dt)
    Q = torch.randn((K, J) if qks else (J, K), device=dev, generator=g).to(dt)
    bias = torch.randn(J, device=dev, generator=g) if not pks else None
    s = _lib.stream_ptr(dev)
    if pks and qks:
        out = torch.zeros((I, J), device=dev, dtype=torch.float32)
        slab = torch.empty((splitk, I, J), device=dev, dtype=torch.float32)
        cs = torch.zeros(I, device=dev, dtype=torch.float32)

        def fn():
            _lib.check(lib.v4h_op_gemm_wgrad_slab(MODE, _lib.ptr(P), P.stride(0), _lib.ptr(Q), Q.stride(0), _lib.ptr(slab), _lib.ptr(out), I, J, K, splitk, _lib.ptr(cs), s))

        def ref():
            return P.float().t() @ Q.float(), P.float().sum(0)

        def reset():
            out.zero_(); cs.zero_()

        return fn, ref, (lambda: (out, cs)), reset
    out = torch.zeros((I, J), device=dev, dtype=dt)

    def fn():
        _lib.check(lib.v4h_op_gemm(MODE, _lib.ptr(P), P.stride(0), int(pks), _lib.ptr(Q), Q.stride(0), int(qks), _lib.ptr(bias), _lib.ptr(out), out.stride(0), 0, I, J, K, 1,
                                   None, s))

    def ref():
        return (P.float() @ (Q.float() if qks else Q.float().t()) + bias,)

    return fn, ref, (lambda: (out,)), (lambda: out.zero_())


shapes = [("fwd qkv", BT, 3 * D, D, 0, 0, 1), ("fwd proj", BT, D, D, 0, 0, 1), ("fwd fc1", BT, M, D, 0, 0, 1), ("fwd fc2", BT, D, M, 0, 0, 1),
          ("dgrad qkv", BT, D, 3 * D, 0, 1, 1), ("dgrad proj", BT, D, D, 0, 1, 1), ("dgrad fc1", BT, D, M, 0, 1, 1), ("dgrad fc2", BT, M, D, 0, 1, 1),
          ("wgrad qkv", 3 * D, D, BT, 1, 1, 8), ("wgrad proj", D, D, BT, 1, 1, 8), ("wgrad fc1", M, D, BT, 1, 1, 8), ("wgrad fc2", D, M, BT, 1, 1, 8),
          ("wgrad fc1 s4", M, D, BT, 1, 1, 4), ("wgrad qkv s14", 3 * D, D, BT, 1, 1, 14), ("wgrad qkv s16", 3 * D, D, BT, 1, 1, 16), ("wgrad fc1 s10", M, D, BT, 1, 1, 10),
          ("wgrad fc2 s10", D, M, BT, 1, 1, 10), ("wgrad fc1 s16", M, D, BT, 1, 1, 16), ("wgrad proj s16", D, D, BT, 1, 1, 16), ("wgrad proj s24", D, D, BT, 1, 1, 24),
          ("wgrad proj s32", D, D, BT, 1, 1, 32), ("wgrad proj s40", D, D, BT, 1, 1, 40)]
only = os.environ.get("ONLY")
for nm, I, J, K, pks, qks, sk in shapes:
    if only and only not in nm:
        continue
    fn, ref, outs, reset = make(nm, I, J, K, pks, qks, sk)
    r = ref()
    res = {}
    for ver in (1, 2, 9):
        lib.v4h_debug_set_gemm_cfg(0, 1000 * ver)  # 1000 = two-workgroup kernel, 2000 = ring kernel (lock-step), 9000 = ring kernel, ping-pong schedule
        reset()
        fn()
        torch.cuda.synchronize()
        errs = []
        for o, rr in zip(outs(), r):
            errs.append(((o.float() - rr).abs().max() / rr.abs().max()).item())
        res[ver] = [max(errs)]
    abl = (3, 4, 5, 6, 7, 8, 10, 11, 12) if (not pks and not qks and os.environ.get("ABL")) else ()  # 10000/11000/12000: ping-pong schedule without DMA in the loop / without stores / without both
    #   # 3000: no DMA in loop; 4000: no stores; 5000: neither; 6000: + no fragment reads; 7000: no DMA/stores/barrier; 8000: MFMA only
    for ver in abl:
        res[ver] = [0.0]
    for _ in range(ROUNDS):
        for ver in (1, 2, 9) + abl:
            lib.v4h_debug_set_gemm_cfg(0, 1000 * ver)
            res[ver].append(timeit(fn))
    fl = 2.0 * I * J * K
    t1, t2, t9 = sorted(res[1][1:]), sorted(res[2][1:]), sorted(res[9][1:])
    m1, m2, m9 = t1[len(t1) // 2], t2[len(t2) // 2], t9[len(t9) // 2]
    print(f"{nm:16s} I={I:6d} J={J:5d} K={K:6d} split={sk:2d} | old {m1:7.1f} us {fl/m1/1e6:7.1f} TF (min {t1[0]:6.1f}) err {res[1][0]:.1e} | new {m2:7.1f} us {fl/m2/1e6:7.1f} TF "
          f"(min {t2[0]:6.1f}) err {res[2][0]:.1e} | x{m1/m2:.3f} | pp {m9:7.1f} us {fl/m9/1e6:7.1f} TF (min {t9[0]:6.1f}) err {res[9][0]:.1e} | x{m1/m9:.3f}" + "".join(f" | abl{v} {sorted(res[v][1:])[len(res[v][1:]) // 2]:6.1f} us" for v in abl), flush=True)
lib.v4h_debug_set_gemm_cfg(0, -1)
