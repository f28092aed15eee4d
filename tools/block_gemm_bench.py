"""The twelve token-sized contractions of one DiT block (nn/vit.py:312-333,416-420: forward, dgrad, wgrad of qkv / proj / fc1 / fc2, with the fused GELU and
DGELU epilogues) on the two product kernels - 128 x 160 two-workgroup kernel (v4h_gemm.h) and 256 x 160 ring kernel (v4h_gemm2.h) - interleaved rounds in ONE
process, random operands, COLD caches (every call of a shape uses the next of SETS buffer sets), each kernel's result checked against torch on the same operands.
usage (GPU box): python tools/block_gemm_bench.py [BT] [rounds]      ONLY=<substring> restricts the shapes, KERNELS=1,2 the kernels"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vit4hep_amd import _lib

BT = int(sys.argv[1]) if len(sys.argv) > 1 else 17280
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 5
SETS = int(os.environ.get("SETS", "6"))
KERNELS = [int(k) for k in os.environ.get("KERNELS", "1,2").split(",")]
lib = _lib.load()
dev = "cuda:0"
D, M = 480, 1920
dt = torch.bfloat16
MODE = _lib.MODES["bf16"]
s = _lib.stream_ptr(dev)


def gelu_ref(x):
    u = 0.7978845608028654 * (x + 0.044715 * x ** 3)
    t = torch.tanh(u)
    return 0.5 * x * (1 + t), 0.5 * (1 + t) + 0.5 * x * (1 - t * t) * 0.7978845608028654 * (1 + 3 * 0.044715 * x * x)


def make(kind, I, J, K, g):
    """returns (call, check) for one buffer set"""
    if kind == "wgrad":
        P = torch.randn((K, I), device=dev, generator=g).to(dt)
        Q = torch.randn((K, J), device=dev, generator=g).to(dt)
        sk = int(lib.v4h_op_gemm_wgrad_splitk(MODE, I, J, K))
        out = torch.zeros((I, J), device=dev)
        slab = torch.empty((max(sk, 1), I, J), device=dev)
        cs = torch.zeros(I, device=dev)

        def call():
            sk_now = int(lib.v4h_op_gemm_wgrad_splitk(MODE, I, J, K))
            _lib.check(lib.v4h_op_gemm_wgrad_slab(MODE, _lib.ptr(P), I, _lib.ptr(Q), J, _lib.ptr(slab), _lib.ptr(out), I, J, K, min(sk_now, slab.shape[0]), _lib.ptr(cs), s))

        def check():
            out.zero_(); cs.zero_(); call()
            ref = P.float().t() @ Q.float()
            return ((out - ref).abs().max() / ref.abs().max()).item()

        return call, check
    P = torch.randn((I, K), device=dev, generator=g).to(dt)
    qks = kind in ("dgrad", "dgelu")
    Q = (torch.randn((K, J) if qks else (J, K), device=dev, generator=g) * (K ** -0.5)).to(dt)
    out = torch.zeros((I, J), device=dev, dtype=dt)
    if kind in ("fwd", "dgrad"):
        bias = torch.randn(J, device=dev, generator=g)

        def call():
            _lib.check(lib.v4h_op_gemm(MODE, _lib.ptr(P), K, 0, _lib.ptr(Q), Q.stride(0), int(qks), _lib.ptr(bias), _lib.ptr(out), J, 0, I, J, K, 1, None, s))

        def check():
            out.zero_(); call()
            ref = P.float() @ (Q.float() if qks else Q.float().t()) + bias
            return ((out.float() - ref).abs().max() / ref.abs().max()).item()

        return call, check
    if kind == "gelu":
        bias = torch.randn(J, device=dev, generator=g)
        dh = torch.zeros((I, J), device=dev, dtype=dt)

        def call():
            _lib.check(lib.v4h_op_gemm_gelu(MODE, _lib.ptr(P), K, _lib.ptr(Q), K, _lib.ptr(bias), _lib.ptr(out), J, _lib.ptr(dh), J, I, J, K, s))

        def check():
            out.zero_(); dh.zero_(); call()
            y, dy = gelu_ref(P.float() @ Q.float().t() + bias)
            return max(((out.float() - y).abs().max() / y.abs().max()).item(), ((dh.float() - dy).abs().max() / dy.abs().max()).item())

        return call, check
    gg = torch.rand((I, J), device=dev, generator=g).to(dt)  # dgelu

    def call():
        _lib.check(lib.v4h_op_gemm_dgelu(MODE, _lib.ptr(P), K, _lib.ptr(Q), J, _lib.ptr(gg), J, _lib.ptr(out), J, I, J, K, s))

    def check():
        out.zero_(); call()
        ref = (P.float() @ Q.float()) * gg.float()
        return ((out.float() - ref).abs().max() / ref.abs().max()).item()

    return call, check


def time_rot(calls, reps):
    for c in calls:
        c()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(reps):
        calls[k % len(calls)]()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


shapes = [("fwd qkv", "fwd", BT, 3 * D, D), ("fwd proj", "fwd", BT, D, D), ("fwd fc1 GELU", "gelu", BT, M, D), ("fwd fc2", "fwd", BT, D, M),
          ("dgrad fc2 DGELU", "dgelu", BT, M, D), ("dgrad fc1", "dgrad", BT, D, M), ("dgrad proj", "dgrad", BT, D, D), ("dgrad qkv", "dgrad", BT, D, 3 * D),
          ("wgrad fc2", "wgrad", D, M, BT), ("wgrad fc1", "wgrad", M, D, BT), ("wgrad proj", "wgrad", D, D, BT), ("wgrad qkv", "wgrad", 3 * D, D, BT),
          ("fwd fc1 plain", "fwd", BT, M, D), ("dgrad fc2 plain", "dgrad", BT, M, D)]
only = os.environ.get("ONLY")
names = {1: "two-wg", 2: "ring", 3: "w-stat", 0: "auto"}
tot = {k: 0.0 for k in KERNELS}
for nm, kind, I, J, K in shapes:
    if only and only not in nm:
        continue
    g = torch.Generator(device=dev).manual_seed(len(nm) * 7 + I % 13)
    sets = [make(kind, I, J, K, g) for _ in range(SETS)]
    fl = 2.0 * I * J * K
    res, err = {k: [] for k in KERNELS}, {}
    for k in KERNELS:
        _lib.check(lib.v4h_select_contraction_kernel(k))
        err[k] = sets[0][1]()
    for _ in range(ROUNDS):
        for k in KERNELS:
            _lib.check(lib.v4h_select_contraction_kernel(k))
            res[k].append(time_rot([c for c, _ in sets], 18))
    lib.v4h_select_contraction_kernel(0)
    line = f"{nm:16s} I={I:6d} J={J:5d} K={K:6d}"
    for k in KERNELS:
        t = sorted(res[k])
        med = t[len(t) // 2]
        if nm.split()[-1] not in ("plain",):
            tot[k] += med
        line += f" | {names[k]:6s} {med:6.1f} us (min {t[0]:6.1f}) {fl / med / 1e6:6.0f} TF err {err[k]:.1e}"
    print(line, flush=True)
    del sets
    torch.cuda.empty_cache()
print("sum of the twelve block contractions: " + ", ".join(f"{names[k]} {tot[k]:.0f} us" for k in KERNELS))
