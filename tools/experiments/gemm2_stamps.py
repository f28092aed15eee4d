"""Slot timeline of the ping-pong ring contraction (diagnostic build: V4H_EXTRA_FLAGS=-DV4H_GEMM2_STAMPS python -m vit4hep_amd.build --force).
Stamps (shader clock) per wave and stage: 0 load slot starts, 1 fragments in registers / DMA share and (at a seam) epilogue issued, 2 after the counted vmcnt wait
(half 1), 3 after the barrier = matrix slot starts, 4 MFMAs issued, 5 after the counted vmcnt wait (half 0); the next stage's 0 is after the second barrier.
usage (GPU box): python tools/experiments/gemm2_stamps.py [fc1|qkv|fc2|proj|wgrad_qkv|wgrad_fc1|wgrad_fc2] [ver]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from vit4hep_amd import _lib

which = sys.argv[1] if len(sys.argv) > 1 else "fc1"
ver = int(sys.argv[2]) if len(sys.argv) > 2 else 9
lib = _lib.load()
from tools._abl import require_ablation_lib
require_ablation_lib(lib)

raw = C.CDLL(_lib.LIB_PATH)
dev = "cuda:0"
BT = 17280
dt = torch.bfloat16
s = _lib.stream_ptr(dev)
lib.v4h_debug_set_gemm_cfg(0, 1000 * ver)
if which.startswith("wgrad"):  # weight gradient of the block's qkv / fc1 / fc2 Linear: both operands token-major, split-K slabs
    I, J = {"wgrad_qkv": (1440, 480), "wgrad_fc1": (1920, 480), "wgrad_fc2": (480, 1920)}[which]
    K = BT // 8  # (K of one split, for the stage count printed below)
    P = torch.randn((BT, I), device=dev).to(dt)
    Q = torch.randn((BT, J), device=dev).to(dt)
    out = torch.zeros((I, J), device=dev)
    slab = torch.empty((8, I, J), device=dev)
    cs = torch.zeros(I, device=dev)
    args = (_lib.MODES["bf16"], _lib.ptr(P), I, _lib.ptr(Q), J, _lib.ptr(slab), _lib.ptr(out), I, J, BT, 8, _lib.ptr(cs), s)
    call = lib.v4h_op_gemm_wgrad_slab
else:
    J, K = {"fc1": (1920, 480), "qkv": (1440, 480), "fc2": (480, 1920), "proj": (480, 480)}[which]
    P = torch.randn((BT, K), device=dev).to(dt)
    Q = torch.randn((J, K), device=dev).to(dt)
    bias = torch.randn(J, device=dev)
    out = torch.empty((BT, J), device=dev, dtype=dt)
    args = (_lib.MODES["bf16"], _lib.ptr(P), K, 0, _lib.ptr(Q), K, 0, _lib.ptr(bias), _lib.ptr(out), J, 0, BT, J, K, 1, None, s)
    call = lib.v4h_op_gemm
for _ in range(5):
    _lib.check(call(*args))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); _lib.check(call(*args)); e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3
FIRST, NST, NK = 3, 8, 8
buf = np.zeros(256 * 8 * NST * NK, dtype=np.uint32)
assert raw.v4h_debug_gemm2_stamps(buf.ctypes.data_as(C.c_void_p)) == 0
st = buf.reshape(256, 8, NST, NK)
if which.startswith("wgrad"):
    st = st[:((I + 255) // 256) * (J // 160) * 8]  # the workgroups this launch had (the rest of the buffer is from earlier launches)
print(f"{which}: one call {us:.1f} us (stamped build)")
nt = (K + 63) // 64
names = ["load-slot work (frag reads, DMA issue, seam: epilogue)", "vmcnt wait (half 1)", "barrier wait", "MFMA issue (40)", "vmcnt wait (half 0)", "barrier wait"]
dif = lambda x, y: ((x.astype(np.int64) - y.astype(np.int64)) & 0xFFFFFFFF)
for half in (0, 1):
    w = st[:, 4 * half:4 * half + 4]          # (256, 4, NST, 6)
    nxt0 = np.concatenate([w[:, :, 1:, 0], w[:, :, -1:, 5]], axis=2)
    d = [dif(w[..., 1], w[..., 0]), dif(w[..., 2], w[..., 1]), dif(w[..., 3], w[..., 2]), dif(w[..., 4], w[..., 3]), dif(w[..., 5], w[..., 4]), dif(nxt0, w[..., 5])]
    print(f"half {half}: median clocks per stage (t = stage within its tile; {nt} stages per tile), over all workgroups and the half's 4 waves")
    for n in range(NST - 1):
        dma = dif(w[..., 6], w[..., 0])[:, :, n]
        extra = f" [DMA issue {int(np.median(dma))}]" if os.environ.get("DMA_FIRST") else ""  # (stamp 6 exists only in a -DV4H_G2_DMA_FIRST build)
        print(f"  stage {FIRST + n} (t={(FIRST + n) % nt}):{extra} " + "; ".join(f"{nm.split(' (')[0]} {int(np.median(x[:, :, n]))}" for nm, x in zip(names, d)) + f"  | stage total {int(np.median(dif(nxt0, w[..., 0])[:, :, n]))}")
lib.v4h_debug_set_gemm_cfg(0, -1)
