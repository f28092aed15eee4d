"""Slot timeline of the weight-stationary contraction (diagnostic build:
    V4H_BUILD_TAG=st3 V4H_EXTRA_FLAGS=-DV4H_GEMM3_STAMPS python -m vit4hep_amd.build ; VIT4HEP_AMD_LIB=$PWD/vit4hep_amd/libvit4hep_hip_st3.so python tools/experiments/gemm3_stamps.py [qkv|proj|fc1|dproj|dfc2])
Stamps (shader clock) per wave: 0 start, 1 weight prologue (dgrad form) done, 2 ring + weight loads requested, 3 all arrived, 4 first barrier; then per tile
interval i: 5+4i first slot done (half 0: matrix, half 1: auxiliary), 6+4i second slot done, 7+4i counted vmcnt wait done, 8+4i barrier passed."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from vit4hep_amd import _lib

which = sys.argv[1] if len(sys.argv) > 1 else "qkv"
BT = int(sys.argv[2]) if len(sys.argv) > 2 else 17280
lib = _lib.load()
raw = C.CDLL(_lib.LIB_PATH)
dev, dt = "cuda:0", torch.bfloat16
s = _lib.stream_ptr(dev)
J, K, qks = {"qkv": (1440, 480, 0), "proj": (480, 480, 0), "fc1": (1920, 480, 0), "dproj": (480, 480, 1), "dfc2": (1920, 480, 1), "gelu": (1920, 480, 0), "dgelu": (1920, 480, 1)}[which]
P = torch.randn((BT, K), device=dev).to(dt)
Q = torch.randn((K, J) if qks else (J, K), device=dev).to(dt)
bias = torch.randn(J, device=dev)
out = torch.empty((BT, J), device=dev, dtype=dt)
args = (_lib.MODES["bf16"], _lib.ptr(P), K, 0, _lib.ptr(Q), Q.stride(0), qks, _lib.ptr(bias), _lib.ptr(out), J, 0, BT, J, K, 1, None, s)
call = lib.v4h_op_gemm
if which == "gelu":
    dh = torch.empty((BT, J), device=dev, dtype=dt)
    args = (_lib.MODES["bf16"], _lib.ptr(P), K, _lib.ptr(Q), K, _lib.ptr(bias), _lib.ptr(out), J, _lib.ptr(dh), J, BT, J, K, s)
    call = lib.v4h_op_gemm_gelu
if which == "dgelu":
    gg = torch.rand((BT, J), device=dev).to(dt)
    args = (_lib.MODES["bf16"], _lib.ptr(P), K, _lib.ptr(Q), J, _lib.ptr(gg), J, _lib.ptr(out), J, BT, J, K, s)
    call = lib.v4h_op_gemm_dgelu
for _ in range(5):
    _lib.check(call(*args))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); _lib.check(call(*args)); e1.record(); torch.cuda.synchronize()
print(f"{which} I={BT}: one call {e0.elapsed_time(e1) * 1e3:.1f} us (stamped build)")
NST = 128
buf = np.zeros(256 * 8 * NST, dtype=np.uint32)
assert raw.v4h_debug_gemm3_stamps(buf.ctypes.data_as(C.c_void_p)) == 0
st = buf.reshape(256, 8, NST).astype(np.int64)
d = lambda a, b: (a - b) & 0xFFFFFFFF
med = lambda x: int(np.median(x))
print("prologue, median clocks over all waves: start->weight prologue %d, ->requests issued %d, ->arrived %d, ->first barrier %d" % (
    med(d(st[:, :, 1], st[:, :, 0])), med(d(st[:, :, 2], st[:, :, 1])), med(d(st[:, :, 3], st[:, :, 2])), med(d(st[:, :, 4], st[:, :, 3]))))
print("whole kernel per wave (start -> last barrier), median: %d clocks" % med(d(st[:, :, 4 + 4 * 8], st[:, :, 0])))
for half in (0, 1):
    w = st[:, 4 * half:4 * half + 4]
    print(f"half {half}: per interval: first slot ({'matrix' if half == 0 else 'auxiliary'}), second slot, vmcnt wait, barrier wait | interval")
    for i in range(10):
        b = 4 + 4 * i
        print(f"  interval {i}: {med(d(w[..., b + 1], w[..., b]))} {med(d(w[..., b + 2], w[..., b + 1]))} {med(d(w[..., b + 3], w[..., b + 2]))} {med(d(w[..., b + 4], w[..., b + 3]))} | {med(d(w[..., b + 4], w[..., b]))}")
