"""Per-item timeline of the persistent attention forward (diagnostic build: V4H_EXTRA_FLAGS=-DV4H_ATTN_STAMPS python -m vit4hep_amd.build --force).
Stamps of wave 0 (100 MHz constant clock): 0 loop top, 1 after the barrier (K/V landed), 2 after requesting the next item, 3 after QK^T + max, 4 after the softmax,
5 after P V, 6 after the stores.  Prints, per item slot of a workgroup, the median duration of each phase over all workgroups."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests import hiputil as U
from vit4hep_amd import _lib

lib = C.CDLL(_lib.LIB_PATH)
B, T, H, dh = 128, 135, 6, 80
qkv = (torch.randn((B * T, 3 * H * dh)) * 0.7).to(U.DEV).to(torch.bfloat16)
for _ in range(5):
    o, lse = U.attention_fwd("bf16", qkv, B, T, H, dh)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); o, lse = U.attention_fwd("bf16", qkv, B, T, H, dh); e1.record(); torch.cuda.synchronize()
print("one call:", round(e0.elapsed_time(e1) * 1e3, 1), "us")
buf = np.zeros(256 * 4 * 8, dtype=np.uint64)
assert lib.v4h_debug_attn_stamps(buf.ctypes.data_as(C.c_void_p)) == 0
st = buf.reshape(256, 4, 8).astype(np.int64)
t0 = st[:, 0, 0].min()
names = ["barrier wait (K/V landed, stores drained)", "request next item (DMA + row loads)", "QK^T + max", "softmax (exp, sum)", "P V", "stores"]
for n in range(3):
    d = np.diff(st[:, n, :7], axis=1) * 10.0  # ns
    start = (st[:, n, 0] - t0) * 10.0
    vm = (st[:, n, 7] - st[:, n, 0]) * 10.0
    print(f"   of the barrier wait, this wave's own vmcnt(0) (its DMA pieces + row loads + previous stores): median {np.median(vm)/1e3:5.2f} us, max {vm.max()/1e3:5.2f} us")
    print(f"item {n}: starts at median {np.median(start)/1e3:6.2f} us after the first workgroup's start;  " + ";  ".join(f"{nm}: {np.median(d[:, k])/1e3:5.2f} us" for k, nm in enumerate(names)))
print("last stamp (median over workgroups):", round(float(np.median(st[:, 2, 6] - t0)) * 10 / 1e3, 2), "us; max", round(float((st[:, 2, 6] - t0).max()) * 10 / 1e3, 2), "us")

arr = np.zeros(256 * 4 * 16, dtype=np.uint64)
assert lib.v4h_debug_attn_arrivals(arr.ctypes.data_as(C.c_void_p)) == 0
arr = arr.reshape(256, 4, 16).astype(np.int64)[:, :, :9]
for n in (1, 2):
    rel = (arr[:, n, :] - arr[:, n, :].min(axis=1, keepdims=True)) * 10.0 / 1e3   # us after the first wave to arrive
    print(f"arrival at the barrier of item {n}, us after the first wave of the workgroup (median over workgroups), waves 0..8 (SIMD = wave % 4):", np.round(np.median(rel, axis=0), 2))
