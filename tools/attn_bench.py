"""Attention core alone (v4h_op_attention_fwd / _bwd through the C ABI) at the shapes of the workloads: HIP-event time per call, cold caches (buffer sets rotated),
values checked against torch SDPA in f32 on the same bf16 inputs.  Run it under `rocprofv3 --pmc ... -- python3 tools/attn_bench.py` for counters.
usage (GPU box): python tools/attn_bench.py [ds2|ds3|calohad] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vit4hep_amd import _lib

which = sys.argv[1] if len(sys.argv) > 1 else "ds2"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
B, T = {"ds2": (128, 135), "ds3": (64, 450), "calohad": (32, 606), "lemurs": (64, 135)}[which]
H, dh = 6, 80
D = H * dh
lib = _lib.load()
dev = "cuda:0"
dt = torch.bfloat16
MODE = _lib.MODES["bf16"]
s = _lib.stream_ptr(dev)
SETS = int(os.environ.get("SETS", "6"))
g = torch.Generator(device=dev).manual_seed(3)
sets = []
for _ in range(SETS):
    qkv = torch.randn((B * T, 3 * D), device=dev, generator=g).to(dt)
    do = torch.randn((B * T, D), device=dev, generator=g).to(dt)
    o = torch.empty((B * T, D), device=dev, dtype=dt)
    lse = torch.empty((B, H, T), device=dev)
    delta = torch.empty((B, H, T), device=dev)
    dqkv = torch.empty_like(qkv)
    sets.append((qkv, do, o, lse, delta, dqkv))


def fwd(k):
    qkv, do, o, lse, delta, dqkv = sets[k % SETS]
    _lib.check(lib.v4h_op_attention_fwd(MODE, _lib.ptr(qkv), _lib.ptr(o), _lib.ptr(lse), B, T, H, dh, s))


def bwd(k):
    qkv, do, o, lse, delta, dqkv = sets[k % SETS]
    _lib.check(lib.v4h_op_attention_bwd(MODE, _lib.ptr(qkv), _lib.ptr(o), _lib.ptr(do), _lib.ptr(lse), _lib.ptr(delta), _lib.ptr(dqkv), B, T, H, dh, s))


def timeit(fn):
    for k in range(SETS):
        fn(k)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(reps):
        fn(k)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


# values (set 0) against torch on the same inputs
qkv, do, o, lse, delta, dqkv = sets[0]
fwd(0); bwd(0)
q, k, v = [t.reshape(B, T, H, dh).transpose(1, 2).float().requires_grad_(True) for t in qkv.reshape(B * T, 3, D).unbind(1)]
ref = torch.nn.functional.scaled_dot_product_attention(q, k, v)
ref.backward(do.reshape(B, T, H, dh).transpose(1, 2).float())
ro = ref.transpose(1, 2).reshape(B * T, D)
rg = torch.stack([t.grad.transpose(1, 2).reshape(B * T, D) for t in (q, k, v)], 1).reshape(B * T, 3 * D)
e_o = ((o.float() - ro).abs().max() / ro.abs().max()).item()
e_g = ((dqkv.float() - rg).abs().max() / rg.abs().max()).item()
del q, k, v, ref, ro, rg
fl_f = 4.0 * B * H * T * T * dh
tf, tb = [], []
for _ in range(5):
    tf.append(timeit(fwd)); tb.append(timeit(bwd))
tf.sort(); tb.sort()
print(f"{which}: B={B} T={T}  fwd {tf[2]:.1f} us (min {tf[0]:.1f}) = {fl_f / tf[2] / 1e6:.0f} TFLOP/s, err {e_o:.1e} | bwd {tb[2]:.1f} us (min {tb[0]:.1f}) = "
      f"{2.5 * fl_f / tb[2] / 1e6:.0f} TFLOP/s, err {e_g:.1e}", flush=True)
