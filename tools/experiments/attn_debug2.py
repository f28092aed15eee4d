"""Which KEYS of a wrong (batch, head) item carry a wrong probability: v = one-hot of (key mod 80), scores linear in the key index."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vit4hep_amd import _lib
lib = _lib.load()
dev = "cuda:0"; dt = torch.bfloat16; MODE = _lib.MODES["bf16"]; s = _lib.stream_ptr(dev)
B, T, H, dh = int(os.environ.get("B", 128)), 135, 6, 80
D = H * dh
x = torch.zeros((B, T, 3, H, dh), device=dev)
x[:, :, 0, :, 0] = 4.0                                           # q = 4 e_0
x[:, :, 1, :, 0] = (torch.arange(T, device=dev).float() / 16)[None, :, None]  # k_t = (t / 16) e_0  (exact in bf16 for t < 256)
x[:, :, 1, :, 1:] = torch.randn((B, T, H, dh - 1), device=dev)   # the rest of k is random but q is zero there
oh = torch.zeros((T, dh), device=dev); oh[torch.arange(T), torch.arange(T) % dh] = 1
x[:, :, 2] = oh[None, :, None, :]
qkv = x.reshape(B * T, 3 * D).to(dt)
o = torch.zeros((B * T, D), device=dev, dtype=dt); lse = torch.zeros((B, H, T), device=dev)
_lib.check(lib.v4h_op_attention_fwd(MODE, _lib.ptr(qkv), _lib.ptr(o), _lib.ptr(lse), B, T, H, dh, s)); torch.cuda.synchronize()
sc = (4.0 * torch.arange(T, device=dev).float() / 16) / dh ** 0.5
p = torch.softmax(sc, 0)                                          # same for every query
ref = torch.zeros(dh, device=dev); ref.index_add_(0, torch.arange(T, device=dev) % dh, p)
oo = o.float().reshape(B, T, H, dh)
err = (oo - ref).abs().amax(dim=(1, 3))
bad = (err > 2e-3).nonzero().tolist()
print("bad items", len(bad), bad[:12])
print("lse ref", torch.logsumexp(sc, 0).item(), "lse got (item 0,0 / first bad)", lse[0, 0, 0].item(), lse[bad[0][0], bad[0][1], :4].tolist() if bad else None)
if bad:
    b0, h0 = bad[0]
    r = oo[b0, :, h0]                                             # (T, dh)
    print("query 0: o * l_ref / p_ref per d (1 = right; d and d+80 share a column):")
    ratio = (r[0] / ref)
    print([round(v, 2) for v in ratio.tolist()])
    print("query 100:", [round(v, 2) for v in (r[100] / ref).tolist()])
