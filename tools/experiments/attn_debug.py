"""Structured-input checks of the single-chunk attention forward (which part of the kernel is wrong when the random-input comparison fails)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vit4hep_amd import _lib
lib = _lib.load()
dev = "cuda:0"; dt = torch.bfloat16; MODE = _lib.MODES["bf16"]; s = _lib.stream_ptr(dev)
B, T, H, dh = int(os.environ.get("B", 2)), int(os.environ.get("T", 135)), 6, 80
D = H * dh


def run(qkv):
    o = torch.full((B * T, D), 7.0, device=dev, dtype=dt)
    lse = torch.zeros((B, H, T), device=dev)
    _lib.check(lib.v4h_op_attention_fwd(MODE, _lib.ptr(qkv), _lib.ptr(o), _lib.ptr(lse), B, T, H, dh, s))
    torch.cuda.synchronize()
    return o, lse


def ref(qkv):
    q, k, v = [t.reshape(B, T, H, dh).transpose(1, 2).float() for t in qkv.reshape(B * T, 3, D).unbind(1)]
    sc = q @ k.transpose(-1, -2) / dh ** 0.5
    return (torch.softmax(sc, -1) @ v).transpose(1, 2).reshape(B * T, D), torch.logsumexp(sc, -1)


def report(name, qkv):
    o, lse = run(qkv)
    ro, rl = ref(qkv)
    e = (o.float() - ro).abs()
    print(f"{name:28s} max err o {e.max().item():.3e} (ref max {ro.abs().max().item():.3e})  lse err {(lse - rl).abs().max().item():.3e}", flush=True)
    if e.max() > 0.05 * max(ro.abs().max().item(), 1e-3):
        bad = (e > 0.05 * ro.abs().max()).nonzero()
        rows = sorted(set(bad[:, 0].tolist())); cols = sorted(set(bad[:, 1].tolist()))
        print(f"   bad rows {len(rows)} of {B*T}: first {rows[:12]} ... ; bad cols {len(cols)} of {D}: first {cols[:24]}")
        r0, c0 = bad[0].tolist()
        print(f"   o[{r0},{c0}:{c0+8}] = {o[r0, c0:c0+8].float().tolist()}\n   ref            = {ro[r0, c0:c0+8].tolist()}")


g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda: torch.randn((B * T, 3 * D), device=dev, generator=g)
x = rnd(); x[:, :D] = 0; report("q = 0 (uniform softmax)", x.to(dt))
x = rnd(); x[:, 2 * D:] = 1; report("v = 1", x.to(dt))
x = rnd(); x[:, :2 * D] = 0; x[:, 2 * D:] = torch.arange(D, device=dev).float()[None, :] / 64; report("q = k = 0, v = column index", x.to(dt))
x = rnd(); x[:, :D] *= 0; x[:, 2 * D:] = (torch.arange(B * T, device=dev) % T).float()[:, None] / 16; report("q = 0, v = token index", x.to(dt))
x = rnd(); x[:, 64:80] = 0; x[:, D + 64:D + 80] = 0; report("head 0: no head_dim tail", x.to(dt))
report("random", rnd().to(dt))
# which (batch, head) items are wrong
qkv = rnd().to(dt)
o, lse = run(qkv)
ro, rl = ref(qkv)
e = (o.float() - ro).abs().reshape(B, T, H, dh).amax(dim=(1, 3))  # (B, H)
bad = (e > 0.05).nonzero().tolist()
print(f"bad items {len(bad)} of {B*H}; first 40: {bad[:40]}")
el = (lse - rl).abs().amax(dim=2)
print("bad lse items", (el > 0.05).sum().item())
# per token within a bad item
if bad:
    b0, h0 = bad[0]
    et = (o.float() - ro).abs().reshape(B, T, H, dh)[b0, :, h0].amax(dim=1)
    print("tokens wrong in first bad item:", (et > 0.05).nonzero().flatten().tolist()[:60])
    print("lse diff first bad item:", (lse - rl)[b0, h0, :24].tolist())
