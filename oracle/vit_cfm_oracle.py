"""CPU oracle for the ViT-CFM hot path.  TEST INFRASTRUCTURE - never shipped, never timed as product.

A functional restatement (plain PyTorch on CPU, fp32 or fp64, autograd for the
backward) of the algorithm the reference runs for the CaloChallenge *shape* CFM
model.  It works on a flat ``dict[str, Tensor]`` keyed by the reference's
state-dict names instead of on ``nn.Module`` objects, so every formula is
visible in one place.  Each function cites the reference lines it follows
(paths relative to /root/reference).

Pinned against the reference itself: ``oracle/make_golden.py`` imports the
reference's own Python (with stand-ins for the three absent third-party
symbols) in the build container and writes ``tests/golden/*.npz``;
``tests/test_oracle_vs_golden.py`` checks this file against those vectors.
The reference ships no tests/fixtures of its own.
"""

from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np
import torch

# --------------------------------------------------------------------------------------
# configuration (configs/model/cfm/cfm_ds{2,3}_electrons.yaml)
# --------------------------------------------------------------------------------------


@dataclass(frozen=True)
class ViTConfig:
    shape: tuple = (45, 16, 9)  # (L, A, R) voxel grid           cfm_ds2_electrons.yaml:3
    patch_shape: tuple = (3, 16, 1)  # (p1, p2, p3)               cfm_ds2_electrons.yaml:4
    in_channels: int = 1
    condition_dim: int = 46
    hidden_dim: int = 480
    depth: int = 6
    num_heads: int = 6
    mlp_ratio: float = 4.0
    freq_dim: int = 256  # TimestepEmbedder frequency_embedding_size  nn/vit.py:359
    # Multi-segment geometries (CaloChallengeCFM_DS1, CaloGANCFM, CaloHadCFM): a tuple of ((L, A, R), (p1, p2, p3)) per
    # segment; `shape` is then (n_voxels,) (the flat sample of the YAML, e.g. cfm_ds1_photons.yaml:3) and `patch_shape` unused.
    segments: tuple = ()

    @property
    def num_patches(self):
        if self.segments:
            raise ValueError("multi-segment geometry: use seg_num_patches")
        return tuple(s // p for s, p in zip(self.shape, self.patch_shape))

    @property
    def seg_num_patches(self):
        """[(l, a, r), ...] = the reference's num_patches_per_dim (e.g. calohadronic/model.py:36-46)"""
        if not self.segments:
            return [self.num_patches]
        return [tuple(s // p for s, p in zip(shape, patch)) for shape, patch in self.segments]

    @property
    def T(self):
        return sum(l * a * r for l, a, r in self.seg_num_patches)

    @property
    def P(self):
        p1, p2, p3 = self.segments[0][1] if self.segments else self.patch_shape
        return p1 * p2 * p3 * self.in_channels

    @property
    def mlp_hidden(self):
        return int(self.hidden_dim * self.mlp_ratio)


DS2 = ViTConfig()
DS3 = ViTConfig(shape=(45, 50, 18), patch_shape=(3, 10, 3))


def ds2(depth=6):
    return ViTConfig(depth=depth)


def ds3(depth=6):
    return ViTConfig(shape=(45, 50, 18), patch_shape=(3, 10, 3), depth=depth)


def _segmented(list_shape, list_patch_shape, condition_dim, depth):
    segs = tuple((tuple(s), tuple(p)) for s, p in zip(list_shape, list_patch_shape))
    return ViTConfig(shape=(sum(math.prod(s) for s in list_shape),), patch_shape=(), condition_dim=condition_dim, depth=depth, segments=segs)


def ds1_photons(depth=6):
    """configs/model/cfm/cfm_ds1_photons.yaml: 5 layers, 440 voxels -> 88 tokens of 5"""
    shapes = [(1, 8, 5), (1, 16, 10), (1, 19, 10), (1, 5, 5), (1, 5, 5)]
    return _segmented(shapes, [(1, 1, 5)] * 5, 6, depth)


def ds1_pions(depth=6):
    """configs/model/cfm/cfm_ds1_pions.yaml: 7 layers, 625 voxels -> 125 tokens of 5"""
    shapes = [(1, 8, 5), (1, 10, 10), (1, 10, 10), (1, 5, 5), (1, 15, 10), (1, 16, 10), (1, 10, 5)]
    return _segmented(shapes, [(1, 1, 5)] * 7, 8, depth)


def calogan(depth=6):
    """configs/model/cfm_calogan/cfm_eplus.yaml: 3 layers, 504 voxels -> 84 tokens of 6"""
    return _segmented([(1, 96, 3), (1, 12, 12), (1, 6, 12)], [(1, 6, 1), (1, 2, 3), (1, 2, 3)], 4, depth)


def calohad(depth=6):
    """configs/model/cfm_calohad/cfm_calohad.yaml: ECal + HCal, 45450 voxels -> 606 tokens of 75"""
    return _segmented([(10, 15, 15), (48, 30, 30)], [(5, 5, 3), (3, 5, 5)], 59, depth)


def lemurs(depth=6):
    """configs/model/cfm_lemurs/cfm_lemurs.yaml: the ds2 grid with 53 conditions"""
    return ViTConfig(condition_dim=53, depth=depth)


# --------------------------------------------------------------------------------------
# parameter inventory + deterministic fill (build-owned, platform independent)
# --------------------------------------------------------------------------------------


def param_shapes(cfg: ViTConfig) -> dict:
    """State-dict names/shapes of ``CaloChallengeCFM.net`` (nn/vit.py:76-132), in
    ``state_dict()`` order, without the ``net.`` prefix.  Buffers pos_z/y/x excluded."""
    D, P, K, F, M = cfg.hidden_dim, cfg.P, cfg.condition_dim, cfg.freq_dim, cfg.mlp_hidden
    s = {}
    s["pos_embed_freqs"] = (D // 6,)
    s["x_embedder.weight"] = (D, P)
    s["x_embedder.bias"] = (D,)
    s["c_embedder.0.weight"] = (D, K)
    s["c_embedder.0.bias"] = (D,)
    s["c_embedder.2.weight"] = (D, D)
    s["c_embedder.2.bias"] = (D,)
    s["t_embedder.mlp.0.weight"] = (D, F)
    s["t_embedder.mlp.0.bias"] = (D,)
    s["t_embedder.mlp.2.weight"] = (D, D)
    s["t_embedder.mlp.2.bias"] = (D,)
    for i in range(cfg.depth):
        p = f"blocks.{i}."
        s[p + "attn.qkv.weight"] = (3 * D, D)
        s[p + "attn.qkv.bias"] = (3 * D,)
        s[p + "attn.proj.weight"] = (D, D)
        s[p + "attn.proj.bias"] = (D,)
        s[p + "mlp.fc1.weight"] = (M, D)
        s[p + "mlp.fc1.bias"] = (M,)
        s[p + "mlp.fc2.weight"] = (D, M)
        s[p + "mlp.fc2.bias"] = (D,)
        s[p + "adaLN_modulation.1.weight"] = (6 * D, D)
        s[p + "adaLN_modulation.1.bias"] = (6 * D,)
    s["final_layer.linear.weight"] = (P, D)
    s["final_layer.linear.bias"] = (P,)
    s["final_layer.adaLN_modulation.1.weight"] = (2 * D, D)
    s["final_layer.adaLN_modulation.1.bias"] = (2 * D,)
    return s


def _name_seed(name: str) -> int:
    h = 1469598103934665603  # FNV-1a 64
    for ch in name.encode():
        h ^= ch
        h = (h * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def hash_uniform(name: str, n: int) -> np.ndarray:
    """n doubles in [-1, 1): splitmix64 of (FNV(name) + index). Pure integer arithmetic."""
    with np.errstate(over="ignore"):
        z = np.arange(n, dtype=np.uint64) + np.uint64(_name_seed(name))
        z = (z + np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    top = (z >> np.uint64(40)).astype(np.float64)  # 24 bits -> exactly representable in fp32
    return top / float(1 << 23) - 1.0


def golden_fill(cfg: ViTConfig, dtype=torch.float32, bias_scale=0.05) -> dict:
    """Deterministic non-degenerate parameters (also non-zero where the reference
    zero-initialises, nn/vit.py:174-183, so outputs/gradients are informative)."""
    out = {}
    for name, shp in param_shapes(cfg).items():
        n = int(np.prod(shp))
        u = hash_uniform(name, n)
        if name == "pos_embed_freqs":
            v = u * 1.5
        elif name.endswith("weight"):
            fan_out, fan_in = shp
            v = u * math.sqrt(6.0 / (fan_in + fan_out))  # xavier-uniform bound, nn/vit.py:168
        else:
            v = u * bias_scale
        out[name] = torch.from_numpy(v.reshape(shp)).to(dtype)
    return out


def fill_tensor(name: str, shape, dtype=torch.float32, bias_scale=0.05):
    """The golden_fill rule for one tensor of any name / shape (tensors that fine-tuning surgery adds or re-shapes)."""
    u = hash_uniform(name, int(np.prod(shape)))
    v = u * math.sqrt(6.0 / (shape[0] + shape[1])) if len(shape) == 2 else u * bias_scale
    return torch.from_numpy(v.reshape(shape)).to(dtype)


def reference_init(cfg: ViTConfig, generator: torch.Generator, dtype=torch.float32) -> dict:
    """nn/vit.py:164-183: xavier-uniform Linear weights, zero biases, zero adaLN-last /
    final adaLN / final linear; pos_embed_freqs ~ N(0,1) (nn/vit.py:86)."""
    out = {}
    for name, shp in param_shapes(cfg).items():
        if name == "pos_embed_freqs":
            out[name] = torch.randn(shp, generator=generator, dtype=dtype)
        elif name.endswith("bias"):
            out[name] = torch.zeros(shp, dtype=dtype)
        elif "adaLN_modulation" in name or name.startswith("final_layer.linear"):
            out[name] = torch.zeros(shp, dtype=dtype)
        else:
            fan_out, fan_in = shp
            a = math.sqrt(6.0 / (fan_in + fan_out))
            out[name] = (torch.rand(shp, generator=generator, dtype=dtype) * 2 - 1) * a
    return out


# --------------------------------------------------------------------------------------
# patching  (experiments/calochallenge/calochallenge_cfm/model.py:40-60)
# --------------------------------------------------------------------------------------


def _to_patches_grid(x, num_patches, patch_shape):
    B, C = x.shape[0], x.shape[1]
    l, a, r = num_patches
    p1, p2, p3 = patch_shape
    x = x.reshape(B, C, l, p1, a, p2, r, p3)
    x = x.permute(0, 2, 4, 6, 3, 5, 7, 1)  # b l a r p1 p2 p3 c
    return x.reshape(B, l * a * r, p1 * p2 * p3 * C)


def _from_patches_grid(z, num_patches, patch_shape, C):
    B = z.shape[0]
    l, a, r = num_patches
    p1, p2, p3 = patch_shape
    z = z.reshape(B, l, a, r, p1, p2, p3, C)
    z = z.permute(0, 7, 1, 4, 2, 5, 3, 6)  # b c l p1 a p2 r p3
    return z.reshape(B, C, l * p1, a * p2, r * p3)


def to_patches(x, cfg: ViTConfig):
    """(B, C, L*p1, A*p2, R*p3) -> (B, l*a*r, p1*p2*p3*C)   model.py:54-60.
    Multi-segment: (B, C, n_voxels) split at the segment edges, each piece patched on its own grid, tokens concatenated
    (model.py:163-173, calogan/model.py:77-87, calohadronic/model.py:76-86)."""
    if not cfg.segments:
        return _to_patches_grid(x, cfg.num_patches, cfg.patch_shape)
    B, C = x.shape[0], x.shape[1]
    toks, off = [], 0
    for (shape, patch), n in zip(cfg.segments, cfg.seg_num_patches):
        v = math.prod(shape)
        toks.append(_to_patches_grid(x[:, :, off : off + v].reshape(B, C, *shape), n, patch))
        off += v
    return torch.cat(toks, dim=1)


def from_patches(z, cfg: ViTConfig):
    """(B, l*a*r, p1*p2*p3*C) -> (B, C, L, A, R)   model.py:40-52; multi-segment: the inverse of the above, flat (B, C, n_voxels)
    (model.py:146-161)."""
    C = cfg.in_channels
    if not cfg.segments:
        return _from_patches_grid(z, cfg.num_patches, cfg.patch_shape, C)
    pieces, off = [], 0
    for (shape, patch), n in zip(cfg.segments, cfg.seg_num_patches):
        t = n[0] * n[1] * n[2]
        pieces.append(_from_patches_grid(z[:, off : off + t], n, patch, C).flatten(start_dim=2))
        off += t
    return torch.cat(pieces, dim=2)


# --------------------------------------------------------------------------------------
# network pieces  (nn/vit.py)
# --------------------------------------------------------------------------------------


def meshgrid_buffers(cfg: ViTConfig, dtype=torch.float32):
    """pos_z, pos_y, pos_x   nn/vit.py:137-154: the layer coordinate runs over ALL segments (arange(sum_l) / sum_l, sliced per
    segment), the angular / radial coordinates restart in every segment."""
    segs = cfg.seg_num_patches
    sum_l = sum(n[0] for n in segs)
    sum_lgrid = torch.arange(sum_l) / sum_l
    zs, ys, xs, l0 = [], [], [], 0
    for l, a, r in segs:
        z, y, x = torch.meshgrid(sum_lgrid[l0 : l0 + l], torch.arange(a) / a, torch.arange(r) / r, indexing="ij")
        zs.append(z.flatten()); ys.append(y.flatten()); xs.append(x.flatten())
        l0 += l
    return torch.cat(zs).to(dtype), torch.cat(ys).to(dtype), torch.cat(xs).to(dtype)


def pos_embedding(freqs, cfg: ViTConfig):
    """nn/vit.py:156-162 - x (radial) first, z (layer) last, sin before cos."""
    pz, py, px = meshgrid_buffers(cfg, freqs.dtype)
    w = freqs * 2 * math.pi
    z = pz[:, None] * w[None, :]
    y = py[:, None] * w[None, :]
    x = px[:, None] * w[None, :]
    return torch.cat((x.sin(), x.cos(), y.sin(), y.cos(), z.sin(), z.cos()), dim=1)


def timestep_embedding(t, dim=256, max_period=10000):
    """nn/vit.py:368-389 - cos first, then sin; t is (B,1)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=t.dtype) / half)
    args = t * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def linear(x, p, name):
    return x @ p[name + ".weight"].T + p[name + ".bias"]


def silu(x):
    return x * torch.sigmoid(x)


def gelu_tanh(x):
    """nn.GELU(approximate='tanh')   nn/vit.py:314-315"""
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x**3)))


def layernorm(x, eps=1e-6):
    """nn.LayerNorm(D, elementwise_affine=False, eps=1e-6)   nn/vit.py:309"""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps)


def modulate(x, shift, scale):
    """nn/vit.py:457-458"""
    return x * (1 + scale[:, None, :]) + shift[:, None, :]


def attention(u, p, prefix, H):
    """nn/vit.py:425-454 (SDPA branch, no mask, no dropout, Identity q/k norm)."""
    B, N, C = u.shape
    dh = C // H
    qkv = linear(u, p, prefix + "attn.qkv").reshape(B, N, 3, H, dh).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    s = (q @ k.transpose(-1, -2)) * dh**-0.5
    a = torch.softmax(s, dim=-1)
    o = (a @ v).transpose(1, 2).reshape(B, N, C)
    return linear(o, p, prefix + "attn.proj")


def dit_block(x, cond_silu, p, i, cfg: ViTConfig):
    """nn/vit.py:327-333"""
    pre = f"blocks.{i}."
    m = linear(cond_silu, p, pre + "adaLN_modulation.1")
    sh1, sc1, g1, sh2, sc2, g2 = m.chunk(6, dim=1)
    x = x + g1[:, None, :] * attention(modulate(layernorm(x), sh1, sc1), p, pre, cfg.num_heads)
    h = gelu_tanh(linear(modulate(layernorm(x), sh2, sc2), p, pre + "mlp.fc1"))
    x = x + g2[:, None, :] * linear(h, p, pre + "mlp.fc2")
    return x


def conditioning(p, t, c, cfg: ViTConfig):
    """t_embedder + c_embedder, summed   nn/vit.py:197-199, 361-365, 77-81"""
    te = timestep_embedding(t, cfg.freq_dim)
    te = linear(silu(linear(te, p, "t_embedder.mlp.0")), p, "t_embedder.mlp.2")
    if "c_embedder.2.0.weight" in p:  # fine-tuning condition mapper: Sequential(Linear, SiLU, c_embedder)   experiment_finetuning.py:106-119
        c = silu(linear(c, p, "c_embedder.0"))
        ce = linear(silu(linear(c, p, "c_embedder.2.0")), p, "c_embedder.2.2")
    else:
        ce = linear(silu(linear(c, p, "c_embedder.0")), p, "c_embedder.2")
    return te + ce


def vit_forward(p, xp, t, c, cfg: ViTConfig):
    """ViT.forward   nn/vit.py:185-206.  xp (B,T,P), t (B,1), c (B,K) -> (B,T,P)."""
    if "x_embedder.0.weight" in p:  # fine-tuning embedding mapper: Sequential(Linear, SiLU, x_embedder)   experiment_finetuning.py:80-91
        x = linear(silu(linear(xp, p, "x_embedder.0")), p, "x_embedder.2")
    else:
        x = linear(xp, p, "x_embedder")
    x = x + pos_embedding(p["pos_embed_freqs"], cfg)
    cs = silu(conditioning(p, t, c, cfg))
    for i in range(cfg.depth):
        x = dit_block(x, cs, p, i, cfg)
    m = linear(cs, p, "final_layer.adaLN_modulation.1")  # nn/vit.py:347-351
    shift, scale = m.chunk(2, dim=1)
    return linear(modulate(layernorm(x), shift, scale), p, "final_layer.linear")


def cfm_forward(p, x, t, c, cfg: ViTConfig):
    """CaloChallengeCFM.forward   calochallenge_cfm/model.py:62-66"""
    return from_patches(vit_forward(p, to_patches(x, cfg), t, c, cfg), cfg)


def cfm_loss(p, x1, c, t, x0, cfg: ViTConfig):
    """CFM._batch_loss with t and x0 injected   models/base_model.py:203-218,
    linear_trajectory models/trajectories.py:5-8.  t is (B,1,1,1,1) ((B,1,1) for the flat multi-segment samples)."""
    x_t = (1 - t) * x0 + t * x1
    x_t_dot = x1 - x0
    v = cfm_forward(p, x_t, t.view(-1, 1), c, cfg)
    return ((v - x_t_dot) ** 2).mean(), v


# --------------------------------------------------------------------------------------
# fixed-grid ODE sampler (torchdiffeq.odeint semantics; un-vendored, unpinned dependency)
# --------------------------------------------------------------------------------------


def fixed_grid(t0: float, t1: float, step: float, dtype=torch.float32):
    """torchdiffeq FixedGridODESolver grid from options.step_size: ceil((t1-t0)/h + 1) nodes
    at t0 + k*h, last node clamped to t1.  Arithmetic in ``dtype`` like the solver does."""
    a = torch.tensor(t0, dtype=dtype)
    b = torch.tensor(t1, dtype=dtype)
    n = int(torch.ceil((b - a) / step + 1).item())
    g = torch.arange(0, n, dtype=dtype) * step + a
    g[-1] = b
    return g


def ode_step(f, method, t0, t1, y):
    dt = t1 - t0
    if method == "euler":
        return y + dt * f(t0, y)
    if method == "midpoint":
        return y + dt * f(t0 + dt * 0.5, y + f(t0, y) * (dt * 0.5))
    if method in ("heun", "heun2"):
        k1 = f(t0, y)
        k2 = f(t1, y + dt * k1)
        return y + (k1 + k2) * (dt * 0.5)
    if method == "rk4":  # torchdiffeq 'rk4' = 3/8 rule (rk4_alt_step_func)
        k1 = f(t0, y)
        k2 = f(t0 + dt / 3, y + dt * k1 / 3)
        k3 = f(t0 + dt * 2 / 3, y + dt * (k2 - k1 / 3))
        k4 = f(t1, y + dt * (k1 - k2 + k3))
        return y + (k1 + 3 * (k2 + k3) + k4) * (dt * 0.125)
    raise ValueError(method)


@torch.no_grad()
def sample(p, c, x_T, cfg: ViTConfig, method="rk4", step_size=0.05):
    """CaloChallengeCFM.sample_batch with x_T injected   calochallenge_cfm/model.py:68-94"""
    B = c.shape[0]

    def f(t, x):
        return cfm_forward(p, x, t.repeat((B, 1)), c, cfg)

    grid = fixed_grid(0.0, 1.0, step_size, x_T.dtype)
    y = x_T
    for k in range(len(grid) - 1):
        y = ode_step(f, method, grid[k], grid[k + 1], y)
    return y


# --------------------------------------------------------------------------------------
# update step  (experiments/base_experiment.py:555-597, configs/training/default.yaml)
# --------------------------------------------------------------------------------------


@dataclass
class AdamWState:
    lr: float = 1e-4
    betas: tuple = (0.9, 0.999)
    eps: float = 1e-8
    weight_decay: float = 0.1
    clip: float = 1000.0
    iterations: int = 50000  # CosineAnnealingLR T_max, eta_min 0
    step: int = 0
    m: dict = field(default_factory=dict)
    v: dict = field(default_factory=dict)

    def lr_at(self, k):
        """CosineAnnealingLR closed form (eta_min = 0) after k scheduler steps."""
        return self.lr * 0.5 * (1 + math.cos(math.pi * k / self.iterations))


def clip_grad_norm(grads: dict, max_norm: float):
    """torch.nn.utils.clip_grad_norm_: total L2 norm, scale by min(1, max/(norm+1e-6))."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).to(next(iter(grads.values())).dtype)
    coef = min(1.0, max_norm / (float(total) + 1e-6))
    if coef < 1.0:
        for g in grads.values():
            g.mul_(coef)
    return float(total)


def adamw_update(p: dict, grads: dict, st: AdamWState):
    """torch.optim.AdamW single param group, applied to every parameter (base_experiment.py:331-346)."""
    st.step += 1
    lr = st.lr_at(st.step - 1)
    b1, b2 = st.betas
    bc1 = 1 - b1**st.step
    bc2 = 1 - b2**st.step
    for k in p:
        g = grads[k]
        if k not in st.m:
            st.m[k] = torch.zeros_like(p[k])
            st.v[k] = torch.zeros_like(p[k])
        p[k].mul_(1 - lr * st.weight_decay)
        st.m[k].mul_(b1).add_(g, alpha=1 - b1)
        st.v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (st.v[k].sqrt() / math.sqrt(bc2)).add_(st.eps)
        p[k].addcdiv_(st.m[k], denom, value=-lr / bc1)


def loss_and_grads(p, x1, c, t, x0, cfg):
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    loss, v = cfm_loss(leaves, x1, c, t, x0, cfg)
    loss.backward()
    grads = {k: leaves[k].grad for k in leaves}
    return loss.detach(), v.detach(), grads


def train_step(p, st: AdamWState, x1, c, t, x0, cfg):
    """One BaseExperiment._step: loss, backward, clip(1000), AdamW, cosine LR."""
    loss, _, grads = loss_and_grads(p, x1, c, t, x0, cfg)
    gnorm = clip_grad_norm(grads, st.clip)
    if not math.isfinite(gnorm):
        raise RuntimeError("non-finite gradient norm")  # error_if_nonfinite=True
    with torch.no_grad():
        adamw_update(p, grads, st)
    return float(loss), gnorm


# --------------------------------------------------------------------------------------
# synthetic CaloChallenge-shaped batches
# --------------------------------------------------------------------------------------


def synthetic_batch(cfg: ViTConfig, B: int, seed: int = 0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((B, cfg.in_channels, *cfg.shape), generator=g, dtype=dtype)
    c = torch.cat(
        [torch.randn((B, cfg.condition_dim - 1), generator=g, dtype=dtype), torch.rand((B, 1), generator=g, dtype=dtype)],
        dim=1,
    )
    return x, c, g


def synthetic_noise(cfg: ViTConfig, B: int, g: torch.Generator, dtype=torch.float32):
    t = torch.rand((B, 1) + (1,) * len(cfg.shape), generator=g, dtype=dtype)  # broadcastable against (B, C, *shape)
    x0 = torch.randn((B, cfg.in_channels, *cfg.shape), generator=g, dtype=dtype)
    return t, x0
