"""CPU oracle of the ENERGY-model network of luigifvr/vit4hep: the `ParallelTransformer` velocity field that the
reference samples the 45 layer-energy ratios from before the shape model runs (experiments/calochallenge/experiment.py:225-247,
configs/model/cfm/cfm_ds2_energy.yaml).

TEST INFRASTRUCTURE ONLY - imported by tests/, oracle/make_golden.py and the benchmark tools' CPU-baseline legs, never by the
product path (vit4hep_amd/).

Restates, as plain functional PyTorch on the CPU:
  * ParallelTransformer.compute_embedding / forward, `embeds: true` branch with a condition     nn/cfm/transformer_cfm.py:76-119
  * GaussianFourierProjection                                                                    nn/cfm/transformer_cfm.py:153-165
  * torch.nn.Transformer(batch_first=True, norm_first=False, activation relu, dropout 0, layer_norm_eps 1e-5) as the reference
    constructs it (nn/cfm/transformer_cfm.py:55-64).  torch is third-party: requirements.txt pins torch==2.7.0, this image has
    2.10; the published semantics restated here are post-norm encoder/decoder layers, a final LayerNorm on each stack,
    MultiheadAttention with packed in_proj (rows q | k | v), heads split h-major, scores scaled by head_dim^-0.5.
Pinned by tests/golden/energy_*.npz, which oracle/make_golden.py produces by running the reference's own class
(`nn.cfm.transformer_cfm.ParallelTransformer` inside `models.base_model.CFM`).
"""

from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch

from .vit_cfm_oracle import fixed_grid, hash_uniform, ode_step


@dataclass(frozen=True)
class EnergyConfig:
    """Keys of ParallelTransformer's `param` mapping (nn/cfm/transformer_cfm.py:21-37); defaults = cfm_ds2_energy.yaml."""

    dims_in: int = 45
    dims_c: int = 1
    dim_embedding: int = 64
    nhead: int = 4
    num_encoder_layers: int = 4
    num_decoder_layers: int = 4
    dim_feedforward: int = 512
    encode_t_scale: float = 30.0
    encode_t_dim: int = 64

    def __post_init__(self):
        # [time embedding | x embedding] must be d_model wide for nn.Transformer (transformer_cfm.py:45,87-89)
        assert self.encode_t_dim == self.dim_embedding, "embeds=True needs encode_t_dim == dim_embedding"
        assert (2 * self.dim_embedding) % self.nhead == 0

    @property
    def d_model(self):  # embeds: true  ->  2 * dim_embedding  (transformer_cfm.py:45)
        return 2 * self.dim_embedding


def param_shapes(cfg: EnergyConfig) -> dict:
    """Learnable + frozen tensors in the reference module's named_parameters() order (shared `layer` / `layers.0` listed once,
    under the name that comes first: `layer`)."""
    d, e, ff, te = cfg.d_model, cfg.dim_embedding, cfg.dim_feedforward, cfg.encode_t_dim
    s = {
        "time_embed.0.W": (te // 2,),  # GaussianFourierProjection, requires_grad False
        "time_embed.1.weight": (te, te), "time_embed.1.bias": (te,),
        "x_embed.weight": (e, 1), "x_embed.bias": (e,),
        "c_embed.weight": (2 * e, 1), "c_embed.bias": (2 * e,),
        "pos_embed_x.weight": (cfg.dims_in, e),
        "pos_embed_c.weight": (cfg.dims_c, 2 * e),
        "layer.weight": (ff, 3 * e), "layer.bias": (ff,),
    }

    def mha(prefix):
        s[prefix + ".in_proj_weight"] = (3 * d, d)
        s[prefix + ".in_proj_bias"] = (3 * d,)
        s[prefix + ".out_proj.weight"] = (d, d)
        s[prefix + ".out_proj.bias"] = (d,)

    def ffn_norms(prefix, nnorm):
        s[prefix + ".linear1.weight"] = (ff, d); s[prefix + ".linear1.bias"] = (ff,)
        s[prefix + ".linear2.weight"] = (d, ff); s[prefix + ".linear2.bias"] = (d,)
        for k in range(1, nnorm + 1):
            s[prefix + f".norm{k}.weight"] = (d,); s[prefix + f".norm{k}.bias"] = (d,)

    for i in range(cfg.num_encoder_layers):
        p = f"transformer.encoder.layers.{i}"
        mha(p + ".self_attn")
        ffn_norms(p, 2)
    s["transformer.encoder.norm.weight"] = (d,); s["transformer.encoder.norm.bias"] = (d,)
    for i in range(cfg.num_decoder_layers):
        p = f"transformer.decoder.layers.{i}"
        mha(p + ".self_attn")
        mha(p + ".multihead_attn")
        ffn_norms(p, 3)
    s["transformer.decoder.norm.weight"] = (d,); s["transformer.decoder.norm.bias"] = (d,)
    s["layers.2.weight"] = (1, ff); s["layers.2.bias"] = (1,)
    return s


def golden_fill(cfg: EnergyConfig, dtype=torch.float32) -> dict:
    """Deterministic, platform-independent, non-degenerate parameters keyed by name."""
    out = {}
    for name, shp in param_shapes(cfg).items():
        u = hash_uniform("energy/" + name, int(np.prod(shp)))
        if name == "time_embed.0.W":
            v = u * cfg.encode_t_scale
        elif "norm" in name and name.endswith("weight"):
            v = 1.0 + 0.2 * u
        elif name.startswith("pos_embed"):
            v = u
        elif name.endswith("weight") and len(shp) == 2:
            v = u * math.sqrt(6.0 / (shp[0] + shp[1]))
        else:
            v = u * 0.1
        out[name] = torch.from_numpy(np.asarray(v).reshape(shp)).to(dtype)
    return out


def _linear(x, w, b):
    return x @ w.T + b


def _layer_norm(x, w, b, eps=1e-5):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def _mha(p, prefix, q_in, kv_in, H):
    """torch.nn.MultiheadAttention forward, batch_first, no masks, dropout 0."""
    d = q_in.shape[-1]
    dh = d // H
    W, bvec = p[prefix + ".in_proj_weight"], p[prefix + ".in_proj_bias"]
    q = _linear(q_in, W[:d], bvec[:d])
    k = _linear(kv_in, W[d : 2 * d], bvec[d : 2 * d])
    v = _linear(kv_in, W[2 * d :], bvec[2 * d :])
    B, L, S = q.shape[0], q.shape[1], k.shape[1]
    q = q.reshape(B, L, H, dh).transpose(1, 2)
    k = k.reshape(B, S, H, dh).transpose(1, 2)
    v = v.reshape(B, S, H, dh).transpose(1, 2)
    a = torch.softmax((q @ k.transpose(-1, -2)) * dh**-0.5, dim=-1)
    o = (a @ v).transpose(1, 2).reshape(B, L, d)
    return _linear(o, p[prefix + ".out_proj.weight"], p[prefix + ".out_proj.bias"])


def _ffn(p, prefix, x):
    return _linear(torch.relu(_linear(x, p[prefix + ".linear1.weight"], p[prefix + ".linear1.bias"])), p[prefix + ".linear2.weight"], p[prefix + ".linear2.bias"])


def time_embed(p, t):
    """GaussianFourierProjection + Linear   transformer_cfm.py:39-42,161-165 ; t is (B, 1)"""
    proj = t * p["time_embed.0.W"] * 2 * math.pi
    return _linear(torch.cat([torch.sin(proj), torch.cos(proj)], dim=1), p["time_embed.1.weight"], p["time_embed.1.bias"])


def energy_forward(p, x, t, c, cfg: EnergyConfig):
    """ParallelTransformer.forward(x (B, dims_in), t (B, 1), condition (B, dims_c)) -> (B, dims_in)   transformer_cfm.py:101-119"""
    H = cfg.nhead
    temb = time_embed(p, t)  # (B, te)
    # compute_embedding(condition, dims_c): c_embed + positional embedding                      transformer_cfm.py:91-94
    src = c.unsqueeze(-1) * p["c_embed.weight"][:, 0] + p["c_embed.bias"] + p["pos_embed_c.weight"][None]
    # compute_embedding(x, dims_in, t): [time embedding | x_embed + positional embedding]       transformer_cfm.py:84-90
    xe = x.unsqueeze(-1) * p["x_embed.weight"][:, 0] + p["x_embed.bias"] + p["pos_embed_x.weight"][None]
    tgt = torch.cat([temb[:, None, :].expand(-1, cfg.dims_in, -1), xe], dim=-1)
    # nn.Transformer: post-norm encoder stack + final norm
    m = src
    for i in range(cfg.num_encoder_layers):
        pre = f"transformer.encoder.layers.{i}"
        m = _layer_norm(m + _mha(p, pre + ".self_attn", m, m, H), p[pre + ".norm1.weight"], p[pre + ".norm1.bias"])
        m = _layer_norm(m + _ffn(p, pre, m), p[pre + ".norm2.weight"], p[pre + ".norm2.bias"])
    m = _layer_norm(m, p["transformer.encoder.norm.weight"], p["transformer.encoder.norm.bias"])
    h = tgt
    for i in range(cfg.num_decoder_layers):
        pre = f"transformer.decoder.layers.{i}"
        h = _layer_norm(h + _mha(p, pre + ".self_attn", h, h, H), p[pre + ".norm1.weight"], p[pre + ".norm1.bias"])
        h = _layer_norm(h + _mha(p, pre + ".multihead_attn", h, m, H), p[pre + ".norm2.weight"], p[pre + ".norm2.bias"])
        h = _layer_norm(h + _ffn(p, pre, h), p[pre + ".norm3.weight"], p[pre + ".norm3.bias"])
    h = _layer_norm(h, p["transformer.decoder.norm.weight"], p["transformer.decoder.norm.bias"])
    # head: Linear(te + d_model -> ff) on [t | embedding], SiLU, Linear(ff -> 1)                 transformer_cfm.py:66-70,114-119
    z = torch.cat([temb[:, None, :].expand(-1, h.shape[1], -1), h], dim=-1)
    z = _linear(z, p["layer.weight"], p["layer.bias"])
    z = z * torch.sigmoid(z)
    return _linear(z, p["layers.2.weight"], p["layers.2.bias"]).squeeze(-1)


@torch.no_grad()
def energy_sample(p, c, x_T, cfg: EnergyConfig, method="rk4", step_size=0.05):
    """CFM.sample_batch with x_T injected   models/base_model.py:220-244 (fixed-grid solver as in vit_cfm_oracle)."""
    B = c.shape[0]

    def f(t, x):
        return energy_forward(p, x, t.repeat((B, 1)), c, cfg)

    grid = fixed_grid(0.0, 1.0, step_size, x_T.dtype)
    y = x_T
    for k in range(len(grid) - 1):
        y = ode_step(f, method, grid[k], grid[k + 1], y)
    return y


def fwd_flops_per_sample(cfg: EnergyConfig) -> float:
    """2*m*n*k of every contraction the reference's module performs per sample and evaluation."""
    d, ff, L, S, te = cfg.d_model, cfg.dim_feedforward, cfg.dims_in, cfg.dims_c, cfg.encode_t_dim
    mha_self = lambda n: n * (2 * d * 3 * d + 2 * d * d) + 4 * n * n * d
    enc = cfg.num_encoder_layers * (mha_self(S) + S * 4 * d * ff)
    dec = cfg.num_decoder_layers * (mha_self(L) + (L * 2 * d * d + S * 2 * d * 2 * d + L * 2 * d * d + 4 * L * S * d) + L * 4 * d * ff)
    head = L * (2 * (te + d) * ff + 2 * ff) + 2 * te * te
    return float(enc + dec + head)
