"""MI355X host mirror of the reference's ``nn.vit`` (reference nn/vit.py).

Same constructor (``ViT(param)`` with the key-by-key defaults of nn/vit.py:52-73), same attribute
names and state-dict keys (``x_embedder``, ``c_embedder.{0,2}``, ``t_embedder.mlp.{0,2}``,
``pos_embed_freqs``, ``pos_{z,y,x}``, ``blocks.i.{attn.qkv,attn.proj,mlp.fc1,mlp.fc2,adaLN_modulation.1}``,
``final_layer.{linear,adaLN_modulation.1}``), same initialisation (nn/vit.py:164-183) - but the
sub-modules are parameter containers only: ``ViT.forward`` runs the whole network (forward and,
through one autograd node, backward) inside libvit4hep_hip.so.  There is no PyTorch fallback.

Extra, optional ``param`` keys (ignored by the reference, which drops unknown keys):
  ``amd_mode``: "f32" (default; exact-f32 MFMA, matches the reference within 1e-4) or "bf16"
                (bf16 MFMA, f32 accumulate: throughput mode).  Env VIT4HEP_AMD_MODE overrides.
"""

from __future__ import annotations

import math
import os

import torch
import torch.nn as nn

from .. import _lib

_DEFAULTS = {  # nn/vit.py:52-70
    "dim": 3,
    "condition_dim": 46,
    "hidden_dim": 180,
    "out_channels": 1,
    "depth": 2,
    "num_heads": 4,
    "mlp_ratio": 2.0,
    "attn_drop": 0.0,
    "proj_drop": 0.0,
    "pos_embedding_coords": "cartesian",
    "temperature": 10000,
    "learn_pos_embed": True,
    "causal_attn": False,
    "checkpoint_grads": False,
    "patch_dim": 12,
    "num_patches": [[15, 4, 9]],
    "use_torch_sdpa": True,
}


class _Mlp(nn.Module):
    """Parameter container with timm Mlp's names (fc1, fc2); reference nn/vit.py:317-322."""

    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU(approximate="tanh")
        self.fc2 = nn.Linear(hidden, dim)


class Attention(nn.Module):
    """Parameter container (qkv, proj); reference nn/vit.py:397-423."""

    def __init__(self, dim, num_heads=8, qkv_bias=True, **_):
        super().__init__()
        assert dim % num_heads == 0, "dim should be divisible by num_heads"
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.scale = self.head_dim**-0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)


class DiTBlock(nn.Module):
    """Parameter container of one adaLN-Zero block; reference nn/vit.py:302-325."""

    def __init__(self, hidden_size, num_heads, mlp_ratio=4.0, **_):
        super().__init__()
        self.attn = Attention(hidden_size, num_heads=num_heads, qkv_bias=True)
        self.mlp = _Mlp(hidden_size, int(hidden_size * mlp_ratio))
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(hidden_size, 6 * hidden_size, bias=True))


class FinalLayer(nn.Module):
    """Parameter container of the output head; reference nn/vit.py:336-345."""

    def __init__(self, hidden_dim, patch_dim, out_channels=1, x_out=1):
        super().__init__()
        self.linear = nn.Linear(hidden_dim, out_channels * x_out * patch_dim)
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(hidden_dim, 2 * hidden_dim))


class TimestepEmbedder(nn.Module):
    """Parameter container (mlp.0, mlp.2); reference nn/vit.py:354-366."""

    def __init__(self, hidden_size, frequency_embedding_size=256):
        super().__init__()
        self.mlp = nn.Sequential(nn.Linear(frequency_embedding_size, hidden_size), nn.SiLU(), nn.Linear(hidden_size, hidden_size))
        self.frequency_embedding_size = frequency_embedding_size


class ViT(nn.Module):
    def __init__(self, param):
        super().__init__()
        for k, default in _DEFAULTS.items():
            setattr(self, k, param[k] if k in param else default)
        self.amd_mode = os.environ.get("VIT4HEP_AMD_MODE") or (param["amd_mode"] if "amd_mode" in param else "f32")
        if self.amd_mode not in _lib.MODES:
            raise ValueError(f"amd_mode must be one of {sorted(_lib.MODES)}, got {self.amd_mode!r}")

        # what the HIP path implements = what every shape-CFM config uses
        if not self.learn_pos_embed:
            raise NotImplementedError("vit4hep_amd: learn_pos_embed=False (fixed sincos table) is not on the shape-CFM path")
        if self.causal_attn:
            raise NotImplementedError("vit4hep_amd: causal_attn=True is not on the shape-CFM path")
        if float(self.attn_drop) != 0.0 or float(self.proj_drop) != 0.0:
            raise NotImplementedError("vit4hep_amd: dropout > 0 is not on the shape-CFM path")
        self.num_patches = [list(int(v) for v in seg) for seg in self.num_patches]
        if self.dim != 3 or len(self.num_patches) != 1 or len(self.num_patches[0]) != 3:
            raise NotImplementedError("vit4hep_amd: only a single 3-D patch segment [[l, a, r]] is supported (ds2/ds3 shape models)")

        D = int(self.hidden_dim)
        self.x_embedder = nn.Linear(int(self.patch_dim), D)
        self.c_embedder = nn.Sequential(nn.Linear(int(self.condition_dim), D), nn.SiLU(), nn.Linear(D, D))
        self.t_embedder = TimestepEmbedder(D)
        self.pos_embed_freqs = nn.Parameter(torch.randn(D // 6))
        pos_z, pos_y, pos_x = self.create_meshgrid()
        self.register_buffer("pos_z", pos_z)
        self.register_buffer("pos_y", pos_y)
        self.register_buffer("pos_x", pos_x)
        self.blocks = nn.ModuleList([DiTBlock(D, int(self.num_heads), mlp_ratio=self.mlp_ratio) for _ in range(int(self.depth))])
        self.final_layer = FinalLayer(D, int(self.patch_dim), int(self.out_channels), x_out=1)
        self.initialize_weights()

        self._geometry = None  # (shape, patch_shape), set by CaloChallengeCFM; a consistent default otherwise
        self._plan = None
        self._infer_ws = {}

    # ------------------------------------------------------------------ reference-visible helpers
    def create_meshgrid(self):
        """Buffers pos_z/pos_y/pos_x on the patch grid (reference nn/vit.py:137-154), single segment."""
        l, a, r = self.num_patches[0]
        z, y, x = torch.meshgrid(torch.arange(l) / l, torch.arange(a) / a, torch.arange(r) / r, indexing="ij")
        return z.flatten().clone(), y.flatten().clone(), x.flatten().clone()  # real storage (meshgrid returns expanded views)

    def learnable_pos_embedding(self):
        """(T, D) table from pos_embed_freqs, computed by the HIP kernel (reference nn/vit.py:156-162)."""
        plan = self._get_plan()
        freqs = _lib.require_cuda(self.pos_embed_freqs.detach(), "pos_embed_freqs")
        pe = torch.empty((self.num_tokens, int(self.hidden_dim)), dtype=torch.float32, device=freqs.device)
        _lib.check(_lib.load().v4h_op_pos_embed(plan.handle, _lib.ptr(freqs), _lib.ptr(pe), _lib.stream_ptr(freqs.device)), "v4h_op_pos_embed")
        return pe

    def initialize_weights(self):
        """xavier-uniform Linear weights, zero biases, zero adaLN-last / final layer (reference nn/vit.py:164-183)."""
        for mod in self.modules():
            if isinstance(mod, nn.Linear):
                nn.init.xavier_uniform_(mod.weight)
                nn.init.zeros_(mod.bias)
        for blk in self.blocks:
            nn.init.zeros_(blk.adaLN_modulation[-1].weight)
            nn.init.zeros_(blk.adaLN_modulation[-1].bias)
        for lin in (self.final_layer.adaLN_modulation[-1], self.final_layer.linear):
            nn.init.zeros_(lin.weight)
            nn.init.zeros_(lin.bias)

    # ------------------------------------------------------------------ HIP plumbing
    @property
    def num_tokens(self):
        l, a, r = self.num_patches[0]
        return l * a * r

    def set_geometry(self, shape, patch_shape):
        shape, patch_shape = [int(s) for s in shape], [int(s) for s in patch_shape]
        if [s // p for s, p in zip(shape, patch_shape)] != self.num_patches[0] or math.prod(patch_shape) * int(self.out_channels) != int(self.patch_dim):
            raise ValueError(f"geometry shape={shape} patch_shape={patch_shape} does not match num_patches={self.num_patches} patch_dim={self.patch_dim}")
        if self._geometry != (shape, patch_shape):
            self._geometry, self._plan, self._infer_ws = (shape, patch_shape), None, {}

    def geometry(self):
        if self._geometry is None:  # any geometry whose patching is the identity on (T, P) token rows will do
            l, a, r = self.num_patches[0]
            P = int(self.patch_dim)
            self._geometry = ([l * P, a, r], [P, 1, 1])
        return self._geometry

    def _get_plan(self):
        if self._plan is None:
            shape, patch_shape = self.geometry()
            self._plan = _lib.Plan(shape, patch_shape, self.condition_dim, self.hidden_dim, self.depth, self.num_heads,
                                   int(self.hidden_dim * self.mlp_ratio), self.t_embedder.frequency_embedding_size, self.amd_mode)
            got = [tuple(p.shape) for p in self.parameter_list()]
            if got != self._plan.shapes:
                raise RuntimeError(f"parameter inventory differs from the library's: {got} vs {self._plan.shapes}")
        return self._plan

    def parameter_list(self):
        """Learnable tensors in the state_dict() order the C ABI expects (include/vit4hep_hip.h)."""
        ps = [self.pos_embed_freqs, self.x_embedder.weight, self.x_embedder.bias,
              self.c_embedder[0].weight, self.c_embedder[0].bias, self.c_embedder[2].weight, self.c_embedder[2].bias,
              self.t_embedder.mlp[0].weight, self.t_embedder.mlp[0].bias, self.t_embedder.mlp[2].weight, self.t_embedder.mlp[2].bias]
        for b in self.blocks:
            ps += [b.attn.qkv.weight, b.attn.qkv.bias, b.attn.proj.weight, b.attn.proj.bias, b.mlp.fc1.weight, b.mlp.fc1.bias,
                   b.mlp.fc2.weight, b.mlp.fc2.bias, b.adaLN_modulation[1].weight, b.adaLN_modulation[1].bias]
        ps += [self.final_layer.linear.weight, self.final_layer.linear.bias,
               self.final_layer.adaLN_modulation[1].weight, self.final_layer.adaLN_modulation[1].bias]
        return ps

    def inference_workspace(self, B, device):
        key = (int(B), str(device))
        ws = self._infer_ws.get(key)
        if ws is None:
            self._infer_ws = {key: torch.empty(self._get_plan().workspace_bytes(B, False), dtype=torch.uint8, device=device)}
            ws = self._infer_ws[key]
        return ws

    # ------------------------------------------------------------------ forward
    def forward(self, x, t, c):
        """x: patch tokens (B, T, P) as in the reference (nn/vit.py:185-206) -> (B, T, P);
        or voxels (B, 1, L, A, R) -> voxels, which skips two layout passes (used by CaloChallengeCFM)."""
        from ..autograd import vit_apply

        if x.dim() == 3:
            return vit_apply(self, x, t, c, patches_io=True)
        return vit_apply(self, x, t, c, patches_io=False)


def modulate(x, shift, scale):
    """Kept for API compatibility (reference nn/vit.py:457-458); on the HIP path it is fused into the LayerNorm kernel."""
    return x * (1 + scale.unsqueeze(1)) + shift.unsqueeze(1)


def get_sincos_pos_embed(*args, **kwargs):
    raise NotImplementedError("vit4hep_amd: fixed sincos tables (learn_pos_embed=False / fine-tuning re-meshing) are outside the shape-CFM hot path")
