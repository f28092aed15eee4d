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
  ``amd_residual``: storage of the residual stream x and of its gradient inside the library's workspace: "auto" (default: "bf16" in bf16 mode
                where the 16-byte LayerNorm kernels serve the width, else "f32"), "f32", "bf16", "x_bf16" (x only), "dx_bf16" (gradient only).
                Arithmetic on them stays f32 in registers; f32 mode always stores f32.  Env VIT4HEP_AMD_RESIDUAL overrides.
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
    # patch_dim / condition_dim follow the LIVE embedders: fine-tuning re-shapes them after construction (weights interpolated to
    # another input width via `.weight.data = ...`, modules replaced; reference experiment_finetuning.py:75-171), and nn.Linear's own
    # `in_features` attribute is not updated by that - the weight's shape is the truth.
    @property
    def patch_dim(self):
        xe = self._modules.get("x_embedder")
        if isinstance(xe, nn.Linear):
            return int(xe.weight.shape[1])
        if self._is_mapper(xe):  # Sequential(mapper Linear, SiLU, x_embedder): the network sees the mapper's input width
            return int(xe[0].weight.shape[1])
        return self.__dict__.get("_patch_dim0")

    @staticmethod
    def _is_mapper(xe):
        return isinstance(xe, nn.Sequential) and len(xe) == 3 and isinstance(xe[0], nn.Linear) and isinstance(xe[1], nn.SiLU) and isinstance(xe[2], nn.Linear)

    @staticmethod
    def _is_c_embedder(ce):
        return isinstance(ce, nn.Sequential) and len(ce) == 3 and isinstance(ce[0], nn.Linear) and isinstance(ce[1], nn.SiLU) and isinstance(ce[2], nn.Linear)

    @classmethod
    def _is_c_mapper(cls, ce):
        """Sequential(mapper Linear, SiLU, <the backbone's c_embedder>): fine-tuning's `map_c_embedding` (experiment_finetuning.py:106-119)."""
        return isinstance(ce, nn.Sequential) and len(ce) == 3 and isinstance(ce[0], nn.Linear) and isinstance(ce[1], nn.SiLU) and cls._is_c_embedder(ce[2])

    def c_embed_in(self):
        """Input width of the inner c_embedder behind a fine-tuning condition mapper, else 0."""
        ce = self._modules.get("c_embedder")
        return int(ce[2][0].weight.shape[1]) if self._is_c_mapper(ce) else 0

    def x_embed_in(self):
        """Input width of the inner x_embedder Linear behind a fine-tuning embedding mapper, else 0."""
        xe = self._modules.get("x_embedder")
        return int(xe[2].weight.shape[1]) if self._is_mapper(xe) else 0

    @patch_dim.setter
    def patch_dim(self, v):
        self.__dict__["_patch_dim0"] = int(v)

    @property
    def condition_dim(self):
        ce = self._modules.get("c_embedder")
        first = ce[0] if isinstance(ce, nn.Sequential) and len(ce) > 0 else None
        return int(first.weight.shape[1]) if isinstance(first, nn.Linear) else self.__dict__.get("_condition_dim0")

    @condition_dim.setter
    def condition_dim(self, v):
        self.__dict__["_condition_dim0"] = int(v)

    def __init__(self, param):
        super().__init__()
        for k, default in _DEFAULTS.items():
            setattr(self, k, param[k] if k in param else default)
        self.amd_mode = os.environ.get("VIT4HEP_AMD_MODE") or (param["amd_mode"] if "amd_mode" in param else "f32")
        if self.amd_mode not in _lib.MODES:
            raise ValueError(f"amd_mode must be one of {sorted(_lib.MODES)}, got {self.amd_mode!r}")
        self.amd_residual = os.environ.get("VIT4HEP_AMD_RESIDUAL") or (param["amd_residual"] if "amd_residual" in param else "auto")
        if self.amd_residual != "auto" and self.amd_residual not in _lib.RESIDUAL:
            raise ValueError(f"amd_residual must be 'auto' or one of {sorted(_lib.RESIDUAL)}, got {self.amd_residual!r}")

        # what the HIP path implements = what every shape-CFM config uses
        if not self.learn_pos_embed:
            raise NotImplementedError("vit4hep_amd: learn_pos_embed=False (fixed sincos table) is not on the shape-CFM path")
        if self.causal_attn:
            raise NotImplementedError("vit4hep_amd: causal_attn=True is not on the shape-CFM path")
        if float(self.attn_drop) != 0.0 or float(self.proj_drop) != 0.0:
            raise NotImplementedError("vit4hep_amd: dropout > 0 is not on the shape-CFM path")
        if self.checkpoint_grads:
            # nn/vit.py:201-202 wraps every block in torch.utils.checkpoint: a memory / recompute trade-off with no numerical effect.  The library keeps
            # every activation (the whole training workspace is 2.4 GB of 288 GB at ds2 bs 128) and has no recompute path: say so instead of ignoring it.
            raise NotImplementedError("vit4hep_amd: checkpoint_grads=True (block recompute) is not built - the HIP path keeps all activations "
                                      "(2.4 GB at ds2 bs 128); set checkpoint_grads: false, the results are identical")
        self.num_patches = [list(int(v) for v in seg) for seg in self.num_patches]
        if self.dim != 3 or len(self.num_patches) < 1 or any(len(seg) != 3 for seg in self.num_patches):
            raise NotImplementedError("vit4hep_amd: num_patches must be a list of 3-D patch segments [[l, a, r], ...]")

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

        # geometry: ("grid", shape, patch_shape) set by CaloChallengeCFM / LEMURSCFM, or ("map", key, voxels) set by the
        # multi-segment wrappers; a consistent default otherwise
        self._geometry = None
        self._patch_map = None  # host int32 (T, P) index table of a "map" geometry
        self._map_holes = False
        self._plan = None
        self._infer_ws = {}
        self._infer_sig = None
        self.weights_epoch = 0  # bumped by code that rewrites parameters behind torch's back (the fused AdamW kernel)
        self._dev_tables = None

    # ------------------------------------------------------------------ reference-visible helpers
    def create_meshgrid(self):
        """Buffers pos_z/pos_y/pos_x on the patch grid (reference nn/vit.py:137-154): z over the concatenated l-grids of all
        segments, y / x per segment."""
        from ..patching import multi_segment_meshgrid

        z, y, x = multi_segment_meshgrid(self.num_patches)
        return torch.from_numpy(z).clone(), torch.from_numpy(y).clone(), torch.from_numpy(x).clone()

    def learnable_pos_embedding(self):
        """(T, D) table from pos_embed_freqs, computed by the HIP kernel (reference nn/vit.py:156-162)."""
        plan = self._get_plan()
        freqs = _lib.require_cuda(self.pos_embed_freqs.detach(), "pos_embed_freqs")
        _, pos = self.device_tables(freqs.device, force_pos=True)
        pe = torch.empty((self.num_tokens, int(self.hidden_dim)), dtype=torch.float32, device=freqs.device)
        _lib.check(_lib.load().v4h_op_pos_embed(plan.handle, _lib.ptr(freqs), _lib.ptr(pe), _lib.stream_ptr(freqs.device), _lib.ptr(pos)), "v4h_op_pos_embed")
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
        # the position buffers made at construction define the token count, as in the reference (wrappers may overwrite
        # self.num_patches afterwards: calochallenge_cfm/model.py:141)
        return int(self.pos_x.numel())

    def _reset_geometry(self, geom):
        self._geometry, self._plan, self._infer_ws, self._dev_tables, self._infer_sig = geom, None, {}, None, None

    def set_geometry(self, shape, patch_shape):
        """Regular grid: closed-form patch indexing inside the kernels."""
        shape, patch_shape = [int(s) for s in shape], [int(s) for s in patch_shape]
        # not validated against the network here: a fine-tuning run builds the wrapper of the NEW dataset around the backbone's network and
        # re-shapes embedders / position buffers afterwards (experiment_finetuning.py:25-34,75-171); _get_plan() checks at the first forward
        geom = ("grid", shape, patch_shape)
        if self._geometry != geom:
            self._patch_map, self._map_holes = None, False
            self._reset_geometry(geom)

    def set_patch_map(self, patch_map, voxels):
        """General geometry: ``patch_map`` int32 (T, P) = voxel index (within a sample of ``voxels`` values) of every token feature,
        -1 for none (vit4hep_amd.patching.segment_patch_map builds it for the multi-segment wrappers)."""
        import numpy as np

        pm = np.ascontiguousarray(np.asarray(patch_map, dtype=np.int32))
        if pm.ndim != 2:
            raise ValueError(f"patch map must be (tokens, patch_dim), got shape {pm.shape}")
        if pm.size and (int(pm.max()) >= int(voxels) or int(pm.min()) < -1):
            raise ValueError(f"patch map entries must lie in [-1, {int(voxels)})")
        if self._geometry is not None and self._geometry[0] == "map" and self._geometry[2] == int(voxels) and np.array_equal(self._patch_map, pm):
            return
        self._patch_map = pm
        self._map_holes = bool(np.unique(pm[pm >= 0]).size != int(voxels))
        self._reset_geometry(("map", None, int(voxels)))

    def geometry(self):
        if self._geometry is None:  # any geometry whose patching is the identity on (T, P) token rows will do
            P = int(self.patch_dim)
            if len(self.num_patches) == 1 and math.prod(self.num_patches[0]) == self.num_tokens:
                l, a, r = self.num_patches[0]
                self._geometry = ("grid", [l * P, a, r], [P, 1, 1])
            else:
                import numpy as np

                self.set_patch_map(np.arange(self.num_tokens * P, dtype=np.int32).reshape(self.num_tokens, P), self.num_tokens * P)
        return self._geometry

    def voxel_shape(self):
        """Per-sample shape (without batch / channel) of the tensors the fused forward consumes and produces."""
        g = self.geometry()
        return tuple(g[1]) if g[0] == "grid" else (g[2],)

    def map_has_holes(self):
        self.geometry()
        return self._map_holes

    def device_tables(self, device, force_pos=False):
        """(patch map, position table) on ``device`` for a mapped geometry, (None, None) for a regular grid."""
        g = self.geometry()
        if g[0] == "grid" and not force_pos:
            return None, None
        key = (str(device), self.pos_x.data_ptr(), self.pos_x._version, self.pos_y._version, self.pos_z._version, id(self._patch_map))
        if self._dev_tables is None or self._dev_tables[0] != key:
            pos = torch.cat([self.pos_x.detach().reshape(-1), self.pos_y.detach().reshape(-1), self.pos_z.detach().reshape(-1)]).to(device=device, dtype=torch.float32).contiguous()
            pm = None if self._patch_map is None else torch.from_numpy(self._patch_map).to(device).contiguous()
            self._dev_tables = (key, pm, pos)
        return self._dev_tables[1], self._dev_tables[2]

    def _check_embedders(self):
        xe = self._modules.get("x_embedder")
        if not (isinstance(xe, nn.Linear) or self._is_mapper(xe)):
            raise NotImplementedError("vit4hep_amd: x_embedder must be a Linear or the fine-tuning mapper Sequential(Linear, SiLU, Linear) "
                                      "(experiment_finetuning.py:80-91)")
        if self._is_mapper(xe) and int(xe[0].weight.shape[0]) != int(xe[2].weight.shape[1]):
            raise ValueError("embedding mapper: output width of the mapper differs from the input width of the x_embedder")
        ce = self._modules.get("c_embedder")
        if not (self._is_c_embedder(ce) or self._is_c_mapper(ce)):
            raise NotImplementedError("vit4hep_amd: c_embedder must be Sequential(Linear, SiLU, Linear) or the fine-tuning mapper "
                                      "Sequential(Linear, SiLU, Sequential(Linear, SiLU, Linear)) (experiment_finetuning.py:106-119)")
        if self._is_c_mapper(ce) and int(ce[0].weight.shape[0]) != int(ce[2][0].weight.shape[1]):
            raise ValueError("condition mapper: output width of the mapper differs from the input width of the c_embedder")
        if not isinstance(getattr(self.final_layer, "linear", None), nn.Linear):
            raise NotImplementedError("vit4hep_amd: final_layer must be a FinalLayer")

    def _get_plan(self):
        self._check_embedders()
        g = self.geometry()
        T, P = self.num_tokens, int(self.patch_dim)
        key = (T, P, int(self.condition_dim), int(self.final_layer.linear.weight.shape[0]), g[0], tuple(g[1]) if g[0] == "grid" else g[2], id(self._patch_map),
               self.x_embed_in(), self.c_embed_in())
        if self._plan is not None and getattr(self, "_plan_key", None) != key:  # embedders / head / position buffers were re-shaped
            self._plan, self._infer_ws, self._dev_tables, self._infer_sig, self._infer_c = None, {}, None, None, None
        if self._plan is None:
            if int(self.final_layer.linear.weight.shape[0]) != P * int(self.out_channels):
                raise ValueError(f"final_layer emits {int(self.final_layer.linear.weight.shape[0])} features per token, x_embedder takes {P}")
            if g[0] == "grid":
                if math.prod(s // p for s, p in zip(g[1], g[2])) != T or math.prod(g[2]) * int(self.out_channels) != P:
                    raise ValueError(f"geometry shape={g[1]} patch_shape={g[2]} does not match num_tokens={T} patch_dim={P}")
            elif self._patch_map.shape != (T, P):
                raise ValueError(f"patch map shape {self._patch_map.shape} does not match (num_tokens={T}, patch_dim={P})")
            self._plan_key = key
            shape, patch_shape, mapped = (g[1], g[2], None) if g[0] == "grid" else (None, None, (self.num_tokens, int(self.patch_dim), g[2]))
            residual = self.amd_residual
            if residual == "auto":
                D = int(self.hidden_dim)
                residual = "bf16" if _lib.MODES[self.amd_mode] == _lib.MODE_BF16 and D % 8 == 0 and D <= 512 else "f32"
            self._plan = _lib.Plan(shape, patch_shape, self.condition_dim, self.hidden_dim, self.depth, self.num_heads,
                                   int(self.hidden_dim * self.mlp_ratio), self.t_embedder.frequency_embedding_size, self.amd_mode, mapped=mapped,
                                   x_embed_in=self.x_embed_in(), c_embed_in=self.c_embed_in(), residual=residual)
            got = [tuple(p.shape) for p in self.parameter_list()]
            if got != self._plan.shapes:
                raise RuntimeError(f"parameter inventory differs from the library's: {got} vs {self._plan.shapes}")
        return self._plan

    def parameter_list(self):
        """Learnable tensors in the state_dict() order the C ABI expects (include/vit4hep_hip.h)."""
        self._check_embedders()
        mapper = self._is_mapper(self.x_embedder)
        xlin = self.x_embedder[2] if mapper else self.x_embedder
        cmapper = self._is_c_mapper(self.c_embedder)
        cseq = self.c_embedder[2] if cmapper else self.c_embedder
        ps = [self.pos_embed_freqs, xlin.weight, xlin.bias,
              cseq[0].weight, cseq[0].bias, cseq[2].weight, cseq[2].bias,
              self.t_embedder.mlp[0].weight, self.t_embedder.mlp[0].bias, self.t_embedder.mlp[2].weight, self.t_embedder.mlp[2].bias]
        for b in self.blocks:
            ps += [b.attn.qkv.weight, b.attn.qkv.bias, b.attn.proj.weight, b.attn.proj.bias, b.mlp.fc1.weight, b.mlp.fc1.bias,
                   b.mlp.fc2.weight, b.mlp.fc2.bias, b.adaLN_modulation[1].weight, b.adaLN_modulation[1].bias]
        ps += [self.final_layer.linear.weight, self.final_layer.linear.bias,
               self.final_layer.adaLN_modulation[1].weight, self.final_layer.adaLN_modulation[1].bias]
        if mapper:  # the C ABI takes the mapper's two tensors last (include/vit4hep_hip.h: v4h_config.x_embed_in)
            ps += [self.x_embedder[0].weight, self.x_embedder[0].bias]
        if cmapper:  # ... and the condition mapper's after them (v4h_config.c_embed_in)
            ps += [self.c_embedder[0].weight, self.c_embedder[0].bias]
        return ps

    def inference_workspace(self, B, device):
        key = (int(B), str(device))
        ws = self._infer_ws.get(key)
        if ws is None:
            self._infer_ws = {key: torch.empty(self._get_plan().workspace_bytes(B, False), dtype=torch.uint8, device=device)}
            self._infer_sig = None
            self._infer_c = None
            ws = self._infer_ws[key]
        return ws

    def frozen_weights(self):
        """Context manager: inside it the parameters are frozen BY CONSTRUCTION (the ODE solve of ``CFM._sample_from``: 80 network evaluations
        of one batch), so the inference forward may keep its bf16 / padded operand copies and the condition embedding between calls.  The first
        forward inside always recasts; every inference forward OUTSIDE such a scope recasts too - a signature of (data_ptr, _version) cannot
        see writes through ``p.data`` (EMA ``copy_to`` / ``restore``, reference base_experiment.py:630), so it is only a second line of
        defence here, not the licence to reuse.  ``weights_epoch`` is the explicit invalidation hook for code that rewrites parameters through
        raw pointers inside a scope (the fused trainer bumps it)."""
        import contextlib

        @contextlib.contextmanager
        def scope():
            outer = getattr(self, "_frozen_scope", None)
            self._frozen_scope = {"primed": False}
            try:
                yield self
            finally:
                self._frozen_scope = outer
                self._infer_sig, self._infer_c = None, None

        return scope()

    def operands_current(self, params, ws, mark=True):
        """True when `ws` already holds the operand copies of exactly these parameter values: only inside a ``frozen_weights()`` scope, after
        its first forward, and with an unchanged signature (same storage, no tracked in-place update, same ``weights_epoch``)."""
        scope = getattr(self, "_frozen_scope", None)
        if scope is None:
            self._infer_sig, self._infer_c = None, None
            return False
        primed = scope["primed"]
        if mark:
            scope["primed"] = True
        sig = (ws.data_ptr(), self.weights_epoch, self.pos_x.data_ptr(), self.pos_x._version, id(self._plan),
               tuple((p.data_ptr(), p._version) for p in params))
        same = primed and getattr(self, "_infer_sig", None) == sig
        if mark:
            self._infer_sig = sig
        return same

    def condition_current(self, c, ws):
        """True when the previous inference call on `ws` embedded exactly these conditions: the same tensor object (held weakly), not written
        in place since (``_version``).  Only meaningful together with ``operands_current`` (the c_embedder weights must be unchanged too)."""
        import weakref

        try:
            sig = (ws.data_ptr(), c.data_ptr(), c._version, tuple(c.shape))
        except RuntimeError:  # inference tensors do not track versions: never assume
            self._infer_c = None
            return False
        prev = getattr(self, "_infer_c", None)
        same = prev is not None and prev[0]() is c and prev[1] == sig
        self._infer_c = (weakref.ref(c), sig)
        return same

    # ------------------------------------------------------------------ forward
    def forward(self, x, t, c):
        """x: patch tokens (B, T, P) as in the reference (nn/vit.py:185-206) -> (B, T, P);
        or voxels (B, 1, L, A, R) / (B, 1, n_voxels) -> voxels, which skips two layout passes (used by the CFM wrappers)."""
        from ..autograd import vit_apply

        # (B, T, P) tokens; anything else is voxels ((B, 1, n_voxels) of a mapped geometry is 3-D too)
        patches_io = x.dim() == 3 and tuple(x.shape[1:]) == (self.num_tokens, int(self.patch_dim))
        return vit_apply(self, x, t, c, patches_io=patches_io)


def modulate(x, shift, scale):
    """Kept for API compatibility (reference nn/vit.py:457-458); on the HIP path it is fused into the LayerNorm kernel."""
    return x * (1 + scale.unsqueeze(1)) + shift.unsqueeze(1)


def get_sincos_pos_embed(*args, **kwargs):
    raise NotImplementedError("vit4hep_amd: fixed sincos tables (learn_pos_embed=False / fine-tuning re-meshing) are outside the shape-CFM hot path")
