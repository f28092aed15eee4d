"""ctypes binding of libvit4hep_hip.so (C ABI: include/vit4hep_hip.h).

The product path has no fallback: if the library is missing or an entry point fails, a
RuntimeError is raised.  PyTorch is used only for device memory and streams.
"""

from __future__ import annotations

import ctypes as C
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
# VIT4HEP_AMD_LIB: load another build of the same ABI (same-box A/B measurements of kernel changes)
LIB_PATH = os.environ.get("VIT4HEP_AMD_LIB") or os.path.join(HERE, "libvit4hep_hip.so")

ABI_VERSION = 11
MODE_F32 = 0
MODE_BF16 = 1
MODES = {"f32": MODE_F32, "fp32": MODE_F32, "float32": MODE_F32, "bf16": MODE_BF16, "bfloat16": MODE_BF16}


class V4HConfig(C.Structure):
    _fields_ = [
        ("shape", C.c_int32 * 3),
        ("patch_shape", C.c_int32 * 3),
        ("in_channels", C.c_int32),
        ("condition_dim", C.c_int32),
        ("hidden_dim", C.c_int32),
        ("depth", C.c_int32),
        ("num_heads", C.c_int32),
        ("mlp_hidden", C.c_int32),
        ("freq_dim", C.c_int32),
        ("mode", C.c_int32),
        ("x_embed_in", C.c_int32),
        ("c_embed_in", C.c_int32),
    ]


class V4HEnergyConfig(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("dims_in", "dims_c", "dim_embedding", "nhead", "num_encoder_layers", "num_decoder_layers", "dim_feedforward",
                                         "encode_t_dim", "mode")]


class V4HChainSpec(C.Structure):
    _fields_ = [("n_layers", C.c_int32), ("n_voxels", C.c_int64)] + [(k, C.c_float) for k in ("eps", "norm_cut", "factor", "cut", "delta", "mean", "std", "alpha",
                                                                                           "e_min", "e_max")]


_vp, _i32, _i64, _f32, _sz = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t
_pp = C.POINTER(C.c_void_p)

# name -> (restype, argtypes); exactly the declarations of include/vit4hep_hip.h
SIGNATURES = {
    "v4h_abi_version": (_i32, []),
    "v4h_last_error": (C.c_char_p, []),
    "v4h_plan_create": (_i32, [C.POINTER(V4HConfig), _pp]),
    "v4h_plan_create_mapped": (_i32, [C.POINTER(V4HConfig), _i32, _i32, _i64, _pp]),
    "v4h_plan_destroy": (None, [_vp]),
    "v4h_plan_num_params": (_i32, [_vp]),
    "v4h_plan_param_shape": (_i32, [_vp, _i32, C.POINTER(_i32), C.POINTER(_i32)]),
    "v4h_plan_workspace_bytes": (_sz, [_vp, _i32, _i32]),
    "v4h_vit_forward": (_i32, [_vp, _i32, _pp, _vp, _vp, _vp, _vp, _vp, _sz, _i32, _vp, _vp, _vp]),
    "v4h_vit_prepare_operands": (_i32, [_vp, _i32, _pp, _vp, _sz, _i32, _vp, _vp]),
    "v4h_vit_update_ahead": (_i32, [_vp, _i32, _pp, _vp, _vp, _vp, _vp, C.POINTER(C.c_int64), _vp, _sz, _vp, _f32, _f32, _f32, _i32, _f32, _f32, _f32, _f32, _f32,
                                    _vp, _vp, _vp, _vp, _vp, _vp]),
    "v4h_plan_join": (_i32, [_vp, _vp]),
    "v4h_plan_set_gradient_mode": (_i32, [_vp, _i32]),
    "v4h_plan_set_residual_storage": (_i32, [_vp, _i32, _i32]),
    "v4h_plan_residual_storage": (_i32, [_vp]),
    "v4h_vit_backward": (_i32, [_vp, _i32, _pp, _pp, _vp, _vp, _sz, _i32, _i32, _vp, _vp, _vp]),
    "v4h_vit_backward_events": (_i32, [_vp, _i32, _pp, _pp, _vp, _vp, _sz, _vp, _vp, _vp, _pp]),
    "v4h_vit_backward_stage": (_i32, [_vp, _i32, _pp, _pp, _vp, _vp, _sz, _i32, _vp, _vp, _vp, _vp, _i32]),
    "v4h_vit_num_backward_stages": (_i32, [_vp]),
    "v4h_energy_plan_create": (_i32, [C.POINTER(V4HEnergyConfig), _pp]),
    "v4h_energy_plan_destroy": (None, [_vp]),
    "v4h_energy_plan_num_params": (_i32, [_vp]),
    "v4h_energy_plan_param_shape": (_i32, [_vp, _i32, C.POINTER(_i32), C.POINTER(_i32)]),
    "v4h_energy_plan_workspace_bytes": (_sz, [_vp, _i32]),
    "v4h_energy_forward": (_i32, [_vp, _i32, _pp, _vp, _vp, _vp, _vp, _vp, _sz, _i32, _vp]),
    "v4h_shape_preprocess": (_i32, [C.POINTER(V4HChainSpec), _vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "v4h_shape_postprocess": (_i32, [C.POINTER(V4HChainSpec), _vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "v4h_cfm_prepare": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i64, _vp]),
    "v4h_mse_loss": (_i32, [_vp, _vp, _vp, _vp, _i64, _vp]),
    "v4h_sq_norm_accum": (_i32, [_vp, _i64, _vp, _vp]),
    "v4h_adamw_step": (_i32, [_vp, _vp, _vp, _vp, _i64, _vp, _f32, _f32, _f32, _f32, _f32, _f32, _i32, _vp, _vp]),
    "v4h_adamw_step_sched": (_i32, [_vp, _vp, _vp, _vp, _i64, _vp, _f32, _f32, _f32, _i32, _f32, _f32, _f32, _f32, _vp, _vp, _f32, _vp, _vp, _vp]),
    "v4h_adamw_step_sched_ema": (_i32, [_vp, _vp, _vp, _vp, _i64, _vp, _f32, _f32, _f32, _i32, _f32, _f32, _f32, _f32, _vp, _vp, _f32, _vp, _vp, _vp, _vp, _f32]),
    "v4h_cfm_prepare_z": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i64, _vp, _vp, _vp]),
    "v4h_mse_loss_acc": (_i32, [_vp, _vp, _vp, _vp, _i64, _vp]),
    "v4h_axpby": (_i32, [_vp, _vp, _vp, _f32, _f32, _i64, _vp]),
    "v4h_rk4_combine": (_i32, [_vp, _vp, _vp, _vp, _vp, _f32, _i64, _vp]),
    "v4h_op_gemm": (_i32, [_i32, _vp, _i32, _i32, _vp, _i32, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "v4h_op_gemm_gelu": (_i32, [_i32, _vp, _i32, _vp, _i32, _vp, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _vp]),
    "v4h_op_gemm_dgelu": (_i32, [_i32, _vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _vp]),
    "v4h_op_gemm_wgrad_slab": (_i32, [_i32, _vp, _i32, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "v4h_op_gemm_wgrad_splitk": (_i32, [_i32, _i32, _i32, _i32]),
    "v4h_op_attention_fwd": (_i32, [_i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "v4h_op_attention_bwd": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "v4h_op_ln_modulate_fwd": (_i32, [_i32, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "v4h_op_patchify": (_i32, [_vp, _vp, _vp, _i32, _vp, _vp]),
    "v4h_op_unpatchify": (_i32, [_vp, _vp, _vp, _i32, _vp, _vp]),
    "v4h_op_pos_embed": (_i32, [_vp, _vp, _vp, _vp, _vp]),
    "v4h_select_contraction_kernel": (_i32, [_i32]),
    "v4h_selected_contraction_kernel": (_i32, []),
    "v4h_reserve_compute_units": (_i32, [_i32]),
    "v4h_reserved_compute_units": (_i32, []),
    "v4h_calib_mfma_loop": (_i32, [_vp, _vp, _i32, _i32, _vp]),
    "v4h_calib_copy": (_i32, [_vp, _vp, _i64, _vp]),
}
# contraction kernels selectable through v4h_select_contraction_kernel (all exact)
KERNEL_AUTO, KERNEL_TWO_WG, KERNEL_RING, KERNEL_WS = 0, 1, 2, 3

_lib = None


def load():
    """Load the shared library (once).  Raises loudly when it is absent: there is no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        try:  # fresh checkout (the .so is git-ignored): compile it in-tree; rank 0 of a multi-process job builds, the others wait
            _build_once()
        except Exception as e:
            raise RuntimeError(
                f"{LIB_PATH} not found and could not be built ({e}): run `python -m vit4hep_amd.build` (hipcc, gfx950). "
                "vit4hep_amd has no CPU or PyTorch fallback for the ViT-CFM path."
            ) from e
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.v4h_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libvit4hep_hip.so ABI {lib.v4h_abi_version()} != {ABI_VERSION}: rebuild with `python -m vit4hep_amd.build`")
    _lib = lib
    return lib


def _build_once():
    import time

    from .build import build

    lock = LIB_PATH + ".lock"
    try:
        fd = os.open(lock, os.O_CREAT | os.O_EXCL | os.O_WRONLY)
    except FileExistsError:  # another local rank is compiling
        for _ in range(600):
            if os.path.exists(LIB_PATH) and not os.path.exists(lock):
                return
            time.sleep(1.0)
        raise RuntimeError("timed out waiting for another process to build the library")
    try:
        os.close(fd)
        build(verbose=False)
    finally:
        os.remove(lock)


def check(rc, what=""):
    if rc != 0:
        msg = load().v4h_last_error().decode(errors="replace")
        raise RuntimeError(f"libvit4hep_hip {what} failed (code {rc}): {msg}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def on_device(t):
    """Context: make the tensor's GPU the current device for the calls inside.  The library launches on torch's current stream OF THAT DEVICE and
    creates a plan's side stream / events on the device that is current at its first call; with another device current (a model on cuda:1 without
    torch.cuda.set_device) both would land on the wrong GPU."""
    return torch.cuda.device(t.device)


def stream_ptr(device=None):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_cuda(t, name, dtype=torch.float32):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(
            f"{name} must be a tensor on the MI355X (cuda) device: vit4hep_amd runs the ViT-CFM path only through its HIP library"
        )
    if t.dtype != dtype:
        raise RuntimeError(f"{name} must have dtype {dtype}, got {t.dtype}")
    return t.contiguous()


def pointer_table(tensors):
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


class Plan:
    """Host-side plan (sizes + workspace layout) for one (geometry, network, mode).

    Regular grid: ``shape`` / ``patch_shape``.  General geometry: ``mapped=(tokens, patch_dim, voxels)`` (shape / patch_shape
    ignored); forward / backward calls then need the index map and position table (include/vit4hep_hip.h)."""

    def __init__(self, shape, patch_shape, condition_dim, hidden_dim, depth, num_heads, mlp_hidden, freq_dim=256, mode="f32", in_channels=1, mapped=None,
                 x_embed_in=0, c_embed_in=0, residual="f32"):
        lib = load()
        cfg = V4HConfig()
        cfg.shape[:] = [int(v) for v in (shape if mapped is None else (0, 0, 0))]
        cfg.patch_shape[:] = [int(v) for v in (patch_shape if mapped is None else (0, 0, 0))]
        cfg.in_channels = int(in_channels)
        cfg.condition_dim = int(condition_dim)
        cfg.hidden_dim = int(hidden_dim)
        cfg.depth = int(depth)
        cfg.num_heads = int(num_heads)
        cfg.mlp_hidden = int(mlp_hidden)
        cfg.freq_dim = int(freq_dim)
        cfg.mode = MODES[mode] if isinstance(mode, str) else int(mode)
        cfg.x_embed_in = int(x_embed_in)
        cfg.c_embed_in = int(c_embed_in)
        self.mode = cfg.mode
        self.cfg = cfg
        h = C.c_void_p()
        self.mapped = mapped is not None
        if mapped is None:
            check(lib.v4h_plan_create(C.byref(cfg), C.byref(h)), "v4h_plan_create")
        else:
            tokens, patch_dim, voxels = (int(v) for v in mapped)
            check(lib.v4h_plan_create_mapped(C.byref(cfg), tokens, patch_dim, voxels, C.byref(h)), "v4h_plan_create_mapped")
        self.handle = h
        # storage of the residual stream and of its gradient inside the workspace (include/vit4hep_hip.h: v4h_plan_set_residual_storage)
        if residual not in RESIDUAL:
            raise ValueError(f"residual storage must be one of {sorted(RESIDUAL)}, got {residual!r}")
        self.residual = residual
        if residual != "f32":
            check(lib.v4h_plan_set_residual_storage(h, *RESIDUAL[residual]), "v4h_plan_set_residual_storage")
        self.num_params = lib.v4h_plan_num_params(h)
        self.num_stages = lib.v4h_vit_num_backward_stages(h)
        self.shapes = []
        for i in range(self.num_params):
            r, c = _i32(), _i32()
            check(lib.v4h_plan_param_shape(h, i, C.byref(r), C.byref(c)), "v4h_plan_param_shape")
            self.shapes.append((r.value,) if c.value == 0 else (r.value, c.value))

    def workspace_bytes(self, B, training):
        return int(load().v4h_plan_workspace_bytes(self.handle, int(B), 1 if training else 0))

    def __del__(self):
        try:
            if _lib is not None and getattr(self, "handle", None):
                _lib.v4h_plan_destroy(self.handle)
        except Exception:
            pass


RESIDUAL = {"f32": (0, 0), "bf16": (1, 1), "x_bf16": (1, 0), "dx_bf16": (0, 1)}  # (x stored as bf16, dx stored as bf16)
FWD_TRAINING, FWD_REUSE_OPERANDS, FWD_SAME_CONDITION, ENERGY_SAME_CONDITION, ENERGY_COMPOSED = 1, 2, 4, 4, 8  # flag bits of include/vit4hep_hip.h


class EnergyPlan:
    """Host-side plan of the energy-model network (v4h_energy_plan_*)."""

    def __init__(self, dims_in, dims_c, dim_embedding, nhead, num_encoder_layers, num_decoder_layers, dim_feedforward, encode_t_dim, mode="f32"):
        lib = load()
        cfg = V4HEnergyConfig(int(dims_in), int(dims_c), int(dim_embedding), int(nhead), int(num_encoder_layers), int(num_decoder_layers),
                              int(dim_feedforward), int(encode_t_dim), MODES[mode] if isinstance(mode, str) else int(mode))
        self.cfg = cfg
        h = C.c_void_p()
        check(lib.v4h_energy_plan_create(C.byref(cfg), C.byref(h)), "v4h_energy_plan_create")
        self.handle = h
        self.num_params = lib.v4h_energy_plan_num_params(h)
        self.shapes = []
        for i in range(self.num_params):
            r, c = _i32(), _i32()
            check(lib.v4h_energy_plan_param_shape(h, i, C.byref(r), C.byref(c)), "v4h_energy_plan_param_shape")
            self.shapes.append((r.value,) if c.value == 0 else (r.value, c.value))

    def workspace_bytes(self, B):
        return int(load().v4h_energy_plan_workspace_bytes(self.handle, int(B)))

    def __del__(self):
        try:
            if _lib is not None and getattr(self, "handle", None):
                _lib.v4h_energy_plan_destroy(self.handle)
        except Exception:
            pass
