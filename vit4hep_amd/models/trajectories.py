"""x_t / dx_t/dt interpolants (reference models/trajectories.py).  Only ``linear`` is reachable on the
CFM path (models/base_model.py:186-190); it runs as one fused HIP kernel (v4h_cfm_prepare)."""

import torch

from .. import _lib


def linear_trajectory(x_0, x_1, t):
    """x_t = (1 - t) x_0 + t x_1 ;  x_t_dot = x_1 - x_0      (reference models/trajectories.py:5-8)"""
    x_0 = _lib.require_cuda(x_0, "x_0")
    x_1 = _lib.require_cuda(x_1, "x_1")
    t = _lib.require_cuda(t, "t").reshape(-1)
    B = x_1.shape[0]
    if t.numel() != B or x_0.shape != x_1.shape:
        raise RuntimeError("linear_trajectory: t must hold one time per sample and x_0, x_1 the same shape")
    x_t, x_t_dot = torch.empty_like(x_1), torch.empty_like(x_1)
    _lib.check(
        _lib.load().v4h_cfm_prepare(_lib.ptr(x_1), _lib.ptr(x_0), _lib.ptr(t), _lib.ptr(x_t), _lib.ptr(x_t_dot), B, x_1[0].numel(),
                                    _lib.stream_ptr(x_1.device)),
        "v4h_cfm_prepare",
    )
    return x_t, x_t_dot
