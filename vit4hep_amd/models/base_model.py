"""Host mirror of the reference's ``models.base_model`` (BaseModel, CFM) for the MI355X path.

``CFM._batch_loss`` / ``forward`` / ``sample_batch`` keep the reference's signatures and semantics
(models/base_model.py:159-244); trajectory, loss and the ODE-solver vector updates are HIP kernels.
The likelihood model ``CINN`` of the reference is a different model family and not part of this package.
"""

from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from .. import _lib
from .trajectories import linear_trajectory


class BaseModel(nn.Module):
    def __init__(self, shape):
        super().__init__()
        self.shape = shape


class _MSE(torch.autograd.Function):
    """loss = mean((v - target)^2)   (reference models/base_model.py:217-218); one kernel gives loss and dloss/dv."""

    @staticmethod
    def forward(ctx, v, target):
        v = _lib.require_cuda(v, "velocity")
        target = _lib.require_cuda(target, "target")
        loss = torch.empty((), dtype=torch.float32, device=v.device)
        dv = torch.empty_like(v) if ctx.needs_input_grad[0] else None
        _lib.check(_lib.load().v4h_mse_loss(_lib.ptr(v), _lib.ptr(target), _lib.ptr(loss), _lib.ptr(dv), v.numel(), _lib.stream_ptr(v.device)), "v4h_mse_loss")
        ctx.dv = dv
        return loss

    @staticmethod
    def backward(ctx, g):
        dv, ctx.dv = ctx.dv, None
        return dv * g, None


def mse_loss(v, target):
    return _MSE.apply(v, target)


def fixed_grid(t0, t1, step):
    """Time grid of torchdiffeq's fixed-grid solvers for options.step_size (f32 arithmetic like the solver)."""
    a, b, h = np.float32(t0), np.float32(t1), np.float32(step)
    n = int(np.ceil(np.float32((b - a) / h) + np.float32(1.0)))
    g = (np.arange(n, dtype=np.float32) * h + a).astype(np.float32)
    g[-1] = b
    return g


def _axpby(out, a, b, alpha, beta):
    with _lib.on_device(out):
        return _axpby_on(out, a, b, alpha, beta)


def _axpby_on(out, a, b, alpha, beta):
    _lib.check(_lib.load().v4h_axpby(_lib.ptr(out), _lib.ptr(a), _lib.ptr(b), float(alpha), float(beta), out.numel(), _lib.stream_ptr(out.device)), "v4h_axpby")
    return out


def odeint_fixed(f, y0, t0, t1, method="rk4", step_size=0.05):
    """Fixed-grid integration of dy/dt = f(t, y) (torchdiffeq.odeint semantics for 'euler', 'midpoint',
    'heun2'/'heun', 'rk4' = 3/8 rule).  Returns y(t1).  Vector updates are HIP kernels."""
    if method not in ("euler", "midpoint", "heun", "heun2", "rk4"):
        raise ValueError(f"unsupported fixed-grid method {method!r}")
    grid = fixed_grid(t0, t1, step_size)
    y = y0.clone()
    tmp = torch.empty_like(y)
    for k in range(len(grid) - 1):
        ta, tb = grid[k], grid[k + 1]
        dt = np.float32(tb - ta)
        if method == "euler":
            _axpby(y, y, f(ta, y), 1.0, dt)
        elif method == "midpoint":
            _axpby(tmp, y, f(ta, y), 1.0, dt * np.float32(0.5))
            _axpby(y, y, f(np.float32(ta + dt * np.float32(0.5)), tmp), 1.0, dt)
        elif method in ("heun", "heun2"):
            k1 = f(ta, y)
            _axpby(tmp, y, k1, 1.0, dt)
            k2 = f(tb, tmp)
            _axpby(tmp, k1, k2, dt * np.float32(0.5), dt * np.float32(0.5))
            _axpby(y, y, tmp, 1.0, 1.0)
        else:  # rk4, 3/8 rule
            third = np.float32(dt / np.float32(3.0))
            k1 = f(ta, y)
            _axpby(tmp, y, k1, 1.0, third)
            k2 = f(np.float32(ta + third), tmp)
            _axpby(tmp, y, k2, 1.0, dt)
            _axpby(tmp, tmp, k1, 1.0, -third)
            k3 = f(np.float32(ta + np.float32(2.0) * third), tmp)
            _axpby(tmp, y, k1, 1.0, dt)
            _axpby(tmp, tmp, k2, 1.0, -dt)
            _axpby(tmp, tmp, k3, 1.0, dt)
            k4 = f(tb, tmp)
            with _lib.on_device(y):
                _lib.check(_lib.load().v4h_rk4_combine(_lib.ptr(y), _lib.ptr(k1), _lib.ptr(k2), _lib.ptr(k3), _lib.ptr(k4), float(dt), y.numel(),
                                                       _lib.stream_ptr(y.device)), "v4h_rk4_combine")
    return y


class CFM(BaseModel):
    """Conditional flow matching model (reference models/base_model.py:159-247)."""

    def __init__(self, net, time_distribution="uniform", trajectory="linear", odeint_kwargs=None, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.time_distribution = self.get_time_distribution(time_distribution)
        self.trajectory = self.get_trajectory(trajectory)
        self.odeint_kwargs = odeint_kwargs
        self.net = net

    def get_trajectory(self, trajectory):
        if trajectory == "linear":
            return linear_trajectory
        raise ValueError

    def get_time_distribution(self, time_distribution):
        if time_distribution == "uniform":
            return torch.distributions.uniform.Uniform(low=0.0, high=1.0)
        raise ValueError

    def forward(self, x, t, c):
        return self.net(x, t, c)

    def _loss_from_noise(self, x, c, t, x_0):
        """The deterministic part of _batch_loss (t and x_0 given)."""
        x_t, x_t_dot = self.trajectory(x_0, x, t)
        velocity = self.forward(x_t, t.view(-1, 1), c)
        return mse_loss(velocity, x_t_dot)

    def _batch_loss(self, x):
        """reference models/base_model.py:203-218: t ~ U(0,1) on the CPU generator, x_0 ~ N(0,1) on the device generator."""
        x, c = x[0], x[1]
        x = x.to(dtype=self.dtype, device=self.device, non_blocking=True)
        c = c.to(dtype=self.dtype, device=self.device, non_blocking=True)
        t = self.time_distribution.sample([x.shape[0]] + [1] * (x.dim() - 1))
        t = t.to(self.device, self.dtype, non_blocking=True)
        x_0 = torch.randn_like(x)
        return self._loss_from_noise(x, c, t, x_0)

    def _sample_from(self, x_T, batch):
        kw = dict(self.odeint_kwargs or {})
        method = kw.get("method", "rk4")
        step = (kw.get("options") or {}).get("step_size")
        if step is None:
            raise NotImplementedError("vit4hep_amd: only fixed-grid solvers with options.step_size are implemented (every shipped config uses rk4 / 0.05)")
        B = x_T.shape[0]

        def f(t, x_t):
            t_vec = torch.full((B, 1), float(t), dtype=x_T.dtype, device=x_T.device)
            return self.forward(x_t, t_vec, batch)

        core = self.net
        while hasattr(core, "module"):  # DDP(model.net) of the reference (base_experiment.py:161-167)
            core = core.module
        if hasattr(core, "frozen_weights"):  # the solver's evaluations share frozen weights and one condition batch
            with core.frozen_weights():
                return odeint_fixed(f, x_T, 0.0, 1.0, method, float(step))
        return odeint_fixed(f, x_T, 0.0, 1.0, method, float(step))

    @torch.inference_mode()
    def sample_batch(self, batch):
        """reference models/base_model.py:220-244"""
        x_T = torch.randn((batch.shape[0], *self.shape), dtype=batch.dtype, device=batch.device)
        return self._sample_from(x_T, batch)
