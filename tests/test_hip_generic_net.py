"""-m gpu: the CFM host class with a network that is NOT this package's ViT (the energy-model CFM of the
reference is `models.base_model.CFM` around `nn.cfm.transformer_cfm.ParallelTransformer`, configs/model/cfm/cfm_ds2_energy.yaml).
Trajectory, loss and the fixed-grid ODE solver are HIP kernels that do not care what the network is; the network itself then runs
as ordinary PyTorch-ROCm modules.  Checked against the oracle's solver / loss restatement with the same module on the CPU."""

import copy

import pytest
import torch
import torch.nn as nn

from oracle import vit_cfm_oracle as O
from tests import hiputil as U

pytestmark = pytest.mark.gpu


class TinyEnergyNet(nn.Module):
    """Stand-in with the call signature of ParallelTransformer.forward(x, t, condition): (B, 45), (B, 1), (B, 1) -> (B, 45)."""

    def __init__(self, dims_in=45, dims_c=1, hidden=96):
        super().__init__()
        self.inp = nn.Linear(dims_in + 1 + dims_c, hidden)
        self.enc = nn.TransformerEncoderLayer(d_model=hidden, nhead=4, dim_feedforward=128, dropout=0.0, batch_first=True)
        self.out = nn.Linear(hidden, dims_in)

    def forward(self, x, t, condition=None):
        h = self.inp(torch.cat([x, t, condition], dim=1))
        return self.out(self.enc(h.unsqueeze(1)).squeeze(1))


def _models():
    from vit4hep_amd import CFM

    torch.manual_seed(0)
    net = TinyEnergyNet()
    cpu_net = copy.deepcopy(net)
    model = CFM(net, "uniform", "linear", {"method": "rk4", "options": {"step_size": 0.05}}, shape=[45])
    model.device, model.dtype = torch.device(U.DEV), torch.float32
    return model.to(U.DEV), cpu_net


def test_batch_loss_and_gradients_with_a_generic_network():
    model, cpu_net = _models()
    g = torch.Generator().manual_seed(1)
    x, c = torch.randn((16, 45), generator=g), torch.rand((16, 1), generator=g)
    t, x0 = torch.rand((16, 1), generator=g), torch.randn((16, 45), generator=g)
    loss = model._loss_from_noise(x.to(U.DEV), c.to(U.DEV), t.to(U.DEV), x0.to(U.DEV))
    loss.backward()
    v = cpu_net((1 - t) * x0 + t * x, t, c)  # models/trajectories.py:5-8, models/base_model.py:214-218
    ref = ((v - (x - x0)) ** 2).mean()
    ref.backward()
    assert abs(loss.item() - ref.item()) / ref.item() < 1e-5
    for (k, p), q in zip(model.net.named_parameters(), cpu_net.parameters()):
        assert U.rel_err(p.grad, q.grad) < 1e-3, k
    # the reference entry point draws t and x_0 itself
    torch.manual_seed(3)
    out = model._batch_loss([x, c])
    assert out.requires_grad and torch.isfinite(out)


@pytest.mark.parametrize("method,step", [("rk4", 0.05), ("heun2", 0.25), ("euler", 0.1), ("midpoint", 0.2)])
def test_fixed_grid_solver_with_a_generic_network(method, step):
    model, cpu_net = _models()
    model.odeint_kwargs = {"method": method, "options": {"step_size": step}}
    g = torch.Generator().manual_seed(2)
    c, x_T = torch.rand((8, 1), generator=g), torch.randn((8, 45), generator=g)
    with torch.inference_mode():
        got = model._sample_from(x_T.to(U.DEV), c.to(U.DEV))

        def f(tt, y):
            return cpu_net(y, torch.full((8, 1), float(tt)), c)

        grid = O.fixed_grid(0.0, 1.0, step)
        y = x_T.clone()
        for k in range(len(grid) - 1):
            y = O.ode_step(f, method, grid[k], grid[k + 1], y)
    assert U.rel_err(got, y) < 1e-4
    torch.manual_seed(5)
    s = model.sample_batch(c.to(U.DEV))  # base-class entry point: x_T of shape (B, *shape), no channel axis (base_model.py:229)
    assert s.shape == (8, 45) and torch.isfinite(s).all()
