"""-m gpu: the fused pre-/post-processing kernels against golden vectors made by the reference's own transform
classes, against the CPU oracle at other sizes, and through round-trip properties at full batch size.
Tolerance: f32 with libm exp/log; sums are accumulated in another order than torch's -> 2e-5 relative (+ 1e-6 of the tensor maximum for
voxels that sit on the cut)."""

import numpy as np
import pytest
import torch

from oracle import transforms_oracle as TO
from tests import hiputil as U

pytestmark = pytest.mark.gpu
SPECS = {"transforms_ds2_b4": TO.ds2_spec(mean=-1.7, std=2.9),
         "transforms_ds1ph_b6": TO.ChainSpec(layer_boundaries=(0, 8, 168, 358, 363, 368), shape=(368,), mean=-0.8, std=3.3, factor=0.5, cut=1e-6)}


def chain_of(s):
    from vit4hep_amd.transforms import ShapeChain

    return ShapeChain(s.layer_boundaries, s.shape, s.eps, s.norm_cut, s.factor, s.cut, s.delta, s.mean, s.std, s.alpha, s.e_min, s.e_max)


def close(got, want, rtol=2e-5, atol_frac=1e-6):
    got, want = got.detach().double().cpu(), torch.as_tensor(want).double()
    tol = rtol * want.abs() + atol_frac * want.abs().max()
    bad = (got - want).abs() > tol
    assert not bad.any(), f"{int(bad.sum())} of {bad.numel()} elements off, worst {float(((got - want).abs() / (want.abs() + 1e-30))[bad].max()):.3e}"


@pytest.mark.parametrize("name", list(SPECS))
def test_chain_vs_reference_vectors(name, golden):
    g, s = golden(name), SPECS[name]
    ch = chain_of(s)
    x, c = ch.preprocess(torch.from_numpy(g["showers"]).to(U.DEV), torch.from_numpy(g["energy"]).to(U.DEV))
    assert x.shape == g["x"].shape and c.shape == g["c"].shape
    close(x, g["x"], atol_frac=2e-6)
    close(c, g["c"], atol_frac=2e-6)
    sh, e = ch.postprocess(torch.from_numpy(g["samples"]).to(U.DEV), torch.from_numpy(g["cond"]).to(U.DEV))
    assert sh.shape == g["post_showers"].shape and e.shape == g["post_energy"].shape
    close(sh, g["post_showers"])
    close(e, g["post_energy"])
    assert torch.equal(sh.cpu() == 0, torch.from_numpy(g["post_showers"]) == 0) or ((sh.cpu() == 0) != (torch.from_numpy(g["post_showers"]) == 0)).sum() <= 2
    sh, e = ch.postprocess(torch.from_numpy(g["x"]).to(U.DEV), torch.from_numpy(g["c"]).to(U.DEV))
    close(sh, g["roundtrip_showers"])


def test_full_size_round_trip_and_conservation():
    """ds3-sized showers (45 x 900 voxels), 256 of them: reverse(forward(x)) = x where the chain is invertible; layer energies and the
    incident energy come back; the sampled side conserves E_inc * u_0 exactly as the reference's recurrence does."""
    bounds = tuple(range(0, 40501, 900))
    s = TO.ChainSpec(layer_boundaries=bounds, shape=(1, 45, 50, 18), mean=-2.1, std=3.0)
    ch = chain_of(s)
    g = torch.Generator().manual_seed(0)
    B = 256
    dep = torch.exp(torch.randn((B, 40500), generator=g) * 2.0) * (torch.rand((B, 40500), generator=g) < 0.2)
    energy = torch.exp(torch.rand((B, 1), generator=g) * (s.e_max - s.e_min) + s.e_min)
    dep = dep / dep.sum(1, keepdim=True) * energy * 0.8
    x, c = ch.preprocess(dep.to(U.DEV), energy.to(U.DEV))
    assert torch.isfinite(x).all() and torch.isfinite(c).all() and x.shape == (B, 1, 45, 50, 18) and c.shape == (B, 46)
    back, e = ch.postprocess(x, c)
    assert float((e.cpu() / energy - 1).abs().max()) < 1e-5
    layers = lambda t: t.reshape(B, 45, 900).sum(-1)
    assert float(((layers(back.cpu()) - layers(dep)).abs() / (layers(dep) + 1e-3 * energy)).max()) < 2e-4
    keep = dep > 1e-5 * dep.max(1, keepdim=True).values
    assert float((back.cpu()[keep] / dep[keep] - 1).abs().max()) < 2e-3
    # against the oracle on a slice of the same inputs
    xo, co = TO.preprocess(dep[:8], energy[:8], s)
    close(x[:8], xo, atol_frac=2e-6)
    close(c[:8], co, atol_frac=2e-6)
    so, eo = TO.postprocess(xo, co, s)
    close(back[:8], so, rtol=1e-4)


def test_edge_cases_and_errors():
    from vit4hep_amd import _lib

    s = SPECS["transforms_ds1ph_b6"]
    ch = chain_of(s)
    # an all-zero shower: every layer empty -> finite outputs, zero shower back
    x, c = ch.preprocess(torch.zeros((2, 368), device=U.DEV), torch.full((2, 1), 5000.0, device=U.DEV))
    assert torch.isfinite(x).all() and torch.isfinite(c).all()
    back, e = ch.postprocess(x, c)
    assert float(back.abs().max()) == 0.0 and float((e - 5000.0).abs().max()) < 0.1
    with pytest.raises(RuntimeError, match="bad shapes"):
        ch.preprocess(torch.zeros((2, 367), device=U.DEV), torch.ones((2, 1), device=U.DEV))
    with pytest.raises(RuntimeError, match="bad shapes"):
        ch.postprocess(torch.zeros((2, 368), device=U.DEV), torch.zeros((2, 5), device=U.DEV))
    spec = ch._spec()
    spec.n_layers = 1000
    import ctypes as C
    rc = _lib.load().v4h_shape_postprocess(C.byref(spec), _lib.ptr(x), _lib.ptr(x), _lib.ptr(c), _lib.ptr(x), _lib.ptr(c), 2, _lib.stream_ptr())
    assert rc != 0 and b"n_layers" in _lib.load().v4h_last_error()
