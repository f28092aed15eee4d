"""-m gpu: the energy-model velocity field through v4h_energy_forward, against golden vectors made by the
reference's own ParallelTransformer + CFM and against the CPU oracle.  f32 mode <= 1e-4, bf16 mode <= 3e-2 (as everywhere)."""

import numpy as np
import pytest
import torch

from oracle import energy_oracle as E
from tests import hiputil as U

pytestmark = pytest.mark.gpu
CASES = {"energy_ds2_b5": E.EnergyConfig(),
         "energy_small_b3": E.EnergyConfig(dims_in=30, dim_embedding=32, nhead=2, num_encoder_layers=1, num_decoder_layers=2, dim_feedforward=256, encode_t_dim=32)}
TOL = {"f32": 1e-4, "bf16": 3e-2}
METHOD = {"rk4": "rk4", "heun": "heun2", "rk4_coarse": "rk4"}


def build(cfg, mode, fill=None):
    from vit4hep_amd import CFM
    from vit4hep_amd.nn.cfm.transformer_cfm import ParallelTransformer

    net = ParallelTransformer({"dims_in": cfg.dims_in, "dims_c": cfg.dims_c, "dim_embedding": cfg.dim_embedding, "nhead": cfg.nhead,
                               "num_encoder_layers": cfg.num_encoder_layers, "num_decoder_layers": cfg.num_decoder_layers,
                               "dim_feedforward": cfg.dim_feedforward, "embeds": True, "encode_t_scale": cfg.encode_t_scale,
                               "encode_t_dim": cfg.encode_t_dim, "amd_mode": mode})
    model = CFM(net, "uniform", "linear", {"method": "rk4", "options": {"step_size": 0.05}}, shape=[cfg.dims_in])
    sd = model.state_dict()
    for k, v in (fill or E.golden_fill(cfg)).items():
        assert sd["net." + k].shape == v.shape, k
        sd["net." + k] = v.clone()
    sd["net.layers.0.weight"], sd["net.layers.0.bias"] = sd["net.layer.weight"].clone(), sd["net.layer.bias"].clone()
    model.load_state_dict(sd)
    model.device, model.dtype = torch.device(U.DEV), torch.float32
    return model.to(U.DEV).eval()


@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("name", list(CASES))
def test_velocity_vs_golden(name, mode, golden):
    g, cfg = golden(name), CASES[name]
    model = build(cfg, mode)
    x, t, c = (torch.from_numpy(g[k]).to(U.DEV) for k in ("x", "t", "c"))
    with torch.no_grad():
        v = model.forward(x, t, c)
        assert v.shape == g["velocity"].shape and U.rel_err(v, torch.from_numpy(g["velocity"])) < TOL[mode]
        t2, x0 = torch.from_numpy(g["loss_t"]).to(U.DEV), torch.from_numpy(g["loss_x0"]).to(U.DEV)
        loss = model._loss_from_noise(x, c, t2, x0)  # trajectory + MSE kernels around the network
    assert abs(loss.item() - float(g["loss"])) / float(g["loss"]) < TOL[mode]


@pytest.mark.parametrize("name,tag", [("energy_ds2_b5", "rk4"), ("energy_ds2_b5", "heun"), ("energy_small_b3", "rk4_coarse")])
def test_sampler_vs_golden(name, tag, golden):
    g, cfg = golden(name), CASES[name]
    model = build(cfg, "f32")
    model.odeint_kwargs = {"method": METHOD[tag], "options": {"step_size": float(g["sample_meta/" + tag][0])}}
    with torch.inference_mode():
        s = model._sample_from(torch.from_numpy(g["x_T"]).to(U.DEV), torch.from_numpy(g["c"]).to(U.DEV))
    assert U.rel_err(s, torch.from_numpy(g["sample/" + tag])) < 1e-4
    torch.manual_seed(1)
    a = model.sample_batch(torch.from_numpy(g["c"]).to(U.DEV))  # reference entry point (models/base_model.py:220-244)
    assert a.shape == g["x_T"].shape and torch.isfinite(a).all()


@pytest.mark.parametrize("B", [1, 7, 64, 300])
def test_batch_sizes_vs_oracle(B):
    cfg = CASES["energy_ds2_b5"]
    fill = E.golden_fill(cfg)
    model = build(cfg, "f32", fill)
    g = torch.Generator().manual_seed(B)
    x, t, c = torch.randn((B, 45), generator=g), torch.rand((B, 1), generator=g), torch.rand((B, 1), generator=g)
    with torch.no_grad():
        v = model.forward(x.to(U.DEV), t.to(U.DEV), c.to(U.DEV))
    assert U.rel_err(v, E.energy_forward(fill, x, t, c, cfg)) < 1e-4


@pytest.mark.parametrize("dims_in", [45, 5, 7])
def test_bf16_resident_decoder_and_composed_path_agree(dims_in):
    """bf16 mode runs the one-launch resident decoder; the kernel-per-operator path is forced for comparison: both against the f32 oracle
    and against each other on the same samples (they round at different places, so not bit-equal).  dims_in 5 / 7 = the ds1 energy models."""
    cfg = E.EnergyConfig(dims_in=dims_in)
    fill = E.golden_fill(cfg)
    model = build(cfg, "bf16", fill)
    g = torch.Generator().manual_seed(dims_in)
    B = 700
    x, t, c = torch.randn((B, dims_in), generator=g), torch.rand((B, 1), generator=g), torch.rand((B, 1), generator=g)
    ref = E.energy_forward(fill, x, t, c, cfg)
    with torch.no_grad():
        fused = model.forward(x.to(U.DEV), t.to(U.DEV), c.to(U.DEV))
        model.net.force_composed = True                                               # V4H_ENERGY_COMPOSED
        composed = model.forward(x.to(U.DEV), t.to(U.DEV), c.to(U.DEV))
        model.net.force_composed = False
    assert U.rel_err(composed, ref) < 3e-2 and U.rel_err(fused, ref) < 3e-2
    assert U.rms_err(fused, ref) < 1e-2 and U.rms_err(composed, ref) < 1e-2
    assert U.rel_err(fused, composed) < 3e-2 and not torch.equal(fused, composed)


def test_condition_and_weight_caches_never_go_stale():
    """The encoder output / cross-attention terms are cached per condition TENSOR OBJECT, the operand copies per parameter version."""
    cfg = CASES["energy_ds2_b5"]
    fill = E.golden_fill(cfg)
    model = build(cfg, "bf16", fill)
    ref32 = build(cfg, "bf16", fill)
    net = model.net
    g = torch.Generator().manual_seed(0)
    x, t = torch.randn((16, 45), generator=g).to(U.DEV), torch.rand((16, 1), generator=g).to(U.DEV)
    c1 = torch.rand((16, 1), generator=g).to(U.DEV)
    with torch.no_grad():
        a = model.forward(x, t, c1)
        b = model.forward(x, t, c1)           # same object: encoder skipped
        assert torch.equal(a, b)
        c2 = c1.clone()                        # same values, other object: recomputed, same result
        assert torch.equal(model.forward(x, t, c2), a)
        c1.mul_(0.5)                           # in-place change of the cached object: version bump -> recomputed
        d = model.forward(x, t, c1)
        assert not torch.equal(d, a) and torch.equal(d, ref32.forward(x, t, c1.clone()))
        # a NEW tensor that happens to reuse the old storage address must not be mistaken for the old condition
        addr = c2.data_ptr()
        del c2
        c3 = torch.full((16, 1), 0.123, device=U.DEV)
        e = model.forward(x, t, c3)
        assert torch.equal(e, ref32.forward(x, t, torch.full((16, 1), 0.123, device=U.DEV))), (addr, c3.data_ptr())
        net.transformer.decoder.layers[0].linear1.weight.mul_(1.25)  # weight update -> operand copies refreshed
        f = model.forward(x, t, c3)
        assert not torch.equal(e, f)
        fresh_fill = {k: v.clone() for k, v in fill.items()}
        fresh_fill["transformer.decoder.layers.0.linear1.weight"] *= 1.25
        assert torch.equal(f, build(cfg, "bf16", fresh_fill).forward(x, t, c3))


def test_errors_are_loud():
    from vit4hep_amd import _lib

    cfg = CASES["energy_ds2_b5"]
    model = build(cfg, "f32")
    x, t, c = torch.zeros((4, 45), device=U.DEV), torch.zeros((4, 1), device=U.DEV), torch.zeros((4, 1), device=U.DEV)
    with torch.no_grad():
        with pytest.raises(RuntimeError, match="bad shapes"):
            model.forward(x[:, :44], t, c)
        with pytest.raises(RuntimeError, match="bad shapes"):
            model.forward(x, t, torch.zeros((4, 2), device=U.DEV))
        with pytest.raises(NotImplementedError):
            model.net(x, t, None)
    with pytest.raises(NotImplementedError, match="forward-only"):
        model.forward(x, t, c)  # gradients enabled
    plan = model.net._get_plan()
    ws = torch.empty(16, dtype=torch.uint8, device=U.DEV)
    params = _lib.pointer_table([p.detach() for p in model.net.parameter_list()])
    rc = _lib.load().v4h_energy_forward(plan.handle, 4, params, _lib.ptr(x), _lib.ptr(t), _lib.ptr(c), _lib.ptr(x), _lib.ptr(ws), 16, 0, _lib.stream_ptr())
    assert rc != 0 and b"workspace too small" in _lib.load().v4h_last_error()
    rc = _lib.load().v4h_energy_forward(plan.handle, 4, params, _lib.ptr(x), _lib.ptr(t), _lib.ptr(c), _lib.ptr(x), _lib.ptr(ws), 16, 1, _lib.stream_ptr())
    assert rc != 0 and b"forward-only" in _lib.load().v4h_last_error()
