"""Pins the energy-model oracle (oracle/energy_oracle.py) against vectors produced by the reference's own
nn.cfm.transformer_cfm.ParallelTransformer inside models.base_model.CFM (oracle/make_golden.py), and checks the host mirror's
reference-visible surface.  CPU only."""

import numpy as np
import pytest
import torch

from oracle import energy_oracle as E

CASES = {"energy_ds2_b5": E.EnergyConfig(),
         "energy_small_b3": E.EnergyConfig(dims_in=30, dim_embedding=32, nhead=2, num_encoder_layers=1, num_decoder_layers=2, dim_feedforward=256, encode_t_dim=32)}
NPARAMS = {"energy_ds2_b5": 1958817, "energy_small_b3": 211121}
METHOD = {"rk4": "rk4", "heun": "heun2", "rk4_coarse": "rk4"}


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_matches_the_reference_module(name, golden):
    g, cfg = golden(name), CASES[name]
    p = E.golden_fill(cfg)
    assert sum(int(np.prod(s)) for s in E.param_shapes(cfg).values()) == int(g["nparams"]) == NPARAMS[name]
    x, t, c = (torch.from_numpy(g[k]) for k in ("x", "t", "c"))
    assert rel(E.time_embed(p, t).numpy(), g["t_emb"]) < 2e-5
    assert rel(E.energy_forward(p, x, t, c, cfg).numpy(), g["velocity"]) < 2e-5
    t2, x0 = torch.from_numpy(g["loss_t"]), torch.from_numpy(g["loss_x0"])
    v = E.energy_forward(p, (1 - t2) * x0 + t2 * x, t2, c, cfg)  # models/base_model.py:209-218
    assert abs(float(((v - (x - x0)) ** 2).mean()) - float(g["loss"])) / float(g["loss"]) < 2e-5
    for k in list(g):
        if k.startswith("sample/"):
            tag = k[7:]
            s = E.energy_sample(p, c, torch.from_numpy(g["x_T"]), cfg, METHOD[tag], float(g["sample_meta/" + tag][0]))
            assert rel(s.numpy(), g[k]) < 1e-5, tag


def test_one_token_memory_identities():
    """What the HIP path exploits: with ONE memory token every attention over it is out_proj(v_proj(.)), independent of q and k."""
    cfg = CASES["energy_small_b3"]
    p = E.golden_fill(cfg)
    d = cfg.d_model
    g = torch.Generator().manual_seed(0)
    m = torch.randn((4, 1, d), generator=g)
    h = torch.randn((4, cfg.dims_in, d), generator=g)
    pre = "transformer.decoder.layers.0.multihead_attn"
    full = E._mha(p, pre, h, m, cfg.nhead)
    W, b = p[pre + ".in_proj_weight"], p[pre + ".in_proj_bias"]
    short = ((m @ W[2 * d :].T + b[2 * d :]) @ p[pre + ".out_proj.weight"].T + p[pre + ".out_proj.bias"]).expand(-1, cfg.dims_in, -1)
    assert rel(full.numpy(), short.numpy()) < 1e-6


def test_host_mirror_surface():
    from vit4hep_amd import _lib
    from vit4hep_amd.nn.cfm.transformer_cfm import ParallelTransformer

    param = {"dims_in": 45, "dims_c": 1, "dim_embedding": 64, "nhead": 4, "num_encoder_layers": 4, "num_decoder_layers": 4, "dim_feedforward": 512,
             "dropout": 0.0, "activation": "relu", "embeds": True, "encode_t_scale": 30, "encode_t_dim": 64}
    net = ParallelTransformer(param)
    cfg = CASES["energy_ds2_b5"]
    assert [k for k, _ in net.named_parameters()] == list(E.param_shapes(cfg))
    sd = net.state_dict()
    assert "layers.0.weight" in sd and sd["layers.0.weight"].data_ptr() == sd["layer.weight"].data_ptr()  # shared tensor, both keys as in the reference
    assert not net.time_embed[0].W.requires_grad
    assert net._get_plan().shapes == [tuple(s) for s in E.param_shapes(cfg).values()]
    assert net._get_plan().workspace_bytes(256) > net._get_plan().workspace_bytes(8) > 0
    with pytest.raises(RuntimeError, match="MI355X"):  # no CPU path
        with torch.no_grad():
            net(torch.zeros(2, 45), torch.zeros(2, 1), torch.zeros(2, 1))
    with pytest.raises(NotImplementedError, match="forward-only"):
        net(torch.zeros(2, 45), torch.zeros(2, 1), torch.zeros(2, 1))
    for bad in ({"embeds": False}, {"dropout": 0.1}, {"activation": "gelu"}):
        with pytest.raises(NotImplementedError):
            ParallelTransformer({**param, **bad})
    for bad, msg in (({"dims_c": 2}, "dims_c"), ({"nhead": 8}, "head_dim"), ({"encode_t_dim": 32}, "encode_t_dim"), ({"dims_in": 100}, "dims_in")):
        with pytest.raises(RuntimeError, match=msg):
            _lib.EnergyPlan(*[({**param, **bad})[k] for k in ("dims_in", "dims_c", "dim_embedding", "nhead", "num_encoder_layers", "num_decoder_layers",
                                                             "dim_feedforward", "encode_t_dim")])
