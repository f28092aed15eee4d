"""Pins the CPU oracle (oracle/vit_cfm_oracle.py) against vectors produced by the reference
itself (oracle/make_golden.py -> tests/golden/*.npz).  CPU only."""

import numpy as np
import pytest
import torch

from oracle import vit_cfm_oracle as O

CASES = {"ds2_d2_b2": O.ds2(2), "ds2_d6_b2": O.ds2(6), "ds3_d6_b1": O.ds3(6),
         # the other ViT-CFM geometries: multi-segment patching (DS1 / CaloGAN / CaloHad wrappers of the reference), LEMURS
         "ds1_photons_d2_b3": O.ds1_photons(2), "ds1_pions_d2_b2": O.ds1_pions(2), "calogan_d2_b3": O.calogan(2),
         "calohad_d2_b1": O.calohad(2), "lemurs_d2_b2": O.lemurs(2)}
SEEDS = {"ds2_d2_b2": 11, "ds2_d6_b2": 12, "ds3_d6_b1": 13, "ds1_photons_d2_b3": 21, "ds1_pions_d2_b2": 22, "calogan_d2_b3": 23,
         "calohad_d2_b1": 24, "lemurs_d2_b2": 25}
NPARAMS = {"ds2_d2_b2": 9424928, "ds2_d6_b2": 26042528, "ds3_d6_b1": 26082890, "ds1_photons_d2_b3": 9364405, "ds1_pions_d2_b2": 9365365,
           "calogan_d2_b3": 9364406, "calohad_d2_b1": 9457115, "lemurs_d2_b2": 9428288}
RTOL = 2e-5  # fp32 CPU vs fp32 CPU, different op order only


def rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


@pytest.mark.parametrize("name", list(CASES))
def test_param_inventory(name, golden):
    g = golden(name)
    cfg = CASES[name]
    n = sum(int(np.prod(s)) for s in O.param_shapes(cfg).values())
    assert n == int(g["nparams"])
    assert n == NPARAMS[name]


@pytest.mark.parametrize("name", list(CASES))
def test_forward_loss_grads(name, golden):
    g = golden(name)
    cfg = CASES[name]
    p = O.golden_fill(cfg)
    x, c, t, x0 = (torch.from_numpy(g[k]) for k in ("x", "c", "t", "x0"))
    # the synthetic generator is part of the contract too
    seed = SEEDS[name]
    xs, cs, _ = O.synthetic_batch(cfg, x.shape[0], seed)
    assert torch.equal(xs, x) and torch.equal(cs, c)

    xt = (1 - t) * x0 + t * x
    assert rel(O.to_patches(xt, cfg).numpy(), g["patches"]) == 0.0
    assert rel(O.pos_embedding(p["pos_embed_freqs"], cfg).numpy(), g["pos_embed"]) < RTOL
    te = O.timestep_embedding(t.view(-1, 1), cfg.freq_dim)
    te = O.linear(O.silu(O.linear(te, p, "t_embedder.mlp.0")), p, "t_embedder.mlp.2")
    assert rel(te.numpy(), g["t_emb"]) < RTOL
    assert rel(O.vit_forward(p, O.to_patches(xt, cfg), t.view(-1, 1), c, cfg).numpy(), g["tokens_out"]) < RTOL

    loss, v, grads = O.loss_and_grads(p, x, c, t, x0, cfg)
    assert rel(v.numpy(), g["velocity"]) < RTOL
    assert abs(float(loss) - float(g["loss"])) / float(g["loss"]) < RTOL
    names = list(O.param_shapes(cfg))
    norms = np.array([float(grads[k].double().norm()) for k in names])
    assert np.abs(norms - g["grad_norms"]).max() / g["grad_norms"].max() < 5e-5
    for k in names:
        got = grads[k].flatten().double().numpy()[g["gidx/" + k]]
        scale = max(float(np.abs(g["gval/" + k]).max()), 1e-3 * float(g["grad_norms"][names.index(k)]), 1e-12)
        assert np.abs(got - g["gval/" + k]).max() / scale < 2e-4, k


@pytest.mark.parametrize("name,tag,method", [("ds2_d2_b2", "rk4", "rk4"), ("ds2_d2_b2", "heun", "heun2"),
                                             ("ds2_d6_b2", "rk4_coarse", "rk4"), ("ds3_d6_b1", "rk4_coarse", "rk4"),
                                             ("ds1_photons_d2_b3", "rk4_coarse", "rk4"), ("ds1_pions_d2_b2", "heun", "heun2"),
                                             ("calogan_d2_b3", "rk4_coarse", "rk4"), ("calohad_d2_b1", "rk4_coarse", "rk4"),
                                             ("lemurs_d2_b2", "rk4_coarse", "rk4")])
def test_sampler(name, tag, method, golden):
    g = golden(name)
    cfg = CASES[name]
    p = O.golden_fill(cfg)
    s = O.sample(p, torch.from_numpy(g["c"]), torch.from_numpy(g["x_T"]), cfg, method, float(g[f"sample_meta/{tag}"][0]))
    assert rel(s.numpy(), g[f"sample/{tag}"]) < 1e-4


def test_fixed_grid():
    g = O.fixed_grid(0.0, 1.0, 0.05)
    assert len(g) == 21 and float(g[-1]) == 1.0 and float(g[0]) == 0.0
    assert len(O.fixed_grid(0.0, 1.0, 0.25)) == 5


@pytest.mark.parametrize("name", ["ds2_d2_b2", "ds2_d6_b2", "ds1_photons_d2_b3", "calogan_d2_b3", "calohad_d2_b1"])
def test_update_step_trajectory(name, golden):
    """AdamW + clip + cosine LR as in BaseExperiment._step (base_experiment.py:555-597)."""
    g = golden(name)
    cfg = CASES[name]
    p = O.golden_fill(cfg)
    st = O.AdamWState(iterations=int(g["train/iters"]))
    x, c = torch.from_numpy(g["x"]), torch.from_numpy(g["c"])
    for k in range(len(g["train/losses"])):
        loss, gn = O.train_step(p, st, x, c, torch.from_numpy(g["train/t"][k]), torch.from_numpy(g["train/x0"][k]), cfg)
        assert abs(loss - g["train/losses"][k]) / g["train/losses"][k] < 1e-4, (k, loss)
        assert abs(gn - g["train/gnorms"][k]) / g["train/gnorms"][k] < 1e-3, (k, gn)
    D = cfg.hidden_dim
    for k in ("pos_embed_freqs", "blocks.0.attn.qkv.bias", "final_layer.linear.bias"):
        got, want = p[k].numpy(), g["train/final/" + k]
        if k.endswith("qkv.bias"):
            # softmax is invariant to a shift of all scores of a row, so d loss / d (key bias) is analytically ZERO: what both
            # sides hold there is rounding noise, which Adam normalises to +-lr per step.  Bound it instead of comparing it.
            steps = len(g["train/losses"])
            assert np.abs(got[D : 2 * D] - want[D : 2 * D]).max() <= 2 * 1e-4 * steps
            got, want = np.delete(got, np.s_[D : 2 * D]), np.delete(want, np.s_[D : 2 * D])
        assert rel(got, want) < 1e-4, k


def test_hash_fill_is_platform_independent():
    u = O.hash_uniform("x_embedder.weight", 5)
    # integers only -> these exact values on every platform
    assert np.all(np.abs(u) < 1.0)
    assert (u * (1 << 23)).tolist() == [float(int(v)) for v in (u * (1 << 23))]
    v = O.hash_uniform("x_embedder.weight", 5)
    assert np.array_equal(u, v)
    assert not np.array_equal(u, O.hash_uniform("x_embedder.bias", 5))


def test_embedding_mapper_oracle_vs_reference(golden):
    """Fine-tuning mapper (x_embedder = Sequential(Linear, SiLU, x_embedder)): the oracle's optional mapper branch against vectors from
    the reference's modules after the surgery of experiment_finetuning.py:75-171."""
    g = golden("ft_mapper_d2_b2")
    back, new = O.ds2(2), O.ViTConfig(shape=(45, 16, 9), patch_shape=(3, 8, 1), depth=2)
    p = dict(O.golden_fill(back))
    p["x_embedder.2.weight"], p["x_embedder.2.bias"] = p.pop("x_embedder.weight"), p.pop("x_embedder.bias")
    for k, shp in (("x_embedder.0.weight", (48, 24)), ("x_embedder.0.bias", (48,)), ("final_layer.linear.weight", (24, 480)), ("final_layer.linear.bias", (24,)),
                   ("final_layer.adaLN_modulation.1.weight", (960, 480)), ("final_layer.adaLN_modulation.1.bias", (960,))):
        p[k] = O.fill_tensor("ft/" + k, shp)
    loss, v, grads = O.loss_and_grads(p, *(torch.from_numpy(g[k]) for k in ("x", "c", "t", "x0")), new)
    assert rel(v.numpy(), g["velocity"]) < RTOL and abs(float(loss) - float(g["loss"])) / float(g["loss"]) < RTOL
    names = [str(n) for n in g["names"]]
    norms = np.array([float(grads[k].double().norm()) for k in names])
    assert np.abs(norms - g["grad_norms"]).max() / g["grad_norms"].max() < 5e-5


def test_both_mappers_oracle_vs_reference(golden):
    """`map_x_embedding` and `map_c_embedding` together (experiment_finetuning.py:79-119): conditions of width 51 in front of the backbone's
    46-wide c_embedder, patches of width 24 in front of its 48-wide x_embedder."""
    g = golden("ft_xc_mapper_d2_b3")
    back, new = O.ds2(2), O.ViTConfig(shape=(45, 16, 9), patch_shape=(3, 8, 1), depth=2, condition_dim=51)
    p = dict(O.golden_fill(back))
    p["x_embedder.2.weight"], p["x_embedder.2.bias"] = p.pop("x_embedder.weight"), p.pop("x_embedder.bias")
    for a, b in (("c_embedder.0.weight", "c_embedder.2.0.weight"), ("c_embedder.0.bias", "c_embedder.2.0.bias"),
                 ("c_embedder.2.weight", "c_embedder.2.2.weight"), ("c_embedder.2.bias", "c_embedder.2.2.bias")):
        p[b] = p.pop(a)
    for k, shp in (("x_embedder.0.weight", (48, 24)), ("x_embedder.0.bias", (48,)), ("c_embedder.0.weight", (46, 51)), ("c_embedder.0.bias", (46,)),
                   ("final_layer.linear.weight", (24, 480)), ("final_layer.linear.bias", (24,)),
                   ("final_layer.adaLN_modulation.1.weight", (960, 480)), ("final_layer.adaLN_modulation.1.bias", (960,))):
        p[k] = O.fill_tensor("ft/" + k, shp)
    loss, v, grads = O.loss_and_grads(p, *(torch.from_numpy(g[k]) for k in ("x", "c", "t", "x0")), new)
    assert g["c"].shape == (3, 51)
    assert rel(v.numpy(), g["velocity"]) < RTOL and abs(float(loss) - float(g["loss"])) / float(g["loss"]) < RTOL
    names = [str(n) for n in g["names"]]
    assert "c_embedder.0.weight" in names and "c_embedder.2.2.bias" in names
    norms = np.array([float(grads[k].double().norm()) for k in names])
    assert np.abs(norms - g["grad_norms"]).max() / g["grad_norms"].max() < 5e-5

