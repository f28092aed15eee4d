"""-m gpu: the other ViT-CFM geometries of the reference through the same HIP path:
multi-segment patching (CaloChallengeCFM_DS1 photons / pions, CaloGANCFM, CaloHadCFM) as an index-map geometry, and LEMURSCFM
(the ds2 grid with 53 conditions).  Checked against golden vectors produced by the reference's own wrapper classes
(oracle/make_golden.py) and against the CPU oracle.  Tolerances as in test_hip_network.py.
"""

import numpy as np
import pytest
import torch

from oracle import vit_cfm_oracle as O
from tests import hiputil as U

pytestmark = pytest.mark.gpu
CASES = {  # fixture -> (oracle config, wrapper kind)
    "ds1_photons_d2_b3": (O.ds1_photons(2), "ds1"),
    "ds1_pions_d2_b2": (O.ds1_pions(2), "ds1"),
    "calogan_d2_b3": (O.calogan(2), "calogan"),
    "calohad_d2_b1": (O.calohad(2), "calohad"),
    "lemurs_d2_b2": (O.lemurs(2), "lemurs"),
}
TOL = {"f32": 1e-4, "bf16": 3e-2}
GRAD_TOL = {"f32": 3e-4, "bf16": 6e-2}


def _inputs(g):
    return tuple(torch.from_numpy(g[k]).to(U.DEV) for k in ("x", "c", "t", "x0"))


@pytest.mark.parametrize("name", list(CASES))
def test_patching_and_positions_vs_golden(name, golden):
    """to_patches is a pure permutation: bit-exact against the reference's split / rearrange / cat; from_patches inverts it;
    the positional table follows the module's pos_x/y/z buffers (multi-segment meshgrid)."""
    g = golden(name)
    cfg, kind = CASES[name]
    fill = O.golden_fill(cfg)
    model = U.build_models(cfg, "f32", fill, kind=kind)
    x, c, t, x0 = _inputs(g)
    x_t = (1 - t) * x0 + t * x
    tok = model.to_patches(x_t)
    assert tok.shape == g["patches"].shape and torch.equal(tok.cpu(), torch.from_numpy(g["patches"]))
    assert torch.equal(model.from_patches(tok), x_t)
    core = model._core()
    pz, py, px = O.meshgrid_buffers(cfg)
    assert torch.equal(core.pos_z.cpu(), pz) and torch.equal(core.pos_y.cpu(), py) and torch.equal(core.pos_x.cpu(), px)
    assert U.rel_err(core.learnable_pos_embedding(), torch.from_numpy(g["pos_embed"])) < 2e-5


@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("name", list(CASES))
def test_forward_loss_grads_vs_golden(name, mode, golden):
    g = golden(name)
    cfg, kind = CASES[name]
    model = U.build_models(cfg, mode, O.golden_fill(cfg), kind=kind)
    x, c, t, x0 = _inputs(g)
    model.train()
    loss = model._loss_from_noise(x, c, t, x0)
    loss.backward()
    with torch.no_grad():
        x_t = (1 - t) * x0 + t * x
        v = model.forward(x_t, t.view(-1, 1), c)
        tok = model.net(model.to_patches(x_t), t.view(-1, 1), c)  # the reference's own call form: tokens in, tokens out
    assert v.shape == g["velocity"].shape and U.rel_err(v, torch.from_numpy(g["velocity"])) < TOL[mode]
    assert tok.shape == g["tokens_out"].shape and U.rel_err(tok, torch.from_numpy(g["tokens_out"])) < TOL[mode]
    assert abs(loss.item() - float(g["loss"])) / float(g["loss"]) < TOL[mode]
    grads = U.named_grads(model)
    names = list(O.param_shapes(cfg))
    norms = np.array([float(grads[k].double().norm()) for k in names])
    worst = float(np.abs(norms - g["grad_norms"]).max() / g["grad_norms"].max())
    assert worst < GRAD_TOL[mode], f"gradient norms off by {worst}"
    tot = float(np.sqrt((norms**2).sum()))
    assert abs(tot - float(g["grad_total_norm"])) / float(g["grad_total_norm"]) < GRAD_TOL[mode]
    for k in names:
        got = grads[k].flatten().double().cpu().numpy()[g["gidx/" + k]]
        scale = max(float(np.abs(g["gval/" + k]).max()), 1e-2 * float(g["grad_norms"][names.index(k)]), 1e-12)
        assert np.abs(got - g["gval/" + k]).max() / scale < (2e-3 if mode == "f32" else 0.25), k


@pytest.mark.parametrize("name", ["ds1_photons_d2_b3", "calogan_d2_b3", "calohad_d2_b1"])
def test_full_gradients_vs_oracle(name, golden):
    """Every element of every gradient tensor, f32 mode, against the oracle's autograd on the same inputs."""
    g = golden(name)
    cfg, kind = CASES[name]
    fill = O.golden_fill(cfg)
    model = U.build_models(cfg, "f32", fill, kind=kind)
    x, c, t, x0 = _inputs(g)
    model._loss_from_noise(x, c, t, x0).backward()
    _, _, ref = O.loss_and_grads(fill, *(torch.from_numpy(g[k]) for k in ("x", "c", "t", "x0")), cfg)
    grads = U.named_grads(model)
    gmax = max(float(v.abs().max()) for v in ref.values())
    for k, r in ref.items():
        scale = max(float(r.abs().max()), 1e-3 * gmax)
        assert float((grads[k].cpu() - r).abs().max()) / scale < 1e-3, k


@pytest.mark.parametrize("name,tag", [("ds1_photons_d2_b3", "rk4_coarse"), ("ds1_pions_d2_b2", "heun"), ("calogan_d2_b3", "rk4_coarse"),
                                      ("calohad_d2_b1", "rk4_coarse"), ("lemurs_d2_b2", "rk4_coarse")])
def test_sampler_vs_golden(name, tag, golden):
    g = golden(name)
    cfg, kind = CASES[name]
    model = U.build_models(cfg, "f32", O.golden_fill(cfg), kind=kind).eval()
    model.odeint_kwargs = {"method": {"rk4_coarse": "rk4", "heun": "heun2"}[tag], "options": {"step_size": float(g[f"sample_meta/{tag}"][0])}}
    with torch.inference_mode():
        s = model._sample_from(torch.from_numpy(g["x_T"]).to(U.DEV), torch.from_numpy(g["c"]).to(U.DEV))
    assert s.shape == g[f"sample/{tag}"].shape and U.rel_err(s, torch.from_numpy(g[f"sample/{tag}"])) < 1e-4
    torch.manual_seed(3)
    a = model.sample_batch(torch.from_numpy(g["c"]).to(U.DEV))  # the reference entry point: draws x_T itself
    assert a.shape == g["x_T"].shape and torch.isfinite(a).all()


@pytest.mark.parametrize("name", ["ds1_photons_d2_b3", "calogan_d2_b3", "calohad_d2_b1"])
def test_update_step_trajectory_vs_golden(name, golden):
    """The fused trainer (staged backward, clip, AdamW, cosine LR) on a mapped geometry."""
    from vit4hep_amd.trainer import CFMTrainer

    g = golden(name)
    cfg, kind = CASES[name]
    model = U.build_models(cfg, "f32", O.golden_fill(cfg), kind=kind)
    tr = CFMTrainer(model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.1, clip_grad_norm=1000.0, iterations=int(g["train/iters"]))
    x, c = torch.from_numpy(g["x"]).to(U.DEV), torch.from_numpy(g["c"]).to(U.DEV)
    for k in range(len(g["train/losses"])):
        loss, gn = tr.step(x, c, torch.from_numpy(g["train/t"][k]).to(U.DEV), torch.from_numpy(g["train/x0"][k]).to(U.DEV))
        assert abs(loss.item() - g["train/losses"][k]) / g["train/losses"][k] < 1e-4, (k, loss.item())
        assert abs(gn.item() - g["train/gnorms"][k]) / g["train/gnorms"][k] < 1e-3, k
    sd = model.state_dict()
    for k in ("pos_embed_freqs", "final_layer.linear.bias"):
        assert U.rel_err(sd["net." + k], torch.from_numpy(g["train/final/" + k])) < 1e-4, k


def test_lemurs_batch_loss_takes_the_dataset_layout(golden):
    """LEMURSCFM._batch_loss reorders (B, R, A, L) showers to (B, 1, L, A, R) itself (reference lemurs/model.py:62-65)."""
    g = golden("lemurs_d2_b2")
    cfg, kind = CASES["lemurs_d2_b2"]
    model = U.build_models(cfg, "f32", O.golden_fill(cfg), kind=kind).train()
    x, c, t, x0 = _inputs(g)
    want = model._loss_from_noise(x, c, t, x0).item()
    assert abs(want - float(g["loss"])) / float(g["loss"]) < 1e-4
    torch.manual_seed(0)
    loss = model._batch_loss([x[:, 0].permute(0, 3, 2, 1).contiguous().cpu(), c.cpu()])
    assert loss.requires_grad and torch.isfinite(loss)
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())


def test_standalone_multisegment_vit_on_tokens(golden):
    """A ViT built straight from the YAML's `num_patches` (no wrapper) accepts patch tokens like the reference's."""
    g = golden("calogan_d2_b3")
    cfg, _ = CASES["calogan_d2_b3"]
    net = U.build_net(cfg, "f32")
    sd = net.state_dict()
    for k, v in O.golden_fill(cfg).items():
        sd[k] = v.clone()
    net.load_state_dict(sd)
    net = net.to(U.DEV)
    with torch.no_grad():
        out = net(torch.from_numpy(g["patches"]).to(U.DEV), torch.from_numpy(g["t"]).to(U.DEV).view(-1, 1), torch.from_numpy(g["c"]).to(U.DEV))
    assert U.rel_err(out, torch.from_numpy(g["tokens_out"])) < 1e-4


def test_mapped_geometry_errors_are_loud():
    from vit4hep_amd import _lib

    cfg, kind = CASES["ds1_photons_d2_b3"]
    model = U.build_models(cfg, "f32", O.golden_fill(cfg), kind=kind)
    c = torch.zeros((2, cfg.condition_dim), device=U.DEV)
    t = torch.zeros((2, 1), device=U.DEV)
    with pytest.raises(RuntimeError, match="does not match"):
        model.forward(torch.zeros((2, 1, 441), device=U.DEV), t, c)
    with pytest.raises(RuntimeError, match="does not match"):
        model.to_patches(torch.zeros((2, 1, 439), device=U.DEV))
    # C ABI: a mapped plan refuses to run without its tables, a grid plan refuses a map
    core = model._core()
    plan = core._get_plan()
    assert plan.mapped
    x = torch.zeros((2, 1, 440), device=U.DEV)
    tok = torch.zeros((2, 88, 5), device=U.DEV)
    rc = _lib.load().v4h_op_patchify(plan.handle, _lib.ptr(x), _lib.ptr(tok), 2, _lib.stream_ptr(), None)
    assert rc != 0 and b"d_patch_map" in _lib.load().v4h_last_error()
    good = core._patch_map.copy()
    core.set_patch_map(np.zeros((3, 5), np.int32), 440)  # shape is checked against the live network when the plan is (re)built
    with pytest.raises(ValueError, match="patch map"):
        core._get_plan()
    core.set_patch_map(good, 440)
    assert core._get_plan().mapped
    with pytest.raises(ValueError, match=r"\[-1, 440\)"):
        core.set_patch_map(np.full((88, 5), 440, np.int32), 440)


def test_patch_map_with_holes_leaves_unmapped_voxels_zero():
    """-1 entries (no voxel) read as 0 and are skipped by the scatter; voxels no token covers come out as 0."""
    cfg, kind = CASES["calogan_d2_b3"]
    model = U.build_models(cfg, "f32", O.golden_fill(cfg), kind=kind)
    core = model._core()
    pm = core._patch_map.copy()
    pm[0, :3] = -1
    core.set_patch_map(pm, 504)
    model._patch_map = pm  # keep the wrapper and the net consistent
    assert core.map_has_holes()
    x = torch.randn((2, 1, 504), device=U.DEV)
    tok = model.to_patches(x)
    assert torch.all(tok[:, 0, :3] == 0)
    back = model.from_patches(tok)
    missing = torch.from_numpy(np.setdiff1d(np.arange(504), pm[pm >= 0])).to(U.DEV)
    assert torch.all(back[:, 0, missing] == 0)
    keep = torch.from_numpy(pm[pm >= 0].astype(np.int64)).to(U.DEV)
    assert torch.equal(back[:, 0, keep], x[:, 0, keep])
