"""-m gpu: the fine-tuning surgery the reference performs on a trained network (SURVEY.md 8f row 4;
experiments/calochallenge/calochallenge_cfm/experiment_finetuning.py:75-205) with the flags of four of its five shipped fine-tuning configs
(`interpolate`, `reinitialize_pos_embedding`, `reinitialize_final_layer`): the wrapper of the NEW dataset is built around the BACKBONE's
network, then embedder weights are interpolated to the new input widths, position buffers re-meshed, the head replaced.  The HIP path
follows the live modules; results are checked against the oracle evaluated with the same (post-surgery) tensors."""

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import vit_cfm_oracle as O
from tests import hiputil as U

pytestmark = pytest.mark.gpu


def surgery(model, new_num_patches, new_patch_dim, new_condition_dim, device):
    """The statements of add_embedding_layers() for map_*=False, reinitialize_{x,c}_embedding=False, interpolate=True,
    reinitialize_pos_embedding=True, reinitialize_final_layer=True."""
    from vit4hep_amd.nn.vit import FinalLayer  # the reference imports it from nn.vit (experiment_finetuning.py:15)

    net = model.net
    w = nn.functional.interpolate(net.x_embedder.weight.unsqueeze(1), size=new_patch_dim, mode="linear").squeeze(1)
    net.x_embedder.weight.data = w.data
    cw = nn.functional.interpolate(net.c_embedder[0].weight.unsqueeze(1), size=new_condition_dim, mode="linear").squeeze(1)
    net.c_embedder[0].weight.data = cw.data
    net.num_patches = new_num_patches
    pos_z, pos_y, pos_x = net.create_meshgrid()
    net.pos_z, net.pos_y, net.pos_x = pos_z.to(device, torch.float32), pos_y.to(device, torch.float32), pos_x.to(device, torch.float32)
    net.final_layer = FinalLayer(int(net.hidden_dim), new_patch_dim, int(net.out_channels)).to(device, torch.float32)
    g = torch.Generator().manual_seed(5)  # a zero-initialised head would make every check trivial
    with torch.no_grad():
        for p in net.final_layer.parameters():
            p.copy_((torch.rand(p.shape, generator=g) * 2 - 1).to(device) * 0.05)


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_ds2_backbone_fine_tuned_on_a_ds3_like_grid(mode):
    from vit4hep_amd import CaloChallengeCFM

    back = O.ds2(2)
    new = O.ViTConfig(shape=(45, 50, 18), patch_shape=(3, 10, 3), condition_dim=50, depth=2)
    net = U.build_net(back, mode)  # the backbone's network: 135 tokens of 48, 46 conditions
    model = CaloChallengeCFM(net, list(new.patch_shape), in_channels=1, odeint_kwargs={"method": "rk4", "options": {"step_size": 0.5}}, shape=list(new.shape))
    sd = model.state_dict()
    for k, v in O.golden_fill(back).items():
        sd["net." + k] = v.clone()
    model.load_state_dict(sd)
    model.device, model.dtype = torch.device(U.DEV), torch.float32
    model = model.to(U.DEV)
    x, c, g = O.synthetic_batch(new, 2, 77)
    t, x0 = O.synthetic_noise(new, 2, g)
    with pytest.raises((ValueError, RuntimeError)):  # before the surgery the wrapper's geometry does not fit the backbone network
        model._loss_from_noise(x.to(U.DEV), c.to(U.DEV), t.to(U.DEV), x0.to(U.DEV))
    surgery(model, [[15, 5, 6]], new.P, new.condition_dim, U.DEV)
    assert model.net.patch_dim == 90 and model.net.condition_dim == 50 and model.net.num_tokens == 450

    loss = model._loss_from_noise(x.to(U.DEV), c.to(U.DEV), t.to(U.DEV), x0.to(U.DEV))
    loss.backward()
    params = {k[4:]: v.detach().cpu() for k, v in model.named_parameters()}
    assert list(params) == list(O.param_shapes(new)) and all(tuple(params[k].shape) == tuple(s) for k, s in O.param_shapes(new).items())
    ref_loss, _, ref_grads = O.loss_and_grads(params, x, c, t, x0, new)
    tol = 1e-4 if mode == "f32" else 3e-2
    assert abs(loss.item() - float(ref_loss)) / float(ref_loss) < tol
    grads = U.named_grads(model)
    gmax = max(float(v.abs().max()) for v in ref_grads.values())
    for k, r in ref_grads.items():
        scale = max(float(r.abs().max()), 1e-3 * gmax)
        assert float((grads[k].cpu() - r).abs().max()) / scale < (1e-3 if mode == "f32" else 0.25), k

    # parameter groups with their own learning rates (experiment_finetuning.py:173-205), one optimizer step through the unchanged-caller path
    net = model.net
    groups = [{"params": list(net.t_embedder.parameters()) + list(net.blocks.parameters()), "lr": 1e-4},
              {"params": list(net.final_layer.parameters()), "lr": 5e-4},
              {"params": list(net.x_embedder.parameters()) + list(net.c_embedder.parameters()) + [net.pos_embed_freqs], "lr": 5e-4}]
    assert sum(len(gr["params"]) for gr in groups) == len(list(model.parameters()))
    opt = torch.optim.AdamW(groups, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.1)
    before = net.x_embedder.weight.detach().clone()
    opt.step()
    assert not torch.equal(before, net.x_embedder.weight)
    with torch.inference_mode():
        s = model.sample_batch(c.to(U.DEV))
    assert s.shape == (2, 1, 45, 50, 18) and torch.isfinite(s).all()


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_embedding_mapper_vs_reference_vectors(mode, golden):
    """`map_x_embedding` (configs/calochallenge/finetuning/calochallenge_ds2tods3_ft.yaml:35): x_embedder becomes
    Sequential(Linear(P_new -> P_old), SiLU, x_embedder); golden vectors from the reference's own modules after the same surgery."""
    from vit4hep_amd import CaloChallengeCFM
    from vit4hep_amd.nn.vit import FinalLayer

    g = golden("ft_mapper_d2_b2")
    back, new = O.ds2(2), O.ViTConfig(shape=(45, 16, 9), patch_shape=(3, 8, 1), depth=2)
    net = U.build_net(back, mode)
    model = CaloChallengeCFM(net, list(new.patch_shape), in_channels=1, odeint_kwargs={"method": "rk4", "options": {"step_size": 0.5}}, shape=list(new.shape))
    sd = model.state_dict()
    for k, v in O.golden_fill(back).items():
        sd["net." + k] = v.clone()
    model.load_state_dict(sd)
    model.device, model.dtype = torch.device(U.DEV), torch.float32
    model = model.to(U.DEV)
    net = model.net
    net.x_embedder = nn.Sequential(nn.Linear(new.P, back.P), nn.SiLU(), net.x_embedder).to(U.DEV, torch.float32)  # experiment_finetuning.py:80-91
    net.num_patches = [list(new.num_patches)]
    pos_z, pos_y, pos_x = net.create_meshgrid()
    net.pos_z, net.pos_y, net.pos_x = pos_z.to(U.DEV), pos_y.to(U.DEV), pos_x.to(U.DEV)
    net.final_layer = FinalLayer(back.hidden_dim, new.P, 1).to(U.DEV, torch.float32)
    with torch.no_grad():
        for k, p_ in model.named_parameters():
            if k.startswith("net.x_embedder.0.") or k.startswith("net.final_layer."):
                p_.copy_(O.fill_tensor("ft/" + k[4:], tuple(p_.shape)).to(U.DEV))
    assert net.patch_dim == 24 and net.x_embed_in() == 48 and net.num_tokens == 270
    x, c, t, x0 = (torch.from_numpy(g[k]).to(U.DEV) for k in ("x", "c", "t", "x0"))
    loss = model._loss_from_noise(x, c, t, x0)
    loss.backward()
    tol = 1e-4 if mode == "f32" else 3e-2
    assert abs(loss.item() - float(g["loss"])) / float(g["loss"]) < tol
    with torch.no_grad():
        v = model.forward((1 - t) * x0 + t * x, t.view(-1, 1), c)
    assert U.rel_err(v, torch.from_numpy(g["velocity"])) < tol
    grads = U.named_grads(model)
    names = [str(n) for n in g["names"]]
    assert names == [k[4:] for k, _ in model.named_parameters()]
    norms = np.array([float(grads[k].double().norm()) for k in names])
    assert float(np.abs(norms - g["grad_norms"]).max() / g["grad_norms"].max()) < (3e-4 if mode == "f32" else 6e-2)
    for k in names:
        got = grads[k].flatten().double().cpu().numpy()[g["gidx/" + k]]
        scale = max(float(np.abs(g["gval/" + k]).max()), 1e-2 * float(g["grad_norms"][names.index(k)]), 1e-12)
        assert np.abs(got - g["gval/" + k]).max() / scale < (2e-3 if mode == "f32" else 0.25), k
    with torch.inference_mode():
        s = model.sample_batch(c)
    assert s.shape == (2, 1, 45, 16, 9) and torch.isfinite(s).all()
    from vit4hep_amd.trainer import CFMTrainer

    with pytest.raises(NotImplementedError, match="embedding mapper"):
        CFMTrainer(model)


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_both_mappers_vs_reference_vectors(mode, golden):
    """`map_x_embedding` + `map_c_embedding` (experiment_finetuning.py:79-119): c_embedder becomes Sequential(Linear(51 -> 46), SiLU, c_embedder);
    golden vectors from the reference's own modules after the same surgery."""
    from vit4hep_amd import CaloChallengeCFM
    from vit4hep_amd.nn.vit import FinalLayer

    g = golden("ft_xc_mapper_d2_b3")
    back, new = O.ds2(2), O.ViTConfig(shape=(45, 16, 9), patch_shape=(3, 8, 1), depth=2, condition_dim=51)
    net = U.build_net(back, mode)
    model = CaloChallengeCFM(net, list(new.patch_shape), in_channels=1, odeint_kwargs={"method": "rk4", "options": {"step_size": 0.5}}, shape=list(new.shape))
    sd = model.state_dict()
    for k, v in O.golden_fill(back).items():
        sd["net." + k] = v.clone()
    model.load_state_dict(sd)
    model.device, model.dtype = torch.device(U.DEV), torch.float32
    model = model.to(U.DEV)
    net = model.net
    net.x_embedder = nn.Sequential(nn.Linear(new.P, back.P), nn.SiLU(), net.x_embedder).to(U.DEV, torch.float32)
    net.c_embedder = nn.Sequential(nn.Linear(new.condition_dim, back.condition_dim), nn.SiLU(), net.c_embedder).to(U.DEV, torch.float32)
    net.num_patches = [list(new.num_patches)]
    pos_z, pos_y, pos_x = net.create_meshgrid()
    net.pos_z, net.pos_y, net.pos_x = pos_z.to(U.DEV), pos_y.to(U.DEV), pos_x.to(U.DEV)
    net.final_layer = FinalLayer(back.hidden_dim, new.P, 1).to(U.DEV, torch.float32)
    with torch.no_grad():
        for k, p_ in model.named_parameters():
            if k.startswith("net.x_embedder.0.") or k.startswith("net.c_embedder.0.") or k.startswith("net.final_layer."):
                p_.copy_(O.fill_tensor("ft/" + k[4:], tuple(p_.shape)).to(U.DEV))
    assert net.condition_dim == 51 and net.c_embed_in() == 46 and net.patch_dim == 24 and net.x_embed_in() == 48
    x, c, t, x0 = (torch.from_numpy(g[k]).to(U.DEV) for k in ("x", "c", "t", "x0"))
    loss = model._loss_from_noise(x, c, t, x0)
    loss.backward()
    tol = 1e-4 if mode == "f32" else 3e-2
    assert abs(loss.item() - float(g["loss"])) / float(g["loss"]) < tol
    with torch.no_grad():
        v = model.forward((1 - t) * x0 + t * x, t.view(-1, 1), c)
    assert U.rel_err(v, torch.from_numpy(g["velocity"])) < tol
    grads = U.named_grads(model)
    names = [str(n) for n in g["names"]]
    assert names == [k[4:] for k, _ in model.named_parameters()]
    norms = np.array([float(grads[k].double().norm()) for k in names])
    assert float(np.abs(norms - g["grad_norms"]).max() / g["grad_norms"].max()) < (3e-4 if mode == "f32" else 6e-2)
    for k in names:
        got = grads[k].flatten().double().cpu().numpy()[g["gidx/" + k]]
        scale = max(float(np.abs(g["gval/" + k]).max()), 1e-2 * float(g["grad_norms"][names.index(k)]), 1e-12)
        assert np.abs(got - g["gval/" + k]).max() / scale < (2e-3 if mode == "f32" else 0.25), k
    with torch.inference_mode():
        s = model.sample_batch(c)
    assert s.shape == (3, 1, 45, 16, 9) and torch.isfinite(s).all()


def test_condition_mapper_alone_vs_oracle():
    """`map_c_embedding` without the x mapper, f32: forward and every gradient against the CPU oracle."""
    back = O.ds2(1)
    model = U.build_models(back, "f32", O.golden_fill(back))
    net = model.net
    net.c_embedder = nn.Sequential(nn.Linear(50, 46), nn.SiLU(), net.c_embedder).to(U.DEV)
    with torch.no_grad():
        for k, p_ in net.named_parameters():
            if k.startswith("c_embedder.0."):
                p_.copy_(O.fill_tensor("ft/" + k, tuple(p_.shape)).to(U.DEV))
    new = O.ViTConfig(depth=1, condition_dim=50)
    x, c, gen = O.synthetic_batch(new, 3, 4)
    t, x0 = O.synthetic_noise(new, 3, gen)
    p = {k: v.detach().cpu() for k, v in net.named_parameters()}
    ref_loss, ref_v, ref_grads = O.loss_and_grads(p, x, c, t, x0, new)
    loss = model._loss_from_noise(x.to(U.DEV), c.to(U.DEV), t.to(U.DEV), x0.to(U.DEV))
    loss.backward()
    assert abs(loss.item() - ref_loss.item()) / ref_loss.item() < 1e-4
    grads = U.named_grads(model)
    for k, ref in ref_grads.items():
        assert U.rel_err(grads[k], ref) < 2e-3 or float(ref.abs().max()) < 1e-7, k


def test_unsupported_surgery_is_refused_loudly():
    back = O.ds2(1)
    model = U.build_models(back, "f32", O.golden_fill(back))
    net = model.net
    net.c_embedder = nn.Sequential(nn.Linear(50, 46), nn.ReLU(), net.c_embedder).to(U.DEV)  # not the mapper of experiment_finetuning.py:106-119
    x, c, g = O.synthetic_batch(back, 2, 1)
    with pytest.raises(NotImplementedError, match="c_embedder must be"):
        model.forward(x.to(U.DEV), torch.rand(2, 1, device=U.DEV), torch.zeros(2, 50, device=U.DEV))
