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


def test_mapper_surgery_is_refused_loudly():
    from vit4hep_amd import CaloChallengeCFM

    back = O.ds2(1)
    model = U.build_models(back, "f32", O.golden_fill(back))
    net = model.net
    net.x_embedder = nn.Sequential(nn.Linear(90, 48), nn.SiLU(), net.x_embedder).to(U.DEV)  # map_x_embedding (experiment_finetuning.py:80-91)
    x, c, g = O.synthetic_batch(back, 2, 1)
    with pytest.raises(NotImplementedError, match="map_x_embedding"):
        model.forward(x.to(U.DEV), torch.rand(2, 1, device=U.DEV), c.to(U.DEV))
