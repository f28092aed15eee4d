"""Generate tests/golden/*.npz from the REFERENCE ITSELF.  Build container only.

Imports the reference's own hot-path modules from /root/reference (read-only, never
copied) after registering build-owned stand-ins for the three third-party symbols that
are absent from this image:

  timm.models.vision_transformer.Mlp      -> fc1 / act / fc2 with the same attribute names
  xformers.ops.memory_efficient_attention -> torch SDPA on the transposes
  torchdiffeq.odeint                      -> fixed-grid solver restated from torchdiffeq's
                                            published algorithm ('rk4' = 3/8 rule); this
                                            part is therefore NOT pinned by third-party code.

Then runs the reference classes (nn.vit.ViT, CaloChallengeCFM and its sibling wrappers, nn.cfm.transformer_cfm.ParallelTransformer, CFM._batch_loss,
sample_batch, torch.optim.AdamW + clip_grad_norm_ as in BaseExperiment._step) on seeded
synthetic CaloChallenge-shaped inputs and stores inputs + outputs as small fixtures.
Weights are not stored: both sides fill them with oracle.vit_cfm_oracle.golden_fill.

Usage:  python oracle/make_golden.py     (writes tests/golden/*.npz)
"""

from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
REF = "/root/reference"

from oracle import energy_oracle as E  # noqa: E402
from oracle import transforms_oracle as TO  # noqa: E402
from oracle import vit_cfm_oracle as O  # noqa: E402


# ----------------------------------------------------------------------------- stand-ins
def _install_standins():
    class Mlp(nn.Module):  # timm Mlp: fc1 -> act -> drop1 -> norm(Identity) -> fc2 -> drop2
        def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
            super().__init__()
            out_features = out_features or in_features
            hidden_features = hidden_features or in_features
            self.fc1 = nn.Linear(in_features, hidden_features)
            self.act = act_layer()
            self.drop1 = nn.Dropout(drop)
            self.norm = nn.Identity()
            self.fc2 = nn.Linear(hidden_features, out_features)
            self.drop2 = nn.Dropout(drop)

        def forward(self, x):
            return self.drop2(self.fc2(self.norm(self.drop1(self.act(self.fc1(x))))))

    timm = types.ModuleType("timm")
    timm_models = types.ModuleType("timm.models")
    timm_vt = types.ModuleType("timm.models.vision_transformer")
    timm_vt.Mlp = Mlp
    timm.models = timm_models
    timm_models.vision_transformer = timm_vt
    sys.modules.update({"timm": timm, "timm.models": timm_models, "timm.models.vision_transformer": timm_vt})

    def memory_efficient_attention(q, k, v, p=0.0):  # (B,T,H,dh) in / out
        o = torch.nn.functional.scaled_dot_product_attention(
            q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), dropout_p=p
        )
        return o.transpose(1, 2)

    xf = types.ModuleType("xformers")
    xf_ops = types.ModuleType("xformers.ops")
    xf_ops.memory_efficient_attention = memory_efficient_attention
    xf.ops = xf_ops
    sys.modules.update({"xformers": xf, "xformers.ops": xf_ops})

    def odeint(func, y0, t, method="rk4", options=None, **kw):
        step = (options or {}).get("step_size", None)
        assert step is not None, "stand-in supports fixed-grid solvers with options.step_size only"
        grid = O.fixed_grid(float(t[0]), float(t[-1]), step, y0.dtype)
        y = y0
        for k in range(len(grid) - 1):
            y = O.ode_step(func, method, grid[k], grid[k + 1], y)
        return torch.stack([y0, y])

    tde = types.ModuleType("torchdiffeq")
    tde.odeint = odeint
    sys.modules["torchdiffeq"] = tde


def build_reference(cfg: O.ViTConfig, use_torch_sdpa=True, kind="calochallenge"):
    from experiments.calochallenge.calochallenge_cfm.model import CaloChallengeCFM
    from nn.vit import ViT

    param = {
        "dim": 3,
        "condition_dim": cfg.condition_dim,
        "hidden_dim": cfg.hidden_dim,
        "out_channels": 1,
        "depth": cfg.depth,
        "num_heads": cfg.num_heads,
        "mlp_ratio": cfg.mlp_ratio,
        "attn_drop": 0.0,
        "proj_drop": 0.0,
        "pos_embedding_coords": "cylindrical",
        "temperature": 10000,
        "learn_pos_embed": True,
        "causal_attn": False,
        "checkpoint_grads": False,
        "num_patches": [list(n) for n in cfg.seg_num_patches],
        "patch_dim": cfg.P,
        "use_torch_sdpa": use_torch_sdpa,
    }
    net = ViT(param)
    common = dict(in_channels=1, time_distribution="uniform", trajectory="linear",
                  odeint_kwargs={"method": "rk4", "options": {"step_size": 0.05}}, shape=list(cfg.shape))
    list_shape = [list(s) for s, _ in cfg.segments]
    list_edges = [int(np.prod(s)) for s, _ in cfg.segments]
    list_patch = [list(p) for _, p in cfg.segments]
    if kind == "calochallenge":
        model = CaloChallengeCFM(net, list(cfg.patch_shape), **common)
    elif kind == "ds1":  # experiments/calochallenge/calochallenge_cfm/model.py:97
        from experiments.calochallenge.calochallenge_cfm.model import CaloChallengeCFM_DS1

        assert all(p == list_patch[0] for p in list_patch)
        model = CaloChallengeCFM_DS1(net, list_shape, list_edges, list_patch[0], **common)
    elif kind == "calogan":  # experiments/calogan/model.py:8
        from experiments.calogan.model import CaloGANCFM

        model = CaloGANCFM(net, list_shape, list_edges, list_patch, **common)
    elif kind == "calohad":  # experiments/calohadronic/model.py:8
        from experiments.calohadronic.model import CaloHadCFM

        model = CaloHadCFM(net, list_shape, list_edges, list_patch, **common)
    elif kind == "lemurs":  # experiments/lemurs/model.py:8
        from experiments.lemurs.model import LEMURSCFM

        model = LEMURSCFM(net, list(cfg.patch_shape), **common)
    else:
        raise ValueError(kind)
    model.device = torch.device("cpu")
    model.dtype = torch.float32
    return model


def load_fill(model, cfg):
    fill = O.golden_fill(cfg)
    sd = model.state_dict()
    for k, v in fill.items():
        assert sd["net." + k].shape == v.shape, (k, sd["net." + k].shape, v.shape)
        sd["net." + k] = v.clone()
    model.load_state_dict(sd)
    return fill


def grad_probe(g: torch.Tensor, n=48):
    """A fixed, size-independent sample of a gradient tensor: first 8 + strided 40."""
    f = g.detach().flatten().double()
    idx = np.unique(np.concatenate([np.arange(min(8, f.numel())), np.linspace(0, f.numel() - 1, 40).astype(np.int64)]))
    return idx, f[idx].numpy()


def make_case(name, cfg, B, seed, sample_specs, train_steps, kind="calochallenge"):
    torch.manual_seed(1234)
    model = build_reference(cfg, kind=kind)
    # param inventory check against the oracle's list (names, shapes, order)
    ref_names = [k[4:] for k, _ in model.named_parameters()]
    assert ref_names == list(O.param_shapes(cfg).keys()), "state-dict order/name mismatch"
    nparams = sum(p.numel() for p in model.parameters())
    load_fill(model, cfg)
    model.train()

    x, c, g = O.synthetic_batch(cfg, B, seed)
    out = {"x": x.numpy(), "c": c.numpy(), "nparams": np.int64(nparams)}

    # --- _batch_loss through the reference's own code path (models/base_model.py:203-218).
    # It draws t then x0 from the global CPU generator; re-seeding reproduces both.
    torch.manual_seed(seed + 100)
    if kind == "lemurs":  # LEMURSCFM._batch_loss takes the dataset layout (B, R, A, L) and reorders it itself (lemurs/model.py:62-65)
        loss = model._batch_loss([x[:, 0].permute(0, 3, 2, 1).contiguous(), c])
    else:
        loss = model._batch_loss([x.clone(), c])
    torch.manual_seed(seed + 100)
    t = torch.rand([B] + [1] * (x.dim() - 1))
    if kind == "lemurs":  # randn_like of the permuted VIEW fills in that view's memory order
        x0 = torch.randn_like(x[:, 0].permute(0, 3, 2, 1).contiguous().permute(0, 3, 2, 1).unsqueeze(1)).contiguous()
    else:
        x0 = torch.randn_like(x)
    with torch.no_grad():
        x_t = (1 - t) * x0 + t * x
        v = model.forward(x_t, t.view(-1, 1), c)
        assert torch.allclose(((v - (x - x0)) ** 2).mean(), loss, rtol=0, atol=0), "t/x0 replay mismatch"
    model.zero_grad(set_to_none=True)
    loss.backward()
    out.update({"t": t.numpy(), "x0": x0.numpy(), "velocity": v.numpy(), "loss": np.float64(loss.item())})
    names, norms = [], []
    for k, p_ in model.named_parameters():
        k = k[4:]
        idx, vals = grad_probe(p_.grad)
        out["gidx/" + k] = idx
        out["gval/" + k] = vals
        names.append(k)
        norms.append(float(p_.grad.double().norm()))
    out["grad_norms"] = np.array(norms)
    out["grad_total_norm"] = np.float64(np.sqrt((np.array(norms) ** 2).sum()))

    # --- the xformers-branch stand-in must agree with the SDPA branch (nn/vit.py:432-448)
    # --- token-level intermediates for per-kernel parity (ViT.forward on patches)
    with torch.no_grad():
        xp = model.to_patches(x_t)
        out["patches"] = xp.numpy()
        out["pos_embed"] = model.net.learnable_pos_embedding().numpy()
        out["tokens_out"] = model.net(xp, t.view(-1, 1), c).numpy()
        out["t_emb"] = model.net.t_embedder(t.view(-1, 1)).numpy()
        out["c_emb"] = model.net.c_embedder(c).numpy()

    # --- sampling (calochallenge_cfm/model.py:68-94) with x_T replayed from the global RNG
    for tag, method, step in sample_specs:
        model.odeint_kwargs = {"method": method, "options": {"step_size": step}}
        model.eval()
        torch.manual_seed(seed + 200)
        s = model.sample_batch(c)
        torch.manual_seed(seed + 200)
        x_T = torch.randn((B, 1, *cfg.shape))
        out["x_T"] = x_T.numpy()
        out[f"sample/{tag}"] = s.numpy()
        out[f"sample_meta/{tag}"] = np.array([step])
    model.train()

    # --- update-step trajectory (experiments/base_experiment.py:555-597; configs/training/default.yaml)
    if train_steps:
        load_fill(model, cfg)
        iters = 50
        opt = torch.optim.AdamW(model.parameters(), lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.1)
        sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=iters, eta_min=0)
        gt = torch.Generator().manual_seed(seed + 300)
        losses, gnorms, ts, x0s = [], [], [], []
        for _ in range(train_steps):
            t_k, x0_k = O.synthetic_noise(cfg, B, gt)
            x_t = (1 - t_k) * x0_k + t_k * x
            vel = model.forward(x_t, t_k.view(-1, 1), c)
            l_k = ((vel - (x - x0_k)) ** 2).mean()
            opt.zero_grad(set_to_none=True)
            l_k.backward()
            torch.nn.utils.clip_grad_norm_(model.net.parameters(), float("inf"))
            gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 1000.0, error_if_nonfinite=True)
            opt.step()
            sched.step()
            losses.append(l_k.item())
            gnorms.append(gn.item())
            ts.append(t_k.numpy())
            x0s.append(x0_k.numpy())
        out["train/losses"] = np.array(losses)
        out["train/gnorms"] = np.array(gnorms)
        out["train/t"] = np.stack(ts)
        out["train/x0"] = np.stack(x0s)
        out["train/iters"] = np.int64(iters)
        sd = model.state_dict()
        for k in ("net.pos_embed_freqs", "net.blocks.0.attn.qkv.bias", "net.final_layer.linear.bias"):
            out["train/final/" + k[4:]] = sd[k].numpy()

    path = os.path.join(REPO, "tests", "golden", name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: loss={loss.item():.6f} |g|={out['grad_total_norm']:.6f} nparams={nparams} -> {os.path.getsize(path)/1024:.0f} KiB")


def make_trajectory_case(name, cfg, B, seed, steps):
    """BASELINE.json config 1: `steps` update steps of the reference's own CFM module on its CPU path (synthetic voxels; AdamW 1e-4 / wd 0.1,
    CosineAnnealingLR over `steps`, clip 1000: configs/training/default.yaml, experiments/base_experiment.py:555-597).  Only the seeds and the
    loss / gradient-norm trajectory are stored: the batch and the per-step (t, x_0) are regenerated by the test from the same seeded CPU
    generators (oracle.synthetic_batch / synthetic_noise), which keeps the fixture at a few KiB."""
    torch.manual_seed(1234)
    model = build_reference(cfg)
    load_fill(model, cfg)
    model.train()
    x, c, _ = O.synthetic_batch(cfg, B, seed)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.1)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=steps, eta_min=0)
    gt = torch.Generator().manual_seed(seed + 300)
    losses, gnorms = [], []
    for _ in range(steps):
        t_k, x0_k = O.synthetic_noise(cfg, B, gt)
        x_t = (1 - t_k) * x0_k + t_k * x
        vel = model.forward(x_t, t_k.view(-1, 1), c)
        l_k = ((vel - (x - x0_k)) ** 2).mean()
        opt.zero_grad(set_to_none=True)
        l_k.backward()
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 1000.0, error_if_nonfinite=True)
        opt.step()
        sched.step()
        losses.append(l_k.item())
        gnorms.append(gn.item())
    sd = model.state_dict()
    out = {"B": np.int64(B), "seed": np.int64(seed), "noise_seed": np.int64(seed + 300), "iters": np.int64(steps), "losses": np.array(losses), "gnorms": np.array(gnorms),
           "x_checksum": np.float64(x.double().sum().item()), "final/blocks.1.mlp.fc2.weight": sd["net.blocks.1.mlp.fc2.weight"].numpy()[::16, ::64].copy(),
           "final/pos_embed_freqs": sd["net.pos_embed_freqs"].numpy()}
    path = os.path.join(REPO, "tests", "golden", name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: loss {losses[0]:.6f} -> {losses[-1]:.6f} over {steps} steps -> {os.path.getsize(path)/1024:.0f} KiB")


def check_branches(cfg):
    """SDPA branch vs (stand-in) xformers branch of nn/vit.py:431-449 must agree."""
    a = build_reference(cfg, True)
    b = build_reference(cfg, False)
    load_fill(a, cfg)
    load_fill(b, cfg)
    x, c, g = O.synthetic_batch(cfg, 2, 5)
    t = torch.rand(2, 1, generator=g)
    with torch.no_grad():
        d = (a.forward(x, t, c) - b.forward(x, t, c)).abs().max().item()
    print("sdpa vs xformers-stand-in max abs diff:", d)
    assert d < 1e-5


def make_energy_case(name, cfg: E.EnergyConfig, B, seed, sample_specs):
    """Energy-model CFM (configs/model/cfm/cfm_ds2_energy.yaml): the reference's ParallelTransformer inside models.base_model.CFM."""
    from models.base_model import CFM
    from nn.cfm.transformer_cfm import ParallelTransformer

    torch.manual_seed(4321)
    net = ParallelTransformer({"dims_in": cfg.dims_in, "dims_c": cfg.dims_c, "dim_embedding": cfg.dim_embedding, "nhead": cfg.nhead,
                               "num_encoder_layers": cfg.num_encoder_layers, "num_decoder_layers": cfg.num_decoder_layers,
                               "dim_feedforward": cfg.dim_feedforward, "dropout": 0.0, "activation": "relu", "embeds": True,
                               "encode_t_scale": cfg.encode_t_scale, "encode_t_dim": cfg.encode_t_dim})
    model = CFM(net, "uniform", "linear", {"method": "rk4", "options": {"step_size": 0.05}}, shape=[cfg.dims_in])
    model.device, model.dtype = torch.device("cpu"), torch.float32
    ref_names = [k[4:] for k, _ in model.named_parameters()]
    assert ref_names == list(E.param_shapes(cfg).keys()), (ref_names, list(E.param_shapes(cfg).keys()))
    fill = E.golden_fill(cfg)
    sd = model.state_dict()
    for k, v in fill.items():
        assert sd["net." + k].shape == v.shape, (k, sd["net." + k].shape, v.shape)
        sd["net." + k] = v.clone()
    sd["net.layers.0.weight"], sd["net.layers.0.bias"] = fill["layer.weight"].clone(), fill["layer.bias"].clone()  # the shared tensor's second name
    model.load_state_dict(sd)
    model.eval()
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((B, cfg.dims_in), generator=g)
    c = torch.rand((B, cfg.dims_c), generator=g)
    t = torch.rand((B, 1), generator=g)
    out = {"x": x.numpy(), "c": c.numpy(), "t": t.numpy(), "nparams": np.int64(sum(p.numel() for p in model.parameters()))}
    with torch.no_grad():
        out["velocity"] = model.forward(x, t, c).numpy()
        out["t_emb"] = net.time_embed(t).numpy()
    # loss through the reference's own _batch_loss (t, x0 from the global generator, replayed)
    torch.manual_seed(seed + 100)
    loss = model._batch_loss([x.clone(), c])
    torch.manual_seed(seed + 100)
    t2 = torch.rand([B, 1])
    x0 = torch.randn_like(x)
    with torch.no_grad():
        v2 = model.forward((1 - t2) * x0 + t2 * x, t2, c)
        assert torch.allclose(((v2 - (x - x0)) ** 2).mean(), loss, rtol=0, atol=0), "t/x0 replay mismatch"
    out.update({"loss_t": t2.numpy(), "loss_x0": x0.numpy(), "loss": np.float64(loss.item())})
    for tag, method, step in sample_specs:
        model.odeint_kwargs = {"method": method, "options": {"step_size": step}}
        torch.manual_seed(seed + 200)
        smp = model.sample_batch(c)
        torch.manual_seed(seed + 200)
        x_T = torch.randn((B, cfg.dims_in))
        out["x_T"] = x_T.numpy()
        out[f"sample/{tag}"] = smp.numpy()
        out[f"sample_meta/{tag}"] = np.array([step])
    path = os.path.join(REPO, "tests", "golden", name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: loss={loss.item():.6f} |v|={float(np.abs(out['velocity']).max()):.4f} nparams={int(out['nparams'])} -> {os.path.getsize(path)/1024:.0f} KiB")


def make_mapper_case(name, back: O.ViTConfig, new: O.ViTConfig, B, seed, map_c=False):
    """Fine-tuning with an embedding mapper (experiments/calochallenge/calochallenge_cfm/experiment_finetuning.py:75-171, flags of
    configs/calochallenge/finetuning/calochallenge_ds2tods3_ft.yaml: map_x_embedding, reinitialize_pos_embedding, reinitialize_final_layer):
    the reference's own modules, surgery statements as in add_embedding_layers()."""
    from nn.vit import FinalLayer

    torch.manual_seed(99)
    model = build_reference(back)                       # backbone network ...
    model.shape, model.patch_shape = list(new.shape), list(new.patch_shape)  # ... inside the new dataset's wrapper
    model.num_patches = list(new.num_patches)
    load_fill(model, back)
    net = model.net
    mapper = nn.Linear(new.P, back.P)
    net.x_embedder = nn.Sequential(mapper, nn.SiLU(), net.x_embedder)
    net.num_patches = [list(new.num_patches)]
    pos_z, pos_y, pos_x = net.create_meshgrid()
    net.pos_z, net.pos_y, net.pos_x = pos_z, pos_y, pos_x
    net.final_layer = FinalLayer(back.hidden_dim, new.P, 1)
    if map_c:  # map_c_embedding as well (experiment_finetuning.py:106-119): conditions of the new dataset's width in front of the backbone's embedder
        net.c_embedder = nn.Sequential(nn.Linear(new.condition_dim, back.condition_dim), nn.SiLU(), net.c_embedder)
    with torch.no_grad():
        for k, p_ in model.named_parameters():
            if k.startswith("net.x_embedder.0.") or k.startswith("net.final_layer.") or (map_c and k.startswith("net.c_embedder.0.")):
                p_.copy_(O.fill_tensor("ft/" + k[4:], tuple(p_.shape)))
    model.train()
    x, c, g = O.synthetic_batch(new, B, seed)
    t, x0 = O.synthetic_noise(new, B, g)
    x_t = (1 - t) * x0 + t * x
    v = model.forward(x_t, t.view(-1, 1), c)
    loss = ((v - (x - x0)) ** 2).mean()
    model.zero_grad(set_to_none=True)
    loss.backward()
    out = {"x": x.numpy(), "c": c.numpy(), "t": t.numpy(), "x0": x0.numpy(), "velocity": v.detach().numpy(), "loss": np.float64(loss.item())}
    names, norms = [], []
    for k, p_ in model.named_parameters():
        k = k[4:]
        idx, vals = grad_probe(p_.grad)
        out["gidx/" + k], out["gval/" + k] = idx, vals
        names.append(k)
        norms.append(float(p_.grad.double().norm()))
    out["grad_norms"] = np.array(norms)
    out["names"] = np.array(names)
    path = os.path.join(REPO, "tests", "golden", name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: loss={loss.item():.6f} |g|={float(np.sqrt((np.array(norms) ** 2).sum())):.6f} -> {os.path.getsize(path)/1024:.0f} KiB")


def make_transforms_case(name, layers, shape, B, seed, spec_kw):
    """Pre-/post-processing chain of the shape models (configs/calochallenge/cfm/calochallenge_ds2.yaml:15-28) through the reference's
    own transform classes.  `layers` = [(n_alpha, n_r), ...]: a synthetic binning file of the CaloChallenge XML format is written to a
    temporary directory (the real ones are dataset assets, not part of the reference repository)."""
    import tempfile

    import experiments.calochallenge.transforms as T

    tmp = tempfile.mkdtemp()
    xml = os.path.join(tmp, "binning.xml")
    rows = [f'<Layer id="{i}" r_edges="{",".join(str(float(k)) for k in range(nr + 1))}" n_bin_alpha="{na}"/>' for i, (na, nr) in enumerate(layers)]
    with open(xml, "w") as fh:
        fh.write('<Bins>\n<Bin pid="11" etaMin="0" etaMax="130" name="electron">\n' + "\n".join(rows) + "\n</Bin>\n</Bins>\n")
    bounds = tuple(int(v) for v in np.cumsum([0] + [a * r for a, r in layers]))
    spec = TO.ChainSpec(layer_boundaries=bounds, shape=tuple(shape), **spec_kw)
    np.save(os.path.join(tmp, "means.npy"), np.float32(spec.mean))
    np.save(os.path.join(tmp, "stds.npy"), np.float32(spec.std))
    n_layers, V = spec.n_layers, spec.n_voxels
    chain = [T.NormalizeByElayer(ptype=xml, xml_file="electron", cut=spec.norm_cut, eps=spec.eps),  # (sic) ptype carries the path: transforms.py:337-339
             T.ScaleTotalEnergy(factor=spec.factor, n_layers=n_layers), T.CutValues(cut=spec.cut, n_layers=n_layers),
             T.ExclusiveLogitTransform(delta=spec.delta, rescale=True), T.GlobalStandardizeFromFile(model_dir=tmp, eps=1.0e-6),
             T.LogEnergy(alpha=spec.alpha), T.ScaleEnergy(e_min=spec.e_min, e_max=spec.e_max), T.AddFeaturesToCond(split_index=V), T.Reshape(shape=list(shape))]
    assert tuple(int(v) for v in chain[0].layer_boundaries) == bounds
    g = torch.Generator().manual_seed(seed)
    # synthetic showers: sparse, positive, a few orders of magnitude; incident energies log-uniform in [1e3, 1e6] MeV
    dep = torch.exp(torch.randn((B, V), generator=g) * 2.0) * (torch.rand((B, V), generator=g) < 0.3)
    dep[0, bounds[1] : bounds[2]] = 0.0  # one empty layer
    energy = torch.exp(torch.rand((B, 1), generator=g) * (spec.e_max - spec.e_min) + spec.e_min)
    dep = dep / dep.sum(1, keepdim=True) * energy * (0.5 + 0.4 * torch.rand((B, 1), generator=g))
    out = {"showers": dep.numpy().copy(), "energy": energy.numpy().copy(), "bounds": np.array(bounds)}
    x, c = dep.clone(), energy.clone()
    for fn in chain:
        x, c = fn(x, c)
    out["x"], out["c"] = x.numpy().copy(), c.numpy().copy()
    xr, cr = x.clone(), c.clone()
    for fn in chain[::-1]:
        xr, cr = fn(xr, cr, rev=True)
    out["roundtrip_showers"], out["roundtrip_energy"] = xr.numpy().copy(), cr.numpy().copy()
    # model-like samples: standard-normal voxels and u's, condition energy in [0, 1]
    smp = torch.randn((B, *shape), generator=g) * 1.5
    cond = torch.cat([torch.randn((B, n_layers), generator=g) * 1.5, torch.rand((B, 1), generator=g)], dim=1)
    out["samples"], out["cond"] = smp.numpy().copy(), cond.numpy().copy()
    xs, cs = smp.clone(), cond.clone()
    for fn in chain[::-1]:
        xs, cs = fn(xs, cs, rev=True)
    out["post_showers"], out["post_energy"] = xs.numpy().copy(), cs.numpy().copy()
    path = os.path.join(REPO, "tests", "golden", name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: V={V} layers={n_layers} |x|max={float(np.abs(out['x']).max()):.3f} sum(post)={float(out['post_showers'].sum()):.4e} -> {os.path.getsize(path)/1024:.0f} KiB")


def main():
    _install_standins()
    sys.path.insert(0, REF)
    torch.set_num_threads(8)
    only = set(sys.argv[1:])  # optional: names of the cases to (re)generate
    if only:
        g = globals()
        for fn in ("make_case", "make_energy_case", "make_mapper_case", "make_transforms_case", "make_trajectory_case"):
            g[fn] = (lambda f: lambda name, *a, **k: f(name, *a, **k) if name in only else None)(g[fn])
        g["check_branches"] = lambda *a, **k: None
    check_branches(O.ds2(2))
    make_case("ds2_d2_b2", O.ds2(2), 2, 11, [("rk4", "rk4", 0.05), ("heun", "heun2", 0.25)], 5)
    make_case("ds2_d6_b2", O.ds2(6), 2, 12, [("rk4_coarse", "rk4", 0.25)], 3)
    make_case("ds3_d6_b1", O.ds3(6), 1, 13, [("rk4_coarse", "rk4", 0.5)], 0)
    make_trajectory_case("ds2_d2_b8_50it", O.ds2(2), 8, 61, 50)  # BASELINE.json configs[0]
    # the other ViT-CFM geometries: multi-segment patching + position buffers
    make_case("ds1_photons_d2_b3", O.ds1_photons(2), 3, 21, [("rk4_coarse", "rk4", 0.25)], 3, kind="ds1")
    make_case("ds1_pions_d2_b2", O.ds1_pions(2), 2, 22, [("heun", "heun2", 0.25)], 0, kind="ds1")
    make_case("calogan_d2_b3", O.calogan(2), 3, 23, [("rk4_coarse", "rk4", 0.25)], 3, kind="calogan")
    make_case("calohad_d2_b1", O.calohad(2), 1, 24, [("rk4_coarse", "rk4", 0.5)], 2, kind="calohad")
    make_case("lemurs_d2_b2", O.lemurs(2), 2, 25, [("rk4_coarse", "rk4", 0.25)], 0, kind="lemurs")
    # the energy-model CFM
    make_energy_case("energy_ds2_b5", E.EnergyConfig(), 5, 31, [("rk4", "rk4", 0.05), ("heun", "heun2", 0.25)])
    # fine-tuning with an embedding mapper
    make_mapper_case("ft_mapper_d2_b2", O.ds2(2), O.ViTConfig(shape=(45, 16, 9), patch_shape=(3, 8, 1), depth=2), 2, 51)
    make_mapper_case("ft_xc_mapper_d2_b3", O.ds2(2), O.ViTConfig(shape=(45, 16, 9), patch_shape=(3, 8, 1), depth=2, condition_dim=51), 3, 52, map_c=True)
    # pre-/post-processing chain
    make_transforms_case("transforms_ds2_b4", [(16, 9)] * 45, (1, 45, 16, 9), 4, 41, dict(mean=-1.7, std=2.9))
    make_transforms_case("transforms_ds1ph_b6", [(1, 8), (10, 16), (10, 19), (1, 5), (1, 5)], (368,), 6, 42, dict(mean=-0.8, std=3.3, factor=0.5, cut=1.0e-6))
    make_energy_case("energy_small_b3", E.EnergyConfig(dims_in=30, dim_embedding=32, nhead=2, num_encoder_layers=1, num_decoder_layers=2,
                                                       dim_feedforward=256, encode_t_dim=32), 3, 32, [("rk4_coarse", "rk4", 0.25)])


if __name__ == "__main__":
    main()
