"""-m gpu: the whole path (forward, loss, gradients, sampler, update step) through the host mirror classes and
the C ABI, against (a) the CPU oracle on the same seeded inputs and (b) the committed golden vectors that were
produced by the reference itself.

Tolerance (north star): f32 mode <= 1e-4 relative on losses / velocities / samples.  bf16 mode (throughput mode):
bf16 operands with f32 accumulation, checked at 3e-2 relative to the f32 result.
"""

import numpy as np
import pytest
import torch

from oracle import vit_cfm_oracle as O
from tests import hiputil as U

pytestmark = pytest.mark.gpu
CASES = {"ds2_d2_b2": O.ds2(2), "ds2_d6_b2": O.ds2(6), "ds3_d6_b1": O.ds3(6)}
TOL = {"f32": 1e-4, "bf16": 3e-2}
GRAD_TOL = {"f32": 3e-4, "bf16": 6e-2}


def _inputs(g):
    return tuple(torch.from_numpy(g[k]).to(U.DEV) for k in ("x", "c", "t", "x0"))


@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("name", list(CASES))
def test_forward_loss_grads_vs_golden(name, mode, golden):
    g = golden(name)
    cfg = CASES[name]
    model = U.build_models(cfg, mode, O.golden_fill(cfg))
    x, c, t, x0 = _inputs(g)
    model.train()
    loss = model._loss_from_noise(x, c, t, x0)
    loss.backward()
    with torch.no_grad():
        x_t = (1 - t) * x0 + t * x
        v = model.forward(x_t, t.view(-1, 1), c)
    assert U.rel_err(v, torch.from_numpy(g["velocity"])) < TOL[mode]
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


@pytest.mark.parametrize("name", ["ds2_d2_b2", "ds3_d6_b1"])
def test_full_gradients_vs_oracle(name, golden):
    """Every element of every gradient tensor, f32 mode, against the oracle's autograd on the same inputs."""
    g = golden(name)
    cfg = CASES[name]
    fill = O.golden_fill(cfg)
    model = U.build_models(cfg, "f32", fill)
    x, c, t, x0 = _inputs(g)
    model._loss_from_noise(x, c, t, x0).backward()
    _, _, ref = O.loss_and_grads(fill, x.cpu(), c.cpu(), t.cpu(), x0.cpu(), cfg)
    grads = U.named_grads(model)
    for k, r in ref.items():
        e = float((grads[k].cpu().double() - r.double()).abs().max() / (r.double().abs().max() + 1e-12))
        assert e < 1e-3, (k, e)
        assert U.rms_err(grads[k], r) < 2e-4, k


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_patch_token_api_matches_voxel_api(mode, golden):
    """ViT.forward on (B,T,P) tokens (the reference's nn/vit.py:185 signature) == fused voxel path."""
    g = golden("ds2_d2_b2")
    cfg = CASES["ds2_d2_b2"]
    model = U.build_models(cfg, mode, O.golden_fill(cfg))
    x, c, t, _ = _inputs(g)
    with torch.no_grad():
        v = model.forward(x, t.view(-1, 1), c)
        tok = model.net(model.to_patches(x), t.view(-1, 1), c)
        assert tok.shape == (2, cfg.T, cfg.P)
        assert torch.equal(model.from_patches(tok), v)
    if mode == "f32":
        assert U.rel_err(tok, torch.from_numpy(g["tokens_out"])) < 2e-4 or True  # tokens_out used x_t, checked below
        x_t = (1 - t) * torch.from_numpy(g["x0"]).to(U.DEV) + t * x
        tok = model.net(model.to_patches(x_t), t.view(-1, 1), c)
        assert U.rel_err(tok, torch.from_numpy(g["tokens_out"])) < 1e-4


def test_fresh_init_outputs_zero():
    """Zero-initialised adaLN / final layer (nn/vit.py:174-183): a fresh model outputs exactly 0, loss ~ 2."""
    from vit4hep_amd import CaloChallengeCFM, ViT

    torch.manual_seed(0)
    net = ViT({"hidden_dim": 480, "depth": 2, "num_heads": 6, "mlp_ratio": 4, "patch_dim": 48, "num_patches": [[15, 1, 9]]})
    model = CaloChallengeCFM(net, [3, 16, 1], shape=[45, 16, 9]).to(U.DEV)
    model.device, model.dtype = torch.device(U.DEV), torch.float32
    x, c, _ = O.synthetic_batch(O.ds2(2), 8, 0)
    with torch.no_grad():
        v = model.forward(x.to(U.DEV), torch.rand(8, 1, device=U.DEV), c.to(U.DEV))
    assert float(v.abs().max()) == 0.0
    loss = model._batch_loss([x, c])
    assert 1.8 < loss.item() < 2.2
    loss.backward()
    assert all(torch.isfinite(p.grad).all() for p in model.parameters())


@pytest.mark.parametrize("name,tag,method", [("ds2_d2_b2", "rk4", "rk4"), ("ds2_d2_b2", "heun", "heun2"), ("ds2_d6_b2", "rk4_coarse", "rk4"),
                                             ("ds3_d6_b1", "rk4_coarse", "rk4")])
def test_sampler_vs_golden(name, tag, method, golden):
    g = golden(name)
    cfg = CASES[name]
    model = U.build_models(cfg, "f32", O.golden_fill(cfg)).eval()
    model.odeint_kwargs = {"method": method, "options": {"step_size": float(g[f"sample_meta/{tag}"][0])}}
    with torch.inference_mode():
        s = model._sample_from(torch.from_numpy(g["x_T"]).to(U.DEV), torch.from_numpy(g["c"]).to(U.DEV))
    assert U.rel_err(s, torch.from_numpy(g[f"sample/{tag}"])) < 1e-4


def test_sample_batch_shape_and_determinism():
    cfg = O.ds2(2)
    model = U.build_models(cfg, "bf16", O.golden_fill(cfg)).eval()
    model.odeint_kwargs = {"method": "rk4", "options": {"step_size": 0.5}}
    _, c, _ = O.synthetic_batch(cfg, 4, 1)
    torch.manual_seed(7)
    a = model.sample_batch(c.to(U.DEV))
    torch.manual_seed(7)
    b = model.sample_batch(c.to(U.DEV))
    assert a.shape == (4, 1, 45, 16, 9) and torch.equal(a, b) and torch.isfinite(a).all()


def test_frozen_weight_operand_copies_are_reused_and_invalidated():
    """Inference forwards keep the bf16 / padded operand copies of the weights in the persistent workspace (V4H_FWD_REUSE_OPERANDS) only
    inside a ``frozen_weights()`` scope (the ODE solve); outside, every forward recasts - so a write through ``p.data`` (EMA copy_to / restore,
    reference base_experiment.py:630), which no version counter sees, can never leave stale operands behind."""
    from vit4hep_amd.trainer import CFMTrainer

    cfg = O.ds2(2)
    fill = O.golden_fill(cfg)
    model = U.build_models(cfg, "bf16", fill).eval()
    net = model.net
    x, c, _ = O.synthetic_batch(cfg, 4, 3)
    x, c = x.to(U.DEV), c.to(U.DEV)
    t = torch.full((4, 1), 0.3, device=U.DEV)
    with torch.no_grad():
        a = model.forward(x, t, c)
        b = model.forward(x, t, c)
        assert net._last_fwd_flags == 0 and torch.equal(a, b)  # no scope: recast, same result
        w = net.blocks[0].mlp.fc1.weight
        ver = w._version
        w.data.copy_(w.data * 1.5)  # the EMA way: not tracked
        assert w._version == ver
        d = model.forward(x, t, c)
        assert net._last_fwd_flags == 0 and not torch.equal(a, d)
        fresh_fill = {k: v.clone() for k, v in fill.items()}
        fresh_fill["blocks.0.mlp.fc1.weight"] *= 1.5
        fresh = U.build_models(cfg, "bf16", fresh_fill).eval()
        assert torch.equal(d, fresh.forward(x, t, c))
        # other batch size -> other workspace -> still right
        assert torch.equal(model.forward(x[:2], t[:2], c[:2]), fresh.forward(x[:2], t[:2], c[:2]))
        with net.frozen_weights():
            e1 = model.forward(x, t, c)
            assert net._last_fwd_flags == 0  # the first forward of a scope always recasts
            e2 = model.forward(x, t, c)
            assert net._last_fwd_flags == 2 | 4 and torch.equal(e1, e2) and torch.equal(e1, d)
            net.blocks[0].mlp.fc1.weight.mul_(2.0)  # a tracked in-place update inside the scope is still noticed
            e3 = model.forward(x, t, c)
            assert net._last_fwd_flags == 0 and not torch.equal(e1, e3)
        model.forward(x, t, c)
        assert net._last_fwd_flags == 0  # scope left: back to recasting
    # the fused trainer rewrites parameters through raw pointers and bumps weights_epoch
    model.train()
    tr = CFMTrainer(model, iterations=10)
    with torch.no_grad():
        before = model.forward(x, t, c).clone()
    tr.step(x, c)
    with torch.no_grad():
        after = model.forward(x, t, c)
    assert not torch.equal(before, after)


def test_condition_embedding_is_kept_across_evaluations_of_the_same_conditions():
    """V4H_FWD_SAME_CONDITION: the ODE solver evaluates the network many times for one condition tensor; the c_embedder term (independent
    of t) is computed once.  Another tensor, an in-place write or changed weights recompute it.  (Inside a frozen_weights() scope only.)"""
    cfg = O.ds2(2)
    fill = O.golden_fill(cfg)
    for mode in ("f32", "bf16"):
        model = U.build_models(cfg, mode, fill).eval()
        net = model.net
        x, c, _ = O.synthetic_batch(cfg, 4, 3)
        x, c = x.to(U.DEV), c.to(U.DEV)
        t1, t2 = torch.full((4, 1), 0.3, device=U.DEV), torch.full((4, 1), 0.7, device=U.DEV)
        with torch.no_grad(), net.frozen_weights():
            model.forward(x, t1, c)
            assert net._last_fwd_flags == 0
            a = model.forward(x, t2, c)          # same conditions, other time
            assert net._last_fwd_flags == 2 | 4
            c2 = c.clone()
            b = model.forward(x, t2, c2)         # equal values, other tensor: recomputed
            assert net._last_fwd_flags == 2 and torch.equal(a, b)
            c2[:, 0] += 1.0                      # in-place write: recomputed
            d = model.forward(x, t2, c2)
            assert net._last_fwd_flags == 2 and not torch.equal(a, d)
            e = model.forward(x, t2, c2)
            assert net._last_fwd_flags == 2 | 4 and torch.equal(d, e)
            fresh = U.build_models(cfg, mode, fill).eval()
            assert torch.equal(d, fresh.forward(x, t2, c2))
            net.c_embedder[2].bias.add_(0.5)     # weights changed: operands and the condition term are refreshed
            f = model.forward(x, t2, c2)
            assert net._last_fwd_flags == 0 and not torch.equal(e, f)
        s1 = model.sample_batch(c)               # the whole sampler: flag set from the second evaluation on
        assert net._last_fwd_flags == 2 | 4 and torch.isfinite(s1).all()


@pytest.mark.parametrize("name", ["ds2_d2_b2", "ds2_d6_b2"])
def test_update_step_trajectory_vs_golden(name, golden):
    """_step semantics (base_experiment.py:555-597) with the fused clip + AdamW kernels: loss trajectory and final weights."""
    from vit4hep_amd.trainer import CFMTrainer

    g = golden(name)
    cfg = CASES[name]
    model = U.build_models(cfg, "f32", O.golden_fill(cfg))
    tr = CFMTrainer(model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.1, clip_grad_norm=1000.0, iterations=int(g["train/iters"]))
    x, c = torch.from_numpy(g["x"]).to(U.DEV), torch.from_numpy(g["c"]).to(U.DEV)
    for k in range(len(g["train/losses"])):
        loss, gn = tr.step(x, c, torch.from_numpy(g["train/t"][k]).to(U.DEV), torch.from_numpy(g["train/x0"][k]).to(U.DEV))
        assert abs(loss.item() - g["train/losses"][k]) / g["train/losses"][k] < 1e-4, (k, loss.item())
        assert abs(gn.item() - g["train/gnorms"][k]) / g["train/gnorms"][k] < 1e-3, k
    sd = model.state_dict()
    D = cfg.hidden_dim
    for k in ("pos_embed_freqs", "blocks.0.attn.qkv.bias", "final_layer.linear.bias"):
        got, want = sd["net." + k].cpu(), torch.from_numpy(g["train/final/" + k])
        if k.endswith("qkv.bias"):  # key-bias gradient is analytically zero (softmax shift invariance): Adam-normalised noise, bounded not compared
            assert (got[D : 2 * D] - want[D : 2 * D]).abs().max() <= 2 * 1e-4 * len(g["train/losses"])
            keep = torch.cat([torch.arange(0, D), torch.arange(2 * D, 3 * D)])
            got, want = got[keep], want[keep]
        assert U.rel_err(got, want) < 1e-4, k


def test_torch_optimizer_dropin_matches_native_trainer(golden):
    """The unchanged-caller path: torch.optim.AdamW + clip_grad_norm_ on the module's parameters (what
    BaseExperiment._step does) gives the same losses as the fused trainer."""
    g = golden("ds2_d2_b2")
    cfg = CASES["ds2_d2_b2"]
    model = U.build_models(cfg, "f32", O.golden_fill(cfg))
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.1)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=int(g["train/iters"]), eta_min=0)
    x, c = torch.from_numpy(g["x"]).to(U.DEV), torch.from_numpy(g["c"]).to(U.DEV)
    for k in range(3):
        loss = model._loss_from_noise(x, c, torch.from_numpy(g["train/t"][k]).to(U.DEV), torch.from_numpy(g["train/x0"][k]).to(U.DEV))
        opt.zero_grad(set_to_none=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1000.0, error_if_nonfinite=True)
        opt.step()
        sched.step()
        assert abs(loss.item() - g["train/losses"][k]) / g["train/losses"][k] < 1e-4


def test_backward_with_collectives_matches_single_call():
    """The multi-rank code paths on one GPU: process group of one rank (RCCL), bucketed all-reduce on the communication stream.
    (a) one backward call that records an event per stage, collectives enqueued behind the events (the product path);
    (b) one call per stage, each ending in a stream join (V4H_STAGED_CALLS=1) - both must give what the plain single-call path gives."""
    import os

    import torch.distributed as dist

    from vit4hep_amd.trainer import CFMTrainer

    cfg = O.ds2(2)
    fill = O.golden_fill(cfg)
    x, c, g = O.synthetic_batch(cfg, 4, 9)
    noise = [O.synthetic_noise(cfg, 4, g) for _ in range(3)]
    x, c = x.to(U.DEV), c.to(U.DEV)

    def run(mode="f32"):
        model = U.build_models(cfg, mode, fill)
        tr = CFMTrainer(model, iterations=20)
        out = [tr.step(x, c, t.to(U.DEV), x0.to(U.DEV)) for t, x0 in noise]
        return [float(l) for l, _ in out], [float(n) for _, n in out], {k: v.detach().clone() for k, v in model.state_dict().items()}

    l0, n0, w0 = run()
    lb0, nb0, wb0 = run("bf16")
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29533", "RANK": "0", "WORLD_SIZE": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    dist.init_process_group("nccl", init_method="env://", device_id=torch.device(U.DEV))
    os.environ["V4H_FORCE_COLLECTIVES"] = "1"
    try:
        from vit4hep_amd.parallel import collectives_enabled

        assert collectives_enabled()
        l1, n1, w1 = run()
        lb1, nb1, wb1 = run("bf16")
        os.environ["V4H_STAGED_CALLS"] = "1"
        l2, n2, w2 = run()
    finally:
        os.environ.pop("V4H_FORCE_COLLECTIVES", None)
        os.environ.pop("V4H_STAGED_CALLS", None)
        dist.destroy_process_group()
    assert np.allclose(l0, l2, rtol=1e-6) and np.allclose(n0, n2, rtol=1e-5)
    for k in w0:
        assert U.rel_err(w2[k], w0[k]) < 1e-4, k
    # bf16 mode: the single-rank pass batches the adaLN backward (different summation order of bf16 products), so compare loosely
    assert np.allclose(lb0, lb1, rtol=2e-3) and np.allclose(nb0, nb1, rtol=2e-2)
    assert np.allclose(l0, l1, rtol=1e-6) and np.allclose(n0, n1, rtol=1e-5)
    for k in w0:  # small-batch conditioning tensors are accumulated with f32 atomics (order varies run to run), Adam normalises the difference
        assert U.rel_err(w1[k], w0[k]) < 1e-4, k
