"""-m gpu, round 4: checkpoint / resume of the fused trainer in torch.optim.AdamW's layout, max_grad_norm step skip on the device, the per-stage
autograd nodes of the data-parallel route (DDP bucket hooks fire stage by stage), full-size gradients against the oracle (tests/test_hip_fullsize.py)."""

import io
import os
import socket

import numpy as np
import pytest
import torch

from oracle import vit_cfm_oracle as O
from tests import hiputil as U
from vit4hep_amd import _lib

pytestmark = pytest.mark.gpu


def _data(cfg, B, seed, steps):
    x, c, g = O.synthetic_batch(cfg, B, seed)
    noise = [O.synthetic_noise(cfg, B, g) for _ in range(steps)]
    return x.to(U.DEV), c.to(U.DEV), [(t.to(U.DEV), x0.to(U.DEV)) for t, x0 in noise]


# ---------------------------------------------------------------------------------------------------------------- checkpoint / resume
def _reference_step(model, opt, sched, x, c, t, x0):
    """BaseExperiment._step (reference experiments/base_experiment.py:555-597) on the autograd route, unchanged semantics."""
    loss = model._loss_from_noise(x, c, t, x0)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.net.parameters(), float("inf")).cpu().item()
    gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 1000.0, error_if_nonfinite=True).cpu().item()
    opt.step()
    sched.step()
    return loss.item(), gn


def test_trainer_checkpoint_resumes_in_trainer_and_in_torch_adamw():
    """5 updates -> save in the reference's checkpoint layout ({model, optimizer, scheduler, ema}, base_experiment.py:661-677) -> continue (i) in a fresh
    CFMTrainer, (ii) with a real torch.optim.AdamW + CosineAnnealingLR on the autograd route (the reference's warm start, :374-388, :420-431), and
    (iii) the other way round: torch's own state_dict into a fresh CFMTrainer.  All three must follow the uninterrupted 10-update run."""
    from vit4hep_amd.trainer import CFMTrainer

    cfg = O.ds2(2)
    fill = O.golden_fill(cfg)
    T_MAX = 40
    x, c, noise = _data(cfg, 4, 31, 10)

    model0 = U.build_models(cfg, "f32", fill)
    tr0 = CFMTrainer(model0, iterations=T_MAX)
    ref = [tr0.step(x, c, t, x0) for t, x0 in noise]
    ref_l, ref_n = [float(l) for l, _ in ref], [float(n) for _, n in ref]
    ref_w = {k: v.detach().clone() for k, v in model0.state_dict().items()}

    model1 = U.build_models(cfg, "f32", fill)
    tr1 = CFMTrainer(model1, iterations=T_MAX)
    first = [float(tr1.step(x, c, t, x0)[0]) for t, x0 in noise[:5]]
    assert np.allclose(first, ref_l[:5], rtol=1e-6)
    buf = io.BytesIO()
    torch.save(tr1.checkpoint(), buf)  # through the file format, like _save_model
    buf.seek(0)
    ck = torch.load(buf, map_location="cpu", weights_only=False)
    assert set(ck) == {"model", "optimizer", "scheduler", "ema"}
    assert ck["scheduler"]["last_epoch"] == 5 and ck["scheduler"]["T_max"] == T_MAX
    assert all(float(s["step"]) == 5.0 for s in ck["optimizer"]["state"].values())
    assert len(ck["optimizer"]["state"]) == len(list(model1.parameters())) == len(ck["optimizer"]["param_groups"][0]["params"])

    # losses / gradient norms to f32 rounding; weights to 1e-3 of the tensor's scale: the conditioning-path gradients are sums of f32 atomics whose order
    # varies from run to run, and Adam turns a last-bit difference of a near-zero gradient into +-lr per step (the uninterrupted run against ITSELF
    # differs by 0.5 - 1.3e-4 on c_embedder.0.weight over these 10 steps).  What the file must carry exactly is checked bit for bit right after loading.
    def close(losses, norms, weights, tol_l, tol_w, who):
        assert np.allclose(losses, ref_l[5:], rtol=tol_l), (who, losses, ref_l[5:])
        assert np.allclose(norms, ref_n[5:], rtol=10 * tol_l), (who, norms, ref_n[5:])
        for k, v in ref_w.items():
            assert U.rel_err(weights[k], v) < tol_w, (who, k, U.rel_err(weights[k], v))

    # (i) a fresh trainer constructed with OTHER hyper-parameters: everything comes from the file
    model2 = U.build_models(cfg, "f32", fill)
    tr2 = CFMTrainer(model2, lr=3e-3, betas=(0.5, 0.9), weight_decay=0.0, iterations=7)
    tr2.load_state_dict(ck)
    assert tr2.sync_counters() == {"optimizer_steps": 5, "scheduler_steps": 5, "skipped_max_grad_norm": 0}
    assert torch.equal(tr2.flat_m, tr1.flat_m) and torch.equal(tr2.flat_v, tr1.flat_v) and torch.equal(tr2.flat_p, tr1.flat_p)
    assert (tr2.lr, tr2.betas, tr2.eps, tr2.wd, tr2.iterations) == (tr1.lr, tr1.betas, tr1.eps, tr1.wd, tr1.iterations)
    out = [tr2.step(x, c, t, x0) for t, x0 in noise[5:]]
    close([float(l) for l, _ in out], [float(n) for _, n in out], model2.state_dict(), 1e-6, 1e-3, "trainer")

    # (ii) the reference's own optimizer and scheduler objects, warm-started from the file, on the unchanged-_step route
    model3 = U.build_models(cfg, "f32", fill)
    model3.load_state_dict(ck["model"])
    opt = torch.optim.AdamW([{"params": model3.parameters(), "lr": 1e-4}], betas=(0.9, 0.999), eps=1e-8, weight_decay=0.1)
    opt.load_state_dict(ck["optimizer"])
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=T_MAX, eta_min=0)
    sched.load_state_dict(ck["scheduler"])
    out = [_reference_step(model3, opt, sched, x, c, t, x0) for t, x0 in noise[5:]]
    close([l for l, _ in out], [n for _, n in out], model3.state_dict(), 2e-6, 1e-3, "torch.optim.AdamW")

    # (iii) torch -> trainer: 5 reference steps from scratch, its state_dict()s into a fresh CFMTrainer
    model4 = U.build_models(cfg, "f32", fill)
    opt4 = torch.optim.AdamW([{"params": model4.parameters(), "lr": 1e-4}], betas=(0.9, 0.999), eps=1e-8, weight_decay=0.1)
    sched4 = torch.optim.lr_scheduler.CosineAnnealingLR(opt4, T_max=T_MAX, eta_min=0)
    for t, x0 in noise[:5]:
        _reference_step(model4, opt4, sched4, x, c, t, x0)
    tr4 = CFMTrainer(model4, iterations=3)
    tr4.load_state_dict({"optimizer": opt4.state_dict(), "scheduler": sched4.state_dict()})
    out = [tr4.step(x, c, t, x0) for t, x0 in noise[5:]]
    close([float(l) for l, _ in out], [float(n) for _, n in out], model4.state_dict(), 2e-6, 1e-3, "torch -> trainer")


def test_max_grad_norm_skips_updates_on_the_device_and_clip_grad_value_raises():
    """training.max_grad_norm (base_experiment.py:586-591): after MIN_STEP_SKIP iterations an update whose gradient norm exceeds it is skipped and
    neither the optimizer's step index nor the scheduler advances."""
    from vit4hep_amd.trainer import CFMTrainer

    cfg = O.ds2(2)
    x, c, noise = _data(cfg, 4, 33, 6)
    model = U.build_models(cfg, "f32", O.golden_fill(cfg))
    with pytest.raises(NotImplementedError, match="clip_grad_value"):
        CFMTrainer(model, clip_grad_value=0.5)
    tr = CFMTrainer(model, iterations=50, max_grad_norm=1e-9)  # every gradient norm exceeds this
    tr.MIN_STEP_SKIP = 2  # the reference's constant is 1000 iterations; same rule, shorter test
    snaps = []
    for t, x0 in noise:
        _, gn = tr.step(x, c, t, x0)
        assert np.isfinite(gn.item()) and gn.item() > 1e-9
        snaps.append(tr.flat_p.clone())
    # loop indices 0, 1, 2 are applied (`step > MIN_STEP_SKIP` is false), 3, 4, 5 skipped
    assert not torch.equal(snaps[1], snaps[0]) and not torch.equal(snaps[2], snaps[1])
    assert torch.equal(snaps[3], snaps[2]) and torch.equal(snaps[5], snaps[2])
    assert tr.sync_counters() == {"optimizer_steps": 3, "scheduler_steps": 3, "skipped_max_grad_norm": 3}
    assert tr.step_count == 3 and tr.scheduler_state_dict()["last_epoch"] == 3
    # a generous limit skips nothing and gives the plain trajectory
    m1, m2 = U.build_models(cfg, "f32", O.golden_fill(cfg)), U.build_models(cfg, "f32", O.golden_fill(cfg))
    ta, tb = CFMTrainer(m1, iterations=50, max_grad_norm=1e9), CFMTrainer(m2, iterations=50)
    ta.MIN_STEP_SKIP = 0
    for t, x0 in noise[:3]:
        la, _ = ta.step(x, c, t, x0)
        lb, _ = tb.step(x, c, t, x0)
        assert abs(la.item() - lb.item()) <= 1e-6 * abs(lb.item())
    assert ta.sync_counters()["skipped_max_grad_norm"] == 0


# ---------------------------------------------------------------------------------------------------------------- per-stage autograd nodes
def _grads(model, x, c, t, x0):
    model.zero_grad(set_to_none=True)
    loss = model._loss_from_noise(x, c, t, x0)
    loss.backward()
    return loss.detach(), {k.replace("net.module.", "net."): p.grad.clone() for k, p in model.named_parameters()}


_DETERMINISTIC = ("attn.qkv.weight", "attn.proj.weight", "mlp.fc1.weight", "mlp.fc2.weight")  # split-K slabs + ordered reduce: no float atomics


@pytest.mark.parametrize("mode,late", [("f32", "1"), ("bf16", "1"), ("f32", "0")])
def test_staged_autograd_nodes_match_the_single_node(mode, late, monkeypatch):
    """The chain of per-stage nodes (one forward call, backward stage by stage) gives the gradients of the one-node form: the block weight gradients
    bit for bit (deterministic kernels), the rest within the run-to-run spread of the atomically summed tensors.  late = 1 (round 5, the default with every
    parameter trainable): no join of the library's streams per stage, a node hands on the gradients of the stage before it behind that stage's event
    (v4h_vit_backward_stage); late = 0: the round-4 form, every stage joins and returns its own gradients."""
    import vit4hep_amd.autograd as AG

    monkeypatch.setenv("V4H_STAGED_LATE", late)

    cfg = O.ds2(3)
    fill = O.golden_fill(cfg)
    x, c, noise = _data(cfg, 4, 35, 1)
    t, x0 = noise[0]
    monkeypatch.setenv("V4H_STAGED_AUTOGRAD", "0")
    l0, g0 = _grads(U.build_models(cfg, mode, fill), x, c, t, x0)
    _, g0b = _grads(U.build_models(cfg, mode, fill), x, c, t, x0)
    monkeypatch.setenv("V4H_STAGED_AUTOGRAD", "1")
    log = []
    monkeypatch.setattr(AG, "STAGE_LOG", log)
    l1, g1 = _grads(U.build_models(cfg, mode, fill), x, c, t, x0)
    assert log == [("stage", s) for s in range(cfg.depth + 2)]  # final layer, blocks depth-1 .. 0, embedders: each exactly once, in order
    assert abs(l0.item() - l1.item()) <= 1e-6 * abs(l0.item())  # (the loss sum itself is one atomic per workgroup)
    for k in g0:
        spread = U.rel_err(g0b[k], g0[k])
        if any(k.endswith(d) for d in _DETERMINISTIC):
            assert spread == 0.0, (k, spread)
            assert torch.equal(g1[k], g0[k]), (mode, k, U.rel_err(g1[k], g0[k]))
        else:  # adaLN tensors: grouped contraction (one node) vs per-block contractions (staged) sum in another order; per-sample sums are atomics
            assert U.rel_err(g1[k], g0[k]) <= max(1e-5 if mode == "f32" else 5e-3, 2.0 * spread), (mode, k, U.rel_err(g1[k], g0[k]), spread)
    # a frozen backbone below a trainable head: only the final-layer node runs
    model = U.build_models(cfg, mode, fill)
    for k, p in model.named_parameters():
        p.requires_grad_("final_layer" in k)
    del log[:]
    _, gf = _grads_frozen(model, x, c, t, x0)
    assert log == [("stage", 0)]
    for k, v in gf.items():
        assert torch.equal(v, g1[k]) or U.rel_err(v, g1[k]) < (1e-5 if mode == "f32" else 5e-3), k


def _grads_frozen(model, x, c, t, x0):
    model.zero_grad(set_to_none=True)
    loss = model._loss_from_noise(x, c, t, x0)
    loss.backward()
    return loss.detach(), {k: p.grad.clone() for k, p in model.named_parameters() if p.requires_grad}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _ddp_rank(rank, world, port, staged, out):
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "WORLD_SIZE": str(world)})
    if not staged:
        os.environ["V4H_STAGED_AUTOGRAD"] = "0"
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP

    import vit4hep_amd.autograd as AG
    from vit4hep_amd.parallel import shard_rows

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = O.ds2(3)
        B = 8
        x, c, g = O.synthetic_batch(cfg, B, 37)
        t, x0 = O.synthetic_noise(cfg, B, g)
        lo, hi = shard_rows(B, rank, world)
        model = U.build_models(cfg, "f32", O.golden_fill(cfg))
        # the reference's wrapping (base_experiment.py:161-167); small buckets so that a 3-block network has several
        model.net = DDP(model.net, device_ids=[0], broadcast_buffers=False, bucket_cap_mb=6)
        log = []
        AG.STAGE_LOG = log

        def hook(state, bucket):  # DDP calls this when every gradient of a bucket is ready
            log.append(("bucket", bucket.index()))
            fut = dist.all_reduce(bucket.buffer(), async_op=True).get_future()
            return fut.then(lambda f: f.value()[0].div_(world))

        model.net.register_comm_hook(None, hook)
        # DDP keeps ALL gradients in one bucket during its first iteration (find_unused_parameters=False) and rebuilds the buckets - in the order in which
        # the gradients became ready, with bucket_cap_mb - before the second: the second iteration is what every later one looks like.
        for it in range(2):
            del log[:]
            model.zero_grad(set_to_none=True)
            loss = model._loss_from_noise(x[lo:hi].to(U.DEV), c[lo:hi].to(U.DEV), t[lo:hi].to(U.DEV), x0[lo:hi].to(U.DEV))
            loss.backward()
            torch.cuda.synchronize()
        grads = {k.replace("net.module.", "net."): p.grad.cpu().numpy() for k, p in model.named_parameters()}
        out[rank] = (float(loss.detach()), list(log), grads)
    finally:
        dist.destroy_process_group()


def test_ddp_bucket_hooks_fire_stage_by_stage_on_the_dropin_route():
    """DDP(model.net) with two ranks (both on cuda:0, gloo transport: RCCL refuses two ranks on one device).  With the per-stage nodes DDP's reducer sees
    a bucket complete - and starts its all-reduce - while later backward stages have not even been enqueued; with the single node every bucket comes
    after the whole pass.  Gradients = the single-process gradients of the whole batch."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    results = {}
    for staged in (True, False):
        out = ctx.Manager().dict()
        port = _free_port()
        procs = [ctx.Process(target=_ddp_rank, args=(r, 2, port, staged, out)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(600)
            assert p.exitcode == 0
        results[staged] = dict(out)
    cfg = O.ds2(3)
    x, c, g = O.synthetic_batch(cfg, 8, 37)
    t, x0 = O.synthetic_noise(cfg, 8, g)
    plain = U.build_models(cfg, "f32", O.golden_fill(cfg))
    l0, g0 = _grads(plain, x.to(U.DEV), c.to(U.DEV), t.to(U.DEV), x0.to(U.DEV))
    last_stage = ("stage", cfg.depth + 1)
    for r in (0, 1):
        loss, log, grads = results[True][r]
        buckets = [i for i, e in enumerate(log) if e[0] == "bucket"]
        stages = [e[1] for e in log if e[0] == "stage"]
        assert stages == list(range(cfg.depth + 2)), log
        assert len(buckets) >= 3, log
        # Round 5, shifted hand-over: a node returns the gradients of the stage that ran BEFORE it (no join of the library's streams per stage), so a bucket
        # reaches the communication backend one stage later than in round 4: the first one (final layer + the head of the last block's parameters at this
        # bucket size) before the fourth backward stage is even enqueued, and everything but the gradients of block 0 and of the embedders - which the last
        # node returns together - before the embedder stage.
        assert buckets[0] < log.index(("stage", 3)), log
        before_last = sum(1 for i in buckets if i < log.index(last_stage))
        assert before_last >= 2 and before_last >= (len(buckets) * (cfg.depth - 1)) // (cfg.depth + 1), log
        for k, v in g0.items():
            assert U.rel_err(torch.from_numpy(grads[k]), v) < 1e-4, (r, k, U.rel_err(torch.from_numpy(grads[k]), v))
        # single node under the same wrapper: no stage entries, and the same gradients
        _, log1, grads1 = results[False][r]
        assert log1 and all(e[0] == "bucket" for e in log1)  # (how DDP re-buckets depends on the order in which the gradients became ready)
        for k in _DETERMINISTIC:
            name = f"net.blocks.1.{k}"
            assert np.array_equal(grads[name], grads1[name]), name
    la, lb = results[True][0][0], results[True][1][0]
    assert abs(0.5 * (la + lb) - l0.item()) / l0.item() < 1e-6


# ---------------------------------------------------------------------------------------------------------------- step plumbing
def test_operand_copies_ahead_of_the_forward_change_nothing():
    """v4h_vit_prepare_operands: the casts of the weights on the side stream at the start of a step, the forward waiting for them - the same kernels on
    another queue, so the trajectory is the one of the in-forward casts."""
    from vit4hep_amd.trainer import CFMTrainer

    cfg = O.ds2(2)
    x, c, noise = _data(cfg, 4, 39, 4)
    runs = []
    for ahead in (True, False):
        model = U.build_models(cfg, "bf16", O.golden_fill(cfg))
        tr = CFMTrainer(model, iterations=30)
        tr.prepare_ahead = ahead
        out = [tr.step(x, c, t, x0) for t, x0 in noise]
        runs.append(([float(l) for l, _ in out], {k: v.detach().clone() for k, v in model.state_dict().items()}))
    assert np.allclose(runs[0][0], runs[1][0], rtol=2e-3), (runs[0][0], runs[1][0])  # bf16 mode: atomically summed tensors vary in the last bit run to run
    for k, v in runs[1][1].items():
        got = runs[0][1][k]
        if k.endswith("attn.qkv.bias"):  # the key third's gradient is analytically zero: rounding noise that Adam turns into +-lr per step (docs/history_r01-r04.md section 2)
            D = cfg.hidden_dim
            got, v = torch.cat([got[:D], got[2 * D :]]), torch.cat([v[:D], v[2 * D :]])
        assert U.rel_err(got, v) < 5e-3, k


def test_whole_step_hipgraph_replays_the_update():
    """Opt-in (use_graph / V4H_STEP_GRAPH=1): noise, trajectory, forward, two-stream backward, norm and AdamW captured once and replayed - possible because the
    optimizer's step index and LR position live on the device.  (Slower than the eager launch sequence on ROCm 7.2: docs/history_r01-r04.md section 5; kept as a tested option.)"""
    from vit4hep_amd.trainer import CFMTrainer

    cfg = O.ds2(2)
    x, c, _ = _data(cfg, 4, 41, 1)
    model = U.build_models(cfg, "f32", O.golden_fill(cfg))
    tr = CFMTrainer(model, lr=1e-3, iterations=100)
    tr.use_graph = True
    torch.manual_seed(7)
    losses, snaps = [], []
    for k in range(8):  # two eager warm-up steps, the capture, five replays
        loss, gn = tr.step(x, c)
        losses.append(float(loss))
        assert np.isfinite(float(gn))
        snaps.append(tr.flat_p.clone())
    assert tr._graph is not None and "graph" in tr._graph
    assert tr.sync_counters() == {"optimizer_steps": 8, "scheduler_steps": 8, "skipped_max_grad_norm": 0} and tr.step_count == 8
    assert all(not torch.equal(a, b) for a, b in zip(snaps, snaps[1:]))  # every replay applied an update
    assert all(np.isfinite(losses)) and min(losses[4:]) < losses[0], losses
    # and the checkpoint of a graphed run continues in an eager trainer
    ck = tr.checkpoint()
    model2 = U.build_models(cfg, "f32", O.golden_fill(cfg))
    tr2 = CFMTrainer(model2, iterations=3)
    tr2.load_state_dict(ck)
    assert tr2.sync_counters()["optimizer_steps"] == 8 and torch.equal(tr2.flat_m, tr.flat_m)


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_pipelined_update_gives_the_in_line_trajectory(mode):
    """pipeline_update: AdamW + operand casts staged on the side stream beside the next step's head (v4h_vit_update_ahead) - the same arithmetic per
    element as the single in-line launch, so losses and weights follow the in-line run; the counters, the checkpoint and a mid-run batch-size change
    (the update in flight was prepared for another workspace layout) behave."""
    from vit4hep_amd.trainer import CFMTrainer

    cfg = O.ds2(3)
    x, c, noise = _data(cfg, 6, 43, 8)
    runs = []
    for pipe in (False, True):
        model = U.build_models(cfg, mode, O.golden_fill(cfg))
        tr = CFMTrainer(model, lr=3e-4, iterations=40, pipeline_update=pipe)
        losses = []
        for k, (t, x0) in enumerate(noise):
            if k == 5:  # a smaller batch in the middle of the run
                l, _ = tr.step(x[:4].contiguous(), c[:4].contiguous(), t[:4].contiguous(), x0[:4].contiguous())
            else:
                l, _ = tr.step(x, c, t, x0)
            losses.append(float(l))
        assert tr.sync_counters()["optimizer_steps"] == len(noise)
        sd = {k: v.detach().clone() for k, v in model.state_dict().items()}  # (sync_counters joined the update in flight)
        runs.append((losses, sd, tr.flat_m.clone()))
    tol = 2e-6 if mode == "f32" else 2e-3
    assert np.allclose(runs[0][0], runs[1][0], rtol=tol), (runs[0][0], runs[1][0])
    D = cfg.hidden_dim
    for k, v in runs[0][1].items():
        got = runs[1][1][k]
        if k.endswith("attn.qkv.bias"):  # the key third has an analytically zero gradient: Adam-normalised rounding noise on both sides (docs/history_r01-r04.md section 2)
            keep = torch.cat([torch.arange(0, D), torch.arange(2 * D, 3 * D)]).to(v.device)
            got, v = got[keep], v[keep]
        # bf16: the two runs are separate processes of float atomics; where a gradient element is rounding noise Adam turns a last-bit difference into up to
        # lr per step (8 steps x 3e-4 = 2.4e-3 absolute), i.e. per cent of a small tensor's scale for the handful of elements it hits (5.96e-3 seen once on
        # t_embedder.mlp.0.weight): the bound is on that, the losses above are held to 2e-3
        assert U.rel_err(got, v) < (1e-3 if mode == "f32" else 2e-2), k
    assert U.rel_err(runs[1][2], runs[0][2]) < (1e-3 if mode == "f32" else 2e-2)


def test_pipelined_update_is_bit_identical_on_identical_gradients():
    """One update from the same parameters, moments and gradients through both forms: every element bit for bit."""
    from vit4hep_amd.trainer import CFMTrainer

    cfg = O.ds2(2)
    x, c, noise = _data(cfg, 4, 45, 1)
    t, x0 = noise[0]
    outs = []
    grads = None
    for pipe in (False, True):
        model = U.build_models(cfg, "f32", O.golden_fill(cfg))
        tr = CFMTrainer(model, iterations=40, pipeline_update=pipe)
        tr.step(x, c, t, x0)  # (builds workspace and moments)
        tr.finish()
        if grads is None:
            grads = (tr.flat_g.clone(), tr.flat_p.clone(), tr.flat_m.clone(), tr.flat_v.clone())
        # second update on pinned inputs: overwrite state and gradient with the first run's, then apply the update alone through the trainer's own path
        tr.flat_p.copy_(grads[1]); tr.flat_m.copy_(grads[2]); tr.flat_v.copy_(grads[3])
        orig = tr.loss_and_grads

        def fixed(*a, **k):
            out = orig(*a, **k)
            tr.flat_g.copy_(grads[0])
            return out

        tr.loss_and_grads = fixed
        tr.step(x, c, t, x0)
        tr.finish()
        outs.append((tr.flat_p.clone(), tr.flat_m.clone(), tr.flat_v.clone()))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)


# ---------------------------------------------------------------------------------------------------------------- gradients written, not accumulated
@pytest.mark.parametrize("mode,batch", [("f32", 4), ("bf16", 4), ("bf16", 40)])
def test_backward_overwrites_a_poisoned_gradient_buffer(mode, batch):
    """v4h_plan_set_gradient_mode(1) (what CFMTrainer uses): the pass writes every gradient, so a gradient buffer full of NaN / stale values gives the
    gradients of the zero-fill + accumulate form - bit for bit where that form is deterministic (the weight gradients reduced from split-K slabs), to
    float-atomic reordering elsewhere; the padding between the tensors stays zero; and the plan is back in accumulate mode afterwards."""
    from vit4hep_amd.trainer import CFMTrainer

    cfg = O.ds2(2)
    x, c, noise = _data(cfg, batch, 47, 1)
    t, x0 = noise[0]
    res = {}
    for overwrite in (False, True):
        model = U.build_models(cfg, mode, O.golden_fill(cfg))
        tr = CFMTrainer(model, iterations=40)
        tr.overwrite_grads = overwrite
        if overwrite:
            for lo, p in zip(tr.offsets, tr.params):
                tr.flat_g[lo : lo + p.numel()] = float("nan")  # (the padding keeps its zeros, as in a run)
        loss = tr.loss_and_grads(x, c, t, x0)
        res[overwrite] = (loss.item(), tr.flat_g.clone(), [n for n, _ in model.named_parameters()], tr)
    (l0, g0, names, tr0), (l1, g1, _, tr1) = res[False], res[True]
    assert abs(l0 - l1) <= 1e-6 * abs(l0)  # (the bf16 forward sums the conditioning terms with float atomics: last-bit differences between two runs)
    assert torch.isfinite(g1).all()
    order = {p.data_ptr(): n for n, p in tr1.model.named_parameters()}
    for lo, p in zip(tr1.offsets, tr1.params):
        a, b = g0[lo : lo + p.numel()], g1[lo : lo + p.numel()]
        name = order.get(p.data_ptr(), "?")
        if mode == "f32" and p.dim() == 2 and ".blocks." in name and "adaLN" not in name and batch * 135 >= 256:  # split-K slabs: deterministic
            assert torch.equal(a, b), name
        else:
            assert U.rms_err(b, a) < (1e-5 if mode == "f32" else 1e-3), name
    pad = torch.ones_like(g1, dtype=torch.bool)
    for lo, p in zip(tr1.offsets, tr1.params):
        pad[lo : lo + p.numel()] = False
    assert float(g1[pad].abs().max()) == 0.0 if pad.any() else True
    # the shared plan accumulates again: the autograd node on the same network still matches
    model = tr1.model
    model.zero_grad(set_to_none=True)
    la = model._loss_from_noise(x.to(U.DEV), c.to(U.DEV), t.to(U.DEV), x0.to(U.DEV))
    la.backward()
    for n, p in model.named_parameters():
        lo = tr1.offsets[[q.data_ptr() for q in tr1.params].index(p.data_ptr())]
        assert U.rms_err(p.grad.reshape(-1), g1[lo : lo + p.numel()]) < (1e-5 if mode == "f32" else 1e-3), n


def test_bounded_host_run_ahead_changes_nothing():
    """CFMTrainer.max_steps_ahead (default 2): the host waits for the step issued that many steps ago before it enqueues the next one - same trajectory as
    the unbounded loop, never more than that many events outstanding."""
    from vit4hep_amd.trainer import CFMTrainer

    cfg = O.ds2(2)
    x, c, noise = _data(cfg, 4, 57, 5)
    out = []
    for ahead in (0, 1, 2):
        model = U.build_models(cfg, "f32", O.golden_fill(cfg))
        tr = CFMTrainer(model, iterations=40)
        tr.max_steps_ahead = ahead
        rec = []
        for t, x0 in noise:
            loss, gn = tr.step(x, c, t, x0)
            assert len(tr._ahead_events) <= ahead
            rec.append((loss, gn))
        out.append([(float(a), float(b)) for a, b in rec])
    for other in out[1:]:
        for (l0, g0), (l1, g1) in zip(out[0], other):
            assert abs(l0 - l1) <= 2e-6 * abs(l0) and abs(g0 - g1) <= 2e-5 * abs(g0)


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_written_gradients_stage_by_stage_equal_the_whole_pass(mode):
    """Gradient mode 1 with the backward issued one stage per call (each call zeroes and writes only its own stage's tensors; the adaLN gradients are then
    per-block launches instead of the whole pass's grouped contraction): every tensor equals the single-call pass on a poisoned buffer."""
    from vit4hep_amd.autograd import run_backward, run_forward
    from vit4hep_amd.trainer import CFMTrainer

    cfg = O.ds2(2)
    x, c, noise = _data(cfg, 4, 71, 1)
    t, x0 = noise[0]
    model = U.build_models(cfg, mode, O.golden_fill(cfg))
    tr = CFMTrainer(model, iterations=40)
    lib = _lib.load()
    plan_h = tr.net._get_plan().handle
    xt = (1 - t.view(-1, 1, 1, 1, 1)) * x0 + t.view(-1, 1, 1, 1, 1) * x
    nst = len(tr.stage_slices)
    res = []
    for staged in (False, True):
        with torch.no_grad():
            v, ws = run_forward(tr.net, tr.p_views, xt, t.reshape(-1), c, True)
            dv = torch.full_like(v, 1e-3)
            tr.flat_g.fill_(float("nan"))
            pad = torch.ones_like(tr.flat_g, dtype=torch.bool)
            for lo, p in zip(tr.offsets, tr.params):
                pad[lo : lo + p.numel()] = False
            tr.flat_g[pad] = 0.0
            _lib.check(lib.v4h_plan_set_gradient_mode(plan_h, 1), "mode")
            try:
                if staged:
                    for st in range(nst):
                        run_backward(tr.net, tr.p_views, tr.g_views, dv, ws, st, st)
                else:
                    run_backward(tr.net, tr.p_views, tr.g_views, dv, ws, 0, nst - 1)
            finally:
                lib.v4h_plan_set_gradient_mode(plan_h, 0)
            torch.cuda.synchronize()
            res.append(tr.flat_g.clone())
    assert torch.isfinite(res[0]).all() and torch.isfinite(res[1]).all()
    names = {p.data_ptr(): n for n, p in model.named_parameters()}
    for lo, p in zip(tr.offsets, tr.params):
        a, b = res[0][lo : lo + p.numel()], res[1][lo : lo + p.numel()]
        assert U.rms_err(b, a) < (1e-5 if mode == "f32" else 2e-3), names.get(p.data_ptr())
