"""CPU, world_size 2, gloo: the data-parallel path of the CFM step (vit4hep_amd/parallel.py).

(1) BucketReducer sums contiguous slices of the flat gradient buffer across ranks, stage by stage, like the trainer does.
(2) Folding 1/world into the loss-gradient seed and SUM-reducing equals the gradient of the global-batch mean loss
    (what DDP's averaging gives, reference experiments/base_experiment.py:161-167) - checked with the CPU oracle.
"""

import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fn, out):
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "WORLD_SIZE": str(world)})
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fn(rank, world, out)
    finally:
        dist.destroy_process_group()


def _run(fn, world=2):
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, fn, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    return dict(out)


def _bucket_job(rank, world, out):
    from vit4hep_amd.parallel import BucketReducer, shard_rows

    torch.manual_seed(rank)
    flat = torch.randn(1000)
    mine = flat.clone()
    red = BucketReducer(flat)
    slices = [(900, 1000), (500, 900), (100, 500), (0, 100)]  # backward order: last parameters first
    for lo, hi in slices:
        red.reduce_slice(lo, hi)
    red.finish()
    gathered = [torch.zeros(1000) for _ in range(world)]
    dist.all_gather(gathered, mine)
    assert torch.allclose(flat, sum(gathered), atol=1e-6)
    rows = [shard_rows(100003, r, world) for r in range(world)]
    assert rows[0][0] == 0 and rows[-1][1] == 100003 and all(a[1] == b[0] for a, b in zip(rows, rows[1:]))
    out[rank] = True


def test_bucket_reducer_sums_slices():
    assert _run(_bucket_job) == {0: True, 1: True}


def _dp_equivalence_job(rank, world, out):
    from oracle import vit_cfm_oracle as O
    from vit4hep_amd.parallel import BucketReducer

    torch.set_num_threads(2)
    cfg = O.ds2(1)
    p = O.golden_fill(cfg)
    B = 2  # per rank
    x, c, g = O.synthetic_batch(cfg, B * world, 3)
    t, x0 = O.synthetic_noise(cfg, B * world, g)
    sl = slice(rank * B, (rank + 1) * B)
    # local gradient of the local-mean loss, seed scaled by 1/world as CFMTrainer does
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    loss, _ = O.cfm_loss(leaves, x[sl], c[sl], t[sl], x0[sl], cfg)
    (loss / world).backward()
    names = list(p)
    sizes = [(leaves[k].numel() + 63) // 64 * 64 for k in names]
    flat = torch.zeros(sum(sizes))
    offs, o = [], 0
    for k, n in zip(names, sizes):
        flat[o : o + leaves[k].numel()] = leaves[k].grad.flatten()
        offs.append(o)
        o += n
    red = BucketReducer(flat)
    stage_bounds = [(offs[11 + 10 * cfg.depth], o), (offs[11], offs[11 + 10 * cfg.depth]), (0, offs[11])]  # final layer, block 0, embedders
    for lo, hi in stage_bounds:
        red.reduce_slice(lo, hi)
    red.finish()
    # reference: gradient of the global-batch mean loss
    _, _, ref = O.loss_and_grads(p, x, c, t, x0, cfg)
    worst = 0.0
    for k, off in zip(names, offs):
        got = flat[off : off + ref[k].numel()].view_as(ref[k])
        worst = max(worst, float((got - ref[k]).abs().max() / (ref[k].abs().max() + 1e-12)))
    out[rank] = worst


def test_sum_of_scaled_local_grads_equals_global_mean_grad():
    res = _run(_dp_equivalence_job)
    assert set(res) == {0, 1} and max(res.values()) < 1e-4, res


def _bucket_bf16_job(rank, world, out):
    from vit4hep_amd.parallel import BucketReducer

    torch.manual_seed(rank)
    flat = torch.randn(1000)
    mine = flat.clone()
    red = BucketReducer(flat, compress="bf16")
    for lo, hi in [(900, 1000), (500, 900), (100, 500), (0, 100)]:
        red.reduce_slice(lo, hi)
    red.finish()
    gathered = [torch.zeros(1000) for _ in range(world)]
    dist.all_gather(gathered, mine)
    exact = sum(gathered)
    # what the compressed path is defined to give: every rank's slice rounded to bf16, summed, the sum rounded to bf16 again, kept as f32
    model = sum(g.to(torch.bfloat16).float() for g in gathered).to(torch.bfloat16).float()
    assert flat.dtype == torch.float32 and torch.equal(flat, model), float((flat - model).abs().max())
    assert float((flat - exact).abs().max()) <= 2.0 ** -7 * float(exact.abs().max())  # bf16 rounding of the f32 path
    out[rank] = True


def test_bucket_reducer_bf16_all_reduce_keeps_an_f32_master_gradient():
    """compress="bf16" (reference gradient exchange experiments/base_experiment.py:161-167): half the bytes on the wire, the result
    equals the f32 sum to bf16 rounding and lands in the f32 buffer the optimizer reads."""
    assert _run(_bucket_bf16_job) == {0: True, 1: True}
    with pytest.raises(ValueError):
        from vit4hep_amd.parallel import BucketReducer

        BucketReducer(torch.zeros(4), compress="fp8")
