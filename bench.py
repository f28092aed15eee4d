"""bench.py - CFM train steps/s of the CaloChallenge-ds2 shape ViT (depth 6, hidden 480, 6 heads, mlp 1920; bs = 128 per GPU)
on N MI355X, through the HIP path, plus the MFMA roofline fraction and the CPU-oracle baseline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode bf16|f32] [--workload ds2|ds3|ds2_d2|lemurs|ds1_photons|ds1_pions|calogan|calohad]
                    [--no-cpu-baseline] [--no-op-rates]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one BaseExperiment._step of the reference (experiments/base_experiment.py:555-602) on one synthetic batch per GPU:
t ~ U(0,1), x0 ~ N(0,1), x_t / target, ViT forward, MSE, backward, gradient all-reduce (N > 1), grad-norm clip(1000), AdamW,
cosine LR.  Inputs x, c are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.
`value` is the whole-job rate in units of bs=128 steps: N * K / time (weak scaling: every GPU keeps 128 samples per step).
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

# GPU_MAX_HW_QUEUES is left at the runtime's default.  Rounds 3-5 set it to 8 here ("every stream its own hardware queue"); the overlap does not need it - the
# library's weight-gradient stream and the reducer's communication stream are priority streams with queues of their own, 4 / 8 / 16 measured the same at N = 1
# (profiles/r04_notes.md) - and with 8 the two-rank rehearsal of the full-size step DEADLOCKED in the reducer's finish() (both ranks on one card, gloo's staging
# streams on top of the library's: 4 and 24 queues finish, 8 hangs: tools/experiments/rehearsal_bisect.sh, profiles/r05_notes.md section 10).  A value a run beside
# RCCL's own streams could hang on is not worth a neutral knob.

import torch  # noqa: E402

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

# HBM bytes per update step come from the PMC counters (separate rocprofv3 --pmc passes over THIS program, tools/pmc_step.sh; FETCH_SIZE
# doubled per MI355X_MICROARCH.md because the reads are 16-byte-per-lane streams, WRITE_SIZE as is).  The profiling script writes
# profiles/step_hbm_traffic.json together with the digest of the kernel sources it measured; a figure whose digest is not the digest of
# the library being run is NOT reported (traffic: null) - a stale constant is worse than none.
TRAFFIC_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "step_hbm_traffic.json")


def measured_traffic(workload, mode):
    try:
        from vit4hep_amd.build import kernel_digest as _digest

        with open(TRAFFIC_FILE) as fh:
            rec = json.load(fh)
        ent = rec.get(f"{workload}/{mode}")
        if ent and ent.get("kernel_digest") == _digest():
            return float(ent["bytes_per_step"]), None
        return None, "profiles/step_hbm_traffic.json was measured on other kernel sources (digest mismatch): re-run tools/pmc_step.sh"
    except (OSError, ValueError, KeyError) as e:
        return None, f"no usable traffic file: {e}"


BF16_DENSE_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
F32_MFMA_PEAK_TFLOPS = 157.3  # f32-input MFMA = vector rate

def _grid(shape, patch, depth, B, desc, cond=46, kind="calochallenge"):
    return {"kind": kind, "shape": tuple(shape), "segments": [(tuple(shape), tuple(patch))], "depth": depth, "B": B, "cond": cond, "desc": desc}


def _segs(kind, list_shape, list_patch, cond, B, desc, depth=6):
    return {"kind": kind, "shape": (sum(s[0] * s[1] * s[2] for s in list_shape),), "segments": [(tuple(s), tuple(p)) for s, p in zip(list_shape, list_patch)],
            "depth": depth, "B": B, "cond": cond, "desc": desc}


WORKLOADS = {
    # the headline (BASELINE.json) and its siblings: regular voxel grids
    "ds2": _grid((45, 16, 9), (3, 16, 1), 6, 128, "CaloChallenge-ds2 shape CFM, full ViT (depth 6), bs=128 per GPU"),
    "ds3": _grid((45, 50, 18), (3, 10, 3), 6, 64, "CaloChallenge-ds3 shape CFM, full ViT (depth 6), bs=64 per GPU"),
    "ds2_d2": _grid((45, 16, 9), (3, 16, 1), 2, 8, "CaloChallenge-ds2 shape CFM, ViT depth 2, bs=8 (reference CPU-runnable case)"),
    # the other ViT-CFM geometries of the reference; batch sizes of configs/training/cfm/shape*.yaml
    "lemurs": _grid((45, 16, 9), (3, 16, 1), 6, 64, "LEMURS shape CFM (ds2 grid, 53 conditions), bs=64 per GPU", cond=53, kind="lemurs"),
    "ds1_photons": _segs("ds1", [(1, 8, 5), (1, 16, 10), (1, 19, 10), (1, 5, 5), (1, 5, 5)], [(1, 1, 5)] * 5, 6, 64,
                         "CaloChallenge-ds1 photons shape CFM (5 layer segments, 88 tokens), bs=64 per GPU"),
    "ds1_pions": _segs("ds1", [(1, 8, 5), (1, 10, 10), (1, 10, 10), (1, 5, 5), (1, 15, 10), (1, 16, 10), (1, 10, 5)], [(1, 1, 5)] * 7, 8, 64,
                       "CaloChallenge-ds1 pions shape CFM (7 layer segments, 125 tokens), bs=64 per GPU"),
    "calogan": _segs("calogan", [(1, 96, 3), (1, 12, 12), (1, 6, 12)], [(1, 6, 1), (1, 2, 3), (1, 2, 3)], 4, 64,
                     "CaloGAN e+ shape CFM (3 layer segments, 84 tokens), bs=64 per GPU"),
    "calohad": _segs("calohad", [(10, 15, 15), (48, 30, 30)], [(5, 5, 3), (3, 5, 5)], 59, 32,
                     "CaloHadronic shape CFM (ECal + HCal segments, 606 tokens of 75), bs=32 per GPU"),
}


def tokens_and_patch_dim(w):
    T = sum((s[0] // p[0]) * (s[1] // p[1]) * (s[2] // p[2]) for s, p in w["segments"])
    p = w["segments"][0][1]
    return T, p[0] * p[1] * p[2]


def fwd_flops_per_sample(T, P, depth, D=480, M=1920, K=46, F=256):
    """2*m*n*k of every contraction of one forward, per sample."""
    return depth * T * (2 * D * 3 * D + 2 * D * D + 4 * D * M + 4 * T * D) + 4 * P * D * T + (depth * 12 * D * D + 4 * D * D + 2 * (K * D + D * D) + 2 * (F * D + D * D))


def build_model(w, mode, device):
    from vit4hep_amd import CaloChallengeCFM, ViT

    T, P = tokens_and_patch_dim(w)
    num_patches = [[s[0] // p[0], s[1] // p[1], s[2] // p[2]] for s, p in w["segments"]]
    net = ViT({"dim": 3, "condition_dim": w["cond"], "hidden_dim": 480, "out_channels": 1, "depth": w["depth"], "num_heads": 6, "mlp_ratio": 4, "attn_drop": 0.0,
               "proj_drop": 0.0, "pos_embedding_coords": "cylindrical", "temperature": 10000, "learn_pos_embed": True, "causal_attn": False,
               "checkpoint_grads": False, "num_patches": num_patches, "patch_dim": P, "use_torch_sdpa": False, "amd_mode": mode})
    common = dict(in_channels=1, time_distribution="uniform", trajectory="linear", odeint_kwargs={"method": "rk4", "options": {"step_size": 0.05}},
                  shape=list(w["shape"]))
    list_shape = [list(s) for s, _ in w["segments"]]
    list_edges = [s[0] * s[1] * s[2] for s in list_shape]
    list_patch = [list(p) for _, p in w["segments"]]
    if w["kind"] == "calochallenge":
        model = CaloChallengeCFM(net, list_patch[0], **common)
    elif w["kind"] == "lemurs":
        from vit4hep_amd.experiments.lemurs.model import LEMURSCFM

        model = LEMURSCFM(net, list_patch[0], **common)
    elif w["kind"] == "ds1":
        from vit4hep_amd.experiments.calochallenge.calochallenge_cfm.model import CaloChallengeCFM_DS1

        model = CaloChallengeCFM_DS1(net, list_shape, list_edges, list_patch[0], **common)
    elif w["kind"] == "calogan":
        from vit4hep_amd.experiments.calogan.model import CaloGANCFM

        model = CaloGANCFM(net, list_shape, list_edges, list_patch, **common)
    else:
        from vit4hep_amd.experiments.calohadronic.model import CaloHadCFM

        model = CaloHadCFM(net, list_shape, list_edges, list_patch, **common)
    # random-init weights of that architecture; the zero-initialised adaLN / output tensors (nn/vit.py:174-183) are given small
    # random values so the step does the same arithmetic as a model a few hundred iterations into training (no zero operands).
    g = torch.Generator().manual_seed(1234)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if "adaLN_modulation" in name or "final_layer.linear" in name:
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)
    model = model.to(device)
    model.device, model.dtype = torch.device(device), torch.float32
    return model


def synthetic(shape, B, seed, device, cond=46):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((B, 1, *shape), generator=g)
    c = torch.cat([torch.randn((B, cond - 1), generator=g), torch.rand((B, 1), generator=g)], dim=1)
    return x.to(device), c.to(device)


OP_RATE_SETS = 6  # buffer sets each timed operator rotates over: with >= 300 MB touched between two uses of a line nothing is served from the 256 MiB Infinity Cache


def _time_rotating(calls, reps):
    """Mean HIP-event time per call of `calls[k % len(calls)]()`: consecutive calls use different buffer sets, so operands and output lines are cold (as they are
    inside the update step; a loop over ONE buffer set re-uses cache-resident output lines and reads 6-9 % fast - docs/history_r01-r04.md section 5 item 5)."""
    for k in range(len(calls)):
        calls[k]()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(reps):
        calls[k % len(calls)]()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def op_rates(mode, BT, device, reps=24):
    """Live HIP-event timing of the four block GEMM shapes (forward form) through the C ABI: TFLOP/s each, cold caches (buffer sets rotated)."""
    from vit4hep_amd import _lib

    lib = _lib.load()
    dt = torch.bfloat16 if mode == "bf16" else torch.float32
    out = {"_note": f"every operator rotates over {OP_RATE_SETS} buffer sets (activations and outputs cold, weights shared): rates as inside the step, not warm-cache rates"}
    s = _lib.stream_ptr(device)
    for name, (J, K) in {"qkv 480->1440": (1440, 480), "proj 480->480": (480, 480), "fc1 480->1920": (1920, 480), "fc2 1920->480": (480, 1920)}.items():
        Q = torch.randn((J, K), device=device).to(dt)
        keep, calls = [], []
        for _ in range(OP_RATE_SETS):
            P = torch.randn((BT, K), device=device).to(dt)
            o = torch.empty((BT, J), device=device, dtype=dt)
            keep.append((P, o))
            args = (_lib.MODES[mode], _lib.ptr(P), K, 0, _lib.ptr(Q), K, 0, None, _lib.ptr(o), J, 0, BT, J, K, 1, None, s)
            calls.append(lambda a=args: _lib.check(lib.v4h_op_gemm(*a)))
        us = _time_rotating(calls, reps)
        out[name] = {"us": round(us, 2), "tflops": round(2.0 * BT * J * K / us / 1e6, 1)}
        del keep, calls
    # the ViT attention block the north star names (qkv projection, softmax(q k^T / sqrt(dh)) v, output projection; forward), launch to launch
    if BT % 135 == 0:  # (ds2 / LEMURS token count; other workloads: per-kernel figures only)
        H, dh, D = 6, 80, 480
        Tn = 135
        Bn = BT // Tn
        wq = torch.randn((3 * D, D), device=device).to(dt)
        wp = torch.randn((D, D), device=device).to(dt)
        keep, calls = [], []
        for _ in range(OP_RATE_SETS):
            u = torch.randn((BT, D), device=device).to(dt)
            qkv = torch.empty((BT, 3 * D), device=device, dtype=dt)
            o = torch.empty((BT, D), device=device, dtype=dt)
            y = torch.empty((BT, D), device=device, dtype=dt)
            lse = torch.empty((Bn, H, Tn), device=device, dtype=torch.float32)
            keep.append((u, qkv, o, y, lse))

            def block(u=u, qkv=qkv, o=o, y=y, lse=lse):
                _lib.check(lib.v4h_op_gemm(_lib.MODES[mode], _lib.ptr(u), D, 0, _lib.ptr(wq), D, 0, None, _lib.ptr(qkv), 3 * D, 0, BT, 3 * D, D, 1, None, s))
                _lib.check(lib.v4h_op_attention_fwd(_lib.MODES[mode], _lib.ptr(qkv), _lib.ptr(o), _lib.ptr(lse), Bn, Tn, H, dh, s))
                _lib.check(lib.v4h_op_gemm(_lib.MODES[mode], _lib.ptr(o), D, 0, _lib.ptr(wp), D, 0, None, _lib.ptr(y), D, 0, BT, D, D, 1, None, s))

            calls.append(block)
        us = _time_rotating(calls, reps)
        fl = 2.0 * BT * D * 3 * D + 4.0 * Bn * H * Tn * Tn * dh + 2.0 * BT * D * D
        out["attention block fwd (qkv + attention + proj)"] = {"us": round(us, 2), "tflops": round(fl / us / 1e6, 1),
                                                                "frac_of_spec_peak": round(fl / us / 1e6 / (BF16_DENSE_PEAK_TFLOPS if mode == "bf16" else F32_MFMA_PEAK_TFLOPS), 4)}
        del keep, calls
    # the dominant kernel of the step: the split-K weight gradients (dW = dY^T X over the tokens; partial slabs + ordered reduce, bias-gradient column sums on)
    for name, (I, J) in {"wgrad qkv 1440x480": (1440, 480), "wgrad proj 480x480": (480, 480), "wgrad fc1 1920x480": (1920, 480), "wgrad fc2 480x1920": (480, 1920)}.items():
        sk = int(lib.v4h_op_gemm_wgrad_splitk(_lib.MODES[mode], I, J, BT))  # the split the backward pass itself uses
        o = torch.zeros((I, J), device=device)
        slab = torch.empty((sk, I, J), device=device)
        cs = torch.zeros(I, device=device)
        keep, calls = [], []
        for _ in range(OP_RATE_SETS):
            P = torch.randn((BT, I), device=device).to(dt)
            Q = torch.randn((BT, J), device=device).to(dt)
            keep.append((P, Q))
            args = (_lib.MODES[mode], _lib.ptr(P), I, _lib.ptr(Q), J, _lib.ptr(slab), _lib.ptr(o), I, J, BT, sk, _lib.ptr(cs), s)
            calls.append(lambda a=args: _lib.check(lib.v4h_op_gemm_wgrad_slab(*a)))
        us = _time_rotating(calls, reps)
        out[name] = {"us": round(us, 2), "tflops": round(2.0 * BT * I * J / us / 1e6, 1), "splitk": sk}
        del keep, calls
    return out


# V4H_BENCH_REHEARSAL=1: rehearse the N-rank path on a box with ONE GPU - every rank computes on cuda:0 and the collectives go over gloo (RCCL refuses two ranks on
# one device).  Exercises the launcher, the rendezvous, the bucketed all-reduce behind the library's stage events and the max-over-ranks timing; its numbers mean
# nothing (N ranks share one card) and the JSON line says "rehearsal": true.
REHEARSAL = os.environ.get("V4H_BENCH_REHEARSAL") == "1"


def launch_ranks(n):
    """`python bench.py --gpus N` outside a launcher: start N ranks (one process per GPU, RCCL over xGMI) the way the reference's main.py:9-26
    spawns its own, as CHILD processes of a parent that never touches the GPU, pass rank 0's JSON line through and return the children's code."""
    import socket
    import subprocess

    have = torch.cuda.device_count()  # (counting devices does not initialise the runtime)
    if have < n and not REHEARSAL:
        print(f"bench.py: --gpus {n} but only {have} GPU(s) visible", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__), *sys.argv[1:]]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    for ln in r.stdout.splitlines():
        if not ln.startswith('{"metric"'):
            print(ln, file=sys.stderr)
    if r.returncode != 0 or not line:
        print(f"bench.py: the {n}-rank run failed (rc {r.returncode})", file=sys.stderr)
        return r.returncode or 1
    print(line[-1], flush=True)
    return 0


def sampling_rates(model, w, device, peak, batch=256, reps=8):
    """BASELINE config 5 (ds2 shape sampling, one GPU): CaloChallengeCFM.sample_batch (calochallenge_cfm/model.py:68-94) at the reference's
    sampling batch of 256 with its default fixed-grid RK4 (step 0.05 = 80 network evaluations) and with Heun (40), timed after the training region."""
    T, P = tokens_and_patch_dim(w)
    flops = fwd_flops_per_sample(T, P, w["depth"], K=w["cond"])
    _, c = synthetic(w["shape"], batch, 7, device, cond=w["cond"])
    was_training = model.training
    model.eval()
    saved = model.odeint_kwargs
    out = {"batch": batch, "workload": "ds2 shape sampling, bf16 network evaluations, conditions resident in HBM"}
    for method, nfe in (("rk4", 80), ("heun2", 40)):
        model.odeint_kwargs = {"method": method, "options": {"step_size": 0.05}}
        model.sample_batch(c)
        torch.cuda.synchronize()
        times = []
        for _ in range(reps):  # every batch timed on its own (host clock around a synchronised batch): the MEDIAN is reported, min / max beside it
            t0 = time.perf_counter()
            s = model.sample_batch(c)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        assert bool(torch.isfinite(s).all())
        dt = sorted(times)[len(times) // 2]
        tf = nfe * batch * flops / dt / 1e12
        out[method] = {"nfe": nfe, "batches": reps, "showers_per_s": round(batch / dt, 1), "s_per_100k": round(1e5 / batch * dt, 2), "tflops": round(tf, 1),
                       "frac": round(tf / peak, 4), "batch_ms_min_median_max": [round(min(times) * 1e3, 2), round(dt * 1e3, 2), round(max(times) * 1e3, 2)]}
    model.odeint_kwargs = saved
    model.train(was_training)
    return out


def box_calibration(device):
    """What THIS card does today on (a) a loop of nothing but bf16 MFMAs on random operands and (b) a 16-byte-per-lane copy (csrc/v4h_calib.hip):
    the boxes of the pool differ by 2-4 % in step rate, and these two figures say which kind of box a line was measured on."""
    from vit4hep_amd import _lib

    lib = _lib.load()
    s = _lib.stream_ptr(device)
    blocks = 512  # two 4-wave workgroups per CU: two waves per SIMD, as in the contraction kernels
    rnd = torch.randn(blocks * 256 * 32, device=device).to(torch.bfloat16)
    sink = torch.zeros(64, device=device)
    iters = 100000

    def timed(fn, reps):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / reps

    dt = timed(lambda: _lib.check(lib.v4h_calib_mfma_loop(_lib.ptr(rnd), _lib.ptr(sink), iters, blocks, s), "v4h_calib_mfma_loop"), 3)
    mfma = blocks * 4 * iters * 16 * (2.0 * 16 * 16 * 32) / dt / 1e12
    nbytes = 1 << 29  # 512 MiB each way: twice the Infinity Cache
    src = torch.empty(nbytes, dtype=torch.uint8, device=device).random_(0, 255)
    dst = torch.empty_like(src)
    dc = timed(lambda: _lib.check(lib.v4h_calib_copy(_lib.ptr(src), _lib.ptr(dst), nbytes, s), "v4h_calib_copy"), 5)
    del src, dst
    return {"mfma_loop_tflops": round(mfma, 1), "copy_tb_per_s": round(2.0 * nbytes / dc / 1e12, 3),
            "note": "bf16 MFMA-only loop on random register operands (2 waves / SIMD, 0.14 s) and a 2 x 512 MiB float4 copy, measured in this run after the timed region"}


def other_rates(mode, device):
    """On the driver's clock too: BASELINE config 3 (ds3, bs = 64) through the fused trainer, and config 2 through the UNCHANGED-_step route - the
    autograd node with torch.optim.AdamW, two clip_grad_norm_().cpu().item() and loss.item() per step (reference experiments/base_experiment.py:555-602)."""
    from vit4hep_amd.trainer import CFMTrainer

    out = {}
    w = WORKLOADS["ds3"]
    model = build_model(w, mode, device)
    tr = CFMTrainer(model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.1, clip_grad_norm=1000.0, iterations=50000,
                    pipeline_update=os.environ.get("V4H_PIPELINE_UPDATE", "0") == "1")
    x, c = synthetic(w["shape"], w["B"], seed=3, device=device, cond=w["cond"])
    for _ in range(3):
        tr.step(x, c)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        _, gn = tr.step(x, c)
    tr.finish()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    CFMTrainer.check_finite(gn)
    T, P = tokens_and_patch_dim(w)
    out["ds3_b64_steps_per_s"] = round(1.0 / dt, 2)
    out["ds3_b64_frac_of_spec_peak"] = round(3.0 * w["B"] * fwd_flops_per_sample(T, P, w["depth"], K=w["cond"]) / dt / 1e12 / (BF16_DENSE_PEAK_TFLOPS if mode == "bf16" else F32_MFMA_PEAK_TFLOPS), 4)
    del tr, model, x, c
    torch.cuda.empty_cache()

    w = WORKLOADS["ds2"]
    model = build_model(w, mode, device)
    opt = torch.optim.AdamW([{"params": model.parameters(), "lr": 1e-4}], betas=(0.9, 0.999), eps=1e-8, weight_decay=0.1)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=50000, eta_min=0)
    x, c = synthetic(w["shape"], w["B"], seed=4, device=device, cond=w["cond"])
    model.train()

    def ref_step():  # BaseExperiment._step, line by line
        loss = model._batch_loss([x, c])
        opt.zero_grad(set_to_none=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.net.parameters(), float("inf")).cpu().item()
        gnorm = torch.nn.utils.clip_grad_norm_(model.parameters(), 1000.0, error_if_nonfinite=True).cpu().item()
        opt.step()
        sched.step()
        return loss.item(), gnorm

    for _ in range(3):
        ref_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        ref_step()
    torch.cuda.synchronize()
    out["dropin_step_steps_per_s"] = round(10.0 / (time.perf_counter() - t0), 2)
    out["dropin_step_note"] = ("ds2 bs=128, unchanged BaseExperiment._step: autograd node + torch.optim.AdamW + CosineAnnealingLR, 2 x clip_grad_norm_().cpu().item(), "
                               "loss.item(); 10 timed steps")
    del opt, sched, model
    torch.cuda.empty_cache()
    return out


def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as fh:
            for ln in fh:
                if ln.startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(w, budget_s=16.0):
    """The CPU oracle (kind 'port') timed on this box's host cores on a bounded sample of the same workload."""
    from oracle import vit_cfm_oracle as O

    depth, B = w["depth"], w["B"]
    if len(w["shape"]) == 3:
        cfg = O.ViTConfig(shape=w["shape"], patch_shape=w["segments"][0][1], depth=depth, condition_dim=w["cond"])
    else:
        cfg = O.ViTConfig(shape=w["shape"], patch_shape=(), depth=depth, condition_dim=w["cond"], segments=tuple(w["segments"]))
    # the GPU box gives one GPU's share of the host: 16 cores (its os.cpu_count() reports the whole machine)
    threads = int(os.environ.get("V4H_CPU_THREADS", min(cpu_share(), 16)))
    torch.set_num_threads(threads)
    p = O.golden_fill(cfg)
    st = O.AdamWState()
    x, c, g = O.synthetic_batch(cfg, B, 0)
    n, t0 = 0, time.perf_counter()
    while True:
        t, x0 = O.synthetic_noise(cfg, B, g)
        O.train_step(p, st, x, c, t, x0, cfg)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 50:
            break
    rec = {"value": round(n / el, 4), "unit": "steps/s", "cores": threads, "kind": "port", "cpu": cpu_model_name(),
           "sample": f"{n} full update steps of the same workload (B={B}, depth {depth}) with the PyTorch-CPU oracle, {el:.1f} s"}
    # one thread, on a slice of the batch (a whole bs=128 step takes about a minute on one core): B/16 samples, scaled
    torch.set_num_threads(1)
    Bs = max(1, B // 16)
    xs, cs, gs = O.synthetic_batch(cfg, Bs, 1)
    ts, x0s = O.synthetic_noise(cfg, Bs, gs)
    t1 = time.perf_counter()
    O.train_step(O.golden_fill(cfg), O.AdamWState(), xs, cs, ts, x0s, cfg)
    e1 = time.perf_counter() - t1
    rec["one_thread"] = {"value": round(Bs / B / e1, 5), "unit": "steps/s", "cores": 1,
                         "sample": f"one update step on {Bs} of the {B} samples ({e1:.1f} s), scaled to the full batch"}
    torch.set_num_threads(threads)
    return rec


def cgroup_cpu_stat():
    for pth in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        if os.path.exists(pth):
            return {ln.split()[0]: int(ln.split()[1]) for ln in open(pth)}
    return {}


def cpu_share():
    """CPUs this process may really use: the container's CFS quota (cgroup cpu.max / cfs_quota_us) when there is one, else its affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        if os.path.exists("/sys/fs/cgroup/cpu.max"):
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()
            if q != "max":
                n = min(n, max(1, int(q) // int(per)))
        elif os.path.exists("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
    except (OSError, ValueError):
        pass
    return n


def main():
    # The GPU box shows 256 CPUs and grants 16 (CFS quota): left alone, PyTorch sizes its intra-op pool to 128 threads, the host-side set-up of this
    # script (parameter init, synthetic batch) burns the quota of several periods at once, and the kernel then freezes the WHOLE process for 40-70 ms
    # some periods later - measured inside the timed region, one step() call of 1 ms taking 45 ms with the device idle (profiles/r04_notes.md, section 6).
    # A rank uses the share it has (its part of it when several ranks run on the node).
    share = max(1, cpu_share() // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))))
    torch.set_num_threads(max(1, min(torch.get_num_threads(), share)))
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--workload", default="ds2", choices=list(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-op-rates", action="store_true")
    ap.add_argument("--no-sampling", action="store_true")
    ap.add_argument("--no-box", action="store_true", help="skip the box calibration (MFMA-only loop, copy)")
    ap.add_argument("--no-other", action="store_true", help="skip the ds3 and drop-in-_step rates")
    ap.add_argument("--lean", action="store_true", help="the timed region only (A/B runs): no op rates, sampling, other rates, CPU baseline; box calibration stays")
    args = ap.parse_args()
    if args.lean:
        args.no_cpu_baseline = args.no_op_rates = args.no_sampling = args.no_other = True
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))  # before anything touches the GPU in this process

    # V4H_BENCH_WATCHDOG=<seconds>: dump every thread's Python stack and exit if the run is still going by then (a hung collective or kernel then leaves a
    # traceback instead of a silent timeout of the caller)
    if os.environ.get("V4H_BENCH_WATCHDOG"):
        import faulthandler

        faulthandler.dump_traceback_later(int(os.environ["V4H_BENCH_WATCHDOG"]), exit=True)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} but --gpus {args.gpus}")
    assert torch.cuda.is_available(), "bench.py needs MI355X devices"
    if REHEARSAL:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"
    import torch.distributed as dist

    if world > 1 or os.environ.get("V4H_FORCE_COLLECTIVES") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if REHEARSAL:
            dist.init_process_group("gloo", init_method="env://")
        else:
            dist.init_process_group("nccl", init_method="env://", device_id=torch.device(device))

    from vit4hep_amd.trainer import CFMTrainer

    w = WORKLOADS[args.workload]
    shape, depth, B, desc = w["shape"], w["depth"], w["B"], w["desc"]
    model = build_model(w, args.mode, device)
    if dist.is_initialized():  # same initial weights everywhere, like DDP's constructor broadcast (base_experiment.py:163)
        for p in model.parameters():
            dist.broadcast(p.data, 0)
    # V4H_PIPELINE_UPDATE=1: AdamW + operand casts on the library's side stream beside the next step's head (same arithmetic).  Measured neutral
    # (4.192 vs 4.198 ms per step: the update's 0.73 GB cross the same HBM either way) and therefore off.
    pipelined = os.environ.get("V4H_PIPELINE_UPDATE", "0") == "1"
    trainer = CFMTrainer(model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.1, clip_grad_norm=1000.0, iterations=50000, pipeline_update=pipelined)
    x, c = synthetic(shape, B, seed=rank, device=device, cond=w["cond"])
    torch.manual_seed(1000 + rank)

    def sync():
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        loss, gn = trainer.step(x, c)
    sync()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    cg0 = cgroup_cpu_stat() if os.environ.get("V4H_BENCH_STEP_EVENTS") == "1" else None
    t0 = time.perf_counter()
    e0.record()
    step_marks = [] if os.environ.get("V4H_BENCH_STEP_EVENTS") == "1" else None  # diagnostic: one event per step (a barrier packet each, ~0.1 %)
    host_marks = []
    host_only = os.environ.get("V4H_BENCH_HOST_TIMES") == "1"  # diagnostic: print the host time of every step() call
    # Always: the host time of every step() call (one perf_counter each, nothing on the stream) -> `host_max_step_call_ms` exposes a stalled call (a 35-80 ms
    # stall of the runtime inside one launch was seen in round 4); and one timing event per step (a marker packet each, < 0.1 % of a step) ->
    # `device_ms_per_step_p50` beside the mean: a run whose mean is pulled up by one slow step shows it.
    call_ms, dev_marks = [], []
    for _ in range(args.steps):
        tc = time.perf_counter()
        loss, gn = trainer.step(x, c)
        call_ms.append((time.perf_counter() - tc) * 1e3)
        dev_marks.append(torch.cuda.Event(enable_timing=True))
        dev_marks[-1].record()
        if step_marks is not None:
            step_marks.append(dev_marks[-1])
        if step_marks is not None or host_only:
            host_marks.append(time.perf_counter())
    trainer.finish()  # (a pipelined update of the last step is ordered into the timed stream: both clocks below include it)
    e1.record()
    sync()
    wall = time.perf_counter() - t0
    dev_ms = e0.elapsed_time(e1)
    if step_marks:
        ts = [e0.elapsed_time(m) for m in step_marks]
        per = [round(b - a, 3) for a, b in zip([0.0] + ts[:-1], ts)]
        print(f"bench.py: device ms per step (rank {rank}): {per[:64]}", file=sys.stderr)
        print(f"bench.py: device steps over 6 ms (index, ms): {[(i, v) for i, v in enumerate(per[:-1]) if v > 6.0]}", file=sys.stderr)
        hp = [round((b - a) * 1e3, 3) for a, b in zip([t0] + host_marks[:-1], host_marks)]
        print(f"bench.py: host ms per step() call (rank {rank}): {hp[:64]}", file=sys.stderr)
        print(f"bench.py: host calls over 10 ms (index, ms since the start of the timed region, ms): {[(i, round((host_marks[i] - t0) * 1e3, 1), v) for i, v in enumerate(hp) if v > 10.0]}",
              file=sys.stderr)
        cg1 = cgroup_cpu_stat()
        print(f"bench.py: container cpu.stat over the timed region: {({k: cg1[k] - cg0[k] for k in cg0 if cg1.get(k) != cg0[k]})}; torch threads {torch.get_num_threads()}",
              file=sys.stderr)
    if host_only:
        hp = [round((b - a) * 1e3, 3) for a, b in zip([t0] + host_marks[:-1], host_marks)]
        print(f"bench.py: host calls over 10 ms (index, ms since the start of the timed region, ms): {[(i, round((host_marks[i] - t0) * 1e3, 1), v) for i, v in enumerate(hp) if v > 10.0]}",
              file=sys.stderr)
    CFMTrainer.check_finite(gn)
    tmax = torch.tensor([wall], dtype=torch.float64, device=device)
    rank_walls = None
    if dist.is_initialized():
        gathered = [torch.zeros_like(tmax) for _ in range(world)]
        dist.all_gather(gathered, tmax)
        rank_walls = [float(g.item()) for g in gathered]
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    wall = float(tmax.item())

    if rank == 0:
        T, P = tokens_and_patch_dim(w)
        flop_step = 3.0 * B * fwd_flops_per_sample(T, P, depth, K=w["cond"])  # per GPU
        ms_step = wall * 1e3 / args.steps
        dev_ms_step = dev_ms / args.steps
        achieved = flop_step / (dev_ms_step * 1e-3) / 1e12
        peak = BF16_DENSE_PEAK_TFLOPS if args.mode == "bf16" else F32_MFMA_PEAK_TFLOPS
        traffic, traffic_note = measured_traffic(args.workload, args.mode)
        rec = {
            "metric": "CFM train steps/sec (ds2 shape ViT, bs=128)" if args.workload == "ds2" else f"CFM train steps/sec ({args.workload})",
            "value": round(world * args.steps / wall, 3),
            "unit": "steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.mode,
            "data": "synthetic",
            "config": {"workload": desc, "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world}",
                       "update": "pipelined (AdamW + operand casts beside the next step's head)" if trainer.pipeline_update and world == 1 else "in line"},
            "details": {"tokens": T, "patch_dim": P, "depth": depth, "params": sum(p.numel() for p in model.parameters()),
                        "init": "random (xavier; zero-init tensors perturbed N(0,0.02))"},
            "loss": round(float(loss), 5),
            "grad_norm": round(float(gn), 5),
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic,
                         "kernel": "whole update step (all launches of one step)", "flop_per_launch": flop_step, "launch_ms": round(dev_ms_step, 4)},
            "mfma_util_pct": round(100.0 * achieved / peak, 2),
        }
        ts = [e0.elapsed_time(m) for m in dev_marks]
        per_step = sorted(b - a for a, b in zip([0.0] + ts[:-1], ts))
        rec["device_ms_per_step_mean"] = round(dev_ms_step, 4)
        rec["device_ms_per_step_p50"] = round(per_step[len(per_step) // 2], 4)
        rec["device_ms_per_step_max"] = round(per_step[-1], 4)
        rec["host_max_step_call_ms"] = round(max(call_ms), 3)
        rec["host_step_call_ms_p50"] = round(sorted(call_ms)[len(call_ms) // 2], 3)
        if not args.lean and world == 1:  # (profiling runs count the steps of the process: no extra ones; with several ranks a step is a collective)
            # host time to ENQUEUE one step on an idle device (the queue empty, nothing to wait for): below ms_per_step = the device sets the pace
            sync()
            th = time.perf_counter()
            for _ in range(3):
                trainer.step(x, c)
            rec["host_enqueue_ms_per_step"] = round((time.perf_counter() - th) * 1e3 / 3, 3)
            sync()
        rec["roofline"]["traffic_source"] = ("profiles/step_hbm_traffic.json (builder-measured PMC passes over this program, digest-matched to the kernel "
                                             "sources of this build; not measured in this run)")
        if traffic_note:
            rec["roofline"]["traffic_note"] = traffic_note
        if dist.is_initialized():  # self-describing N > 1 lines: what transport the ranks really used, and how far apart they finished
            rec["rccl_ranks"] = dist.get_world_size() if dist.get_backend() == "nccl" else 0
            rec["backend"] = dist.get_backend()
            rec["rank_ms_per_step"] = {"min": round(min(rank_walls) * 1e3 / args.steps, 4), "max": round(max(rank_walls) * 1e3 / args.steps, 4),
                                       "per_rank": [round(v * 1e3 / args.steps, 4) for v in rank_walls]}
        if REHEARSAL:
            rec["rehearsal"] = True  # N ranks on one GPU over gloo: a functional check of the multi-rank path, not a measurement
        if world == 1 and not args.no_box:
            rec["box"] = box_calibration(device)
        if world == 1 and not args.no_op_rates:
            rec["gemm_ops"] = op_rates(args.mode, B * T, device)
            # the step's dominant kernel (18.9 % of its kernel time: the split-K weight gradients of qkv / fc1 / fc2 on the ring kernel), timed in THIS run
            dom = [v for k, v in rec["gemm_ops"].items() if k.startswith("wgrad") and "proj" not in k]
            if dom:
                us = sum(v["us"] for v in dom) / len(dom)
                tfl = sum(v["tflops"] * v["us"] for v in dom) / sum(v["us"] for v in dom)
                rec["roofline"]["dominant"] = {"kernel": "gemm2<bf16,wgrad,256x160x64,SLAB_F32,colsum,ping-pong> + slab_reduce (dW = dY^T X of attn.qkv, mlp.fc1, mlp.fc2; cold operands)",
                                               "us_per_call": round(us, 2), "tflops": round(tfl, 1), "frac": round(tfl / peak, 4), "calls_per_step": 3 * depth}
        if world == 1 and not args.no_sampling and args.workload == "ds2":
            rec["sampling"] = sampling_rates(model, w, device, peak)
        if world == 1 and not args.no_other and args.workload == "ds2":
            del trainer
            torch.cuda.empty_cache()
            rec["other"] = other_rates(args.mode, device)
        if world == 1 and not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline(w)
            rec["speedup_vs_cpu"] = round(rec["value"] / rec["cpu_baseline"]["value"], 1)
        print(json.dumps(rec), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
