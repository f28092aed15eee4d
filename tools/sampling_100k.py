"""BASELINE config 5 as a real run: 100 000 ds2 showers through CaloChallengeCFM.sample_batch (reference calochallenge_cfm/model.py:68-94) in the reference's
sampling batches of 256 (configs/training/default.yaml:3; 391 batches, the last of 160), fixed-grid step 0.05, with the reference's default solver (torchdiffeq
'rk4' = 3/8 rule, 80 network evaluations) and with Heun (40) - wall time of the whole loop incl. host-side noise and condition hand-over - plus the parity of
the first batch: the same x_T and conditions through the CPU oracle (rows 0-1; the oracle needs about 0.3 s per evaluation of two rows).
usage (GPU box): python tools/sampling_100k.py [n_showers] > profiles/r05_sampling_100k.md"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import WORKLOADS, build_model, fwd_flops_per_sample, tokens_and_patch_dim, BF16_DENSE_PEAK_TFLOPS
from oracle import vit_cfm_oracle as O  # checker of the first batch only

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
dev = "cuda:0"
w = WORKLOADS["ds2"]
cfg = O.ds2(6)
model = build_model(w, "bf16", dev).eval()
fill = {k[4:]: v.detach().float().cpu() for k, v in model.state_dict().items() if k.startswith("net.") and k[4:] not in ("pos_x", "pos_y", "pos_z")}
T, P = tokens_and_patch_dim(w)
flops = fwd_flops_per_sample(T, P, w["depth"], K=w["cond"])
g = torch.Generator().manual_seed(5)
cond = torch.cat([torch.randn((N, 45), generator=g), torch.rand((N, 1), generator=g)], 1)  # 45 standardised-logit u-ratios, scaled log-energy last
cond_dev = cond.to(dev)  # conditions resident in HBM (the energy model's output in a real run)
print(f"# ds2 shape sampling, {N} showers, batches of 256, bf16 network evaluations (MI355X, one GPU)\n")
print("| solver | NFE | batches | wall s | showers/s | TFLOP/s | frac of 2.5 PFLOP/s | first batch vs oracle (rows 0-1): max abs / rel to max |")
print("|---|---|---|---|---|---|---|---|")
for method, nfe in (("rk4", 80), ("heun2", 40)):
    model.odeint_kwargs = {"method": method, "options": {"step_size": 0.05}}
    with torch.no_grad():
        model.sample_batch(cond_dev[:256])  # warm-up (workspace, operand copies)
        torch.manual_seed(11)
        xT0 = torch.randn((256, 1, 45, 16, 9), device=dev)
        first = model._sample_from(xT0.clone(), cond_dev[:256])
        ref = O.sample(fill, cond[:2], xT0[:2].cpu(), cfg, method=method, step_size=0.05)
        err = (first[:2].float().cpu() - ref).abs().max().item()
        rel = err / ref.abs().max().item()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out, nb = [], 0
        for lo in range(0, N, 256):
            out.append(model.sample_batch(cond_dev[lo : lo + 256]))
            nb += 1
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    assert sum(o.shape[0] for o in out) == N and all(bool(torch.isfinite(o).all()) for o in out[:: max(1, nb // 8)])
    tf = nfe * N * flops / dt / 1e12
    print(f"| {method} | {nfe} | {nb} | {dt:.2f} | {N / dt:.0f} | {tf:.0f} | {tf / BF16_DENSE_PEAK_TFLOPS:.3f} | {err:.2e} / {rel:.2e} |", flush=True)
    del out
print("\nParity bar of the bf16 sampler against reference-generated samples: 3e-2 of the sample's scale (tests/test_hip_round2.py); the f32 mode is held to 1e-4.")
