"""Energy-model sampling throughput: samples/s of the reference's default solver (RK4, step 0.05 = 80 network
evaluations) through v4h_energy_forward, next to (a) the same network as PyTorch-ROCm eager ops on the same GPU (the oracle's
functional restatement run on device tensors = what the reference's nn.Transformer module does there) and (b) the CPU oracle.
usage (GPU box): python tools/energy_bench.py [bf16|f32] [B ...]        prints one JSON line per batch size"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import energy_oracle as E
from vit4hep_amd import CFM
from vit4hep_amd.nn.cfm.transformer_cfm import ParallelTransformer

mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
batches = [int(b) for b in sys.argv[2:]] or [256, 2048]
dev = "cuda:0"
cfg = E.EnergyConfig()
net = ParallelTransformer({"dims_in": 45, "dims_c": 1, "dim_embedding": 64, "nhead": 4, "num_encoder_layers": 4, "num_decoder_layers": 4,
                           "dim_feedforward": 512, "embeds": True, "encode_t_dim": 64, "amd_mode": mode})
model = CFM(net, "uniform", "linear", {"method": "rk4", "options": {"step_size": 0.05}}, shape=[45]).to(dev).eval()
model.device, model.dtype = torch.device(dev), torch.float32
params_dev = {k: v.detach() for k, v in net.named_parameters()}
flops = E.fwd_flops_per_sample(cfg)
NFE = 80


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, out


for B in batches:
    g = torch.Generator().manual_seed(B)
    c = torch.rand((B, 1), generator=g).to(dev)
    x_T = torch.randn((B, 45), generator=g).to(dev)
    with torch.inference_mode():
        dt, s = timed(lambda: model._sample_from(x_T, c), 5)
        # one network evaluation, HIP events
        x, t = torch.randn((B, 45), device=dev), torch.rand((B, 1), device=dev)
        for _ in range(3): net(x, t, c)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): net(x, t, c)
        e1.record(); torch.cuda.synchronize()
        eval_ms = e0.elapsed_time(e1) / 50
        def torch_eager():
            from oracle.vit_cfm_oracle import fixed_grid, ode_step
            grid = fixed_grid(0.0, 1.0, 0.05)
            y = x_T
            for k in range(len(grid) - 1):
                y = ode_step(lambda tt, yy: E.energy_forward(params_dev, yy, torch.full((B, 1), float(tt), device=dev), c, cfg), "rk4", float(grid[k]), float(grid[k + 1]), y)
            return y
        dt_torch, s_ref = timed(torch_eager, 1)
    err = float((s - s_ref).abs().max() / s_ref.abs().max())
    rec = {"metric": "energy-model samples/s (RK4, 80 NFE)", "value": round(B / dt, 1), "unit": "samples/s", "dtype": mode, "batch": B,
           "ms_per_batch": round(dt * 1e3, 2), "ms_per_eval": round(eval_ms, 4), "tflops": round(NFE * B * flops / dt / 1e12, 2),
           "algorithmic_flop_per_sample_eval": flops, "torch_rocm_eager_same_gpu": {"samples_per_s": round(B / dt_torch, 1), "ms_per_batch": round(dt_torch * 1e3, 1)},
           "speedup_vs_torch_eager": round(dt_torch / dt, 1), "max_rel_diff_vs_torch": err, "100k_samples_s": round(1e5 / (B / dt), 2)}
    if B == batches[0]:
        torch.set_num_threads(min(len(os.sched_getaffinity(0)), 16))
        pc = {k: v.cpu() for k, v in params_dev.items()}
        cc, xc = c.cpu()[:64], x_T.cpu()[:64]
        t0 = time.perf_counter()
        E.energy_sample(pc, cc, xc, cfg, "rk4", 0.25)  # 16 NFE on 64 samples
        el = time.perf_counter() - t0
        rec["cpu_baseline"] = {"value": round(64 * 16 / NFE / el, 1), "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"64 samples x 16 evaluations with the PyTorch-CPU oracle ({el:.1f} s), scaled to 80 evaluations"}
    print(json.dumps(rec), flush=True)
