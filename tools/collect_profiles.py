"""Copy the summaries of a tools/refresh_profiles.sh run (gpurun_out/refresh_<tag>/) into profiles/ under round-stable names and rebuild the all-workload table.
usage: python tools/collect_profiles.py <tag> [round prefix, default r03]"""
import json, os, shutil, sys

tag = sys.argv[1]
rp = sys.argv[2] if len(sys.argv) > 2 else "r05"
R = f"gpurun_out/refresh_{tag}"
for src, dst in (("step_serialized.md", f"{rp}_step_final_bf16_serialized.md"), ("step_overlapped.md", f"{rp}_step_final_bf16_overlapped.md"), ("step_gaps.txt", f"{rp}_step_final_gaps.txt"), ("step_timeline.txt", f"{rp}_step_timeline.txt"),
                 ("step_pmc_counters.md", f"{rp}_step_pmc_counters.md"), ("step_hbm_traffic.json", "step_hbm_traffic.json"), ("bench_final_bf16.json", f"{rp}_bench_final_bf16.json"),
                 ("ab_in_context.txt", f"{rp}_ab_in_context_{tag}.txt"), ("ds3_step_serialized.md", f"{rp}_ds3_step_serialized.md"),
                 ("block_gemm_bench.txt", f"{rp}_block_gemm_bench.txt"), ("attn_bench.txt", f"{rp}_attn_bench.txt"), ("ddp_route.txt", f"{rp}_ddp_route_rehearsal.txt"),
                 ("comm_interference.txt", f"{rp}_comm_interference_raw.txt")):
    if os.path.exists(f"{R}/{src}"):
        shutil.copy(f"{R}/{src}", f"profiles/{dst}")
if os.path.exists(f"{R}/step_pmc_counters.md") and os.path.exists("tools/pmc_counters_header.md"):  # the audit's reading in front of the raw tables
    open(f"profiles/{rp}_step_pmc_counters.md", "w").write(open("tools/pmc_counters_header.md").read() + "```\n" + open(f"{R}/step_pmc_counters.md").read() + "```\n")
if os.path.exists(f"{R}/ab_in_context.txt"):
    open(f"profiles/{rp}_ab_in_context_{tag}.txt", "w").write(
        "# In-context A/B of the switches the library keeps (tools/refresh_profiles.sh, part 3): two interleaved rounds on one box, each line one full `bench.py` run\n"
        "# (40 timed steps; sampling at batch 256).  V4H_GEMM2=-1 is the default build.  Round-5 levers switched off one at a time: VIT4HEP_AMD_RESIDUAL=f32 / x_bf16 /\n"
        "# dx_bf16 (storage of the residual stream and of its gradient; default bf16 for both), V4H_GEMM3=0 / 3 / 39 / 63 (no class / qkv + proj forward /\n"
        "# + both GELU forwards / every K = 480 class on the weight-stationary kernel; default 55 = all but the plain dgrad), V4H_STOP_EVENTS=0 (recorded events instead of completion signals for the backward's forks); then the older ones: gradient buffer zero-filled\n"
        "# and accumulated into, 128 x 160 tiles where the GELU / DGELU classes run on the two-workgroup kernel,\n"
        "# round-2 LayerNorm backward, pipelined update, 0 / 8 = two-workgroup / ring kernel everywhere, whole-K kernels off, round-2 attention, 4 K-splits, per-block adaLN,\n"
        "# no weight-gradient stream.\n"
        + open(f"{R}/ab_in_context.txt").read())
if os.path.exists(f"{R}/gemm2_ablation.txt"):
    old = open(f"profiles/{rp}_gemm2_ablation.txt").read() if os.path.exists(f"profiles/{rp}_gemm2_ablation.txt") else ""
    hdr = old.split("\n\n")[0] + "\n\n" if old.startswith("#") else ""  # (keep a hand-written header, never an old body)
    body = "\n".join(l for l in open(f"{R}/gemm2_ablation.txt").read().splitlines() if "amdgpu.ids" not in l)
    open(f"profiles/{rp}_gemm2_ablation.txt", "w").write(hdr + body + "\n")
names = {"ds2": "bench_final_bf16.json", "ds3": "bench_ds3.json", "ds2_d2": "bench_ds2_d2.json", "lemurs": "bench_lemurs.json", "ds1_photons": "bench_ds1_photons.json",
         "ds1_pions": "bench_ds1_pions.json", "calogan": "bench_calogan.json", "calohad": "bench_calohad.json", "ds2 f32 mode": "bench_ds2_f32.json",
         "ds2, collectives forced on (1 rank)": "bench_ds2_forced_collectives.json"}
out = [f"# bench.py on one MI355X, final build of the round (tools/refresh_profiles.sh {tag}; bf16 mode unless noted)", "",
       "| workload | per-GPU batch | steps/s | ms/step | TFLOP/s | fraction of the dense MFMA spec peak |", "|---|---|---|---|---|---|"]
for k, f in names.items():
    if not os.path.exists(f"{R}/{f}"):
        continue
    r = json.loads(open(f"{R}/{f}").read().strip().splitlines()[-1])
    out.append(f"| {k} ({r['config']['workload']}) | {r['config']['per_gpu_batch']} | {r['value']} | {r['ms_per_step']} | {r['roofline']['achieved']} | {r['roofline']['frac']} |")
r = json.loads(open(f"{R}/bench_final_bf16.json").read().strip().splitlines()[-1])
s = r["sampling"]
out += ["", f"Sampling (BASELINE config 5; ds2, batch 256, bf16): RK4 step 0.05 (80 evaluations) {s['rk4']['showers_per_s']} showers/s = 100 k in {s['rk4']['s_per_100k']} s, "
            f"{s['rk4']['tflops']} TFLOP/s; Heun (40 evaluations) {s['heun2']['showers_per_s']} showers/s = 100 k in {s['heun2']['s_per_100k']} s.",
        f"CPU oracle on the same box ({r['cpu_baseline']['cpu']}): {r['cpu_baseline']['value']} steps/s on {r['cpu_baseline']['cores']} threads; "
        f"{r['cpu_baseline']['one_thread']['value']} steps/s on one thread ({r['cpu_baseline']['one_thread']['sample']}).",
        f"HBM traffic of one update step (PMC, same build): {r['roofline']['traffic']/1e9:.2f} GB.",
        "Contraction rates inside this run (`gemm_ops`): " + "; ".join(f"{k} {v['us']} us = {v['tflops']} TFLOP/s" for k, v in r["gemm_ops"].items() if isinstance(v, dict)) + "." + (" (" + r["gemm_ops"]["_note"] + ")" if "_note" in r["gemm_ops"] else "")]
open(f"profiles/{rp}_bench_all_workloads.md", "w").write("\n".join(out) + "\n")
print("\n".join(out))
