"""Summarise the rocprofv3 --pmc passes of tools/pmc_step.sh (bench.py itself under the profiler).

  * FETCH_SIZE / WRITE_SIZE (KiB units) summed over every kernel of the run, per update step -> profiles/step_hbm_traffic.json, stamped with
    the digest of the kernel sources (bench.py refuses a figure measured on other sources).  FETCH_SIZE is doubled: on gfx950 it reports half
    the bytes of 16-byte-per-lane streaming reads (MI355X_MICROARCH.md, HBM).
  * per kernel: MFMA-busy fraction of SQ busy cycles (a counter, not FLOP / time / peak), bf16 MFMA ops, LDS bank-conflict share, L2 hit rate,
    device clock (GRBM_GUI_ACTIVE / duration) -> stdout (kept as profiles/<tag>_step_pmc_counters.md).
usage: python tools/pmc_step_summary.py <steps incl. warmup> [tag]"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
steps = int(sys.argv[1])
tag = sys.argv[2] if len(sys.argv) > 2 else "r04"
BASE = "gpurun_out/pmc_step"


def _short_gemm(n):
    import re
    m = re.match(r"_Z15v4h_gemm_kernelI7GemmCfgI(DF16b|f)(DF16b|f)Lb(\d)ELb(\d)ELi(\d+)ELi(\d+)ELi(\d+)ELi\d+ELi\d+ELi(\d+)ELb(\d)", n)
    if m:
        t, _, pks, qks, bi, bj, bk, epi, cs = m.groups()
        epis = ["STORE", "STORE_F32", "SILU", "COND_SUM", "EMBED", "GATE_RESID", "GELU", "DGELU", "DSILU", "ATOMIC_F32", "ACCUM_F32", "UNPATCH", "SLAB_F32", "RELU", "ROWADD_SILU"]
        lay = {("0", "0"): "fwd", ("0", "1"): "dgrad", ("1", "1"): "wgrad"}[(pks, qks)]
        return f"gemm<{'bf16' if t == 'DF16b' else 'f32'},{lay},{bi}x{bj}x{bk},{epis[int(epi)]}{',colsum' if cs == '1' else ''}>"
    m = re.match(r"(?:void )?v4h_gemm2_kernel<Gemm2Cfg<(true|false), (true|false), (\d+), (true|false), (\d+)(?:, (true|false))? ?> ?>", n)
    if m:  # the 256 x 160 ring kernel (v4h_gemm2.h)
        pks, qks, epi, cs, dbg, pp = m.groups()
        epis = ["STORE", "STORE_F32", "SILU", "COND_SUM", "EMBED", "GATE_RESID", "GELU", "DGELU", "DSILU", "ATOMIC_F32", "ACCUM_F32", "UNPATCH", "SLAB_F32", "RELU", "ROWADD_SILU"]
        lay = {("false", "false"): "fwd", ("false", "true"): "dgrad", ("true", "true"): "wgrad"}[(pks, qks)]
        return f"gemm2<bf16,{lay},256x160x64,{epis[int(epi)]}{',colsum' if cs == 'true' else ''},{'ping-pong' if pp == 'true' else 'lock-step'}>"
    m = re.match(r"_ZN(?:3v4h)?12_GLOBAL__N_1(\d+)([A-Za-z_0-9]+)", n)
    if m:
        return m.group(2)[:int(m.group(1))]
    return None


def short(k):
    g = _short_gemm(k)
    if g:
        return g
    k = k.replace("void ", "")
    for a, b in (("v4h_gemm_kernel<GemmCfg<", "gemm<"), ("(anonymous namespace)::", ""), ("v4h::", "")):
        k = k.replace(a, b)
    return k[:72]


def load(name):
    f = glob.glob(f"{BASE}/{name}/**/*counter_collection.csv", recursive=True)
    per = collections.defaultdict(lambda: collections.Counter())
    dur = collections.Counter()
    calls = collections.Counter()
    if not f:
        return per, dur, calls
    seen = set()
    for r in csv.DictReader(open(max(f, key=os.path.getmtime))):
        k = short(r["Kernel_Name"])
        per[r["Counter_Name"]][k] += float(r["Counter_Value"])
        key = (r.get("Dispatch_Id"), r["Counter_Name"])
        if r.get("Start_Timestamp") and (r.get("Dispatch_Id"), "t") not in seen:
            seen.add((r.get("Dispatch_Id"), "t"))
            dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
            calls[k] += 1
    return per, dur, calls


fetch, _, fcalls = load("FETCH_SIZE")
write, _, _ = load("WRITE_SIZE")
rq, _, _ = load("TCC_EA0_RDREQ_sum+TCC_EA0_RDREQ_32B_sum+TCC_EA0_RDREQ_64B_sum+TCC_EA0_RDREQ_128B_sum")


def exact_fetch(k):
    """bytes fetched by kernel k from the read requests by size class (32 / 64 / 128 bytes): what FETCH_SIZE would be if it priced each request at its size"""
    n, n32, n64, n128 = (rq[c].get(k, 0.0) for c in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"))
    if n == 0:
        return None
    other = max(n - n32 - n64 - n128, 0.0)  # requests in no size class (none seen so far): priced at 64 bytes like FETCH_SIZE does
    return 32 * n32 + 64 * n64 + 128 * n128 + 64 * other


# Algorithmic HBM bytes per call of the ds2 bs = 128 update step (BT = 17280 tokens, D = 480, M = 1920, bf16 activations, bf16 residual streams - round 5): every
# operand read once, every result written once, with the activation-materialising structure of the path (each Linear writes its output).
BT, D, M = 17280, 480, 1920
ALGO = {  # short kernel name -> (read MB per call, written MB per call, what)
    "ln_modulate_fwd8_kernel": ((BT * D * 4) / 1e6, (BT * D * 4) / 1e6, "x bf16 + y bf16 -> x' bf16 + u bf16"),
    "ln_modulate_bwd8_kernel": ((BT * D * 12) / 1e6, (BT * D * 6) / 1e6, "du bf16 + x f32 + dx f32 + y bf16 -> dx f32 + dy bf16"),
    "ln_modulate_bwd8v2_kernel": ((BT * D * 8) / 1e6, (BT * D * 4) / 1e6, "du bf16 + x bf16 + dx bf16 + y bf16 -> dx bf16 + dy bf16"),
    "v4h_gemm3_kernel<Gemm3Cfg<false, 0, 3": ((BT * D * 2 + 3 * D * D * 2) / 1e6, (BT * 3 * D * 2) / 1e6, "qkv forward, weight-stationary: u1 + W -> qkv"),
    "v4h_gemm3_kernel<Gemm3Cfg<false, 0, 2": ((BT * D * 2 + D * D * 2) / 1e6, (BT * D * 2) / 1e6, "attn.proj forward, weight-stationary: o + W -> y1"),
    "v4h_gemm3_kernel<Gemm3Cfg<false, 6, 2": ((BT * D * 2 + M * D * 2) / 1e6, (2 * BT * M * 2) / 1e6, "fc1 + GELU forward, weight-stationary: u2 + W -> h + gelu'"),
    "v4h_gemm3_kernel<Gemm3Cfg<true, 7, 2": ((BT * D * 2 + M * D * 2 + BT * M * 2) / 1e6, (BT * M * 2) / 1e6, "fc2 dgrad x gelu', weight-stationary: dy + W + gelu' -> dh"),
    "adamw_sched_kernel": (26042528 * 16 / 1e6, 26042528 * 12 / 1e6, "p, g, m, v -> p, m, v"),
    "gemm<bf16,fwd,128x128x64,GELU>": ((BT * D * 2 + M * D * 2) / 1e6, (2 * BT * M * 2) / 1e6, "u2 + W -> h + gelu' (128 x 128 tiles)"),
    "gemm<bf16,dgrad,128x128x64,DGELU>": ((BT * D * 2 + M * D * 2 + BT * M * 2) / 1e6, (BT * M * 2) / 1e6, "dy + W + gelu' -> dh (128 x 128 tiles)"),
    "attn_fwd_dense_kernel": ((BT * 3 * D * 2) / 1e6, (BT * D * 2) / 1e6, "qkv -> o"),
    "attn_bwd_fused_kernel": ((BT * 5 * D * 2) / 1e6, (BT * 3 * D * 2) / 1e6, "qkv + o + dO -> dqkv"),
    "adamw_kernel": (26042528 * 16 / 1e6, 26042528 * 12 / 1e6, "p, g, m, v -> p, m, v"),
    "gemm<bf16,fwd,128x160x64,GELU>": ((BT * D * 2 + M * D * 2) / 1e6, (2 * BT * M * 2) / 1e6, "u2 + W -> h + gelu'"),
    "gemm<bf16,dgrad,128x160x64,DGELU>": ((BT * D * 2 + M * D * 2 + BT * M * 2) / 1e6, (BT * M * 2) / 1e6, "dy + W + gelu' -> dh"),
    "gemm2<bf16,fwd,256x160x64,STORE,ping-pong>": ((BT * M * 2 + D * M * 2) / 1e6, (BT * D * 2) / 1e6, "fc2 forward: h + W -> y2"),
    "gemm2<bf16,dgrad,256x160x64,STORE,ping-pong>": ((BT * (3 * D + D + M) * 2 + (3 * D * D + D * D + D * M) * 2) / 3e6, (BT * 3 * D * 2) / 3e6, "mean of d qkv, d proj, d fc1"),
    "gemm2<bf16,wgrad,256x160x64,SLAB_F32,colsum,ping-pong>": ((BT * ((3 * D + D) + (M + D) + (D + M)) * 2) / 3e6, (8 * (3 * D * D + 2 * D * M) * 4) / 3e6, "mean of qkv, fc1, fc2: dY + X -> 8 f32 slabs"),
}

tot_f = sum(fetch["FETCH_SIZE"].values()) * 1024 / steps
tot_w = sum(write["WRITE_SIZE"].values()) * 1024 / steps
tot_x = sum(v for v in (exact_fetch(k) for k in fetch["FETCH_SIZE"]) if v) / steps
print(f"HBM traffic per update step: FETCH_SIZE raw {tot_f/1e9:.3f} GB (x2 = {2*tot_f/1e9:.3f} GB: FETCH_SIZE prices every read request at 64 bytes, the requests are 128 bytes - "
      f"tools/experiments/fetch_calib.hip; priced by size class: {tot_x/1e9:.3f} GB), WRITE_SIZE {tot_w/1e9:.3f} GB  ->  {(2*tot_f+tot_w)/1e9:.3f} GB/step")
print("\naudit: algorithmic bytes per call next to the counters (fetched = 2 x FETCH_SIZE, [by request size class]; written = WRITE_SIZE)")
print(f"{'kernel':60s} {'calls':>5s} {'algo read':>10s} {'fetched':>9s} {'[sized]':>9s} {'ratio':>6s} {'algo write':>10s} {'written':>9s} {'ratio':>6s}  what")
per_kernel = {}
for k in sorted(fetch["FETCH_SIZE"], key=lambda k: -(2 * fetch["FETCH_SIZE"][k] + write["WRITE_SIZE"].get(k, 0.0))):
    f_mb = 2 * fetch["FETCH_SIZE"][k] * 1024 / 1e6 / steps
    w_mb = write["WRITE_SIZE"].get(k, 0.0) * 1024 / 1e6 / steps
    per_kernel[k] = {"fetched_MB_per_step": round(f_mb, 1), "written_MB_per_step": round(w_mb, 1)}
    a = next((v for kk, v in ALGO.items() if kk in k), None)
    ncalls = int(round(fcalls[k] / steps))
    if a is None or f_mb + w_mb < 50 or ncalls == 0:
        continue
    xf = exact_fetch(k)
    per_kernel[k].update({"calls_per_step": ncalls, "algorithmic_read_MB_per_call": round(a[0], 1), "algorithmic_written_MB_per_call": round(a[1], 1)})
    print(f"{k[:60]:60s} {ncalls:5d} {a[0]:10.1f} {f_mb/ncalls:9.1f} {(xf/1e6/steps/ncalls if xf else float('nan')):9.1f} {f_mb/ncalls/a[0]:6.2f} {a[1]:10.1f} {w_mb/ncalls:9.1f} {w_mb/ncalls/a[1]:6.2f}  {a[2]}")
if tot_f > 0 and tot_w > 0:
    from vit4hep_amd.build import kernel_digest as _digest

    path = "profiles/step_hbm_traffic.json"
    rec = json.load(open(path)) if os.path.exists(path) else {}
    rec["ds2/bf16"] = {"bytes_per_step": 2 * tot_f + tot_w, "fetch_size_raw": tot_f, "fetch_by_request_size": tot_x, "write_size": tot_w, "kernel_digest": _digest(),
                       "steps_profiled": steps, "per_kernel": per_kernel,
                       "how": "tools/pmc_step.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE / TCC_EA0_RDREQ by size class (separate passes) -- python3 "
                              "bench.py; FETCH_SIZE doubled: it prices each (128-byte) read request at 64 bytes - calibrated on known byte counts for every access "
                              "shape of the path with tools/experiments/fetch_calib.hip (profiles/r03_fetch_size_calibration.md)"}
    json.dump(rec, open(path, "w"), indent=1)
    print(f"wrote {path}")

mf, dur, calls = load("SQ_VALU_MFMA_BUSY_CYCLES+SQ_BUSY_CYCLES")
ops, _, _ = load("SQ_INSTS_VALU_MFMA_MOPS_BF16+SQ_WAVE_CYCLES")
lds, _, _ = load("SQ_LDS_BANK_CONFLICT+SQ_LDS_IDX_ACTIVE")
l2, _, _ = load("TCC_HIT_sum+TCC_MISS_sum")
clk, cdur, _ = load("GRBM_GUI_ACTIVE")
print(f"\nper kernel, {steps} steps (profiler build of the same run; durations under counter collection are longer than in a plain run)")
NSIMD = 1024  # 256 CUs x 4
print("MFMA busy % = SQ_VALU_MFMA_BUSY_CYCLES (summed over the 1024 SIMDs) / (1024 x kernel duration x 2.4 GHz): the share of the peak-clock cycle budget in which "
      "a SIMD's matrix pipe was busy - a counter, comparable with FLOP / time / peak; the chip clocks BELOW 2.4 GHz under this load (last column: GRBM_GUI_ACTIVE / 8 / "
      "duration, which reads high on dispatches shorter than ~0.3 ms - MI355X_MICROARCH.md, DVFS).")
print(f"{'kernel':72s} {'calls/step':>10s} {'us/call':>8s} {'MFMA busy %':>11s} {'bf16 MOPS/step':>14s} {'LDS confl %':>11s} {'L2 hit %':>8s} {'MB fetched/step':>15s} {'MB written/step':>15s} {'clock MHz':>9s}")
keys = sorted(dur, key=lambda k: -dur[k])
for k in keys[:28]:
    mb = 100.0 * mf["SQ_VALU_MFMA_BUSY_CYCLES"].get(k, 0.0) / (NSIMD * dur[k] * 2400.0) if dur.get(k) else float("nan")
    la = lds["SQ_LDS_IDX_ACTIVE"].get(k, 0.0)
    lc = 100.0 * lds["SQ_LDS_BANK_CONFLICT"].get(k, 0.0) / la if la else float("nan")
    h, m = l2["TCC_HIT_sum"].get(k, 0.0), l2["TCC_MISS_sum"].get(k, 0.0)
    hit = 100.0 * h / (h + m) if h + m else float("nan")
    mhz = clk["GRBM_GUI_ACTIVE"].get(k, 0.0) / 8.0 / cdur[k] if cdur.get(k) else float("nan")
    print(f"{k:72s} {calls[k]/steps:10.1f} {dur[k]/max(calls[k],1):8.1f} {mb:11.1f} {ops['SQ_INSTS_VALU_MFMA_MOPS_BF16'].get(k,0.0)/steps:14.3e} {lc:11.2f} {hit:8.1f} "
          f"{2*fetch['FETCH_SIZE'].get(k,0.0)*1024/steps/1e6:15.1f} {write['WRITE_SIZE'].get(k,0.0)*1024/steps/1e6:15.1f} {mhz:9.0f}")
