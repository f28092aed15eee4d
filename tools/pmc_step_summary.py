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
tag = sys.argv[2] if len(sys.argv) > 2 else "r02"
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
    for r in csv.DictReader(open(f[0])):
        k = short(r["Kernel_Name"])
        per[r["Counter_Name"]][k] += float(r["Counter_Value"])
        key = (r.get("Dispatch_Id"), r["Counter_Name"])
        if r.get("Start_Timestamp") and (r.get("Dispatch_Id"), "t") not in seen:
            seen.add((r.get("Dispatch_Id"), "t"))
            dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
            calls[k] += 1
    return per, dur, calls


fetch, _, _ = load("FETCH_SIZE")
write, _, _ = load("WRITE_SIZE")
tot_f = sum(fetch["FETCH_SIZE"].values()) * 1024 / steps
tot_w = sum(write["WRITE_SIZE"].values()) * 1024 / steps
print(f"HBM traffic per update step: FETCH_SIZE raw {tot_f/1e9:.3f} GB (x2 for 16-B/lane streams = {2*tot_f/1e9:.3f} GB), WRITE_SIZE {tot_w/1e9:.3f} GB  ->  {(2*tot_f+tot_w)/1e9:.3f} GB/step")
if tot_f > 0 and tot_w > 0:
    from vit4hep_amd.build import kernel_digest as _digest

    path = "profiles/step_hbm_traffic.json"
    rec = json.load(open(path)) if os.path.exists(path) else {}
    rec["ds2/bf16"] = {"bytes_per_step": 2 * tot_f + tot_w, "fetch_size_raw": tot_f, "write_size": tot_w, "kernel_digest": _digest(), "steps_profiled": steps,
                       "how": "tools/pmc_step.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python3 bench.py; FETCH doubled per MI355X_MICROARCH.md"}
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
