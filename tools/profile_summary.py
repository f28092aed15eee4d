"""Summarise a rocprofv3 --kernel-trace --stats CSV directory: per-kernel time per step.
usage: python tools/profile_summary.py <dir with *_kernel_stats.csv> <steps incl. warmup> [out.md]"""
import csv, glob, re, sys
d, steps = sys.argv[1], int(sys.argv[2])
f = glob.glob(d + "/**/*_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
lines = [f"source: {f}", f"total kernel time {tot/1e6:.3f} ms over {steps} steps = {tot/1e6/steps:.3f} ms/step", "",
         "| ms/step | calls/step | avg us | % | kernel |", "|---|---|---|---|---|"]
def short(n):
    n = re.sub(r"\(anonymous namespace\)::|v4h::|void ", "", n)
    m = re.match(r"_Z15v4h_gemm_kernelI7GemmCfgI(DF16b|f)(DF16b|f)Lb(\d)ELb(\d)ELi(\d+)ELi(\d+)ELi(\d+)ELi\d+ELi\d+ELi(\d+)ELb(\d)", n)
    if not m:
        m = re.match(r"_Z15v4h_gemm_kernelI7GemmCfgI(DF16b|f)(DF16b|f)Lb(\d)ELb(\d)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb(\d)", n)
    if m:
        t, _, pks, qks, bi, bj, bk, epi, cs = m.groups()
        epis = ["STORE","STORE_F32","SILU","COND_SUM","EMBED","GATE_RESID","GELU","DGELU","DSILU","ATOMIC_F32","ACCUM_F32","UNPATCH","SLAB_F32","RELU","ROWADD_SILU"]
        lay = {("0","0"):"fwd",("0","1"):"dgrad",("1","1"):"wgrad"}[(pks,qks)]
        return f"gemm<{'bf16' if t=='DF16b' else 'f32'},{lay},{bi}x{bj}x{bk},{epis[int(epi)]}{',colsum' if cs=='1' else ''}>"
    m = re.match(r"v4h_gemm2_kernel<Gemm2Cfg<(true|false), (true|false), (\d+), (true|false), (\d+)(?:, (true|false))? ?> ?>", n)
    if m:  # the 256 x 160 ring kernel (v4h_gemm2.h): lock-step or ping-pong schedule
        pks, qks, epi, cs, dbg, pp = m.groups()
        epis = ["STORE","STORE_F32","SILU","COND_SUM","EMBED","GATE_RESID","GELU","DGELU","DSILU","ATOMIC_F32","ACCUM_F32","UNPATCH","SLAB_F32","RELU","ROWADD_SILU"]
        lay = {("false","false"):"fwd",("false","true"):"dgrad",("true","true"):"wgrad"}[(pks,qks)]
        return f"gemm2<bf16,{lay},256x160x64,{epis[int(epi)]}{',colsum' if cs=='true' else ''},{'ping-pong' if pp=='true' else 'lock-step'}{',dbg'+dbg if dbg!='0' else ''}>"
    m = re.match(r"_ZN(?:3v4h)?12_GLOBAL__N_1(\d+)([a-z_0-9]+)I(DF16b|f)", n)
    if m: return f"{m.group(2)[:int(m.group(1))]}<{'bf16' if m.group(3)=='DF16b' else 'f32'}>"
    return n[:90]
for r in rows[:40]:
    lines.append(f'| {float(r["TotalDurationNs"])/1e6/steps:.3f} | {int(r["Calls"])/steps:.1f} | {float(r["AverageNs"])/1e3:.1f} | {float(r["Percentage"]):.1f} | {short(r["Name"])} |')
out = "\n".join(lines)
print(out)
if len(sys.argv) > 3: open(sys.argv[3], "w").write(out + "\n")
