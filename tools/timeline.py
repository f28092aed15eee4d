"""Timeline of one steady-state update step from a rocprofv3 --kernel-trace CSV: start, duration, queue and kernel, per launch.
usage: python tools/timeline.py <dir with *_kernel_trace.csv> [step index from the end, default 2]"""
import csv, glob, re, sys
d = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
f = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
adam = [i for i, r in enumerate(rows) if "sq_norm_kernel" in r[3]]  # one per update step (the update itself may be several staged launches)
lo, hi = adam[-back - 1] + 1, adam[-back] + 1
t0 = rows[lo][0]
queues = sorted({r[2] for r in rows[lo:hi]})
def short(n):
    n = re.sub(r"\(anonymous namespace\)::|v4h::|void ", "", n)
    m = re.match(r"_Z15v4h_gemm_kernelI7GemmCfgI(DF16b|f)(DF16b|f)Lb(\d)ELb(\d)ELi(\d+)ELi(\d+)ELi(\d+)ELi\d+ELi\d+ELi(\d+)ELb(\d)", n)
    if m:
        t, _, pks, qks, bi, bj, bk, epi, cs = m.groups()
        lay = {("0", "0"): "fwd", ("0", "1"): "dgrad", ("1", "1"): "wgrad"}[(pks, qks)]
        return f"gemm {lay} epi{epi}"
    m = re.match(r"v4h_gemm2_kernel<Gemm2Cfg<(true|false), (true|false), (\d+), (true|false), (\d+)(?:, (true|false))? ?> ?>", n)
    if m:
        lay = {("false", "false"): "fwd", ("false", "true"): "dgrad", ("true", "true"): "wgrad"}[(m.group(1), m.group(2))]
        return f"gemm2 {lay} epi{m.group(3)} {'pp' if m.group(6) == 'true' else 'ls'}"
    m = re.match(r"_ZN(?:3v4h)?12_GLOBAL__N_1(\d+)([a-z_0-9]+)", n)
    if m: return m.group(2)[:int(m.group(1))]
    return n[:50]
last_end = {q: None for q in queues}
for s, e, q, n in rows[lo:hi]:
    col = queues.index(q)
    gap = "" if last_end[q] is None else f"(+{(s - last_end[q]) / 1e3:6.1f})"
    last_end[q] = e
    print(f"{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:7.1f} us  {'                         ' * col}q{col} {gap:10s} {short(n)}")
