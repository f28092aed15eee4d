"""GPU idle time between kernels of the update step, from a rocprofv3 --kernel-trace CSV.
usage: python tools/gap_analysis.py <dir with *_kernel_trace.csv> <steps incl. warmup> [skip_steps]
Busy time = union of all kernel intervals (both streams); idle = span - busy.  Also prints the gap histogram per predecessor kernel."""
import csv, glob, re, sys
from collections import defaultdict
d, steps = sys.argv[1], int(sys.argv[2])
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 5
f = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# steady-state window: from the (skip)-th adamw kernel to the last one
adam = [i for i, r in enumerate(rows) if "sq_norm_kernel" in r[2]]  # one per update step (the update itself may be several staged launches)
lo, hi = adam[skip - 1] + 1, adam[-1] + 1
win = rows[lo:hi]
nsteps = len(adam) - skip
span = win[-1][1] - win[0][0]
busy, cur_s, cur_e = 0, win[0][0], win[0][1]
gaps = defaultdict(lambda: [0, 0])
prev_name = win[0][2]
for s, e, n in win[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        g = gaps[re.sub(r"<.*", "", prev_name)[:60]]
        g[0] += s - cur_e; g[1] += 1
        cur_s, cur_e = s, e
        prev_name = n
    elif e > cur_e:
        cur_e = e
        prev_name = n
busy += cur_e - cur_s
print(f"{nsteps} steps: span {span/1e6/nsteps:.3f} ms/step, busy {busy/1e6/nsteps:.3f} ms/step, idle {(span-busy)/1e6/nsteps:.3f} ms/step ({100*(span-busy)/span:.1f} %), "
      f"{len(win)/nsteps:.0f} launches/step, sum of kernel durations {sum(e-s for s,e,_ in win)/1e6/nsteps:.3f} ms/step")
for k, (t, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"  idle after {k:60s} {t/1e3/nsteps:8.1f} us/step  {c/nsteps:5.1f} gaps/step  avg {t/c/1e3:.1f} us")
