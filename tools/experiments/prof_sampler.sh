mkdir -p gpurun_out/r05s2
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05s2/prof -o p -- python3 tools/sample_bench.py bf16 2 > gpurun_out/r05s2/log.txt 2>&1
python3 - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/r05s2/prof/**/p_kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms',tot/1e6)
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:22]:
    print(f"{float(r['TotalDurationNs'])/tot*100:5.1f}% {int(r['Calls']):6d} calls {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:110]}")
PY
rm -rf gpurun_out/r05s2/prof/*/*_kernel_trace.csv gpurun_out/r05s2/prof/*_kernel_trace.csv
