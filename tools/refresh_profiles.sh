#!/bin/bash
# Re-measure everything profiles/ quotes for the update step, on the GPU box:  bash tools/refresh_profiles.sh <tag>
# Writes gpurun_out/refresh_<tag>/...; copy what should be judged into profiles/.
set -e -o pipefail
tag=${1:-r01}
out=gpurun_out/refresh_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
python3 bench.py --steps 60 --warmup 10 > $out/bench_final_bf16.json
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_ovl -o p -- python3 bench.py --steps 12 --warmup 5 > $out/prof_ovl.log 2>&1
python3 tools/profile_summary.py $out/prof_ovl 17 $out/step_overlapped.md > /dev/null
python3 tools/gap_analysis.py $out/prof_ovl 17 > $out/step_gaps.txt
echo "overlapped profile done"
V4H_WGRAD_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_ser -o p -- python3 bench.py --steps 12 --warmup 5 > $out/prof_ser.log 2>&1
python3 tools/profile_summary.py $out/prof_ser 17 $out/step_serialized.md > /dev/null
echo "serialized profile done"
rm -rf $out/prof_ovl/*/*_kernel_trace.csv $out/prof_ser/*/*_kernel_trace.csv $out/prof_ovl/*_kernel_trace.csv $out/prof_ser/*_kernel_trace.csv 2>/dev/null || true
for w in ds2 ds3 ds2_d2 lemurs ds1_photons ds1_pions calogan calohad; do
  python3 bench.py --workload $w --steps 20 --warmup 5 > $out/bench_$w.json
  echo "workload $w done"
done
python3 tools/sample_bench.py > $out/sample_bench.log 2>&1
echo "sampling done"
