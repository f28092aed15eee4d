#!/bin/bash
# quick kernel-trace profiles of the step (overlapped and serialized streams) -> gpurun_out/<tag>/
tag=${1:-prof}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_ovl -o p -- python3 bench.py --steps 12 --warmup 5 --lean --no-box > $out/prof_ovl.log 2>&1
python3 tools/profile_summary.py $out/prof_ovl 17 $out/step_overlapped.md > /dev/null
python3 tools/gap_analysis.py $out/prof_ovl 17 > $out/step_gaps.txt
python3 tools/timeline.py $out/prof_ovl 3 > $out/step_timeline.txt
V4H_WGRAD_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_ser -o p -- python3 bench.py --steps 12 --warmup 5 --lean --no-box > $out/prof_ser.log 2>&1
python3 tools/profile_summary.py $out/prof_ser 17 $out/step_serialized.md > /dev/null
rm -rf $out/prof_ovl/*/*_kernel_trace.csv $out/prof_ser/*/*_kernel_trace.csv $out/prof_ovl/*_kernel_trace.csv $out/prof_ser/*_kernel_trace.csv 2>/dev/null || true
head -3 $out/step_gaps.txt; head -30 $out/step_serialized.md
