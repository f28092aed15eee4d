set -e -o pipefail
out=gpurun_out/refresh_r01f; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
python3 bench.py --steps 60 --warmup 10 > $out/bench_final_bf16.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_ovl -o p -- python3 bench.py --steps 12 --warmup 5 > $out/prof_ovl.log 2>&1
python3 tools/profile_summary.py $out/prof_ovl 17 $out/step_overlapped.md > /dev/null
python3 tools/gap_analysis.py $out/prof_ovl 17 > $out/step_gaps.txt
python3 tools/timeline.py $out/prof_ovl > $out/timeline.txt
V4H_WGRAD_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_ser -o p -- python3 bench.py --steps 12 --warmup 5 > $out/prof_ser.log 2>&1
python3 tools/profile_summary.py $out/prof_ser 17 $out/step_serialized.md > /dev/null
tail -1 $out/bench_final_bf16.json | cut -c1-160
head -2 $out/step_gaps.txt; head -2 $out/step_serialized.md
