#!/bin/bash
# Re-measure everything profiles/ quotes for the update step, on the GPU box:  bash tools/refresh_profiles.sh <tag>
# Writes gpurun_out/refresh_<tag>/...; copy what should be judged into profiles/ (python tools/collect_profiles.py <tag> does).
set -e -o pipefail
tag=${1:-r02}
out=gpurun_out/refresh_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
bash tools/pmc_step.sh 4 $tag > $out/pmc_step.log 2>&1 || echo "pmc failed"
cp gpurun_out/pmc_step/summary.txt $out/step_pmc_counters.md 2>/dev/null || true
cp profiles/step_hbm_traffic.json $out/ 2>/dev/null || true
rm -rf gpurun_out/pmc_step/*/
echo "pmc done"
python3 bench.py --steps 60 --warmup 10 > $out/bench_final_bf16.json
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_ovl -o p -- python3 bench.py --steps 12 --warmup 5 --no-cpu-baseline --no-op-rates --no-sampling > $out/prof_ovl.log 2>&1
python3 tools/profile_summary.py $out/prof_ovl 17 $out/step_overlapped.md > /dev/null
python3 tools/gap_analysis.py $out/prof_ovl 17 > $out/step_gaps.txt
echo "overlapped profile done"
V4H_WGRAD_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_ser -o p -- python3 bench.py --steps 12 --warmup 5 --no-cpu-baseline --no-op-rates --no-sampling > $out/prof_ser.log 2>&1
python3 tools/profile_summary.py $out/prof_ser 17 $out/step_serialized.md > /dev/null
echo "serialized profile done"
rm -rf $out/prof_ovl/*/*_kernel_trace.csv $out/prof_ser/*/*_kernel_trace.csv $out/prof_ovl/*_kernel_trace.csv $out/prof_ser/*_kernel_trace.csv 2>/dev/null || true
for w in ds3 ds2_d2 lemurs ds1_photons ds1_pions calogan calohad; do
  python3 bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-op-rates > $out/bench_$w.json
  echo "workload $w done"
done
python3 bench.py --mode f32 --steps 10 --warmup 3 --no-cpu-baseline --no-op-rates --no-sampling > $out/bench_ds2_f32.json
V4H_FORCE_COLLECTIVES=1 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-op-rates --no-sampling > $out/bench_ds2_forced_collectives.json 2> $out/forced_collectives.err || echo "forced-collectives run failed"
[ -f vit4hep_amd/libvit4hep_hip_abl.so ] && VIT4HEP_AMD_LIB=$PWD/vit4hep_amd/libvit4hep_hip_abl.so ABL=1 python3 tools/gemm2_bench.py > $out/gemm2_ablation.txt 2>&1 || true
# in-context A/B of the alternatives that are kept behind switches (interleaved, same box)
for r in 1 2; do
  for v in "V4H_GEMM2=-1" "V4H_GEMM2=0" "V4H_GEMM2_PP=0" "V4H_GEMM2_PP=63" "V4H_GEMM2_PP=53" "V4H_WGRAD_WGS=256" "V4H_LN_RESID=0"; do
    echo -n "$v  " >> $out/ab_in_context.txt
    env $v python3 bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-op-rates 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], 'steps/s', r['ms_per_step'], 'ms', r['sampling']['rk4']['showers_per_s'], 'showers/s (RK4)')" >> $out/ab_in_context.txt
  done
done
echo "all done"
