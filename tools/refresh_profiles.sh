#!/bin/bash
# Re-measure everything profiles/ quotes for the update step, on the GPU box:  bash tools/refresh_profiles.sh <tag> [part ...]
#   part 1: PMC passes + traffic json, the full bench line, overlapped / serialized kernel profiles, gaps     (~8 min)
#   part 2: the other workloads, f32, forced collectives, ds3 serialized profile, ablation build               (~6 min)
#   part 3: in-context A/B of the kept switches                                                                (~10 min)
#   part 4: block contraction bench, attention bench, comm rehearsal, DDP route rehearsal                      (~8 min)
# Writes gpurun_out/refresh_<tag>/...; copy what should be judged into profiles/ (python tools/collect_profiles.py <tag> does).
set -e -o pipefail
tag=${1:-r05}
shift || true
parts="${*:-1 2 3 4}"
out=gpurun_out/refresh_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
has() { case " $parts " in *" $1 "*) return 0;; *) return 1;; esac; }
if has 1; then
bash tools/pmc_step.sh 4 $tag > $out/pmc_step.log 2>&1 || echo "pmc failed"
cp gpurun_out/pmc_step/summary.txt $out/step_pmc_counters.md 2>/dev/null || true
cp profiles/step_hbm_traffic.json $out/ 2>/dev/null || true
rm -rf gpurun_out/pmc_step/*/
echo "pmc done"
python3 bench.py --steps 60 --warmup 10 > $out/bench_final_bf16.json
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_ovl -o p -- python3 bench.py --steps 12 --warmup 5 --lean --no-box > $out/prof_ovl.log 2>&1
python3 tools/profile_summary.py $out/prof_ovl 17 $out/step_overlapped.md > /dev/null
python3 tools/gap_analysis.py $out/prof_ovl 17 > $out/step_gaps.txt
python3 tools/timeline.py $out/prof_ovl 3 > $out/step_timeline.txt
echo "overlapped profile done"
V4H_WGRAD_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_ser -o p -- python3 bench.py --steps 12 --warmup 5 --lean --no-box > $out/prof_ser.log 2>&1
python3 tools/profile_summary.py $out/prof_ser 17 $out/step_serialized.md > /dev/null
echo "serialized profile done"
rm -rf $out/prof_ovl/*/*_kernel_trace.csv $out/prof_ser/*/*_kernel_trace.csv $out/prof_ovl/*_kernel_trace.csv $out/prof_ser/*_kernel_trace.csv 2>/dev/null || true
fi
if has 2; then
V4H_WGRAD_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_ds3 -o p -- python3 bench.py --workload ds3 --steps 8 --warmup 3 --lean --no-box > $out/prof_ds3.log 2>&1
python3 tools/profile_summary.py $out/prof_ds3 11 $out/ds3_step_serialized.md > /dev/null
rm -rf $out/prof_ds3/*/*_kernel_trace.csv $out/prof_ds3/*_kernel_trace.csv 2>/dev/null || true
echo "ds3 profile done"
for w in ds3 ds2_d2 lemurs ds1_photons ds1_pions calogan calohad; do
  python3 bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-op-rates --no-other --no-box > $out/bench_$w.json
  echo "workload $w done"
done
python3 bench.py --mode f32 --steps 10 --warmup 3 --lean --no-box > $out/bench_ds2_f32.json
V4H_FORCE_COLLECTIVES=1 python3 bench.py --steps 30 --warmup 5 --lean --no-box > $out/bench_ds2_forced_collectives.json 2> $out/forced_collectives.err || echo "forced-collectives run failed"
# (the ablation build does not travel to the GPU box - .gpurunignore - and is made here, on request: ABL_BENCH=1)
if [ "${ABL_BENCH:-0}" = "1" ]; then
  V4H_BUILD_TAG=abl V4H_EXTRA_FLAGS=-DV4H_ABLATIONS python3 -m vit4hep_amd.build > $out/abl_build.log 2>&1 && \
  VIT4HEP_AMD_LIB=$PWD/vit4hep_amd/libvit4hep_hip_abl.so ABL=1 python3 tools/gemm2_bench.py > $out/gemm2_ablation.txt 2>&1 || true
fi
fi
if has 3; then
# in-context A/B of the alternatives that are kept behind switches (interleaved, same box)
for r in 1 2; do
  for v in "V4H_GEMM2=-1" "VIT4HEP_AMD_RESIDUAL=f32" "VIT4HEP_AMD_RESIDUAL=x_bf16" "VIT4HEP_AMD_RESIDUAL=dx_bf16" "V4H_GEMM3=0" "V4H_GEMM3=3" "V4H_GEMM3=39" "V4H_GEMM3=63" "V4H_STOP_EVENTS=0" "V4H_OVERWRITE_GRADS=0" "V4H_MLP_TILE=0" "V4H_LNB_V2=0" "V4H_PIPELINE_UPDATE=1" "V4H_GEMM2=0" "V4H_GEMM2=8" "V4H_GEMM_SMALL=0" "V4H_ATTN_DENSE=0" "V4H_WGRAD_WGS=-4" "V4H_BATCH_ADALN=0" "V4H_WGRAD_OVERLAP=0"; do
    echo -n "$v  " >> $out/ab_in_context.txt
    env $v python3 bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-op-rates --no-other --no-box 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], 'steps/s', r['ms_per_step'], 'ms', r['sampling']['rk4']['showers_per_s'], 'showers/s (RK4)')" >> $out/ab_in_context.txt
  done
done
echo "A/B done"
fi
if has 4; then
KERNELS=1,2,3 python3 tools/block_gemm_bench.py 17280 5 > $out/block_gemm_bench.txt 2>&1 || echo "block bench failed"
python3 tools/attn_bench.py > $out/attn_bench.txt 2>&1 || echo "attn bench failed"
python3 tools/comm_interference.py --steps 30 --rounds 2 > $out/comm_interference.txt 2> $out/comm_interference.err || echo "comm rehearsal failed"
python3 tools/ddp_route_bench.py --steps 20 --rounds 2 > $out/ddp_route.txt 2> $out/ddp_route.err || echo "DDP route rehearsal failed"
fi
echo "all done"
