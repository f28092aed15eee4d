#!/bin/bash
# HBM traffic and MFMA counters of the whole training step, measured on bench.py itself (program directly after `--`): separate rocprofv3 --pmc
# passes, kernel-trace only.  Writes gpurun_out/pmc_step/... and, from them, profiles/step_hbm_traffic.json (+ the digest of the kernel sources
# measured, which bench.py checks before it reports `roofline.traffic`) and gpurun_out/pmc_step/summary.txt (per-kernel table).
#   bash tools/pmc_step.sh [steps] [tag]
steps=${1:-4}
tag=${2:-r03}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/pmc_step
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  name=$(echo $c | tr ' ' '+')
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$name -- python3 $root/bench.py --steps $steps --warmup 1 --lean --no-box > $out/$name.log 2>&1 || echo "pass $name failed"
  echo "pass $name done"
done
cd $root && python3 tools/pmc_step_summary.py $((steps + 1)) $tag > $out/summary.txt 2>&1
cp profiles/step_hbm_traffic.json $out/step_hbm_traffic.json  # (profiles/ is not merged back from a GPU box: copy it from here)
tail -5 $out/summary.txt
