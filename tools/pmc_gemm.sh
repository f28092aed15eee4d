#!/bin/bash
# usage (GPU box): bash tools/pmc_gemm.sh  -> gpurun_out/pmc_*/  (one rocprofv3 --pmc pass per counter group)
cd /tmp && export TMPDIR=/tmp
export CFGS=${CFGS:-1} WCFGS=${WCFGS:-5}
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /root/repo/gpurun_out/pmc_$name -- python3 /root/repo/tools/gemm_bench.py bf16 > /root/repo/gpurun_out/pmc_$name.log 2>&1; }
run l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_WAVES
