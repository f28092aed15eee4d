#!/bin/bash
# HBM traffic of the whole training step: separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), kernel-trace only.
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /root/repo/gpurun_out/pmc_step_$c -- python3 /root/repo/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-op-rates > /root/repo/gpurun_out/pmc_step_$c.log 2>&1
done
