#!/bin/bash
# which setting lets the 2-rank rehearsal of bench.py (both ranks on cuda:0, gloo) finish at full size?  usage: bash tools/experiments/rehearsal_bisect.sh "<ENV=.. args>" ...
mkdir -p gpurun_out/rehb; n=0
for v in "$@"; do
  n=$((n+1)); port=$((29540+n))
  envs=$(echo "$v" | tr ' ' '\n' | grep '=' | tr '\n' ' '); args=$(echo "$v" | tr ' ' '\n' | grep -v '=' | tr '\n' ' ')
  t0=$(date +%s)
  env $envs V4H_BENCH_WATCHDOG=60 V4H_BENCH_REHEARSAL=1 timeout -k 10 100 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $port bench.py --gpus 2 --steps 6 --warmup 2 --lean --no-box $args > gpurun_out/rehb/$n.out 2> gpurun_out/rehb/$n.err
  rc=$?
  echo "[$v] rc=$rc $(( $(date +%s) - t0 )) s  $(grep -o '"value": [0-9.]*' gpurun_out/rehb/$n.out | head -1)  $(grep -c 'in finish' gpurun_out/rehb/$n.err) stacks in finish()"
done
