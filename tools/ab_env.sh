#!/bin/bash
# Interleaved A/B of environment settings inside the update step, on one box.  usage: bash tools/ab_env.sh <tag> "<ENV=.. ENV=..>" "<ENV=..>" ...
# (each argument after the tag is one variant: a space-separated list of VAR=value; "-" = the defaults)
tag=$1; shift
mkdir -p gpurun_out/$tag
for i in 1 2; do
  n=0
  for v in "$@"; do
    n=$((n+1))
    if [ "$v" = "-" ]; then envs=""; else envs="$v"; fi
    env $envs timeout -k 10 200 python bench.py --steps 40 --warmup 10 ${AB_ARGS:---no-sampling} > gpurun_out/$tag/v${n}_$i.json 2> gpurun_out/$tag/v${n}_$i.err || exit 1
    echo "[$v] round $i: $(grep -o '"value": [0-9.]*' gpurun_out/$tag/v${n}_$i.json | head -1) $(grep -o '"showers_per_s[a-z_0-9]*": [0-9.]*' gpurun_out/$tag/v${n}_$i.json | head -2 | tr '\n' ' ')"
  done
done
