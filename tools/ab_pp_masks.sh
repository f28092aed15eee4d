#!/bin/bash
# A/B of the ping-pong dispatch mask inside the update step (interleaved rounds on one box).  usage: bash tools/ab_pp_masks.sh <tag> "<masks>"
tag=$1; masks=${2:-"31 0 30 29 27 23 15"}
mkdir -p gpurun_out/$tag
for i in 1 2; do
  for m in $masks; do
    V4H_GEMM2_PP=$m timeout -k 10 200 python bench.py --steps 40 --warmup 10 --no-sampling > gpurun_out/$tag/m${m}_$i.json 2> gpurun_out/$tag/m${m}_$i.err || exit 1
    echo "mask $m round $i: $(grep -o '"value": [0-9.]*' gpurun_out/$tag/m${m}_$i.json | head -1)"
  done
done
