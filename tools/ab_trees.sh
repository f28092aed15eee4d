#!/bin/bash
# Interleaved A/B of two source trees on one box: bash tools/ab_trees.sh <tag> <other tree> [rounds]   (the other tree has its own built library)
tag=$1; other=$2; rounds=${3:-3}
mkdir -p gpurun_out/$tag
here=$PWD
for i in $(seq 1 $rounds); do
  for t in "$here" "$here/$other"; do
    n=$(basename $t)
    (cd $t && timeout -k 10 200 python bench.py --steps 40 --warmup 10 --lean --no-box > $here/gpurun_out/$tag/${n}_$i.json 2> $here/gpurun_out/$tag/${n}_$i.err) || exit 1
    echo "[$n] round $i: $(grep -o '"value": [0-9.]*' gpurun_out/$tag/${n}_$i.json | head -1)"
  done
done
