#!/bin/bash
# round-4 baseline: how the driver's short run (--steps 20 --warmup 5) compares with the builder's longer A/B runs on one box, per-step device times, K-split A/B
out=gpurun_out/r04a; mkdir -p $out
B="--lean --no-box"
val() { grep -o '"value": [0-9.]*' $1 | head -1; }
for i in 1 2 3; do
  python3 bench.py --steps 20 --warmup 5 $B > $out/d_$i.json 2> $out/d_$i.err || exit 1; echo "20/5 run $i: $(val $out/d_$i.json)"
  python3 bench.py --steps 40 --warmup 10 $B > $out/l_$i.json 2> $out/l_$i.err || exit 1; echo "40/10 run $i: $(val $out/l_$i.json)"
  V4H_WGRAD_WGS=-4 python3 bench.py --steps 40 --warmup 10 $B > $out/k4_$i.json 2> $out/k4_$i.err || exit 1; echo "40/10 ksplit4 run $i: $(val $out/k4_$i.json)"
done
V4H_BENCH_STEP_EVENTS=1 python3 bench.py --steps 20 --warmup 5 $B > $out/ev.json 2> $out/ev.err || exit 1; echo "events: $(val $out/ev.json)"; grep "device ms" $out/ev.err
V4H_BENCH_STEP_EVENTS=1 python3 bench.py --steps 30 --warmup 0 $B > $out/ev0.json 2> $out/ev0.err || exit 1; echo "events w0: $(val $out/ev0.json)"; grep "device ms" $out/ev0.err
