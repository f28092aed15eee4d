# diagnostic: one-time host stall at the start of the timed region - which configuration shows it?  Interleaved: bash tools/experiments/evrun.sh <rounds> "<ENV..>" "<ENV..>" ...
root=$PWD
rounds=$1; shift
for i in $(seq 1 $rounds); do
  n=0
  for v in "$@"; do
    n=$((n+1))
    if [ "$v" = "-" ]; then envs="X=1"; else envs="$v"; fi
    env $envs ${EVMODE:-V4H_BENCH_STEP_EVENTS}=1 python bench.py --steps 20 --warmup 5 --lean --no-box > $root/gpurun_out/ev_${n}_$i.json 2> $root/gpurun_out/ev_${n}_$i.err
    python3 - $root/gpurun_out/ev_${n}_$i.err $root/gpurun_out/ev_${n}_$i.json "$v" $i <<'PY'
import sys, json, re
err = open(sys.argv[1]).read()
m = re.search(r"host calls over 10 ms \(index, ms since the start of the timed region, ms\): (.*)", err)
val = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])["value"]
print(f"[{sys.argv[3]}] round {sys.argv[4]}: {val} steps/s; host calls over 10 ms: {m.group(1) if m else '?'}")
PY
  done
done
