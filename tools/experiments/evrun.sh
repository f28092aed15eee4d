# diagnostic: where in the timed region does a one-time stall land?  (per-step device times, V4H_BENCH_STEP_EVENTS)
root=$PWD
run() { tag=$1; steps=$2; shift; shift; for i in 1 2 3; do env "$@" V4H_BENCH_STEP_EVENTS=1 python bench.py --steps $steps --warmup 10 --lean --no-box > $root/gpurun_out/ev_${tag}_$i.json 2> $root/gpurun_out/ev_${tag}_$i.err; python3 - $root/gpurun_out/ev_${tag}_$i.err $root/gpurun_out/ev_${tag}_$i.json $tag $i <<'PY'
import sys, json, re
err = open(sys.argv[1]).read()
m = re.search(r"device ms per step \(rank 0\): \[(.*?)\]", err)
v = [float(x) for x in m.group(1).split(",")]
val = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])["value"]
big = [(i, x) for i, x in enumerate(v[:-1]) if x > 6.0]
print(sys.argv[3], sys.argv[4], val, "steps/s; steps over 6 ms:", big, "median", sorted(v)[len(v) // 2])
PY
done; }
run default 40 X=1
run ahead1 40 V4H_MAX_STEPS_AHEAD=1
run ahead2 40 V4H_MAX_STEPS_AHEAD=2
run ahead4 40 V4H_MAX_STEPS_AHEAD=4
