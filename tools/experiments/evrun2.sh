# diagnostic: the driver's own command several times in a row, host time of every step() call logged
for i in $(seq 1 ${1:-8}); do
  V4H_BENCH_HOST_TIMES=1 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/drv_$i.json 2> gpurun_out/drv_$i.err
  echo "full $i: $(grep -o '"value": [0-9.]*' gpurun_out/drv_$i.json | head -1) $(grep 'host calls over' gpurun_out/drv_$i.err | sed 's/.*ms): //')"
done
