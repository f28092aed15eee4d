timeout -k 10 300 python -m pytest tests/test_hip_network.py tests/test_hip_ops.py -x -q 2>&1 | tail -2
AB_ARGS="--steps 40 --warmup 8 --no-cpu-baseline --no-op-rates --no-other --no-box" bash tools/ab_env.sh r05w_ab "-" "V4H_UNPATCH_FUSED=1"
