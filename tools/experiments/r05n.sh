mkdir -p gpurun_out/r05n
python -m pytest tests -m gpu -x -q -s > gpurun_out/r05n/tests.log 2>&1; echo "tests rc=$?"
grep -a "residual storage\|passed\|failed" gpurun_out/r05n/tests.log | tail -8
AB_ARGS="--lean" bash tools/ab_env.sh r05n_ab "-" "V4H_WGRAD_WGS=-4" "V4H_WGRAD_WGS=-6"
