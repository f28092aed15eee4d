mkdir -p gpurun_out/r05a
python -m pytest tests -m gpu -x -q > gpurun_out/r05a/tests.log 2>&1; echo "tests rc=$?" 
tail -15 gpurun_out/r05a/tests.log
AB_ARGS="--lean" bash tools/ab_env.sh r05a_ab "VIT4HEP_AMD_RESIDUAL=f32" "VIT4HEP_AMD_RESIDUAL=bf16" "VIT4HEP_AMD_RESIDUAL=x_bf16" "VIT4HEP_AMD_RESIDUAL=dx_bf16"
