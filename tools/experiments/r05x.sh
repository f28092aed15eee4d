AB_ARGS="--lean" bash tools/ab_env.sh r05x_ab "-" "V4H_ATTN_DENSE=0" "-" "V4H_ATTN_DENSE=0"
V4H_ATTN_DENSE=0 bash tools/prof_step.sh r05x_persist > gpurun_out/r05x_persist.log 2>&1
