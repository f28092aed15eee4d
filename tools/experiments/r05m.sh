AB_ARGS="--lean" bash tools/ab_env.sh r05m_ab "V4H_GEMM3=0" "V4H_GEMM3=1" "V4H_GEMM3=3" "V4H_GEMM3=11"
