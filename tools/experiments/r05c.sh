mkdir -p gpurun_out/r05c
KERNELS=1,2,3 timeout -k 10 500 python tools/block_gemm_bench.py 17280 3 2>&1 | tee gpurun_out/r05c/block_bench.txt
AB_ARGS="--lean" bash tools/ab_env.sh r05c_ab "V4H_GEMM3=0" "V4H_GEMM3=1"
