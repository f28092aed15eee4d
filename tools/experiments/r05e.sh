mkdir -p gpurun_out/r05e
for bt in 8640 17280 34560 69120; do
KERNELS=2,3 ONLY="fwd qkv" timeout -k 10 300 python tools/block_gemm_bench.py $bt 3 2>&1 | grep "fwd qkv" | tee -a gpurun_out/r05e/qkv_vs_rows.txt
KERNELS=2,3 ONLY="fwd proj" timeout -k 10 300 python tools/block_gemm_bench.py $bt 3 2>&1 | grep "fwd proj" | tee -a gpurun_out/r05e/qkv_vs_rows.txt
KERNELS=2,3 ONLY="fc1 plain" timeout -k 10 300 python tools/block_gemm_bench.py $bt 3 2>&1 | grep "fc1 plain" | tee -a gpurun_out/r05e/qkv_vs_rows.txt
done
