mkdir -p gpurun_out/r05b
timeout -k 10 600 python -m pytest tests/test_hip_round5.py -x -q -s -k "weight_stationary" > gpurun_out/r05b/tests.log 2>&1; echo "tests rc=$?"
tail -25 gpurun_out/r05b/tests.log
KERNELS=2,3 ONLY="fwd qkv" timeout -k 10 300 python tools/block_gemm_bench.py 17280 3 2>&1 | tee gpurun_out/r05b/bench_qkv.txt
