mkdir -p gpurun_out/r05q
V4H_STAGED_LATE=0 timeout -k 10 300 python -m pytest tests/test_hip_round2.py -x -q -k "ddp_wrapped" > gpurun_out/r05q/ddp_late0.log 2>&1; echo "late0 rc=$?"; grep -a "AssertionError:\|passed\|failed" gpurun_out/r05q/ddp_late0.log | head -3
V4H_STAGED_LATE=1 timeout -k 10 300 python -m pytest tests/test_hip_round2.py -x -q -k "ddp_wrapped" > gpurun_out/r05q/ddp_late1.log 2>&1; echo "late1 rc=$?"; grep -a "AssertionError:\|passed\|failed" gpurun_out/r05q/ddp_late1.log | head -3
timeout -k 10 300 python -m pytest tests/test_hip_round5.py -x -q -k "weight_stationary" > gpurun_out/r05q/tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r05q/tests.log
KERNELS=1,3 ONLY="GELU" timeout -k 10 300 python tools/block_gemm_bench.py 17280 3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05q/block_gelu.txt
AB_ARGS="--lean" bash tools/ab_env.sh r05q_ab "-" "V4H_GEMM3=19" "V4H_GEMM3=23" "V4H_LNB_ROWS=16" "V4H_LNB_ROWS=24"
