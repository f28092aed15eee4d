V4H_GEMM3=1 bash tools/prof_step.sh r05d_ws > gpurun_out/r05d_ws.log 2>&1
V4H_GEMM3=0 bash tools/prof_step.sh r05d_base > gpurun_out/r05d_base.log 2>&1
echo done
