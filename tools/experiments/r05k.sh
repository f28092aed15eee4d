mkdir -p gpurun_out/r05k
timeout -k 10 300 python tools/experiments/gemm3_warm.py J1152 J1536 J1440 J1488 J768 J384 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05k/warm.txt
