mkdir -p gpurun_out/r05i
timeout -k 10 200 python tools/experiments/gemm3_warm.py qkv proj fc1 dproj dfc2 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05i/warm.txt
