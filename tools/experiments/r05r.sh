mkdir -p gpurun_out/r05r
timeout -k 10 600 python -m pytest tests/test_hip_round5.py -x -q -k "weight_stationary" > gpurun_out/r05r/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r05r/tests.log
for w in qkv proj; do VIT4HEP_AMD_LIB=$PWD/vit4hep_amd/libvit4hep_hip_st3.so timeout -k 10 120 python tools/experiments/gemm3_stamps.py $w 2>&1 | grep -v amdgpu.ids | grep -v "interval [0-46-9]" | tee -a gpurun_out/r05r/stamps.txt; done
timeout -k 10 200 python tools/experiments/gemm3_warm.py J1152 qkv proj fc1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05r/warm.txt
KERNELS=2,3 ONLY="fwd" timeout -k 10 300 python tools/block_gemm_bench.py 17280 3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05r/block.txt
