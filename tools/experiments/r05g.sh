mkdir -p gpurun_out/r05g
timeout -k 10 600 python -m pytest tests/test_hip_round5.py -x -q -k "weight_stationary" > gpurun_out/r05g/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r05g/tests.log
for w in qkv proj; do VIT4HEP_AMD_LIB=$PWD/vit4hep_amd/libvit4hep_hip_st3.so timeout -k 10 120 python tools/experiments/gemm3_stamps.py $w 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r05g/stamps.txt; done
KERNELS=2,3 timeout -k 10 500 python tools/block_gemm_bench.py 17280 3 2>&1 | grep -v "wgrad\|amdgpu.ids" | tee gpurun_out/r05g/block_bench.txt
