mkdir -p gpurun_out/r05j
timeout -k 10 600 python -m pytest tests/test_hip_round5.py -x -q -k "weight_stationary" > gpurun_out/r05j/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r05j/tests.log
for w in qkv; do VIT4HEP_AMD_LIB=$PWD/vit4hep_amd/libvit4hep_hip_st3.so timeout -k 10 120 python tools/experiments/gemm3_stamps.py $w 2>&1 | grep -v amdgpu.ids | grep -v "interval [0-46-9]" | tee -a gpurun_out/r05j/stamps.txt; done
timeout -k 10 200 python tools/experiments/gemm3_warm.py qkv proj fc1 dproj dfc2 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05j/warm.txt
