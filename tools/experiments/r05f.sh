mkdir -p gpurun_out/r05f
V4H_BUILD_TAG=st3 V4H_EXTRA_FLAGS=-DV4H_GEMM3_STAMPS python -m vit4hep_amd.build > gpurun_out/r05f/build.log 2>&1 || { tail gpurun_out/r05f/build.log; exit 1; }
for w in qkv proj fc1 dproj; do VIT4HEP_AMD_LIB=$PWD/vit4hep_amd/libvit4hep_hip_st3.so timeout -k 10 120 python tools/experiments/gemm3_stamps.py $w 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r05f/stamps.txt; done
