mkdir -p gpurun_out/r05l
for w in gelu dgelu; do VIT4HEP_AMD_LIB=$PWD/vit4hep_amd/libvit4hep_hip_st3.so timeout -k 10 120 python tools/experiments/gemm3_stamps.py $w 2>&1 | grep -v amdgpu.ids | grep -v "interval [0-46-9]" | tee -a gpurun_out/r05l/stamps.txt; done
