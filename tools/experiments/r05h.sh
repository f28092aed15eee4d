mkdir -p gpurun_out/r05h
for t in st3a; do echo "== $t"; VIT4HEP_AMD_LIB=$PWD/vit4hep_amd/libvit4hep_hip_$t.so timeout -k 10 120 python tools/experiments/gemm3_stamps.py qkv 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r05h/stamps3.txt; done
