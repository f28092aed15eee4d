mkdir -p gpurun_out/r05v
for r in 1 2; do for v in 3 35 34; do echo -n "V4H_GEMM3=$v: "; V4H_GEMM3=$v timeout -k 10 200 python tools/sample_bench.py bf16 4 2>&1 | grep -v amdgpu.ids | tr '\n' ' '; echo; done; done | tee gpurun_out/r05v/sampling_ab.txt
