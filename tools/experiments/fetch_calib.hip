// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 against KNOWN byte counts, per access shape: the guide says FETCH_SIZE
// reads exactly half the bytes of a 16-byte-per-lane coalesced stream; which of the path's kernels have that shape?  Each kernel reads (or writes) a buffer
// of `bytes` exactly once; buffers are 1 GiB, far beyond the 256 MiB Infinity Cache, and every launch uses a fresh quarter of a 4 GiB arena.
//   hipcc --offload-arch=gfx950 -O3 -o tools/experiments/fetch_calib tools/experiments/fetch_calib.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- tools/experiments/fetch_calib      (then again with --pmc WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

__global__ void read16_stream(const f32x4* __restrict__ p, float* out, long n16) {  // lane i: 16 bytes at 16 i (1 KiB per wave-instruction)
  f32x4 s = {0, 0, 0, 0};
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x) s += p[i];
  if (s[0] + s[1] + s[2] + s[3] == 123.456f) out[0] = 1.f;
}
__global__ void read2x16_rows(const f32x4* __restrict__ p, float* out, long nrows) {  // LayerNorm shape: a wave per 1920-byte row, lane l reads 16 B at 32 l and at 32 l + 16
  f32x4 s = {0, 0, 0, 0};
  const int lane = threadIdx.x & 63;
  for (long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6); r < nrows; r += (long)gridDim.x * 4) {
    if (lane < 60) {
      const f32x4* row = p + r * 120;
      s += row[2 * lane];
      s += row[2 * lane + 1];
    }
  }
  if (s[0] + s[1] + s[2] + s[3] == 123.456f) out[0] = 1.f;
}
__global__ void read8_stream(const f32x2* __restrict__ p, float* out, long n8) {  // 8 bytes per lane
  f32x2 s = {0, 0};
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) s += p[i];
  if (s[0] + s[1] == 123.456f) out[0] = 1.f;
}
__global__ void read4_stream(const float* __restrict__ p, float* out, long n4) {  // 4 bytes per lane
  float s = 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) s += p[i];
  if (s == 123.456f) out[0] = 1.f;
}
__global__ void dma16_stream(const f32x4* __restrict__ p, float* out, long n16) {  // global_load_lds_dwordx4: 16 bytes per lane straight into LDS
  __shared__ __attribute__((aligned(16))) char lds[4 * 1024];
  const int wave = threadIdx.x >> 6;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + i), (__attribute__((address_space(3))) void*)(lds + wave * 1024), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (((float*)lds)[threadIdx.x] == 123.456f) out[0] = 1.f;
}
__global__ void read_slices(const char* __restrict__ p, float* out, long nrows) {  // attention shape: 160-byte slices (10 lanes x 16 B) of 2880-byte rows, 6 slices of a row by 6 workgroups
  f32x4 s = {0, 0, 0, 0};
  const int h = blockIdx.y;
  for (long u = (long)blockIdx.x * blockDim.x + threadIdx.x; u < nrows * 10; u += (long)gridDim.x * blockDim.x) {
    const long r = u / 10, ch = u % 10;
    s += *(const f32x4*)(p + r * 2880 + 960 + h * 160 + ch * 16);
  }
  if (s[0] + s[1] + s[2] + s[3] == 123.456f) out[0] = 1.f;
}
__global__ void write16_stream(f32x4* __restrict__ p, long n16) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x) p[i] = f32x4{1, 2, 3, 4};
}
__global__ void write2x16_rows(f32x4* __restrict__ p, long nrows) {
  const int lane = threadIdx.x & 63;
  for (long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6); r < nrows; r += (long)gridDim.x * 4)
    if (lane < 60) {
      p[r * 120 + 2 * lane] = f32x4{1, 2, 3, 4};
      p[r * 120 + 2 * lane + 1] = f32x4{1, 2, 3, 4};
    }
}
__global__ void write_slices(char* __restrict__ p, long nrows) {  // attention output shape: 160-byte slices of 960-byte rows
  const int h = blockIdx.y;
  for (long u = (long)blockIdx.x * blockDim.x + threadIdx.x; u < nrows * 10; u += (long)gridDim.x * blockDim.x) {
    const long r = u / 10, ch = u % 10;
    *(f32x4*)(p + r * 960 + h * 160 + ch * 16) = f32x4{1, 2, 3, 4};
  }
}
__global__ void atomic_rows(float* __restrict__ p, long n4) {  // float atomics, 256 contiguous bytes per wave-instruction
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) atomicAdd(p + i, 1.0f);
}

int main() {
  const long Q = 1L << 30;  // bytes per launch
  char* arena;
  float* out;
  if (hipMalloc(&arena, 4 * Q) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(arena, 0, 4 * Q);
  hipDeviceSynchronize();
  int k = 0;
  auto buf = [&]() { return arena + (long)(k++ & 3) * Q; };
  const dim3 g(2048), b(256);
  read16_stream<<<g, b>>>((const f32x4*)buf(), out, Q / 16);
  read2x16_rows<<<g, b>>>((const f32x4*)buf(), out, Q / 1920);
  read8_stream<<<g, b>>>((const f32x2*)buf(), out, Q / 8);
  read4_stream<<<g, b>>>((const float*)buf(), out, Q / 4);
  dma16_stream<<<g, b>>>((const f32x4*)buf(), out, Q / 16);
  read_slices<<<dim3(512, 6), b>>>(buf(), out, Q / 2880);  // reads 960 of every 2880 bytes: Q / 3 bytes
  write16_stream<<<g, b>>>((f32x4*)buf(), Q / 16);
  write2x16_rows<<<g, b>>>((f32x4*)buf(), Q / 1920);
  write_slices<<<dim3(512, 6), b>>>(buf(), Q / 960);
  atomic_rows<<<g, b>>>((float*)buf(), Q / 16);  // Q / 4 bytes of adds
  hipDeviceSynchronize();
  printf("bytes per launch: read16 %ld read2x16 %ld read8 %ld read4 %ld dma16 %ld read_slices %ld write16 %ld write2x16 %ld write_slices %ld atomic %ld\n", Q, (Q / 1920) * 1920, Q, Q,
         Q, (Q / 2880) * 960, Q, (Q / 1920) * 1920, (Q / 960) * 960, Q / 4);
  return 0;
}
