// Stand-in for RCCL's ring all-reduce kernel on a box with ONE GPU (tools/comm_interference.py).  What matters for the step beside
// it is not what the collective computes but what it OCCUPIES: a handful of long-lived workgroups ("channels"), each holding a CU's wave slots, some LDS and
// registers for as long as the links need to move the bucket, streaming the gradient slice out of HBM and writing what arrives from the neighbour with
// system-scope stores.  This kernel does that and nothing else: `nwg` workgroups of 256 lanes copy `bytes` bytes (wrapping over the source as often as
// needed - a ring all-reduce moves 2 (N-1)/N of the message through every rank) into a scratch buffer, 16 bytes per lane, stores `sc0 sc1`, paced so the
// whole copy takes bytes / rate like the links would.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/libcomm_standin.so tools/comm_standin.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ void store_sys(u32x4* p, u32x4 v) {  // a store a peer must see: write through to system scope
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}

constexpr int UNROLL = 16;                      // 16-byte loads in flight per lane (64 KiB per workgroup: what it takes for 8 workgroups to reach link rate)
constexpr long CHUNK16 = 256L * UNROLL;         // 16-byte units a workgroup moves between two looks at the clock (64 KiB)

__global__ __launch_bounds__(256) void ring_copy_kernel(u32x4* __restrict__ dst, const u32x4* __restrict__ src, long src16, long total16, long ticks_per_chunk_x256) {
  extern __shared__ char lds[];  // only held, like a channel's staging space
  const long per = ((total16 + gridDim.x - 1) / gridDim.x + CHUNK16 - 1) / CHUNK16 * CHUNK16;
  const long lo = blockIdx.x * per, hi = lo + per < total16 ? lo + per : total16;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();  // constant 100 MHz
  long nchunk = 0;
  for (long base = lo; base < hi; base += CHUNK16, ++nchunk) {
    u32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const long i = base + u * 256 + threadIdx.x;
      v[u] = i < hi ? src[i % src16] : u32x4{0, 0, 0, 0};
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const long i = base + u * 256 + threadIdx.x;
      if (i < hi) store_sys(dst + i, v[u]);
    }
    if (ticks_per_chunk_x256 > 0) {  // the links' pace: chunk n may not start before t0 + n * ticks
      const unsigned long long due = t0 + (unsigned long long)(((nchunk + 1) * ticks_per_chunk_x256) >> 8);
      while (__builtin_amdgcn_s_memrealtime() < due) __builtin_amdgcn_s_sleep(8);
    }
  }
  if (lds[threadIdx.x] == 77 && total16 < 0) dst[0] = u32x4{1, 1, 1, 1};  // (keeps the LDS allocation referenced)
}

// dst: scratch of at least `bytes`; src: `src_bytes` of gradients; both 16-byte aligned, sizes multiples of 16.  gbps <= 0: unpaced.
extern "C" int comm_standin_copy(void* dst, const void* src, long src_bytes, long bytes, int nwg, int lds_bytes, double gbps, void* stream) {
  if (!dst || !src || src_bytes < 16 || bytes < 16 || nwg < 1 || nwg > 64 || lds_bytes < 0 || lds_bytes > 64 * 1024) return 1;
  const long total16 = bytes / 16;
  long ticks_x256 = 0;
  if (gbps > 0) {  // a workgroup moves CHUNK16 * 16 bytes per chunk at gbps / nwg: seconds = bytes * nwg / (gbps 1e9); ticks at 100 MHz, 8 fractional bits
    const double sec = (double)CHUNK16 * 16.0 * nwg / (gbps * 1e9);
    ticks_x256 = (long)(sec * 100e6 * 256.0);
  }
  static bool attr = false;
  if (!attr && lds_bytes > 48 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&ring_copy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) != hipSuccess) return 2;
    attr = true;
  }
  hipLaunchKernelGGL(ring_copy_kernel, dim3(nwg), dim3(256), (size_t)lds_bytes, (hipStream_t)stream, (u32x4*)dst, (const u32x4*)src, src_bytes / 16, total16, ticks_x256);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}
