// Box calibration for bench.py (not on the hot path): what THIS card, at its clocks today, does on (a) a loop of nothing but bf16 MFMAs on random
// operands and (b) a 16-byte-per-lane streaming copy.  Box classes of the pool differ by 2-4 % in step rate (docs/history_r01-r04.md section 5); the two figures go
// into the bench line next to the step rate so that rates measured on different boxes can be compared.
#include "v4h_common.h"
#include "v4h_ops.h"

namespace v4h {
namespace {

// One wave per SIMD slot, NACC independent accumulator tiles per wave (an MFMA 16x16x32 bf16 has 8 passes of latency: independent tiles keep the pipe
// full), operands random bf16 held in registers - operand data decides the power the matrix pipe draws and with it the clock it gets.
constexpr int CAL_NACC = 8;
__global__ __launch_bounds__(256) void calib_mfma_kernel(const bf16* __restrict__ rnd, float* __restrict__ sink, int iters) {
  const int lane = threadIdx.x & 63;
  Frag<bf16> a[2], b[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    a[k].v = *reinterpret_cast<const bf16x8*>(rnd + ((size_t)(blockIdx.x * 256 + threadIdx.x) * 4 + k) * 8);
    b[k].v = *reinterpret_cast<const bf16x8*>(rnd + ((size_t)(blockIdx.x * 256 + threadIdx.x) * 4 + 2 + k) * 8);
  }
  f32x4 acc[CAL_NACC];
#pragma unroll
  for (int n = 0; n < CAL_NACC; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
  // in-place accumulation spelled out: left to itself the register allocator rotates the eight tiles through 56 accumulator moves per iteration
#define V4H_CAL_MFMA(n, A, B) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[n]) : "v"(A.v), "v"(B.v))
  for (int it = 0; it < iters; ++it) {
    V4H_CAL_MFMA(0, a[0], b[0]); V4H_CAL_MFMA(1, a[1], b[0]); V4H_CAL_MFMA(2, a[0], b[1]); V4H_CAL_MFMA(3, a[1], b[1]);
    V4H_CAL_MFMA(4, a[0], b[0]); V4H_CAL_MFMA(5, a[1], b[0]); V4H_CAL_MFMA(6, a[0], b[1]); V4H_CAL_MFMA(7, a[1], b[1]);
    V4H_CAL_MFMA(0, a[1], b[1]); V4H_CAL_MFMA(1, a[0], b[1]); V4H_CAL_MFMA(2, a[1], b[0]); V4H_CAL_MFMA(3, a[0], b[0]);
    V4H_CAL_MFMA(4, a[1], b[1]); V4H_CAL_MFMA(5, a[0], b[1]); V4H_CAL_MFMA(6, a[1], b[0]); V4H_CAL_MFMA(7, a[0], b[0]);
  }
#undef V4H_CAL_MFMA
  // the compiler inserts no wait states between an inline-assembly MFMA and a visible read of its result: the wait states, tied to the tiles by data
  asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]));
  f32x4 s = acc[0];
#pragma unroll
  for (int n = 1; n < CAL_NACC; ++n) s += acc[n];
  if (s[0] + s[1] + s[2] + s[3] == 12345.678f) sink[lane] = s[0];  // keeps the loop alive; never true for accumulators of random products
}

__global__ __launch_bounds__(256) void calib_copy_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, long n4) {
  const long stride = (long)gridDim.x * blockDim.x;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {  // four independent 16-byte loads in flight per lane
    const f32x4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
    dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
  }
  for (; i < n4; i += stride) dst[i] = src[i];
}

}  // namespace

int calib_mfma_loop(const void* rnd, float* sink, int iters, int blocks, hipStream_t s) {
  hipLaunchKernelGGL(calib_mfma_kernel, dim3(blocks), dim3(256), 0, s, (const bf16*)rnd, sink, iters);
  V4H_CHECK_LAUNCH("calib_mfma");
  return V4H_OK;
}
int calib_copy(const void* src, void* dst, long bytes, hipStream_t s) {
  hipLaunchKernelGGL(calib_copy_kernel, dim3(2048), dim3(256), 0, s, (const f32x4*)src, (f32x4*)dst, bytes / 16);
  V4H_CHECK_LAUNCH("calib_copy");
  return V4H_OK;
}

}  // namespace v4h
