// Batch-row contractions of the conditioning path: Out[i][j] = sum_k P[i][k] Q[j][k] with only B rows (the t / c embedder MLPs nn/vit.py:77-81,361-365
// forward, and their input gradients): 128 x 480 x 480 is 59 MFLOP, yet the tiled kernel of v4h_gemm.h needs 12-25 us for it - three workgroups walking
// eight K-steps, each a global->LDS round trip (profiles/r03_step_timeline.txt: the forward waits 80 us for this chain before its first block, the
// backward ends with it).  Here the WHOLE K extent of a 64 x 32 output tile is requested at once (at most 96 KB of LDS, one round trip), then one
// barrier, K / 32 slabs of two MFMAs per wave, the fused epilogue of v4h_gemm.h on 8-column chunks: 30 workgroups for a 128 x 480 output.
#pragma once
#include "v4h_gemm.h"

namespace v4h_small {

constexpr int SM_BI = 64, SM_BJ = 32, SM_KMAX = 512, SM_NT = 256;

// The same idea for the weight gradients of those Linears, dW[i][j] += sum_t dY[t][i] X[t][j] over only B tokens (both operands K-strided, whole K in LDS,
// no K split: every output element has ONE writer, so the accumulation is a plain read-modify-write; bias gradient = column sums of dY by the j-tile 0).
__global__ __launch_bounds__(SM_NT) void v4h_smallk_wgrad_kernel(const GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [P image: K x 64 | Q image: K x 32]
  const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntj = a.J / SM_BJ;
  const int ti = blockIdx.x / ntj, tj = blockIdx.x - ti * ntj;
  const int i0 = ti * SM_BI, j0 = tj * SM_BJ, K = a.K;
  const bf16* gP = reinterpret_cast<const bf16*>(a.P);
  const bf16* gQ = reinterpret_cast<const bf16*>(a.Q);
  char* iP = smem;
  char* iQ = smem + K * SM_BI * 2;
  for (int u0 = wave * 64; u0 < K * 8; u0 += SM_NT) {  // P[k][i0 .. i0 + 63]: 8 chunks per row; columns beyond I from the zero page
    const int u = u0 + lane, r = u >> 3, ch = u & 7;
    const void* src = (i0 + ch * 8 + 8 <= a.I) ? (const void*)(gP + (size_t)r * a.ldp + i0 + ch * 8) : (const void*)v4h_zero_page;
    dma16(src, iP + u0 * 16);
  }
  for (int u0 = wave * 64; u0 < K * 4; u0 += SM_NT) {
    const int u = u0 + lane, r = u >> 2, ch = u & 3;
    dma16(gQ + (size_t)r * a.ldq + j0 + ch * 8, iQ + u0 * 16);
  }
  __syncthreads();
  const bf16* sP = reinterpret_cast<const bf16*>(iP);
  const bf16* sQ = reinterpret_cast<const bf16*>(iQ);
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
  float cs = 0.f;
  for (int k0 = 0; k0 < K; k0 += 32) {
    const Frag<bf16> pf = frag_kstrided(sP, SM_BI, k0, wave * 16, lane);
    acc0 = mma(frag_kstrided(sQ, SM_BJ, k0, 0, lane), pf, acc0);
    acc1 = mma(frag_kstrided(sQ, SM_BJ, k0, 16, lane), pf, acc1);
    if (tj == 0) {
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) cs += (float)pf.v[jj];
    }
  }
  acc0 *= 1.0f;
  acc1 *= 1.0f;
  asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc0), "+v"(acc1));
  const f32x8 v = swap_pair(acc0, acc1);
  const int i = i0 + wave * 16 + c, j = j0 + (g & 1) * 16 + (g >> 1) * 8;
  if (i < a.I) {
    float* o = reinterpret_cast<float*>(a.e.out) + (size_t)i * a.e.ldo + j;
    store8(o, add8(load8(o), v));
  }
  if (tj == 0 && a.colsum != nullptr) {
    cs += __shfl_xor(cs, 16, 64);
    cs += __shfl_xor(cs, 32, 64);
    if (g == 0 && i < a.I) a.colsum[i] += cs;
  }
}

template <bool QKS, int EPI> __global__ __launch_bounds__(SM_NT) void v4h_smallm_kernel(const GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [P image: 64 rows x K | Q image: 32 x K (K-contiguous) or K x 32 (K-strided)]
  const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntj = a.J / SM_BJ;
  const int ti = blockIdx.x / ntj, tj = blockIdx.x - ti * ntj;
  const int i0 = ti * SM_BI, j0 = tj * SM_BJ, K = a.K, cpr = K / 8;  // 16-byte chunks per K-contiguous row
  const bf16* gP = reinterpret_cast<const bf16*>(a.P);
  const bf16* gQ = reinterpret_cast<const bf16*>(a.Q);
  char* iP = smem;
  char* iQ = smem + SM_BI * K * 2;
  // ---- request everything: dense images in DMA order (unit u = 16 bytes; rows beyond I read row 0: their results are never stored)
  {
    const int unitsP = SM_BI * cpr;
    for (int u0 = wave * 64; u0 < unitsP; u0 += SM_NT) {
      const int u = u0 + lane, r = u / cpr, ch = u - r * cpr, gi = i0 + r;
      const void* src = u < unitsP ? (const void*)(gP + (size_t)(gi < a.I ? gi : 0) * a.ldp + ch * 8) : (const void*)v4h_zero_page;
      dma16(src, iP + u0 * 16);
    }
    if constexpr (!QKS) {
      const int unitsQ = SM_BJ * cpr;
      for (int u0 = wave * 64; u0 < unitsQ; u0 += SM_NT) {
        const int u = u0 + lane, r = u / cpr, ch = u - r * cpr;
        const void* src = u < unitsQ ? (const void*)(gQ + (size_t)(j0 + r) * a.ldq + ch * 8) : (const void*)v4h_zero_page;
        dma16(src, iQ + u0 * 16);
      }
    } else {  // Q[k][j]: image rows = k, 4 chunks (32 columns) per row
      const int unitsQ = K * 4;
      for (int u0 = wave * 64; u0 < unitsQ; u0 += SM_NT) {
        const int u = u0 + lane, r = u >> 2, ch = u & 3;
        const void* src = u < unitsQ ? (const void*)(gQ + (size_t)r * a.ldq + j0 + ch * 8) : (const void*)v4h_zero_page;
        dma16(src, iQ + u0 * 16);
      }
    }
  }
  __syncthreads();  // (vmcnt(0) + barrier)
  // ---- wave w: rows i0 + 16 w .. + 15, all 32 columns
  const bf16* sP = reinterpret_cast<const bf16*>(iP);
  const bf16* sQ = reinterpret_cast<const bf16*>(iQ);
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
  for (int k0 = 0; k0 < K; k0 += 32) {
    const Frag<bf16> pf = frag_kcontig(sP, K, wave * 16, k0, lane);
    Frag<bf16> q0, q1;
    if constexpr (!QKS) {
      q0 = frag_kcontig(sQ, K, 0, k0, lane);
      q1 = frag_kcontig(sQ, K, 16, k0, lane);
    } else {
      q0 = frag_kstrided(sQ, SM_BJ, k0, 0, lane);
      q1 = frag_kstrided(sQ, SM_BJ, k0, 16, lane);
    }
    acc0 = mma(q0, pf, acc0);
    acc1 = mma(q1, pf, acc1);
  }
  // ---- epilogue: lanes with even g hold 8 consecutive columns of tile 0, odd g of tile 1 (v_permlane16_swap).  The exchange is inline assembly,
  // which gets no wait states after an MFMA: pass the results through a vector instruction the compiler sees first.
  acc0 *= 1.0f;
  acc1 *= 1.0f;
  asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc0), "+v"(acc1));
  f32x8 v = swap_pair(acc0, acc1);
  const int i = i0 + wave * 16 + c, j = j0 + (g & 1) * 16 + (g >> 1) * 8;
  using Epi = Epilogue<EPI, bf16, bf16>;
  if (i < a.I) {
    if constexpr (Epi::HAS_BIAS) {
      if (a.e.bias != nullptr) v = add8(v, load8(a.e.bias + j));
    }
    const typename Epi::Ops ops = Epi::load(a.e, i, j);
    Epi::finish(a.e, i, j, v, ops);
  }
}

inline bool smallm_eligible(const GemmArgs& a) {
  return a.I <= 512 && a.J % SM_BJ == 0 && a.K % 32 == 0 && a.K <= SM_KMAX && a.ldp % 8 == 0 && a.ldq % 8 == 0 && ((uintptr_t)a.P % 16) == 0 &&
         ((uintptr_t)a.Q % 16) == 0;
}
template <bool QKS, int EPI> int smallm_launch(const GemmArgs& a, hipStream_t s, const char* name) {
  const size_t lds = (size_t)(SM_BI + SM_BJ) * a.K * 2;
  static DeviceOnce lds_attr;
  if (int rc = lds_attr.ensure([&]() -> hipError_t {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&v4h_smallm_kernel<QKS, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, (SM_BI + SM_BJ) * SM_KMAX * 2);
      }, name, "reserve the LDS of the tile")) return rc;
  const int grid = ((a.I + SM_BI - 1) / SM_BI) * (a.J / SM_BJ);
  hipLaunchKernelGGL((v4h_smallm_kernel<QKS, EPI>), dim3(grid), dim3(SM_NT), lds, s, a);
  V4H_CHECK_LAUNCH(name);
  return V4H_OK;
}
inline bool smallk_wgrad_eligible(const GemmArgs& a) {
  return a.K <= SM_KMAX && a.K % 32 == 0 && a.J % SM_BJ == 0 && a.I % 8 == 0 && a.ldp % 8 == 0 && a.ldq % 8 == 0 && a.e.ldo % 4 == 0 && a.e.group_rows == 0 &&
         ((uintptr_t)a.P % 16) == 0 && ((uintptr_t)a.Q % 16) == 0 && ((uintptr_t)a.e.out % 16) == 0;
}
inline int smallk_wgrad_launch(const GemmArgs& a, hipStream_t s) {
  static DeviceOnce lds_attr;
  if (int rc = lds_attr.ensure([&]() -> hipError_t {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&v4h_smallk_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (SM_BI + SM_BJ) * SM_KMAX * 2);
      }, "gemm_small/wgrad", "reserve the LDS of the tile")) return rc;
  const int grid = ((a.I + SM_BI - 1) / SM_BI) * (a.J / SM_BJ);
  hipLaunchKernelGGL(v4h_smallk_wgrad_kernel, dim3(grid), dim3(SM_NT), (size_t)(SM_BI + SM_BJ) * a.K * 2, s, a);
  V4H_CHECK_LAUNCH("gemm_small/wgrad");
  return V4H_OK;
}
// (explicit instantiations: see v4h_attention_dense.h - host stubs of kernel templates reached only through a launcher template can go missing)
#define V4H_SMALLM(QKS, EPI) template __global__ void v4h_smallm_kernel<QKS, EPI>(const GemmArgs);
V4H_SMALLM(false, EPI_SILU) V4H_SMALLM(false, EPI_COND_SUM) V4H_SMALLM(true, EPI_DSILU)
#undef V4H_SMALLM

}  // namespace v4h_small
