// Multi-head self-attention core for the DiT blocks: softmax(q k^T / sqrt(dh)) v, no mask, no dropout
// (reference nn/vit.py:425-451, both the SDPA and the xformers branch), forward and backward.
//
// Data stays token-major exactly as the reference's qkv Linear produces it: qkv[b*T + t][s*D + h*dh + d]
// (s = 0,1,2 for q,k,v; the reshape (B,N,3,H,dh) of nn/vit.py:427), o[b*T + t][h*dh + d] (nn/vit.py:451).
//
// One workgroup = 4 waves works on one (batch, head); each wave owns 16-row tiles of the "lane side" sequence
// (queries for forward / dQ, keys for dK/dV) and streams the other sequence through LDS in chunks of 160 rows
// with an online softmax, so T = 135 (ds2) is a single chunk and T = 450 (ds3) three.  Scores are produced
// with the streamed sequence on the accumulator rows, so a lane holds the scores of ONE lane-side row: the row
// reductions are in-lane plus two wavefront shuffles (xor 16, 32), and the probability accumulators are re-used
// directly as the lane-side operand of the next product (frag_from_acc), its other operand being read
// transposed from the row-major LDS image (ds_read_b64_tr_b16 in bf16 mode).  head_dim 80 is padded to 96 in
// the contraction by zeroing fragment lanes, never in memory.
#include "v4h_common.h"
#include "v4h_gemm.h"  // TileStage
#include <stdlib.h>

#include "v4h_ops.h"

#ifdef V4H_ATTN_STAMPS
// Diagnostic build only (V4H_EXTRA_FLAGS=-DV4H_ATTN_STAMPS): wave 0 of every workgroup stamps the constant 100 MHz clock at the phase boundaries of each item
// into a buffer no kernel reads; v4h_debug_attn_stamps copies it out (tools/experiments/attn_stamps.py).
__device__ unsigned long long v4h_attn_stamps[256 * 4 * 8];
__device__ unsigned long long v4h_attn_arrive[256 * 4 * 16];  // [workgroup][item][wave]: arrival at the top-of-item barrier
#define V4H_STAMP(n, k)                                                                                        \
  do {                                                                                                         \
    if (threadIdx.x == 0 && (n) < 4) v4h_attn_stamps[(blockIdx.x * 4 + (n)) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
extern "C" int v4h_debug_attn_arrivals(void* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(v4h_attn_arrive), sizeof(v4h_attn_arrive)) == hipSuccess ? 0 : 1;
}
extern "C" int v4h_debug_attn_stamps(void* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(v4h_attn_stamps), sizeof(v4h_attn_stamps)) == hipSuccess ? 0 : 1;
}
#else
#define V4H_STAMP(n, k) do {} while (0)
#endif

namespace {

// Row store of NDT 16-column accumulator tiles (lane (c, g): row c, columns 16 dt + 4g .. + 3): pairs of tiles are exchanged between the
// 16-lane groups (v_permlane16_swap) so that a lane writes 8 consecutive columns - 16 bytes of bf16 - instead of 8-byte pieces.
// `rowp` = start of the lane's row for EVEN g and for odd g alike (the swap moves tiles, not rows).  EXEC must be full (the swap crosses lanes).
template <typename T, int NDT> V4H_DEV void store_row_tiles(T* rowp, const f32x4* t, int g, bool ok, float mul = 1.0f) {
  const int ge = g & 1, gh = g >> 1;
#pragma unroll
  for (int d = 0; d + 1 < NDT; d += 2) {
    f32x8 v = swap_pair(t[d], t[d + 1]);
#pragma unroll
    for (int r = 0; r < 8; ++r) v.v[r] *= mul;
    if (ok) store8(rowp + (d + ge) * 16 + 8 * gh, v);
  }
  if constexpr (NDT & 1) {
    if (ok) store4(rowp + (NDT - 1) * 16 + 4 * g, t[NDT - 1] * mul);
  }
}

constexpr int KC = 160;  // streamed rows per LDS chunk (multiple of 32)

template <typename T, int DH> struct AttnCfg {
  static constexpr int PAD = 16 / (int)sizeof(T);
  static constexpr int LD = DH + PAD;
  static constexpr int NKF = (DH + 31) / 32;  // 32-wide contraction slabs over head_dim
  static constexpr int NDT = DH / 16;         // 16-wide output tiles over head_dim
  static constexpr int NJT = KC / 16;
  static constexpr int TILE_ELEMS = KC * LD;
  static_assert(DH % 16 == 0, "head_dim must be a multiple of 16");
};

// lane-side fragments of a 16-row tile, straight from global memory (rows >= rows_end and d >= DH read as zero)
template <typename T, int DH> V4H_DEV void load_row_frags(Frag<T>* f, const T* base, int ld, int row0, int rows_end, int lane) {
  const int g = lane >> 4, c = lane & 15;
  const int row = row0 + c;
#pragma unroll
  for (int s = 0; s < AttnCfg<T, DH>::NKF; ++s) {
    const int d = 32 * s + 8 * g;
    f[s] = frag_zero<T>();
    if (row < rows_end && d + 8 <= DH) {
      const T* p = base + (size_t)row * ld + d;
      if constexpr (sizeof(T) == 2) {
        f[s].v = *reinterpret_cast<const bf16x8*>(p);
      } else {
        const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
        f[s].v[0] = a.x; f[s].v[1] = a.y; f[s].v[2] = a.z; f[s].v[3] = a.w;
        f[s].v[4] = b.x; f[s].v[5] = b.y; f[s].v[6] = b.z; f[s].v[7] = b.w;
      }
    }
  }
}

// regs-side fragment of streamed rows jt*16.. from the LDS tile, slab s of head_dim, zero beyond DH
template <typename T, int DH> V4H_DEV Frag<T> tile_frag(const T* tile, int jt, int s, int lane) {
  Frag<T> f = frag_kcontig(tile, AttnCfg<T, DH>::LD, jt * 16, 32 * s, lane);
  if (32 * s + 8 * (lane >> 4) + 8 > DH) f = frag_zero<T>();
  return f;
}

// acc[jt] (lane: lane-side row c ; regs: streamed row jt*16 + 4g + r) = sum_d X[c][d] * Y[jt*16+4g+r][d]
template <typename T, int DH> V4H_DEV f32x4 score_tile(const T* tile, int jt, const Frag<T>* x, int lane) {
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < AttnCfg<T, DH>::NKF; ++s) a = mma(tile_frag<T, DH>(tile, jt, s, lane), x[s], a);
  return a;
}

// out[dt] (lane: lane-side row c ; regs: d = dt*16 + 4g + r) += sum_rows W[c][row] * Z[row][d] over the chunk,
// W given as NJT accumulator tiles, Z the row-major LDS tile read transposed.
template <typename T, int DH> V4H_DEV void accumulate_wz(f32x4* out, const f32x4* w, const T* ztile, int lane) {
  using C = AttnCfg<T, DH>;
#pragma unroll
  for (int ks = 0; ks < C::NJT / 2; ++ks) {
    const Frag<T> wf = frag_from_acc(w[2 * ks], w[2 * ks + 1], T());
#pragma unroll
    for (int dt = 0; dt < C::NDT; ++dt) {
      const Frag<T> zf = frag_kstrided2(ztile, C::LD, 32 * ks, 32 * ks + 16, dt * 16, lane);
      out[dt] = mma(zf, wf, out[dt]);
    }
  }
}

// one 32-row step of the above: W given as the two accumulator tiles 2*ks, 2*ks+1
template <typename T, int DH> V4H_DEV void accumulate_wz_step(f32x4* out, f32x4 w0, f32x4 w1, const T* ztile, int ks, int lane) {
  using C = AttnCfg<T, DH>;
  const Frag<T> wf = frag_from_acc(w0, w1, T());
#pragma unroll
  for (int dt = 0; dt < C::NDT; ++dt) {
    const Frag<T> zf = frag_kstrided2(ztile, C::LD, 32 * ks, 32 * ks + 16, dt * 16, lane);
    out[dt] = mma(zf, wf, out[dt]);
  }
}

// ------------------------------------------------------------------------------------------------- forward
template <typename T, int DH, int NW> __global__ __launch_bounds__(64 * NW) void attn_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ o, float* __restrict__ lse,
                                                                                      int Tn, int H, float scale) {
  using C = AttnCfg<T, DH>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* sK = reinterpret_cast<T*>(smem);
  T* sV = sK + C::TILE_ELEMS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int D = H * DH, ld = 3 * D;
  const T* base = qkv + (size_t)b * Tn * ld + h * DH;
  const int ntiles = (Tn + 15) / 16, nchunks = (Tn + KC - 1) / KC;
  const int nrounds = (ntiles + NW * gridDim.y - 1) / (NW * gridDim.y);

  for (int rd = 0; rd < nrounds; ++rd) {
    const int qt = (rd * gridDim.y + blockIdx.y) * NW + wave;
    const bool active = qt < ntiles;  // wave-uniform
    Frag<T> xq[C::NKF];
    load_row_frags<T, DH>(xq, base, ld, qt * 16, active ? Tn : 0, lane);
    float m = -INFINITY, l = 0.f;
    f32x4 oacc[C::NDT];
#pragma unroll
    for (int dt = 0; dt < C::NDT; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int ch = 0; ch < nchunks; ++ch) {
      if (!(nchunks == 1 && rd > 0)) {
        __syncthreads();
        TileStage<T, KC, DH, C::LD, 64 * NW> st;
        st.load(base + D, ld, ch * KC, 0, Tn, DH, tid);
        st.store(sK, tid);
        st.load(base + 2 * D, ld, ch * KC, 0, Tn, DH, tid);
        st.store(sV, tid);
        __syncthreads();
      }
      if (active) {
        f32x4 p[C::NJT];
        float mx = -INFINITY;
#pragma unroll
        for (int jt = 0; jt < C::NJT; ++jt) {
          p[jt] = score_tile<T, DH>(sK, jt, xq, lane);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = ch * KC + jt * 16 + 4 * g + r;
            p[jt][r] = key < Tn ? p[jt][r] * scale : -INFINITY;
            mx = fmaxf(mx, p[jt][r]);
          }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mn = fmaxf(m, mx);
        const float alpha = __expf(m - mn);
        float rs = 0.f;
#pragma unroll
        for (int jt = 0; jt < C::NJT; ++jt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            p[jt][r] = __expf(p[jt][r] - mn);
            rs += p[jt][r];
          }
        rs += __shfl_xor(rs, 16, 64);
        rs += __shfl_xor(rs, 32, 64);
        l = l * alpha + rs;
        m = mn;
#pragma unroll
        for (int dt = 0; dt < C::NDT; ++dt) oacc[dt] *= alpha;
        accumulate_wz<T, DH>(oacc, p, sV, lane);
      }
    }
    const int q = qt * 16 + c;
    if (active && q < Tn) {
      const float inv = 1.0f / l;
      T* orow = o + ((size_t)b * Tn + q) * D + h * DH;
#pragma unroll
      for (int dt = 0; dt < C::NDT; ++dt) store4(orow + dt * 16 + 4 * g, oacc[dt] * inv);
      if (g == 0 && lse) lse[((size_t)b * H + h) * Tn + q] = m + __logf(l);
    }
  }
}

// ------------------------------------------------------------------------------------------------- forward, persistent
// Single-chunk case (T <= 160, i.e. ds2): a persistent workgroup walks (batch, head) items id, id + grid, ... and DMAs the NEXT
// item's K and V (global_load_lds, dense 160-byte rows, rows >= T from the zero page) into the other LDS buffer while it
// computes the current one, so the load -> compute serialisation of the one-shot kernel disappears.  One barrier per item.
// (Round 2 built two more forms of this kernel and measured them inside the step: Q through the DMA as well + counted vmcnt waits + buffer stores
// - no store drain at the barrier - 21.8 -> 21.8 us per call; that plus a third of the vector instructions removed - K = 16 MFMA for the head_dim
// tail instead of masked K = 32 slabs, base-2 softmax with the scale folded in, DMA offsets computed once - 21.6 -> 20.0 us.  Neither the store
// drain nor the instruction count is what bounds it; both forms were dropped again.  What the kernel moves per call is 66 MB in 160-byte row
// slices of 2880-byte rows: 3.0 TB/s.)
V4H_DEV bool dma_duty_balanced() {
#ifdef V4H_ATTN_DMA_ALL_WAVES
  return false;
#else
  return true;
#endif
}
template <typename T, int DH> struct AttnDense {
  static constexpr int CPRD = DH * (int)sizeof(T) / 16;        // 16-byte chunks per row
  static constexpr int UNITS = KC * CPRD, NI = (UNITS + 63) / 64;
  static constexpr int BYTES = NI * 1024;                       // whole DMA instructions
  // stage rows [0, KC) x DH of a token-major tensor (row stride ld elements) into a dense image
  // `wave` of `nw`: which DMA instructions this wave issues.  With nine waves (3 + 2 + 2 + 2 on the four SIMDs) the three waves of SIMD 0 set the pace of an item
  // (profiles/r02_attn_fwd_timeline.txt), so the DMA duty (about 1 us per wave and item: address arithmetic + 60-185 cycles of issue per piece) is given to the six
  // waves of the other SIMDs only.
  static V4H_DEV void stage(char* img, const T* base, int ld, int rows_end, int wave, int nw, int lane) {
    if (nw == 9 && dma_duty_balanced()) {
      if ((wave & 3) == 0) return;
      wave = (wave >> 2) * 3 + (wave & 3) - 1;
      nw = 6;
    }
    for (int inst = wave; inst < NI; inst += nw) {
      const int u = inst * 64 + lane, row = u / CPRD, ch = u % CPRD;
      const void* src = (u < UNITS && row < rows_end) ? (const void*)(base + (size_t)row * ld + ch * (16 / (int)sizeof(T))) : (const void*)v4h_zero_page;
      dma16(src, img + inst * 1024);
    }
  }
};

// Persistent (batch, head) walk, XCD-aware.  Workgroups are dealt round-robin over the 8 XCDs, each with a private L2, and the H heads of one
// sample share every 128-byte line of a qkv row (a head's slice is DH * 2 = 160 bytes of a 3 * D * 2 = 2880-byte row).  The walk therefore
// gives XCD x the samples b = x (mod 8) and lets consecutive workgroups of an XCD take consecutive heads of the same sample at the same
// time, so each line is fetched into ONE L2 once instead of into up to H of them (measured before: 80 MB fetched per call for 50 MB of qkv).
// Speed only: any placement computes the same thing.  n-th item of workgroup `wg` of `nwg` (a multiple of 8), or -1.
V4H_DEV int attn_item(int wg, int nwg, int n, int B, int H) {
  const int xcd = wg & 7, slot = wg >> 3, per = nwg >> 3;        // per: workgroups per XCD
  const int nb = (B - xcd + 7) >> 3;                             // samples of this XCD: xcd, xcd + 8, ...
  const int k = slot + n * per;                                  // index into this XCD's (sample, head) list
  if (k >= nb * H) return -1;
  return (xcd + 8 * (k / H)) * H + k % H;
}

template <typename T, int DH, int NW> __global__ __launch_bounds__(64 * NW) void attn_fwd_persist_kernel(const T* __restrict__ qkv, T* __restrict__ o,
                                                                                                     float* __restrict__ lse, int Tn, int H, int nitems,
                                                                                                     float scale) {
  using C = AttnCfg<T, DH>;
  using DI = AttnDense<T, DH>;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [K0 | V0 | K1 | V1]
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int D = H * DH, ld = 3 * D;
  const int ntiles = (Tn + 15) / 16;
  const bool active = wave < ntiles;
  auto item_base = [&](int it) { return qkv + (size_t)(it / H) * Tn * ld + (it % H) * DH; };
  auto stage_item = [&](int it, int buf) {
    const T* base = item_base(it);
    DI::stage(smem + buf * 2 * DI::BYTES, base + D, ld, Tn, wave, NW, lane);
    DI::stage(smem + buf * 2 * DI::BYTES + DI::BYTES, base + 2 * D, ld, Tn, wave, NW, lane);
  };
  const int Bn = nitems / H;
  int it = attn_item(blockIdx.x, gridDim.x, 0, Bn, H);
  Frag<T> xq[C::NKF], xq_next[C::NKF];  // this wave's 16 query rows: current item, and the next one (fetched a whole item ahead)
  if (it >= 0) {
    stage_item(it, 0);
    load_row_frags<T, DH>(xq_next, item_base(it), ld, wave * 16, active ? Tn : 0, lane);
  }
  for (int n = 0; it >= 0; ++n) {
    const int buf = n & 1;
    const int it_next = attn_item(blockIdx.x, gridDim.x, n + 1, Bn, H);
#pragma unroll
    for (int s2 = 0; s2 < C::NKF; ++s2) xq[s2] = xq_next[s2];
    V4H_STAMP(n, 0);
#ifdef V4H_ATTN_STAMPS
    if ((threadIdx.x & 63) == 0 && n < 4) v4h_attn_arrive[(blockIdx.x * 4 + n) * 16 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    V4H_STAMP(n, 7);
#endif
    __syncthreads();  // this item's K/V landed (vmcnt(0)); everyone is done with the buffer the next DMA overwrites
    V4H_STAMP(n, 1);
    if (it_next >= 0) {
      stage_item(it_next, buf ^ 1);
      load_row_frags<T, DH>(xq_next, item_base(it_next), ld, wave * 16, active ? Tn : 0, lane);
    }
    V4H_STAMP(n, 2);
    if (active) {
      const T* sK = reinterpret_cast<const T*>(smem + buf * 2 * DI::BYTES);
      const T* sV = reinterpret_cast<const T*>(smem + buf * 2 * DI::BYTES + DI::BYTES);
      const int b = it / H, h = it % H;
      f32x4 p[C::NJT];
      float mx = -INFINITY;
#pragma unroll
      for (int jt = 0; jt < C::NJT; ++jt) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s2 = 0; s2 < C::NKF; ++s2) {
          Frag<T> kf = frag_kcontig(sK, DH, jt * 16, 32 * s2, lane);
          if (32 * s2 + 8 * g + 8 > DH) kf = frag_zero<T>();
          a = mma(kf, xq[s2], a);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = jt * 16 + 4 * g + r;
          a[r] = key < Tn ? a[r] * scale : -INFINITY;
          mx = fmaxf(mx, a[r]);
        }
        p[jt] = a;
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      V4H_STAMP(n, 3);
      float rs = 0.f;
#pragma unroll
      for (int jt = 0; jt < C::NJT; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          p[jt][r] = __expf(p[jt][r] - mx);
          rs += p[jt][r];
        }
      rs += __shfl_xor(rs, 16, 64);
      rs += __shfl_xor(rs, 32, 64);
      V4H_STAMP(n, 4);
      f32x4 oacc[C::NDT];
#pragma unroll
      for (int dt = 0; dt < C::NDT; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < C::NJT / 2; ++ks) {
        const Frag<T> wf = frag_from_acc(p[2 * ks], p[2 * ks + 1], T());
#pragma unroll
        for (int dt = 0; dt < C::NDT; ++dt) oacc[dt] = mma(frag_kstrided2(sV, DH, 32 * ks, 32 * ks + 16, dt * 16, lane), wf, oacc[dt]);
      }
      V4H_STAMP(n, 5);
      const int q = wave * 16 + c;
      {
        const float inv = 1.0f / rs;
        T* orow = o + ((size_t)b * Tn + q) * D + h * DH;
        if constexpr (sizeof(T) == 2) {
          store_row_tiles<T, C::NDT>(orow, oacc, g, q < Tn, inv);
        } else {
          if (q < Tn) {
#pragma unroll
            for (int dt = 0; dt < C::NDT; ++dt) store4(orow + dt * 16 + 4 * g, oacc[dt] * inv);
          }
        }
        if (q < Tn && g == 0 && lse) lse[((size_t)b * H + h) * Tn + q] = mx + __logf(rs);
      }
      V4H_STAMP(n, 6);
    }
    it = it_next;
  }
}



// ------------------------------------------------------------------------------------------------- backward
// dQ: lane side = queries, streamed = keys (K and V chunks in LDS)
template <typename T, int DH, int NW> __global__ __launch_bounds__(64 * NW) void attn_bwd_dq_kernel(const T* __restrict__ qkv, const T* __restrict__ o, const T* __restrict__ dout,
                                                                                         const float* __restrict__ lse, float* __restrict__ delta, T* __restrict__ dqkv, int Tn, int H,
                                                                                         float scale) {
  using C = AttnCfg<T, DH>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* sK = reinterpret_cast<T*>(smem);
  T* sV = sK + C::TILE_ELEMS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int D = H * DH, ld = 3 * D;
  const T* base = qkv + (size_t)b * Tn * ld + h * DH;
  const T* dobase = dout + (size_t)b * Tn * D + h * DH;
  const int ntiles = (Tn + 15) / 16, nchunks = (Tn + KC - 1) / KC;
  const int nrounds = (ntiles + NW * gridDim.y - 1) / (NW * gridDim.y);

  for (int rd = 0; rd < nrounds; ++rd) {
    const int qt = (rd * gridDim.y + blockIdx.y) * NW + wave;
    const bool active = qt < ntiles;
    const int q = qt * 16 + c;
    Frag<T> xq[C::NKF], xdo[C::NKF];
    load_row_frags<T, DH>(xq, base, ld, qt * 16, active ? Tn : 0, lane);
    load_row_frags<T, DH>(xdo, dobase, D, qt * 16, active ? Tn : 0, lane);
    // delta[q] = sum_d dO[q][d] * O[q][d]: the lane already holds its slice of the dO row; same slice of O, 2 shuffles
    float lse_q = 0.f, delta_q = 0.f;
    {
      Frag<T> xo[C::NKF];
      load_row_frags<T, DH>(xo, o + (size_t)b * Tn * D + h * DH, D, qt * 16, active ? Tn : 0, lane);
#pragma unroll
      for (int s2 = 0; s2 < C::NKF; ++s2)
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) delta_q += to_f32(xo[s2].v[jj]) * to_f32(xdo[s2].v[jj]);
      delta_q += __shfl_xor(delta_q, 16, 64);
      delta_q += __shfl_xor(delta_q, 32, 64);
    }
    if (active && q < Tn) {
      lse_q = lse[((size_t)b * H + h) * Tn + q];
      if (g == 0) delta[((size_t)b * H + h) * Tn + q] = delta_q;  // for the dK/dV pass
    }
    f32x4 dq[C::NDT];
#pragma unroll
    for (int dt = 0; dt < C::NDT; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int ch = 0; ch < nchunks; ++ch) {
      if (!(nchunks == 1 && rd > 0)) {
        __syncthreads();
        TileStage<T, KC, DH, C::LD, 64 * NW> st;
        st.load(base + D, ld, ch * KC, 0, Tn, DH, tid);
        st.store(sK, tid);
        st.load(base + 2 * D, ld, ch * KC, 0, Tn, DH, tid);
        st.store(sV, tid);
        __syncthreads();
      }
      if (active) {  // two key tiles at a time: scores -> dS -> straight into dQ (keeps the live set small)
#pragma unroll
        for (int ks = 0; ks < C::NJT / 2; ++ks) {
          f32x4 ds2[2];
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            const int jt = 2 * ks + hh;
            const f32x4 s = score_tile<T, DH>(sK, jt, xq, lane);
            const f32x4 dp = score_tile<T, DH>(sV, jt, xdo, lane);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int key = ch * KC + jt * 16 + 4 * g + r;
              const float p = (key < Tn && q < Tn) ? __expf(s[r] * scale - lse_q) : 0.f;
              ds2[hh][r] = p * (dp[r] - delta_q) * scale;
            }
          }
          accumulate_wz_step<T, DH>(dq, ds2[0], ds2[1], sK, ks, lane);
        }
      }
    }
    if (active && q < Tn) {
      T* row = dqkv + ((size_t)b * Tn + q) * ld + h * DH;
#pragma unroll
      for (int dt = 0; dt < C::NDT; ++dt) store4(row + dt * 16 + 4 * g, dq[dt]);
    }
  }
}

// dK, dV: lane side = keys, streamed = queries (Q and dO chunks + their lse / delta in LDS)
template <typename T, int DH, int NW> __global__ __launch_bounds__(64 * NW) void attn_bwd_dkv_kernel(const T* __restrict__ qkv, const T* __restrict__ dout, const float* __restrict__ lse,
                                                                                          const float* __restrict__ delta, T* __restrict__ dqkv, int Tn, int H, float scale) {
  using C = AttnCfg<T, DH>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* sQ = reinterpret_cast<T*>(smem);
  T* sDO = sQ + C::TILE_ELEMS;
  float* sLse = reinterpret_cast<float*>(sDO + C::TILE_ELEMS);
  float* sDelta = sLse + KC;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int D = H * DH, ld = 3 * D;
  const T* base = qkv + (size_t)b * Tn * ld + h * DH;
  const T* dobase = dout + (size_t)b * Tn * D + h * DH;
  const float* lse_bh = lse + ((size_t)b * H + h) * Tn;
  const float* delta_bh = delta + ((size_t)b * H + h) * Tn;
  const int ntiles = (Tn + 15) / 16, nchunks = (Tn + KC - 1) / KC;
  const int nrounds = (ntiles + NW * gridDim.y - 1) / (NW * gridDim.y);

  for (int rd = 0; rd < nrounds; ++rd) {
    const int kt = (rd * gridDim.y + blockIdx.y) * NW + wave;
    const bool active = kt < ntiles;
    const int key = kt * 16 + c;
    Frag<T> xk[C::NKF], xv[C::NKF];
    load_row_frags<T, DH>(xk, base + D, ld, kt * 16, active ? Tn : 0, lane);
    load_row_frags<T, DH>(xv, base + 2 * D, ld, kt * 16, active ? Tn : 0, lane);
    f32x4 dk[C::NDT], dv[C::NDT];
#pragma unroll
    for (int dt = 0; dt < C::NDT; ++dt) {
      dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
      dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int ch = 0; ch < nchunks; ++ch) {
      if (!(nchunks == 1 && rd > 0)) {
        __syncthreads();
        TileStage<T, KC, DH, C::LD, 64 * NW> st;
        st.load(base, ld, ch * KC, 0, Tn, DH, tid);
        st.store(sQ, tid);
        st.load(dobase, D, ch * KC, 0, Tn, DH, tid);
        st.store(sDO, tid);
        if (tid < KC) {
          const int qq = ch * KC + tid;
          sLse[tid] = qq < Tn ? lse_bh[qq] : 0.f;
          sDelta[tid] = qq < Tn ? delta_bh[qq] : 0.f;
        }
        __syncthreads();
      }
      if (active) {  // two query tiles at a time: P^T, dS^T -> straight into dV, dK
#pragma unroll
        for (int ks = 0; ks < C::NJT / 2; ++ks) {
          f32x4 pt2[2], dst2[2];
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            const int jt = 2 * ks + hh;
            const f32x4 s = score_tile<T, DH>(sQ, jt, xk, lane);
            const f32x4 dp = score_tile<T, DH>(sDO, jt, xv, lane);
            const f32x4 ls = *reinterpret_cast<const f32x4*>(sLse + jt * 16 + 4 * g);
            const f32x4 de = *reinterpret_cast<const f32x4*>(sDelta + jt * 16 + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int qq = ch * KC + jt * 16 + 4 * g + r;
              const float p = (qq < Tn && key < Tn) ? __expf(s[r] * scale - ls[r]) : 0.f;
              pt2[hh][r] = p;
              dst2[hh][r] = p * (dp[r] - de[r]) * scale;
            }
          }
          accumulate_wz_step<T, DH>(dv, pt2[0], pt2[1], sDO, ks, lane);
          accumulate_wz_step<T, DH>(dk, dst2[0], dst2[1], sQ, ks, lane);
        }
      }
    }
    if (active && key < Tn) {
      T* row = dqkv + ((size_t)b * Tn + key) * ld + h * DH;
#pragma unroll
      for (int dt = 0; dt < C::NDT; ++dt) {
        store4(row + D + dt * 16 + 4 * g, dk[dt]);
        store4(row + 2 * D + dt * 16 + 4 * g, dv[dt]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------- backward, fused + persistent
// Single-chunk case (T <= 160), bf16: one persistent workgroup per CU walks (batch, head) items and computes dQ, dK and dV of an item in two
// phases that ping-pong two pairs of dense LDS images (AttnDense, global_load_lds):
//   phase 1  dQ      streams K, V from LDS   (lane side: this wave's 16 query rows of Q and dO, from global memory)
//            ... meanwhile the DMA brings this item's Q and dO into the other pair
//   phase 2  dK, dV  streams Q, dO from LDS  (lane side: this wave's 16 key rows of K and V, read from the images before phase 1 ends)
//            ... meanwhile the DMA brings the NEXT item's K and V
// so apart from the first K, V of a workgroup no load is exposed, the scores' ingredients are fetched once per item instead of once per
// kernel, and delta = rowsum(dO * O) and the log-sum-exp go from phase 1 to phase 2 through LDS instead of through HBM.
template <typename T, int DH, int NW> __global__ __launch_bounds__(64 * NW) void attn_bwd_fused_kernel(const T* __restrict__ qkv, const T* __restrict__ o,
                                                                                                   const T* __restrict__ dout, const float* __restrict__ lse,
                                                                                                   T* __restrict__ dqkv, int Tn, int H, int nitems, float scale) {
  using C = AttnCfg<T, DH>;
  using DI = AttnDense<T, DH>;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [K | V | Q | dO | lse[KC] | delta[KC]]
  char* iK = smem;
  char* iV = smem + DI::BYTES;
  char* iQ = smem + 2 * DI::BYTES;
  char* iDO = smem + 3 * DI::BYTES;
  const T *sK = reinterpret_cast<const T*>(iK), *sV = reinterpret_cast<const T*>(iV), *sQ = reinterpret_cast<const T*>(iQ), *sDO = reinterpret_cast<const T*>(iDO);
  float* sLse = reinterpret_cast<float*>(smem + 4 * DI::BYTES);
  float* sDelta = sLse + KC;
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int D = H * DH, ld = 3 * D;
  const int ntiles = (Tn + 15) / 16;
  const bool active = wave < ntiles;
  const int row = wave * 16 + c;  // this lane's lane-side row: a query in phase 1, a key in phase 2
  // The element-wise work between the products is what the busiest SIMD (3 of the 9 waves) spends most of its time on, so it is kept to
  // p = exp2(s * c2 - lse * log2e), ds = p * (dp * scale - delta * scale): two FMAs, one exponential, one multiply per score; the mask of
  // the streamed rows >= T only where a tile can contain such rows.
  const float c2 = scale * 1.4426950408889634f;
  const int full_tiles = Tn / 16;  // streamed tiles jt < full_tiles hold valid rows only
  auto kfrag = [&](const T* img, int jt, int s2) {  // regs-side fragment of streamed rows jt*16.. from a dense image, zero beyond head_dim
    Frag<T> f = frag_kcontig(img, DH, jt * 16, 32 * s2, lane);
    if (32 * s2 + 8 * g + 8 > DH) f = frag_zero<T>();
    return f;
  };
  auto scores = [&](const T* img, int jt, const Frag<T>* x) {
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s2 = 0; s2 < C::NKF; ++s2) a = mma(kfrag(img, jt, s2), x[s2], a);
    return a;
  };
  auto accumulate = [&](f32x4* out, f32x4 w0, f32x4 w1, const T* img, int ks) {
    const Frag<T> wf = frag_from_acc(w0, w1, T());
#pragma unroll
    for (int dt = 0; dt < C::NDT; ++dt) out[dt] = mma(frag_kstrided2(img, DH, 32 * ks, 32 * ks + 16, dt * 16, lane), wf, out[dt]);
  };
  for (int r = ntiles * 16 + tid; r < KC; r += 64 * NW) {  // rows no wave owns: never written again, read (and masked) by phase 2
    sLse[r] = 0.f;
    sDelta[r] = 0.f;
  }
  const int Bn = nitems / H;
  int it = attn_item(blockIdx.x, gridDim.x, 0, Bn, H);
  if (it >= 0) {
    const T* base = qkv + (size_t)(it / H) * Tn * ld + (it % H) * DH;
    DI::stage(iK, base + D, ld, Tn, wave, NW, lane);
    DI::stage(iV, base + 2 * D, ld, Tn, wave, NW, lane);
  }
  f32x4 dk[C::NDT], dv[C::NDT];  // results of phase 2: stored after the NEXT barrier (see the note at barrier B)
  int pend = -1;                 // item whose dK, dV are still in registers
  auto store_dkv = [&](int item) {
    if (active) {  // (wave-uniform: the lane exchange inside needs all lanes)
      T* out = dqkv + ((size_t)(item / H) * Tn + row) * ld + (item % H) * DH;
      store_row_tiles<T, C::NDT>(out + D, dk, g, row < Tn);
      store_row_tiles<T, C::NDT>(out + 2 * D, dv, g, row < Tn);
    }
  };
  for (int n = 0; it >= 0; ++n) {
    const int it_next = attn_item(blockIdx.x, gridDim.x, n + 1, Bn, H);
    const int b = it / H, h = it % H;
    const T* base = qkv + (size_t)b * Tn * ld + h * DH;
    const T* dobase = dout + (size_t)b * Tn * D + h * DH;
    // ---- phase 1: dQ
    Frag<T> xq[C::NKF], xdo[C::NKF];
    load_row_frags<T, DH>(xq, base, ld, wave * 16, active ? Tn : 0, lane);
    load_row_frags<T, DH>(xdo, dobase, D, wave * 16, active ? Tn : 0, lane);
    float lse_q = 0.f, delta_q = 0.f;
    {
      Frag<T> xo[C::NKF];
      load_row_frags<T, DH>(xo, o + (size_t)b * Tn * D + h * DH, D, wave * 16, active ? Tn : 0, lane);
#pragma unroll
      for (int s2 = 0; s2 < C::NKF; ++s2)
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) delta_q += to_f32(xo[s2].v[jj]) * to_f32(xdo[s2].v[jj]);
      delta_q += __shfl_xor(delta_q, 16, 64);
      delta_q += __shfl_xor(delta_q, 32, 64);
    }
    if (active && row < Tn) lse_q = lse[((size_t)b * H + h) * Tn + row] * 1.4426950408889634f;  // in units of log 2
    const float delta_s = delta_q * scale;
    __syncthreads();  // (A) K and V of this item have landed; nobody reads the previous item's Q / dO images, lse or delta any more
    if (pend >= 0) store_dkv(pend);
    DI::stage(iQ, base, ld, Tn, wave, NW, lane);
    DI::stage(iDO, dobase, D, Tn, wave, NW, lane);
    if (active && g == 0) {
      sLse[row] = lse_q;      // (log-2 units) rows >= Tn hold 0 (masked in phase 2)
      sDelta[row] = row < Tn ? delta_s : 0.f;  // delta * scale
    }
    f32x4 dq[C::NDT];
#pragma unroll
    for (int dt = 0; dt < C::NDT; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (active) {
#pragma unroll
      for (int ks = 0; ks < C::NJT / 2; ++ks) {  // two key tiles at a time: scores -> dS -> straight into dQ
        f32x4 ds2[2];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const int jt = 2 * ks + hh;
          const f32x4 sc = scores(sK, jt, xq);
          const f32x4 dp = scores(sV, jt, xdo);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float pr = __builtin_amdgcn_exp2f(sc[r] * c2 - lse_q);
            if (jt >= full_tiles && jt * 16 + 4 * g + r >= Tn) pr = 0.f;  // key rows beyond the sequence (zero rows of the image)
            ds2[hh][r] = pr * (dp[r] * scale - delta_s);
          }
        }
        accumulate(dq, ds2[0], ds2[1], sK, ks);
      }
    }
    // lane-side operands of phase 2: this wave's 16 key rows, from the K / V images (rows >= T are zero) before the next item's DMA overwrites them
    Frag<T> xk[C::NKF], xv[C::NKF];
#pragma unroll
    for (int s2 = 0; s2 < C::NKF; ++s2) {
      xk[s2] = kfrag(sK, wave, s2);
      xv[s2] = kfrag(sV, wave, s2);
    }
    __syncthreads();  // (B) Q and dO have landed, lse / delta are complete; every wave holds its K / V fragments, the images are free
    // results are stored AFTER the barrier that follows their phase: a barrier drains the memory counter (loads, DMA and stores alike), so a
    // store issued just before it is a full round trip of waiting for all nine waves; issued here it completes under the next phase
    if (active) {
      T* out = dqkv + ((size_t)b * Tn + row) * ld + h * DH;
      store_row_tiles<T, C::NDT>(out, dq, g, row < Tn);
    }
    if (it_next >= 0) {
      const int nx = it_next;
      const T* nbase = qkv + (size_t)(nx / H) * Tn * ld + (nx % H) * DH;
      DI::stage(iK, nbase + D, ld, Tn, wave, NW, lane);
      DI::stage(iV, nbase + 2 * D, ld, Tn, wave, NW, lane);
    }
    // ---- phase 2: dK, dV
#pragma unroll
    for (int dt = 0; dt < C::NDT; ++dt) {
      dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
      dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (active) {
#pragma unroll
      for (int ks = 0; ks < C::NJT / 2; ++ks) {  // two query tiles at a time: P^T, dS^T -> straight into dV, dK
        f32x4 pt2[2], dst2[2];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const int jt = 2 * ks + hh;
          const f32x4 sc = scores(sQ, jt, xk);
          const f32x4 dp = scores(sDO, jt, xv);
          const f32x4 ls = *reinterpret_cast<const f32x4*>(sLse + jt * 16 + 4 * g);
          const f32x4 de = *reinterpret_cast<const f32x4*>(sDelta + jt * 16 + 4 * g);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float pr = __builtin_amdgcn_exp2f(sc[r] * c2 - ls[r]);
            if (jt >= full_tiles && jt * 16 + 4 * g + r >= Tn) pr = 0.f;  // query rows beyond the sequence
            pt2[hh][r] = pr;
            dst2[hh][r] = pr * (dp[r] * scale - de[r]);
          }
        }
        accumulate(dv, pt2[0], pt2[1], sDO, ks);
        accumulate(dk, dst2[0], dst2[1], sQ, ks);
      }
    }
    pend = it;
    it = it_next;
  }
  if (pend >= 0) store_dkv(pend);
}

}  // namespace
#include "v4h_attention_dense.h"
namespace {
using v4h_dense::attn_fwd_dense_kernel;
using v4h_dense::DenseImage;
using v4h_dense::attn_fwd_long_kernel;
using v4h_dense::attn_bwd_dq_img_kernel;
using v4h_dense::attn_bwd_dkv_img_kernel;
using v4h_dense::AL_PAD;
using v4h_dense::AL_NT;
using v4h_dense::AL_NW;

template <typename K> int set_lds(K kernel, size_t bytes, const char* name) {
  if (bytes > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
      v4h_set_error("%s: cannot reserve %zu bytes of LDS: %s", name, bytes, hipGetErrorString(e));
      return V4H_ERR_HIP;
    }
  }
  return V4H_OK;
}

// One workgroup per (batch, head): K/V (or Q/dO) are loaded once and every 16-row tile of the lane-side sequence gets a
// wave.  9 waves when the tile count is a multiple of 9 (ds2: T = 135 -> 9 tiles, no idle wave), else 8 (ds3: 29 tiles).
template <typename T, int NW, int DHT = 80> int attn_fwd_launch(const void* qkv, void* o, float* lse, int B, int Tn, int H, int DH, hipStream_t s) {
  using C = AttnCfg<T, DHT>;
  const size_t lds = 2 * (size_t)C::TILE_ELEMS * sizeof(T) + 64;
  int rc = set_lds(attn_fwd_kernel<T, DHT, NW>, lds, "attn_fwd");
  if (rc) return rc;
  // one round of NW query tiles per workgroup: a (batch, head) item with more tiles is spread over gridDim.y workgroups, each staging the
  // K/V chunks once - the same number of chunk loads as one workgroup doing several rounds, but B*H = 192..384 items no longer leave CUs idle
  const int ny = ((Tn + 15) / 16 + NW - 1) / NW;
  hipLaunchKernelGGL((attn_fwd_kernel<T, DHT, NW>), dim3(B * H, ny), dim3(64 * NW), lds, s, (const T*)qkv, (T*)o, lse, Tn, H, 1.0f / sqrtf((float)DH));
  V4H_CHECK_LAUNCH("attn_fwd");
  return V4H_OK;
}
template <int NT, int WPE, int NBUF> int attn_fwd_dense_launch(const void* qkv, void* o, float* lse, int Tn, int H, int nitems, float scale, int grid, hipStream_t s) {
  const size_t lds = NBUF * 2 * (size_t)DenseImage<NT>::BYTES;
  int rc = set_lds(attn_fwd_dense_kernel<NT, WPE, NBUF>, lds, "attn_fwd_dense");
  if (rc) return rc;
  hipLaunchKernelGGL((attn_fwd_dense_kernel<NT, WPE, NBUF>), dim3(grid), dim3(64 * NT), lds, s, (const bf16*)qkv, (bf16*)o, lse, Tn, H, nitems, scale);
  return V4H_OK;
}
template <typename T> int attn_fwd_t(const void* qkv, void* o, float* lse, int B, int Tn, int H, int DH, hipStream_t s) {
  if (DH == 32) {  // energy-model transformer (d_model 128, 4 heads; configs/model/cfm/cfm_ds2_energy.yaml), forward only
    V4H_CHECK_ARG(Tn <= 64, "attention: head_dim 32 is built for sequences of at most 64 tokens (got %d)", Tn);
    return attn_fwd_launch<T, 4, 32>(qkv, o, lse, B, Tn, H, DH, s);
  }
  V4H_CHECK_ARG(DH == 80, "attention: head_dim %d not built (80 = 480/6 for every shape-CFM config, 32 forward-only for the energy model)", DH);
  const int ntiles = (Tn + 15) / 16;
  if constexpr (sizeof(T) == 2) {
    static const int dense = getenv("V4H_ATTN_DENSE") ? atoi(getenv("V4H_ATTN_DENSE")) : 1;  // A/B hook: 0 = round 2's persistent kernel
    if (dense && Tn <= KC && ntiles >= 6 && (long)H * 80 * 2 * 3 * Tn < 0x7FFFFF00L) {  // 81..160 tokens: the instruction-lean, descriptor-addressed form
      // (its key mask covers the LAST tile only, so the tile count must be ceil(T / 16); shorter sequences keep the kernels below)
      const int nitems = B * H;
      const float scale = 1.0f / sqrtf((float)DH);
      const int cus = v4h_compute_units();
      const int rc = ntiles == 6   ? attn_fwd_dense_launch<6, 3, 2>(qkv, o, lse, Tn, H, nitems, scale, cus, s)
                     : ntiles == 7 ? attn_fwd_dense_launch<7, 3, 2>(qkv, o, lse, Tn, H, nitems, scale, cus, s)
                     : ntiles == 8 ? attn_fwd_dense_launch<8, 3, 2>(qkv, o, lse, Tn, H, nitems, scale, cus, s)
                     : ntiles == 9 ? attn_fwd_dense_launch<9, 3, 2>(qkv, o, lse, Tn, H, nitems, scale, cus, s)
                                   : attn_fwd_dense_launch<10, 3, 2>(qkv, o, lse, Tn, H, nitems, scale, cus, s);
      if (rc) return rc;
      V4H_CHECK_LAUNCH("attn_fwd_dense");
      return V4H_OK;
    }
    if (dense && ntiles >= 24 && ntiles <= AL_NT && (long)H * 80 * 2 * 3 * Tn < 0x7FFFFF00L) {  // 369..480 tokens (ds3): whole-item K / V images in LDS
      const size_t lds = 2 * (size_t)DenseImage<AL_NT, AL_NW>::BYTES;
      int rc = set_lds(attn_fwd_long_kernel<2>, lds, "attn_fwd_long");
      if (rc) return rc;
      hipLaunchKernelGGL((attn_fwd_long_kernel<2>), dim3(v4h_compute_units()), dim3(64 * AL_NW), lds, s, (const bf16*)qkv, (bf16*)o, lse, Tn, H, B * H,
                         1.0f / sqrtf((float)DH));
      V4H_CHECK_LAUNCH("attn_fwd_long");
      return V4H_OK;
    }
    static const bool persist = !(getenv("V4H_ATTN_PERSIST") && getenv("V4H_ATTN_PERSIST")[0] == '0');
    if (persist && Tn <= KC && ntiles <= 9) {  // single key chunk: persistent, double-buffered K/V
      constexpr int NW = 9;
      const size_t lds = 4 * (size_t)AttnDense<T, 80>::BYTES + 64;
      int rc = set_lds(attn_fwd_persist_kernel<T, 80, NW>, lds, "attn_fwd_persist");
      if (rc) return rc;
      const int nitems = B * H;
      hipLaunchKernelGGL((attn_fwd_persist_kernel<T, 80, NW>), dim3(v4h_compute_units()), dim3(64 * NW), lds, s, (const T*)qkv, (T*)o, lse, Tn, H, nitems,
                         1.0f / sqrtf((float)DH));
      V4H_CHECK_LAUNCH("attn_fwd_persist");
      return V4H_OK;
    }
  }
  if (sizeof(T) == 4 || ntiles <= 4) return attn_fwd_launch<T, 4>(qkv, o, lse, B, Tn, H, DH, s);  // f32: 8 VGPRs per fragment -> 256-register budget
  if (ntiles % 9 == 0) return attn_fwd_launch<T, 9>(qkv, o, lse, B, Tn, H, DH, s);
  return attn_fwd_launch<T, 8>(qkv, o, lse, B, Tn, H, DH, s);
}

template <typename T, int NW> int attn_bwd_launch(const void* qkv, const void* o, const void* dout, const float* lse, float* delta, void* dqkv, int B, int Tn, int H,
                                                 int DH, hipStream_t s) {
  using C = AttnCfg<T, 80>;
  const float scale = 1.0f / sqrtf((float)DH);
  const size_t lds_q = 2 * (size_t)C::TILE_ELEMS * sizeof(T) + 64;
  int rc = set_lds(attn_bwd_dq_kernel<T, 80, NW>, lds_q, "attn_bwd_dq");
  if (rc) return rc;
  const int ny = ((Tn + 15) / 16 + NW - 1) / NW;  // as in the forward: one round of tiles per workgroup
  hipLaunchKernelGGL((attn_bwd_dq_kernel<T, 80, NW>), dim3(B * H, ny), dim3(64 * NW), lds_q, s, (const T*)qkv, (const T*)o, (const T*)dout, lse, delta, (T*)dqkv, Tn, H, scale);
  V4H_CHECK_LAUNCH("attn_bwd_dq");
  const size_t lds_kv = lds_q + 2 * KC * sizeof(float);
  rc = set_lds(attn_bwd_dkv_kernel<T, 80, NW>, lds_kv, "attn_bwd_dkv");
  if (rc) return rc;
  hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, 80, NW>), dim3(B * H, ny), dim3(64 * NW), lds_kv, s, (const T*)qkv, (const T*)dout, lse, delta, (T*)dqkv, Tn, H, scale);
  V4H_CHECK_LAUNCH("attn_bwd_dkv");
  return V4H_OK;
}
template <typename T> int attn_bwd_t(const void* qkv, const void* o, const void* dout, const float* lse, float* delta, void* dqkv, int B, int Tn, int H, int DH,
                                     hipStream_t s) {
  V4H_CHECK_ARG(DH == 80, "attention: head_dim %d not built (only 80)", DH);
  const int ntiles = (Tn + 15) / 16;
  if constexpr (sizeof(T) == 2) {
    static const bool fused = !(getenv("V4H_ATTN_BWD_FUSED") && getenv("V4H_ATTN_BWD_FUSED")[0] == '0');
    // (Round 4: the same kernel with 2 or 3 lane-side tiles per wave - 5 / 3 waves instead of 9, every streamed LDS fragment feeding 2 / 3 MFMAs, results bit-identical -
    //  measured 61.6 / 72.5 us per call against 54.8 cold, 242 / 239 against 243-246 steps/s: this kernel wants MORE resident waves, not fewer LDS reads.
    //  Kept out of the library: tools/experiments/attn_bwd_tpw.inc, profiles/r04_notes.md.)
    if (fused && Tn <= KC && ntiles <= 9) {  // single chunk: dQ, dK, dV of an item in one persistent, double-buffered kernel
      constexpr int NW = 9;
      const size_t lds = 4 * (size_t)AttnDense<T, 80>::BYTES + 2 * KC * sizeof(float);
      int rc = set_lds(attn_bwd_fused_kernel<T, 80, NW>, lds, "attn_bwd_fused");
      if (rc) return rc;
      const int nitems = B * H;
      V4H_LAUNCH((attn_bwd_fused_kernel<T, 80, NW>), dim3(v4h_compute_units()), dim3(64 * NW), lds, s, (const T*)qkv, (const T*)o, (const T*)dout, lse,
                         (T*)dqkv, Tn, H, nitems, 1.0f / sqrtf((float)DH));
      V4H_CHECK_LAUNCH("attn_bwd_fused");
      return V4H_OK;
    }
  }
  if constexpr (sizeof(T) == 2) {
    static const int dense = getenv("V4H_ATTN_DENSE") ? atoi(getenv("V4H_ATTN_DENSE")) : 1;  // A/B hook: 0 = the chunked kernels
    if (dense && ntiles >= 24 && ntiles <= AL_NT && (long)H * 80 * 2 * 3 * Tn < 0x7FFFFF00L) {  // 369..480 tokens (ds3): whole-item images in LDS
      const size_t img2 = 2 * (size_t)DenseImage<AL_NT, AL_NW>::BYTES;
      const float scale = 1.0f / sqrtf((float)DH);
      int rc = set_lds(attn_bwd_dq_img_kernel<AL_NT, AL_NW, 2, 1>, img2 + AL_PAD, "attn_bwd_long_dq");
      if (rc) return rc;
      hipLaunchKernelGGL((attn_bwd_dq_img_kernel<AL_NT, AL_NW, 2, 1>), dim3(v4h_compute_units()), dim3(64 * AL_NW), img2 + AL_PAD, s, (const bf16*)qkv, (const bf16*)o,
                         (const bf16*)dout, lse, delta, (bf16*)dqkv, Tn, H, B * H, scale);
      V4H_CHECK_LAUNCH("attn_bwd_long_dq");
      const size_t lds2 = img2 + AL_PAD + 2 * AL_NT * 16 * sizeof(float);
      rc = set_lds(attn_bwd_dkv_img_kernel<AL_NT, AL_NW, 2, 1>, lds2, "attn_bwd_long_dkv");
      if (rc) return rc;
      V4H_LAUNCH((attn_bwd_dkv_img_kernel<AL_NT, AL_NW, 2, 1>), dim3(v4h_compute_units()), dim3(64 * AL_NW), lds2, s, (const bf16*)qkv, (const bf16*)dout, lse,
                         (const float*)delta, (bf16*)dqkv, Tn, H, B * H, scale);
      V4H_CHECK_LAUNCH("attn_bwd_long_dkv");
      return V4H_OK;
    }
  }
  if (sizeof(T) == 4 || ntiles <= 4) return attn_bwd_launch<T, 4>(qkv, o, dout, lse, delta, dqkv, B, Tn, H, DH, s);
  if (ntiles % 9 == 0) return attn_bwd_launch<T, 9>(qkv, o, dout, lse, delta, dqkv, B, Tn, H, DH, s);
  return attn_bwd_launch<T, 8>(qkv, o, dout, lse, delta, dqkv, B, Tn, H, DH, s);
}

}  // namespace

namespace v4h {
int attention_fwd(Mode m, const void* qkv, void* o, float* lse, int B, int Tn, int H, int DH, hipStream_t s) {
  return m == MODE_BF16 ? attn_fwd_t<bf16>(qkv, o, lse, B, Tn, H, DH, s) : attn_fwd_t<float>(qkv, o, lse, B, Tn, H, DH, s);
}
int attention_bwd(Mode m, const void* qkv, const void* o, const void* dout, const float* lse, float* delta, void* dqkv, int B, int Tn, int H, int DH, hipStream_t s) {
  return m == MODE_BF16 ? attn_bwd_t<bf16>(qkv, o, dout, lse, delta, dqkv, B, Tn, H, DH, s)
                        : attn_bwd_t<float>(qkv, o, dout, lse, delta, dqkv, B, Tn, H, DH, s);
}
}  // namespace v4h
