// Generic LDS-tiled MFMA contraction for every Linear on the ViT path (reference nn/vit.py: x_embedder :76,
// c_embedder :77-81, t_embedder :361-365, adaLN :323-325/345, qkv :416, proj :420, timm Mlp fc1/fc2 :317-322,
// final linear :344) and for their dgrad / wgrad.
//
//     Out[i][j] = sum_k P[i][k] * Q[j][k]            i: "lane side" (tokens), j: "regs side" (features)
//
// Each operand is either K-contiguous in memory (X[idx][k], row stride ld) or K-strided (X[k][idx]):
//     forward  y = x W^T      : P = x  [tok][K]  contig , Q = W  [N][K]    contig
//     dgrad    dx = dy W      : P = dy [tok][N]  contig , Q = W  [N][Kin]  K-strided  (Q[j=kin][k=n] = W[n][kin])
//     wgrad    dW = dy^T x    : P = dy [tok][N]  strided, Q = x  [tok][Kin] K-strided (k = token)
// so no transposed copy of a weight or an activation is ever written to HBM; K-strided bf16 fragments come from
// ds_read_b64_tr_b16.  A lane ends up with 4 consecutive j of one i -> 8/16-byte epilogue accesses.
//
// Structure: 256 threads = 4 waves as 2(i) x 2(j); block tile BI x BJ; K-step BK (multiple of 32); register-staged
// global->LDS copy with the next tile's loads in flight during the MFMAs; two LDS buffers, one barrier per K-step.
// Tails: rows beyond I / K and 16-byte column chunks beyond the extent are zero-filled on load; stores are guarded.
// Split-K over blockIdx.z with f32 atomics (wgrad).  Optional column sums of P over k (bias gradients) for free.
#pragma once
#include "v4h_common.h"

enum : int {
  EPI_STORE = 0,      // out(TO)[i][j] = acc + bias[j]
  EPI_STORE_F32,      // out(f32)[i][j] = acc + bias[j]
  EPI_SILU,           // out(TO) = silu(acc + bias) ; out2(f32) = acc + bias (pre-activation, if out2)
  EPI_COND_SUM,       // s = acc + bias + (resid? resid[i][j] : 0) ; out(f32) = s ; out2(TO) = silu(s)
  EPI_EMBED,          // out(f32) = acc + bias + rowvec[(i % T)][j]                (x_embedder + pos-emb)
  EPI_GATE_RESID,     // y = acc + bias ; out2(TO) = y ; out(f32) = resid + gate[b(i)][j] * y
  EPI_GELU,           // pre = acc + bias ; out(TO) = pre ; out2(TO) = gelu_tanh(pre)
  EPI_DGELU,          // out(TO) = acc * gelu_tanh'(aux(TO)[i][j])
  EPI_DSILU,          // out(TO) = acc * silu'(auxf(f32)[i][j])
  EPI_ATOMIC_F32,     // atomicAdd(out(f32)[i][j], acc)
  EPI_ACCUM_F32,      // out(f32)[i][j] += acc
  EPI_UNPATCH,        // voxel scatter: out(f32)[b, voxel(n, f=j)] = acc + bias[j]  (final linear + from_patches)
};

struct PatchGeom {  // CaloChallengeCFM.to_patches / from_patches  (calochallenge_cfm/model.py:40-60), C = 1
  int L, A, R;      // voxel grid
  int p1, p2, p3;   // patch shape
  int l, a, r;      // patches per axis
};

struct EpiArgs {
  void* out; int ldo;
  void* out2; int ldo2;
  const float* bias;
  const float* rowvec; int ld_rowvec;  // gate [B][ld] (pointer already offset to the gate chunk) or pos-emb [T][D]
  int T;                                // tokens per sample (b = i / T)
  const float* resid; int ld_resid;
  const void* aux; int ld_aux;          // TO-typed auxiliary (pre-activation for gelu')
  const float* auxf; int ld_auxf;       // f32 auxiliary
  PatchGeom pg; int P;                  // EPI_UNPATCH: real patch_dim (columns >= P are padding)
};

struct GemmArgs {
  const void* P; const void* Q;
  int ldp, ldq;
  int I, J, K;
  int klen;          // K range per blockIdx.z (multiple of BK)
  float* colsum;     // optional: colsum[i] += sum_k P[i][k]   (f32 atomics; only j-tile 0 contributes)
  EpiArgs e;
};

template <int EPI, typename T, typename TO> struct Epilogue {
  static V4H_DEV void apply(const EpiArgs& e, int i, int j, f32x4 v) {
    if constexpr (EPI != EPI_DGELU && EPI != EPI_DSILU && EPI != EPI_ATOMIC_F32 && EPI != EPI_ACCUM_F32) {
      if (e.bias) {
        const f32x4 b = load4(e.bias + j);
        v += b;
      }
    }
    if constexpr (EPI == EPI_STORE) {
      store4(reinterpret_cast<TO*>(e.out) + (size_t)i * e.ldo + j, v);
    } else if constexpr (EPI == EPI_STORE_F32) {
      store4(reinterpret_cast<float*>(e.out) + (size_t)i * e.ldo + j, v);
    } else if constexpr (EPI == EPI_SILU) {
      if (e.out2) store4(reinterpret_cast<float*>(e.out2) + (size_t)i * e.ldo2 + j, v);
      f32x4 s;
#pragma unroll
      for (int r = 0; r < 4; ++r) s[r] = silu_f(v[r]);
      store4(reinterpret_cast<TO*>(e.out) + (size_t)i * e.ldo + j, s);
    } else if constexpr (EPI == EPI_COND_SUM) {
      if (e.resid) v += load4(e.resid + (size_t)i * e.ld_resid + j);
      store4(reinterpret_cast<float*>(e.out) + (size_t)i * e.ldo + j, v);
      f32x4 s;
#pragma unroll
      for (int r = 0; r < 4; ++r) s[r] = silu_f(v[r]);
      store4(reinterpret_cast<TO*>(e.out2) + (size_t)i * e.ldo2 + j, s);
    } else if constexpr (EPI == EPI_EMBED) {
      v += load4(e.rowvec + (size_t)(i % e.T) * e.ld_rowvec + j);
      store4(reinterpret_cast<float*>(e.out) + (size_t)i * e.ldo + j, v);
    } else if constexpr (EPI == EPI_GATE_RESID) {
      if (e.out2) store4(reinterpret_cast<TO*>(e.out2) + (size_t)i * e.ldo2 + j, v);
      const f32x4 g = load4(e.rowvec + (size_t)(i / e.T) * e.ld_rowvec + j);
      const f32x4 x = load4(e.resid + (size_t)i * e.ld_resid + j);
      store4(reinterpret_cast<float*>(e.out) + (size_t)i * e.ldo + j, x + g * v);
    } else if constexpr (EPI == EPI_GELU) {
      store4(reinterpret_cast<TO*>(e.out) + (size_t)i * e.ldo + j, v);
      f32x4 s;
#pragma unroll
      for (int r = 0; r < 4; ++r) s[r] = gelu_tanh_f<T>(v[r]);
      store4(reinterpret_cast<TO*>(e.out2) + (size_t)i * e.ldo2 + j, s);
    } else if constexpr (EPI == EPI_DGELU) {
      const f32x4 pre = load4(reinterpret_cast<const TO*>(e.aux) + (size_t)i * e.ld_aux + j);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] *= dgelu_tanh_f<T>(pre[r]);
      store4(reinterpret_cast<TO*>(e.out) + (size_t)i * e.ldo + j, v);
    } else if constexpr (EPI == EPI_DSILU) {
      const f32x4 pre = load4(e.auxf + (size_t)i * e.ld_auxf + j);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] *= dsilu_f(pre[r]);
      store4(reinterpret_cast<TO*>(e.out) + (size_t)i * e.ldo + j, v);
    } else if constexpr (EPI == EPI_ATOMIC_F32) {
      float* o = reinterpret_cast<float*>(e.out) + (size_t)i * e.ldo + j;
#pragma unroll
      for (int r = 0; r < 4; ++r) atomicAdd(o + r, v[r]);
    } else if constexpr (EPI == EPI_ACCUM_F32) {
      float* o = reinterpret_cast<float*>(e.out) + (size_t)i * e.ldo + j;
      store4(o, load4(o) + v);
    } else if constexpr (EPI == EPI_UNPATCH) {
      // token n = (li*a + ai)*r + ri ; feature f = (pi*p2 + pj)*p3 + pk  ->  voxel (li*p1+pi, ai*p2+pj, ri*p3+pk)
      const PatchGeom& g = e.pg;
      const int b = i / e.T, n = i % e.T;
      const int ri = n % g.r, ai = (n / g.r) % g.a, li = n / (g.r * g.a);
      float* o = reinterpret_cast<float*>(e.out) + (size_t)b * g.L * g.A * g.R;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int f = j + r;
        if (f < e.P) {
          const int pk = f % g.p3, pj = (f / g.p3) % g.p2, pi = f / (g.p3 * g.p2);
          o[((size_t)(li * g.p1 + pi) * g.A + (ai * g.p2 + pj)) * g.R + (ri * g.p3 + pk)] = v[r];
        }
      }
    }
  }
};

// 2-D tile copy global -> registers -> LDS in 16-byte chunks, zero-filled outside [0,rows_end) x [0,cols_end).
template <typename T, int ROWS, int COLS, int LD> struct TileStage {
  static constexpr int CH = 16 / (int)sizeof(T);
  static constexpr int CPR = COLS / CH;
  static constexpr int TOTAL = ROWS * CPR;
  static constexpr int N = (TOTAL + 255) / 256;
  static_assert(COLS % CH == 0, "tile columns must be whole 16-byte chunks");
  uint4 r[N];

  V4H_DEV void load(const T* g, int ld, int row0, int col0, int rows_end, int cols_end, int tid) {
#pragma unroll
    for (int n = 0; n < N; ++n) {
      const int c = tid + n * 256;
      const int tr = c / CPR, tc = (c % CPR) * CH;
      const int gr = row0 + tr, gc = col0 + tc;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (c < TOTAL && gr < rows_end && gc + CH <= cols_end) v = *reinterpret_cast<const uint4*>(g + (size_t)gr * ld + gc);
      r[n] = v;
    }
  }
  V4H_DEV void store(T* s, int tid) const {
#pragma unroll
    for (int n = 0; n < N; ++n) {
      const int c = tid + n * 256;
      if (c < TOTAL) {
        const int tr = c / CPR, tc = (c % CPR) * CH;
        *reinterpret_cast<uint4*>(s + tr * LD + tc) = r[n];
      }
    }
  }
};

template <typename T_, typename TO_, bool PKS_, bool QKS_, int BI_, int BJ_, int BK_, int EPI_, bool COLSUM_> struct GemmCfg {
  using T = T_;
  using TO = TO_;
  static constexpr bool PKS = PKS_, QKS = QKS_, COLSUM = COLSUM_;
  static constexpr int BI = BI_, BJ = BJ_, BK = BK_, EPI = EPI_;
  static constexpr int PAD = 16 / (int)sizeof(T);
  // LDS images: K-contiguous operand tile[idx][BK + PAD]; K-strided operand tile[BK][idx + PAD]
  static constexpr int P_ROWS = PKS ? BK : BI, P_COLS = PKS ? BI : BK, P_LD = P_COLS + PAD;
  static constexpr int Q_ROWS = QKS ? BK : BJ, Q_COLS = QKS ? BJ : BK, Q_LD = Q_COLS + PAD;
  static constexpr int P_ELEMS = P_ROWS * P_LD, Q_ELEMS = Q_ROWS * Q_LD;
  static constexpr size_t LDS_BYTES = 2 * (size_t)(P_ELEMS + Q_ELEMS) * sizeof(T);
  static constexpr int WTI = BI / 2, WTJ = BJ / 2, TI = WTI / 16, TJ = WTJ / 16;
  static_assert(BI % 32 == 0 && BJ % 32 == 0 && BK % 32 == 0, "tile shape");
};

template <class C> __global__ __launch_bounds__(256) void v4h_gemm_kernel(const GemmArgs a) {
  using T = typename C::T;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* const sP0 = reinterpret_cast<T*>(smem);                      // two buffers of P, then two of Q
  T* const sQ0 = reinterpret_cast<T*>(smem) + 2 * C::P_ELEMS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave >> 1, wj = wave & 1;
  const int i0 = blockIdx.x * C::BI, j0 = blockIdx.y * C::BJ;
  const int kb = blockIdx.z * a.klen;
  const int ke = min(a.K, kb + a.klen);
  const int nt = (ke - kb + C::BK - 1) / C::BK;
  const T* gP = reinterpret_cast<const T*>(a.P);
  const T* gQ = reinterpret_cast<const T*>(a.Q);

  TileStage<T, C::P_ROWS, C::P_COLS, C::P_LD> stP;
  TileStage<T, C::Q_ROWS, C::Q_COLS, C::Q_LD> stQ;
  auto gload = [&](int t) {
    const int k0 = kb + t * C::BK;
    if constexpr (C::PKS) stP.load(gP, a.ldp, k0, i0, ke, a.I, tid);
    else stP.load(gP, a.ldp, i0, k0, a.I, ke, tid);
    if constexpr (C::QKS) stQ.load(gQ, a.ldq, k0, j0, ke, a.J, tid);
    else stQ.load(gQ, a.ldq, j0, k0, a.J, ke, tid);
  };

  f32x4 acc[C::TI][C::TJ];
#pragma unroll
  for (int x = 0; x < C::TI; ++x)
#pragma unroll
    for (int y = 0; y < C::TJ; ++y) acc[x][y] = f32x4{0.f, 0.f, 0.f, 0.f};
  float cs[C::TI];
#pragma unroll
  for (int x = 0; x < C::TI; ++x) cs[x] = 0.f;

  if (nt > 0) {
    gload(0);
    stP.store(sP0, tid);
    stQ.store(sQ0, tid);
  }
  __syncthreads();

  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    if (t + 1 < nt) gload(t + 1);
    const T* tp = sP0 + cur * C::P_ELEMS;
    const T* tq = sQ0 + cur * C::Q_ELEMS;
#pragma unroll
    for (int kk = 0; kk < C::BK; kk += 32) {
      Frag<T> pf[C::TI], qf[C::TJ];
#pragma unroll
      for (int x = 0; x < C::TI; ++x) {
        const int idx = wi * C::WTI + x * 16;
        if constexpr (C::PKS) pf[x] = frag_kstrided<T>(tp, C::P_LD, kk, idx, lane);
        else pf[x] = frag_kcontig(tp, C::P_LD, idx, kk, lane);
      }
#pragma unroll
      for (int y = 0; y < C::TJ; ++y) {
        const int idx = wj * C::WTJ + y * 16;
        if constexpr (C::QKS) qf[y] = frag_kstrided<T>(tq, C::Q_LD, kk, idx, lane);
        else qf[y] = frag_kcontig(tq, C::Q_LD, idx, kk, lane);
      }
      if constexpr (C::COLSUM) {
#pragma unroll
        for (int x = 0; x < C::TI; ++x)
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) cs[x] += to_f32(pf[x].v[jj]);
      }
#pragma unroll
      for (int x = 0; x < C::TI; ++x)
#pragma unroll
        for (int y = 0; y < C::TJ; ++y) acc[x][y] = mma(qf[y], pf[x], acc[x][y]);
    }
    if (t + 1 < nt) {
      stP.store(sP0 + (cur ^ 1) * C::P_ELEMS, tid);
      stQ.store(sQ0 + (cur ^ 1) * C::Q_ELEMS, tid);
    }
    __syncthreads();
  }

  const int c = lane & 15, g = lane >> 4;
#pragma unroll
  for (int x = 0; x < C::TI; ++x) {
    const int i = i0 + wi * C::WTI + x * 16 + c;
#pragma unroll
    for (int y = 0; y < C::TJ; ++y) {
      const int j = j0 + wj * C::WTJ + y * 16 + 4 * g;
      if (i < a.I && j < a.J) Epilogue<C::EPI, T, typename C::TO>::apply(a.e, i, j, acc[x][y]);
    }
  }
  if constexpr (C::COLSUM) {
    if (a.colsum != nullptr && blockIdx.y == 0 && wj == 0) {
#pragma unroll
      for (int x = 0; x < C::TI; ++x) {
        float s = cs[x];
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        const int i = i0 + wi * C::WTI + x * 16 + c;
        if (g == 0 && i < a.I) atomicAdd(a.colsum + i, s);
      }
    }
  }
}

template <class C> int v4h_gemm_launch(GemmArgs a, int splitk, hipStream_t stream, const char* name) {
  V4H_CHECK_ARG(a.I > 0 && a.J > 0 && a.K > 0, "%s: empty problem I=%d J=%d K=%d", name, a.I, a.J, a.K);
  constexpr int CH = 16 / (int)sizeof(typename C::T);
  V4H_CHECK_ARG(a.J % 4 == 0, "%s: J=%d must be a multiple of 4", name, a.J);
  V4H_CHECK_ARG(a.ldp % CH == 0 && a.ldq % CH == 0, "%s: operand row strides (%d,%d) must be whole 16-byte chunks", name, a.ldp, a.ldq);
  V4H_CHECK_ARG(C::PKS ? (a.I % CH == 0) : (a.K % CH == 0), "%s: P extent not a whole number of 16-byte chunks", name);
  V4H_CHECK_ARG(C::QKS ? (a.J % CH == 0) : (a.K % CH == 0), "%s: Q extent not a whole number of 16-byte chunks", name);
  V4H_CHECK_ARG(((uintptr_t)a.P % 16) == 0 && ((uintptr_t)a.Q % 16) == 0, "%s: operands must be 16-byte aligned", name);
  if (splitk < 1) splitk = 1;
  if (C::EPI != EPI_ATOMIC_F32) splitk = 1;
  int klen = (a.K + splitk - 1) / splitk;
  klen = (klen + C::BK - 1) / C::BK * C::BK;
  a.klen = klen;
  const int nz = (a.K + klen - 1) / klen;
  dim3 grid((a.I + C::BI - 1) / C::BI, (a.J + C::BJ - 1) / C::BJ, nz);
  static bool attr_set = false;
  if (!attr_set && C::LDS_BYTES > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&v4h_gemm_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)C::LDS_BYTES);
    if (e != hipSuccess) {
      v4h_set_error("%s: cannot reserve %zu bytes of LDS: %s", name, (size_t)C::LDS_BYTES, hipGetErrorString(e));
      return V4H_ERR_HIP;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(v4h_gemm_kernel<C>, grid, dim3(256), C::LDS_BYTES, stream, a);
  V4H_CHECK_LAUNCH(name);
  return V4H_OK;
}
