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
#include <type_traits>

#include "v4h_common.h"

enum : int {
  EPI_STORE = 0,      // out(TO)[i][j] = acc + bias[j]
  EPI_STORE_F32,      // out(f32)[i][j] = acc + bias[j]
  EPI_SILU,           // out(TO) = silu(acc + bias) ; out2(f32) = acc + bias (pre-activation, if out2)
  EPI_COND_SUM,       // s = acc + bias + (resid? resid[i][j] : 0) ; out(f32) = s ; out2(TO) = silu(s)
  EPI_EMBED,          // out(f32, or TO with out_t) = acc + bias + rowvec[(i % T)][j]   (x_embedder + pos-emb)
  EPI_GATE_RESID,     // y = acc + bias ; out2(TO) = y ; out(f32) = resid + gate[b(i)][j] * y
  EPI_GELU,           // pre = acc + bias ; out(TO) = gelu_tanh'(pre) ; out2(TO) = gelu_tanh(pre)   (one tanh serves both)
  EPI_DGELU,          // out(TO) = acc * aux(TO)[i][j]        (aux = gelu_tanh'(pre) saved by the forward)
  EPI_DSILU,          // out(TO) = acc * silu'(auxf(f32)[i][j])
  EPI_ATOMIC_F32,     // atomicAdd(out(f32)[i][j], acc)
  EPI_ACCUM_F32,      // out(f32)[i][j] += acc
  EPI_UNPATCH,        // voxel scatter: out(f32)[b, voxel(n, f=j)] = acc + bias[j]  (final linear + from_patches)
  EPI_SLAB_F32,       // split-K partial: out(f32)[z][i][j] = acc (plain stores; summed afterwards by slab_reduce)
  EPI_RELU,           // out(TO) = max(acc + bias, 0)                               (nn.Transformer feed-forward, energy model)
  EPI_ROWADD_SILU,    // out(TO) = silu(acc + bias + rowvec[b(i)][j])               (energy-model head: per-sample time term)
};

struct PatchGeom {  // CaloChallengeCFM.to_patches / from_patches  (calochallenge_cfm/model.py:40-60), C = 1
  int L, A, R;      // voxel grid
  int p1, p2, p3;   // patch shape
  int l, a, r;      // patches per axis
};

constexpr int V4H_GEMM_MAX_GROUPS = 64;
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
  long slab_stride;                     // EPI_SLAB_F32: elements between the partial results of consecutive K splits
  const int* map; long V;               // EPI_UNPATCH, mapped geometry: voxel index of (token n, feature f) = map[n*P + f] (or -1), V voxels per sample
  // EPI_ATOMIC_F32, grouped output: rows [g * group_rows, (g+1) * group_rows) of the result go to the tensor group_tab[g] (its row 0) and
  // their column sums to group_tab[V4H_GEMM_MAX_GROUPS + g] - one launch for several tensors that share the Q operand (all adaLN weight
  // gradients of a step).  group_tab is a DEVICE table (kernel arguments stay small: 1 KB more of them cost 2 % of the step in launch
  // overhead); group_rows must be a multiple of 16; 0 = off.
  int group_rows;
  float* const* group_tab;
  // EPI_ATOMIC_F32 without K splits: every output element has exactly one writer - `store` makes it a plain store (the output need not be zeroed, and
  // the 35 MB of adaLN weight gradients leave the chip at the store rate instead of the float-atomic rate).  Column sums stay atomic.
  int store;
  int out_t;  // EPI_EMBED: `out` is TO-typed instead of f32 (the residual stream kept in the mode type, round 5)
};

struct GemmArgs {
  const void* P; const void* Q;
  int ldp, ldq;
  int I, J, K;
  int klen;          // K range per split (multiple of BK)
  int stagger_sleeps;  // tuning: s_sleep(127) count (8128 cycles each) for the second half of a persistent grid
  int nti, ntj, nz;  // tiles along i, along j, K splits (filled by the launcher; 1-D grid, XCD-aware decode in the kernel)
  float* colsum;     // optional: colsum[i] += sum_k P[i][k]   (f32 atomics; only j-tile 0 contributes)
  EpiArgs e;
};

// 8-wide vector helpers (one lane owns 8 consecutive output columns of a row)
struct f32x8 { float v[8]; };
// The same two functions on the 8 consecutive columns a lane owns in an epilogue.  bf16 mode: written on PAIRS of elements (ext_vector_type(2) floats =
// v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32, two elements per issue slot); only the exponential and the reciprocal stay one instruction per element.
// Per element and issue slot: 5 packed + 2 transcendental instead of 12 + 2 (training form), 3 + 2 instead of 6 + 2 (inference form) - the
// compiler packs the scalar form only in part.  log2(e) is folded into the polynomial's constants: s = 1 / (1 + 2^(x (a' + b' x^2))).
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <typename T> V4H_DEV void gelu8_and_grad(f32x8& v, f32x8& d) {
#pragma unroll
  for (int r = 0; r < 8; ++r) gelu_and_grad<T>(v.v[r], v.v[r], d.v[r]);
}
template <typename T> V4H_DEV void gelu8_only(f32x8& v) {
#pragma unroll
  for (int r = 0; r < 8; ++r) v.v[r] = gelu_only<T>(v.v[r]);
}
V4H_DEV f32x2 gelu_sigmoid2(f32x2 x, f32x2 x2) {
  const f32x2 w = x * (x2 * (-1.4426950409f * 0.0713548163f) + (-1.4426950409f * 1.5957691216f));
  f32x2 e;
  e.x = __builtin_amdgcn_exp2f(w.x);
  e.y = __builtin_amdgcn_exp2f(w.y);
  e = e + 1.0f;
  f32x2 s;
  s.x = __builtin_amdgcn_rcpf(e.x);
  s.y = __builtin_amdgcn_rcpf(e.y);
  return s;
}
template <> V4H_DEV void gelu8_and_grad<bf16>(f32x8& v, f32x8& d) {
#pragma unroll
  for (int r = 0; r < 8; r += 2) {
    const f32x2 x = {v.v[r], v.v[r + 1]};
    const f32x2 x2 = x * x;
    const f32x2 s = gelu_sigmoid2(x, x2);
    const f32x2 y = x * s;
    const f32x2 dy = (y * (1.0f - s)) * (x2 * 0.2140644489f + 1.5957691216f) + s;
    v.v[r] = y.x; v.v[r + 1] = y.y;
    d.v[r] = dy.x; d.v[r + 1] = dy.y;
  }
}
template <> V4H_DEV void gelu8_only<bf16>(f32x8& v) {
#pragma unroll
  for (int r = 0; r < 8; r += 2) {
    const f32x2 x = {v.v[r], v.v[r + 1]};
    const f32x2 y = x * gelu_sigmoid2(x, x * x);
    v.v[r] = y.x; v.v[r + 1] = y.y;
  }
}
V4H_DEV float& at(f32x8& x, int r) { return x.v[r]; }
V4H_DEV f32x8 make8(f32x4 lo, f32x4 hi) {
  f32x8 x;
#pragma unroll
  for (int r = 0; r < 4; ++r) { x.v[r] = lo[r]; x.v[4 + r] = hi[r]; }
  return x;
}
V4H_DEV f32x8 load8(const float* p) { return make8(load4(p), load4(p + 4)); }
V4H_DEV f32x8 load8(const bf16* p) {
  const bf16x8 o = *reinterpret_cast<const bf16x8*>(p);
  f32x8 x;
#pragma unroll
  for (int r = 0; r < 8; ++r) x.v[r] = (float)o[r];
  return x;
}
V4H_DEV void store8(float* p, const f32x8& x) {
  store4(p, f32x4{x.v[0], x.v[1], x.v[2], x.v[3]});
  store4(p + 4, f32x4{x.v[4], x.v[5], x.v[6], x.v[7]});
}
V4H_DEV void store8(bf16* p, const f32x8& x) {
  bf16x8 o;
#pragma unroll
  for (int r = 0; r < 8; ++r) o[r] = (bf16)x.v[r];
  *reinterpret_cast<bf16x8*>(p) = o;
}
// An output nobody reads before the backward pass (the saved GELU derivative, 66 MB per block) is stored - and later read, once - non-temporally, so
// that it does not displace what the next kernels hit in L2 / Infinity Cache: +0.7 % on the step (235.1 -> 236.8 steps/s, same box, round 3).  The
// same hint on outputs that ARE read next costs 10 % (all contraction outputs non-temporal: 234.3 -> 211.1), and on the backward's last-use reads of
// saved activations in the LayerNorm kernels -0.4 ... -1.3 %: it is right only for this tensor.
V4H_DEV void store8_saved(bf16* p, const f32x8& x) {
  bf16x8 o;
#pragma unroll
  for (int r = 0; r < 8; ++r) o[r] = (bf16)x.v[r];
  __builtin_nontemporal_store(o, reinterpret_cast<bf16x8*>(p));
}
V4H_DEV void store8_saved(float* p, const f32x8& x) { store8(p, x); }
V4H_DEV f32x8 add8(f32x8 a, const f32x8& b) {
#pragma unroll
  for (int r = 0; r < 8; ++r) a.v[r] += b.v[r];
  return a;
}

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
// two accumulator tiles a, b (lane (c, g) holds columns 4g..4g+3 of row c) -> lanes with even g hold columns 8(g>>1)..+7 of
// tile a, lanes with odd g the same columns of tile b (v_permlane16_swap: odd 16-lane rows of the first operand <-> even
// rows of the second).
// (inline assembly: the __builtin_amdgcn_permlane16_swap of this ROCm folds the four swaps of a tile pair into one - wrong code;
// the two wait states a VALU write of either operand needs before the swap reads it are not padded inside asm, hence the s_nop)
V4H_DEV f32x8 swap_pair(f32x4 a, f32x4 b) {
  f32x8 o;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float x = a[r], y = b[r];
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    o.v[r] = x;
    o.v[4 + r] = y;
  }
  return o;
}
// The same exchange on values already rounded to bf16: two swaps per tile pair instead of four (the lane permutation does not look at the contents).
// Returns the 8 consecutive columns of the lane's row as 16 bytes.
V4H_DEV u32x4 swap_pair_bf16(f32x4 a, f32x4 b) {
  typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
  unsigned x01 = __builtin_bit_cast(unsigned, bf16x2_{(bf16)a[0], (bf16)a[1]}), x23 = __builtin_bit_cast(unsigned, bf16x2_{(bf16)a[2], (bf16)a[3]});
  unsigned y01 = __builtin_bit_cast(unsigned, bf16x2_{(bf16)b[0], (bf16)b[1]}), y23 = __builtin_bit_cast(unsigned, bf16x2_{(bf16)b[2], (bf16)b[3]});
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x01), "+v"(y01));
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x23), "+v"(y23));
  return u32x4{x01, x23, y01, y23};
}
V4H_DEV u32x4 pack_bf16x8(const f32x8& x) {
  bf16x8 o;
#pragma unroll
  for (int r = 0; r < 8; ++r) o[r] = (bf16)x.v[r];
  return __builtin_bit_cast(u32x4, o);
}

// Epilogue on 8 consecutive columns j..j+7 of row i (j % 8 == 0, J % 8 == 0 checked by the launcher), in two halves so
// that the kernel can issue EVERY load of a strip before its first store: s_waitcnt vmcnt counts loads and stores in one
// in-order queue, so a load waited for after a store would drain that store's whole round trip.
template <int EPI, typename T, typename TO> struct Epilogue {
  struct Ops { f32x8 a, b; };
  static constexpr bool HAS_BIAS = EPI != EPI_DGELU && EPI != EPI_DSILU && EPI != EPI_ATOMIC_F32 && EPI != EPI_ACCUM_F32 && EPI != EPI_SLAB_F32;

  static V4H_DEV Ops load(const EpiArgs& e, int i, int j) {
    Ops o;
    if constexpr (EPI == EPI_COND_SUM) {
      if (e.resid) o.a = load8(e.resid + (size_t)i * e.ld_resid + j);
    } else if constexpr (EPI == EPI_EMBED) {
      o.a = load8(e.rowvec + (size_t)(i % e.T) * e.ld_rowvec + j);
    } else if constexpr (EPI == EPI_ROWADD_SILU) {
      o.a = load8(e.rowvec + (size_t)(i / e.T) * e.ld_rowvec + j);
    } else if constexpr (EPI == EPI_GATE_RESID) {
      o.a = load8(e.rowvec + (size_t)(i / e.T) * e.ld_rowvec + j);
      o.b = load8(e.resid + (size_t)i * e.ld_resid + j);
    } else if constexpr (EPI == EPI_DGELU) {
      o.a = load8(reinterpret_cast<const TO*>(e.aux) + (size_t)i * e.ld_aux + j);
    } else if constexpr (EPI == EPI_DSILU) {
      o.a = load8(e.auxf + (size_t)i * e.ld_auxf + j);
    } else if constexpr (EPI == EPI_ACCUM_F32) {
      o.a = load8(reinterpret_cast<const float*>(e.out) + (size_t)i * e.ldo + j);
    }
    return o;
  }

  // v = accumulator (+ bias already added by the caller)
  static V4H_DEV void finish(const EpiArgs& e, int i, int j, f32x8 v, const Ops& o) {
    if constexpr (EPI == EPI_STORE) {
      store8(reinterpret_cast<TO*>(e.out) + (size_t)i * e.ldo + j, v);
    } else if constexpr (EPI == EPI_STORE_F32 || EPI == EPI_SLAB_F32) {
      store8(reinterpret_cast<float*>(e.out) + (size_t)i * e.ldo + j, v);
    } else if constexpr (EPI == EPI_SILU) {
      if (e.out2) store8(reinterpret_cast<float*>(e.out2) + (size_t)i * e.ldo2 + j, v);
#pragma unroll
      for (int r = 0; r < 8; ++r) v.v[r] = silu_f(v.v[r]);
      store8(reinterpret_cast<TO*>(e.out) + (size_t)i * e.ldo + j, v);
    } else if constexpr (EPI == EPI_RELU) {
#pragma unroll
      for (int r = 0; r < 8; ++r) v.v[r] = fmaxf(v.v[r], 0.0f);
      store8(reinterpret_cast<TO*>(e.out) + (size_t)i * e.ldo + j, v);
    } else if constexpr (EPI == EPI_ROWADD_SILU) {
#pragma unroll
      for (int r = 0; r < 8; ++r) v.v[r] = silu_f(v.v[r] + o.a.v[r]);
      store8(reinterpret_cast<TO*>(e.out) + (size_t)i * e.ldo + j, v);
    } else if constexpr (EPI == EPI_COND_SUM) {
      if (e.resid) v = add8(v, o.a);
      store8(reinterpret_cast<float*>(e.out) + (size_t)i * e.ldo + j, v);
#pragma unroll
      for (int r = 0; r < 8; ++r) v.v[r] = silu_f(v.v[r]);
      store8(reinterpret_cast<TO*>(e.out2) + (size_t)i * e.ldo2 + j, v);
    } else if constexpr (EPI == EPI_EMBED) {
      if (e.out_t) store8(reinterpret_cast<TO*>(e.out) + (size_t)i * e.ldo + j, add8(v, o.a));
      else store8(reinterpret_cast<float*>(e.out) + (size_t)i * e.ldo + j, add8(v, o.a));
    } else if constexpr (EPI == EPI_GATE_RESID) {
      if (e.out2) store8(reinterpret_cast<TO*>(e.out2) + (size_t)i * e.ldo2 + j, v);
      f32x8 x = o.b;
#pragma unroll
      for (int r = 0; r < 8; ++r) x.v[r] += o.a.v[r] * v.v[r];
      store8(reinterpret_cast<float*>(e.out) + (size_t)i * e.ldo + j, x);
    } else if constexpr (EPI == EPI_GELU) {
      if (e.out) {  // training: the derivative is kept for the backward
        f32x8 d;
        gelu8_and_grad<T>(v, d);
        store8_saved(reinterpret_cast<TO*>(e.out) + (size_t)i * e.ldo + j, d);
      } else {
        gelu8_only<T>(v);
      }
      store8(reinterpret_cast<TO*>(e.out2) + (size_t)i * e.ldo2 + j, v);
    } else if constexpr (EPI == EPI_DGELU) {
#pragma unroll
      for (int r = 0; r < 8; ++r) v.v[r] *= o.a.v[r];
      store8(reinterpret_cast<TO*>(e.out) + (size_t)i * e.ldo + j, v);
    } else if constexpr (EPI == EPI_DSILU) {
#pragma unroll
      for (int r = 0; r < 8; ++r) v.v[r] *= dsilu_f(o.a.v[r]);
      store8(reinterpret_cast<TO*>(e.out) + (size_t)i * e.ldo + j, v);
    } else if constexpr (EPI == EPI_ACCUM_F32) {
      store8(reinterpret_cast<float*>(e.out) + (size_t)i * e.ldo + j, add8(o.a, v));
    } else if constexpr (EPI == EPI_UNPATCH) {
      // token n = (li*a + ai)*r + ri ; feature f = (pi*p2 + pj)*p3 + pk  ->  voxel (li*p1+pi, ai*p2+pj, ri*p3+pk)
      const PatchGeom& g = e.pg;
      const int b = i / e.T, n = i % e.T;
      if (e.map != nullptr) {  // general geometry: table lookup
        float* ov = reinterpret_cast<float*>(e.out) + (size_t)b * e.V;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int f = j + r;
          if (f < e.P) {
            const int vi = e.map[(size_t)n * e.P + f];
            if (vi >= 0) ov[vi] = v.v[r];
          }
        }
        return;
      }
      const int ri = n % g.r, ai = (n / g.r) % g.a, li = n / (g.r * g.a);
      float* op = reinterpret_cast<float*>(e.out) + (size_t)b * g.L * g.A * g.R;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int f = j + r;
        if (f < e.P) {
          const int pk = f % g.p3, pj = (f / g.p3) % g.p2, pi = f / (g.p3 * g.p2);
          op[((size_t)(li * g.p1 + pi) * g.A + (ai * g.p2 + pj)) * g.R + (ri * g.p3 + pk)] = v.v[r];
        }
      }
    }
  }
};

// 2-D tile copy global -> registers -> LDS in 16-byte chunks, zero-filled outside [0,rows_end) x [0,cols_end).
template <typename T, int ROWS, int COLS, int LD, int NT = 256> struct TileStage {
  static constexpr int CH = 16 / (int)sizeof(T);
  static constexpr int CPR = COLS / CH;
  static constexpr int TOTAL = ROWS * CPR;
  static constexpr int N = (TOTAL + NT - 1) / NT;
  static_assert(COLS % CH == 0, "tile columns must be whole 16-byte chunks");
  uint4 r[N];

  V4H_DEV void load(const T* g, int ld, int row0, int col0, int rows_end, int cols_end, int tid) {
#pragma unroll
    for (int n = 0; n < N; ++n) {
      const int c = tid + n * NT;
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
      const int c = tid + n * NT;
      if (c < TOTAL) {
        const int tr = c / CPR, tc = (c % CPR) * CH;
        *reinterpret_cast<uint4*>(s + tr * LD + tc) = r[n];
      }
    }
  }
};

// ---------------------------------------------------------------------------------------------------------------------
// Dense, XOR-swizzled LDS images filled by direct global->LDS DMA (global_load_lds_dwordx4: 1 KiB per wave-instruction,
// LDS destination = wave-uniform base + lane*16, no VGPR / ds_write pass).  The image is linear in DMA order; the swizzle
// is applied to the per-lane SOURCE address and again on the fragment read (same involution), never to the destination.
// Swizzles were chosen with tools/lds_model.py (bank model of MI355X_MICROARCH.md) to make every fragment read
// conflict-free:  K-contiguous rows of 4 chunks: pos = kc ^ ((row & 4) >> 1);  of 8 chunks (bf16): kc ^ (row & 6);
// of 8 chunks (f32, two chunks per lane): kc ^ (((row >> 1) & 1) | (row & 4));  K-strided bf16 (transposed read):
// pos = chunk ^ ((k & 8) >> 2).  Out-of-range chunks are fetched from a zero page, so tails need no masking later.
__device__ uint4 v4h_zero_page[4];

template <typename T, int CPR> V4H_DEV int sw_kcontig(int row) {
  if constexpr (CPR == 4) return (row & 4) >> 1;
  else if constexpr (sizeof(T) == 2) return row & (CPR - 1) & 6;
  else return ((row >> 1) & 1) | (row & 4);
}
// (rows of a multiple of 256 bytes - CPR % 16 == 0 - all start on the same bank, so the four rows q of a transposed read must be
// spread by (k & 3) as well: pos = chunk ^ 2 * ((k & 3) | ((k >> 1) & 4)), conflict-free by tools/lds_model.py)
template <typename T, int CPR = 0> V4H_DEV int sw_kstrided(int k) {
  if constexpr (sizeof(T) == 2 && CPR > 0 && CPR % 16 == 0) return 2 * ((k & 3) | ((k >> 1) & 4));
  else if constexpr (sizeof(T) == 2) return (k & 8) >> 2;
  else return 0;
}

V4H_DEV void dma16(const void* g, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Both image kinds keep, per DMA instruction of this wave, the lane's source pointer for the CURRENT K-tile in registers
// (set up once, advanced by a constant per tile), so the loop spends 2 VALU per KiB moved instead of a 64-bit
// multiply-add chain.  Lanes whose row/column lies beyond the operand extent point at row/column 0 (their products only
// reach output elements that are never stored); lanes beyond the K extent (last, partial tile only) read the zero page.

// K-contiguous operand X[idx][k]: image rows = idx (ROWS), CPR 16-byte chunks of K per row.
template <typename T, int ROWS, int BK, int NW> struct ImgKContig {
  static constexpr int CH = 16 / (int)sizeof(T), CPR = BK / CH, UNITS = ROWS * CPR, NI = UNITS / 64, NPW = (NI + NW - 1) / NW;
  static constexpr int BYTES = UNITS * 16;
  static_assert(UNITS % 64 == 0 && (CPR == 4 || CPR == 8), "image shape");
  static V4H_DEV int count(int wave) { return (NI - wave + NW - 1) / NW; }  // DMA instructions this wave issues per tile
  const char* src[NPW];
  int koff[NPW];  // lane's K offset (elements) inside a tile
  V4H_DEV void init(const T* g, int ld, int idx0, int kb, int idx_end, int wave, int lane) {
#pragma unroll
    for (int n = 0; n < NPW; ++n) {
      const int u = (wave + n * NW) * 64 + lane, row = u / CPR, pos = u % CPR;
      const int gi = idx0 + row;
      koff[n] = (pos ^ sw_kcontig<T, CPR>(row)) * CH;
      src[n] = reinterpret_cast<const char*>(g + (size_t)(gi < idx_end ? gi : 0) * ld + kb + koff[n]);
    }
  }
  // stage the tile starting at k0 into img, then advance to the next tile
  V4H_DEV void stage(char* img, int k0, int k_end, int ld, int wave) {
    // steady state (whole tile inside K): nothing but the DMA and one pointer bump per KiB; only the last, partial tile
    // takes the predicated path.  The branch is on a scalar, so the compiler must not if-convert it into per-lane selects.
    if (__builtin_expect(__builtin_amdgcn_readfirstlane((int)(k0 + BK <= k_end)), 1)) {
#pragma unroll
      for (int n = 0; n < NPW; ++n) {
        const int inst = wave + n * NW;
        if (inst < NI) dma16(src[n], img + inst * 1024);
        src[n] += BK * sizeof(T);
      }
    } else {
#pragma unroll
      for (int n = 0; n < NPW; ++n) {
        const int inst = wave + n * NW;
        if (inst < NI) {
          const void* p = (k0 + koff[n] + CH > k_end) ? (const void*)v4h_zero_page : (const void*)src[n];
          dma16(p, img + inst * 1024);
        }
        src[n] += BK * sizeof(T);
      }
    }
  }
  // canonical fragment of rows idx0.. , K slab kk..kk+31
  static V4H_DEV Frag<T> frag(const char* img, int idx0, int kk, int lane) {
    const int row = idx0 + (lane & 15), g = lane >> 4;
    Frag<T> f;
    if constexpr (sizeof(T) == 2) {
      const int kc = kk / 8 + g;
      f.v = *reinterpret_cast<const bf16x8*>(img + (row * CPR + (kc ^ sw_kcontig<T, CPR>(row))) * 16);
    } else {
      const int kc = kk / 4 + 2 * g, s = sw_kcontig<T, CPR>(row);
      const float4 a = *reinterpret_cast<const float4*>(img + (row * CPR + (kc ^ s)) * 16);
      const float4 b = *reinterpret_cast<const float4*>(img + (row * CPR + ((kc + 1) ^ s)) * 16);
      f.v[0] = a.x; f.v[1] = a.y; f.v[2] = a.z; f.v[3] = a.w;
      f.v[4] = b.x; f.v[5] = b.y; f.v[6] = b.z; f.v[7] = b.w;
    }
    return f;
  }
};

// K-strided operand X[k][idx]: image rows = k (BK), CPR chunks of idx per row.
template <typename T, int COLS, int BK, int NW> struct ImgKStrided {
  static constexpr int CH = 16 / (int)sizeof(T), CPR = COLS / CH, UNITS = BK * CPR, NI = UNITS / 64, NPW = (NI + NW - 1) / NW;
  static constexpr int BYTES = UNITS * 16;
  static_assert(UNITS % 64 == 0 && CPR % 4 == 0, "image shape");
  static V4H_DEV int count(int wave) { return (NI - wave + NW - 1) / NW; }
  const char* src[NPW];
  int krow[NPW];  // lane's row (k offset) inside a tile
  V4H_DEV void init(const T* g, int ld, int idx0, int kb, int idx_end, int wave, int lane) {
#pragma unroll
    for (int n = 0; n < NPW; ++n) {
      const int u = (wave + n * NW) * 64 + lane, row = u / CPR, pos = u % CPR;
      const int gi = idx0 + (pos ^ sw_kstrided<T, CPR>(row)) * CH;
      krow[n] = row;
      src[n] = reinterpret_cast<const char*>(g + (size_t)(kb + row) * ld + (gi + CH <= idx_end ? gi : 0));
    }
  }
  V4H_DEV void stage(char* img, int k0, int k_end, int ld, int wave) {
    const size_t bump = (size_t)BK * ld * sizeof(T);
    if (__builtin_expect(__builtin_amdgcn_readfirstlane((int)(k0 + BK <= k_end)), 1)) {
#pragma unroll
      for (int n = 0; n < NPW; ++n) {
        const int inst = wave + n * NW;
        if (inst < NI) dma16(src[n], img + inst * 1024);
        src[n] += bump;
      }
    } else {
#pragma unroll
      for (int n = 0; n < NPW; ++n) {
        const int inst = wave + n * NW;
        if (inst < NI) {
          const void* p = (k0 + krow[n] >= k_end) ? (const void*)v4h_zero_page : (const void*)src[n];
          dma16(p, img + inst * 1024);
        }
        src[n] += bump;
      }
    }
  }
  static V4H_DEV Frag<T> frag(const char* img, int idx0, int kk, int lane) {
    const int g = lane >> 4;
    Frag<T> f;
    if constexpr (sizeof(T) == 2) {
      const int q = (lane >> 2) & 3, p = lane & 3;
      const int ch = (idx0 + 4 * p) / 8, r0 = kk + 8 * g + q, r1 = r0 + 4;
      const bf16x4 lo = lds_tr_read(reinterpret_cast<const bf16*>(img + (r0 * CPR + (ch ^ sw_kstrided<T, CPR>(r0))) * 16 + 8 * (p & 1)));
      const bf16x4 hi = lds_tr_read(reinterpret_cast<const bf16*>(img + (r1 * CPR + (ch ^ sw_kstrided<T, CPR>(r1))) * 16 + 8 * (p & 1)));
      f.v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    } else {
      const int col = idx0 + (lane & 15);
#pragma unroll
      for (int j = 0; j < 8; ++j) f.v[j] = *reinterpret_cast<const float*>(img + ((kk + 8 * g + j) * CPR + col / 4) * 16 + 4 * (col & 3));
    }
    return f;
  }
};

template <typename T_, typename TO_, bool PKS_, bool QKS_, int BI_, int BJ_, int BK_, int WI_, int WJ_, int EPI_, bool COLSUM_, int DBG_ = 0, int NSTAGE_ = 2> struct GemmCfg {
  static constexpr int NSTAGE = NSTAGE_;  // LDS ring depth: 2 = one tile ahead (vmcnt(0) per step), 4 = three tiles ahead, counted vmcnt
  static constexpr int DBG = DBG_;  // ablation builds (tools/gemm_bench.py only): 1 = staging only, 2 = compute only
  using T = T_;
  using TO = TO_;
  static constexpr bool PKS = PKS_, QKS = QKS_, COLSUM = COLSUM_;
  static constexpr int BI = BI_, BJ = BJ_, BK = BK_, WI = WI_, WJ = WJ_, NW = WI_ * WJ_, NT = 64 * NW, EPI = EPI_;
  using ImgP = typename std::conditional<PKS, ImgKStrided<T, BI, BK, NW>, ImgKContig<T, BI, BK, NW>>::type;
  using ImgQ = typename std::conditional<QKS, ImgKStrided<T, BJ, BK, NW>, ImgKContig<T, BJ, BK, NW>>::type;
  static constexpr int P_BYTES = ImgP::BYTES, Q_BYTES = ImgQ::BYTES;
  static constexpr size_t LDS_BYTES = NSTAGE * (size_t)(P_BYTES + Q_BYTES);
  static_assert(NSTAGE == 2 || NSTAGE == 4, "ring depth");
  static constexpr int WTI = BI / WI, WTJ = BJ / WJ, TI = WTI / 16, TJ = WTJ / 16;
  static_assert(BI % (16 * WI) == 0 && BJ % (16 * WJ) == 0 && BK % 32 == 0, "tile shape");
};

// s_waitcnt vmcnt(n) for a run-time (wave-uniform) n: the instruction needs an immediate
V4H_DEV void wait_vmcnt(int n) {
#define V4H_VM_CASE(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
  switch (n) {
    V4H_VM_CASE(0) V4H_VM_CASE(1) V4H_VM_CASE(2) V4H_VM_CASE(3) V4H_VM_CASE(4) V4H_VM_CASE(5) V4H_VM_CASE(6) V4H_VM_CASE(7) V4H_VM_CASE(8)
    V4H_VM_CASE(9) V4H_VM_CASE(10) V4H_VM_CASE(11) V4H_VM_CASE(12) V4H_VM_CASE(13) V4H_VM_CASE(14) V4H_VM_CASE(15) V4H_VM_CASE(16)
    V4H_VM_CASE(17) V4H_VM_CASE(18) V4H_VM_CASE(19) V4H_VM_CASE(20) V4H_VM_CASE(21) V4H_VM_CASE(22) V4H_VM_CASE(23) V4H_VM_CASE(24)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
#undef V4H_VM_CASE
}

// 4-wave workgroups must fit two per CU (2 waves / SIMD, i.e. <= 256 VGPR+AGPR): several epilogue variants sit just above that
// and would silently halve their occupancy.
template <class C> __global__ __launch_bounds__(C::NT, (C::NT == 256 ? 2 : 1)) void v4h_gemm_kernel(const GemmArgs a) {
  using T = typename C::T;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BUF_BYTES = C::P_BYTES + C::Q_BYTES;  // LDS: [P0 | Q0 | P1 | Q1]; epilogue strips re-use buffer 1

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wave / C::WJ, wj = wave % C::WJ;
  const T* gP = reinterpret_cast<const T*>(a.P);
  const T* gQ = reinterpret_cast<const T*>(a.Q);

  // Virtual block id -> (i-tile, j-tile, k-split).  Workgroups are dealt round-robin over the 8 XCDs (private 4 MiB L2
  // each), so id % 8 labels the XCD (the grid is a multiple of 8).  Speed only - any placement gives the same result.
  //  * no split: XCD x owns the i-tiles (tokens) == x (mod 8) and sweeps all j-tiles of one i-tile back to back: the
  //    activation tile is fetched into that L2 once and the whole weight matrix stays L2-resident.  The workgroup is
  //    PERSISTENT: it walks virtual ids id, id + grid, ... and never waits for its output stores - a workgroup's LDS and
  //    registers are only handed to a successor once its stores have drained, which otherwise costs one store round trip
  //    per tile (measured: as long as the whole K = 480 main loop).
  //  * split-K (wgrad): split z lives on XCD z % 8 and all output tiles of a split run together, so the token rows of
  //    the split are fetched once per XCD and shared by every output tile; one tile per workgroup.
  //  * fewer than 8 i-tiles (batch-row contractions: adaLN modulations, embedders): the i-tile -> XCD pinning would leave whole XCDs
  //    idle (one i-tile = one XCD = 32 of 256 CUs), so output tiles are simply dealt round-robin over the XCDs.
  const bool few_rows = a.nz == 1 && a.nti < 8;
  const int nvirt = few_rows ? a.nti * a.ntj : (a.nz == 1 ? ((a.nti + 7) / 8) * 8 * a.ntj : a.nti * a.ntj * a.nz);
  auto decode = [&](int v, int& ti, int& tj, int& tz) -> bool {
    if (v >= nvirt) return false;
    if (few_rows) {
      ti = v % a.nti;
      tj = v / a.nti;
      tz = 0;
      return true;
    }
    if (a.nz == 1) {
      const int xcd = v & 7, slot = v >> 3;
      tj = slot % a.ntj;
      ti = (slot / a.ntj) * 8 + xcd;
      tz = 0;
      return ti < a.nti;
    }
    tz = v % a.nz;
    const int tile = v / a.nz;
    ti = tile % a.nti;
    tj = tile / a.nti;
    return true;
  };

  // Stagger: the workgroups that share a CU run the same program with the same tile time, i.e. in lockstep (both in
  // their main loop, then both in their store phase).  Holding back the second half of the persistent grid by roughly half
  // a tile lets one workgroup's epilogue overlap its partner's MFMAs.
  if (a.stagger_sleeps > 0 && a.nz == 1 && blockIdx.x >= gridDim.x / 2) {
    for (int n = 0; n < a.stagger_sleeps; ++n) __builtin_amdgcn_s_sleep(127);
  }

  typename C::ImgP stP;
  typename C::ImgQ stQ;
  int ti, tj, tz;
  bool staged = false;  // first K-tile(s) of the current virtual id already in flight / landed (issued before the previous epilogue)
  int pre_issued = 0;   // how many (ring path)
  for (int v = blockIdx.x; decode(v, ti, tj, tz); v += gridDim.x) {
  const int i0 = ti * C::BI, j0 = tj * C::BJ;
  const int kb = tz * a.klen;
  const int ke = min(a.K, kb + a.klen);
  const int nt = (ke - kb + C::BK - 1) / C::BK;
  auto stage = [&](int t, int buf) {  // tiles must be staged in order t = 0, 1, 2, ... (the stagers advance their pointers)
    const int k0 = kb + t * C::BK;
    stP.stage(smem + buf * BUF_BYTES, k0, ke, a.ldp, wave);
    stQ.stage(smem + buf * BUF_BYTES + C::P_BYTES, k0, ke, a.ldq, wave);
  };
  f32x4 acc[C::TI][C::TJ];
#pragma unroll
  for (int x = 0; x < C::TI; ++x)
#pragma unroll
    for (int y = 0; y < C::TJ; ++y) acc[x][y] = f32x4{0.f, 0.f, 0.f, 0.f};
  // Column sums of P (bias gradients of a wgrad).  Summing the P fragments costs 132 VALU instructions per K-step (bf16 unpack +
  // adds, 4 cycles each) next to 30 MFMAs (16 cycles each): done by every wave in every K-step it made the whole wgrad
  // VALU-bound.  Each (row tile, K-step) needs it exactly once, so the duty rotates over the ntj * WJ waves that hold the same
  // P fragments: wave (tj, wj) sums in the K-steps with  step % (ntj * WJ) == tj * WJ + wj  (one scalar branch per K-step),
  // in a sweep of its own so that the MFMA loop exists in one variant only (two variants of it spilled registers).
  float cs[C::TI];
#pragma unroll
  for (int x = 0; x < C::TI; ++x) cs[x] = 0.f;
  const int cs_period = a.ntj * C::WJ, cs_duty = tj * C::WJ + wj;
  int cs_phase = 0;  // K-step index modulo cs_period (every tile of a row panel walks the same K-steps)
  auto compute_impl = [&](int buf) {
    const char* tp = smem + buf * BUF_BYTES;
    const char* tq = tp + C::P_BYTES;
#pragma unroll
    for (int kk = 0; kk < (C::DBG == 1 ? 0 : C::BK); kk += 32) {  // DBG 1: staging only
      Frag<T> pf[C::TI], qf[C::TJ];
#pragma unroll
      for (int x = 0; x < C::TI; ++x) pf[x] = C::ImgP::frag(tp, wi * C::WTI + x * 16, kk, lane);
#pragma unroll
      for (int y = 0; y < C::TJ; ++y) qf[y] = C::ImgQ::frag(tq, wj * C::WTJ + y * 16, kk, lane);
#pragma unroll
      for (int x = 0; x < C::TI; ++x)
#pragma unroll
        for (int y = 0; y < C::TJ; ++y) acc[x][y] = mma(qf[y], pf[x], acc[x][y]);  // (s_setprio around the cluster measured 40 % slower here)
    }
    if constexpr (C::DBG == 7 && sizeof(T) == 2 && C::BK == 64) {
      // Scheduling experiment (tools/gemm_bench.py cfg 28): slab 0's fragment reads first, then slab 1's reads spread between slab 0's MFMAs,
      // then the remaining MFMAs - two lgkmcnt waits per K-step instead of the compiler's four full drains.  Plain-store kernels gain
      // (fc2 shape, K = 1920: +9 %; proj +4 %; qkv / fc1 and transposed-read operands: nothing), but the 18 live fragments push the
      // fused-epilogue variants over the 256-register budget (GATE_RESID: 26 spills) and the sampler ran 6 % SLOWER end to end: not enabled.
      constexpr int NR = C::TI * (C::PKS ? 2 : 1) + C::TJ * (C::QKS ? 2 : 1), NM = C::TI * C::TJ;
      __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, NM / NR, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 2 * NM - (NM / NR) * NR, 0);
    }
  };
  auto compute = [&](int buf) {  // one scalar branch per K-step, outside the MFMA cluster
    if constexpr (C::COLSUM) {
      const bool mine = a.colsum != nullptr && cs_phase == cs_duty;
      if (++cs_phase == cs_period) cs_phase = 0;
      if (mine) {  // a separate sweep over the P image (its fragments are re-read): the MFMA loop below stays one lean variant
        const char* tp = smem + buf * BUF_BYTES;
#pragma unroll
        for (int kk = 0; kk < C::BK; kk += 32)
#pragma unroll
          for (int x = 0; x < C::TI; ++x) {
            const Frag<T> pf = C::ImgP::frag(tp, wi * C::WTI + x * 16, kk, lane);
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) cs[x] += to_f32(pf.v[jj]);
          }
      }
    }
    compute_impl(buf);
  };

  if constexpr (C::NSTAGE == 2) {
    if (!staged) {
      stP.init(gP, a.ldp, i0, kb, a.I, wave, lane);
      stQ.init(gQ, a.ldq, j0, kb, a.J, wave, lane);
      if (nt > 0) stage(0, 0);
    }
    if (C::DBG == 6 && staged) {  // experiment: leave the previous tile's 12 output stores (younger than the prefetch) in flight
      wait_vmcnt(12);
      __builtin_amdgcn_s_barrier();
    } else {
      __syncthreads();  // vmcnt(0) + barrier: tile 0 has landed for every wave, previous tile's epilogue strips are dead
    }
    for (int t = 0; t < nt; ++t) {
      const int cur = t & 1;
      if (t + 1 < nt && (C::DBG != 2) && (C::DBG != 5)) stage(t + 1, cur ^ 1);  // DMA of the next tile flies during this tile's MFMAs
      compute(cur);
      __syncthreads();  // next tile landed (vmcnt(0)) and everyone is done reading the current one
    }
  } else {
    // 4-deep LDS ring, three K-tiles in flight.  Before the barrier of step t a wave waits only until ITS DMA pieces of
    // tile t have landed (counted vmcnt: the younger tiles stay in flight), the barrier then makes every wave's pieces
    // visible; tile t+3 is issued right after the barrier into the buffer tile t-1 was read from.
    const int cnt = C::ImgP::count(wave) + C::ImgQ::count(wave);
    int issued = staged ? pre_issued : 0;
    bool drain_all = staged;  // the previous tile's epilogue loads/stores sit behind the prefetch in the vmcnt queue
    if (!staged) {
      stP.init(gP, a.ldp, i0, kb, a.I, wave, lane);
      stQ.init(gQ, a.ldq, j0, kb, a.J, wave, lane);
      while (issued < nt && issued < 3) { stage(issued, issued); ++issued; }
    }
    for (int t = 0; t < nt; ++t) {
      if (drain_all) {
        wait_vmcnt(0);
        drain_all = false;
      } else {
        wait_vmcnt((issued - 1 - t) * cnt);
      }
      __builtin_amdgcn_s_barrier();
      while (issued < nt && issued <= t + 3) { stage(issued, issued & 3); ++issued; }
      compute(t & 3);
    }
    __builtin_amdgcn_s_barrier();  // every wave is done reading the ring before the epilogue strips / next prefetch overwrite it
  }

  // Prefetch the first K-tile(s) of this workgroup's NEXT output tile before the epilogue: buffer 0 (2-stage) or buffers
  // 0 and 1 (ring); the epilogue strips live in the remaining buffers.
  {
    int nti_, ntj_, ntz_;
    staged = false;
    if (C::EPI != EPI_ATOMIC_F32 && a.nz == 1 && decode(v + gridDim.x, nti_, ntj_, ntz_)) {
      stP.init(gP, a.ldp, nti_ * C::BI, 0, a.I, wave, lane);
      stQ.init(gQ, a.ldq, ntj_ * C::BJ, 0, a.J, wave, lane);
      const int ntn = (a.K + C::BK - 1) / C::BK;
      pre_issued = 0;
      for (int t = 0; t < (C::NSTAGE == 2 ? 1 : 2) && t < ntn; ++t) {
        stP.stage(smem + t * BUF_BYTES, t * C::BK, a.K, a.ldp, wave);
        stQ.stage(smem + t * BUF_BYTES + C::P_BYTES, t * C::BK, a.K, a.ldq, wave);
        ++pre_issued;
      }
      staged = true;
    }
  }

  const int c = lane & 15, g = lane >> 4;
  if constexpr (C::EPI == EPI_ATOMIC_F32) {
    // Split-K accumulation with f32 atomics.  Float atomics run at the memory side at full rate only when one
    // wave-instruction covers >= 128 contiguous bytes per row (MI355X_MICROARCH.md, Global float atomics); the MFMA
    // accumulator layout (16 rows x isolated dwords per instruction) is the ~17x slower shape.  So each wave passes its
    // tile through a private LDS strip, 16 rows at a time, and adds 64 consecutive floats of a row per instruction.
    constexpr int SLD = C::WTJ + 4;
    static_assert(C::NW * 16 * SLD * sizeof(float) <= C::LDS_BYTES, "atomic staging strip must fit the operand LDS");
    float* strip = reinterpret_cast<float*>(smem) + wave * (16 * SLD);  // no prefetch in flight on this path
#pragma unroll
    for (int x = 0; x < C::TI; ++x) {
#pragma unroll
      for (int y = 0; y < C::TJ; ++y) *reinterpret_cast<f32x4*>(strip + c * SLD + y * 16 + 4 * g) = acc[x][y];
      __syncthreads();
      const int ib = i0 + wi * C::WTI + x * 16, jb = j0 + wj * C::WTJ;
      float* outp = reinterpret_cast<float*>(a.e.out) + (size_t)ib * a.e.ldo;  // row ib of the output
      if (a.e.group_rows > 0 && ib < a.I) {
        const int gi = ib / a.e.group_rows;  // a 16-row strip never straddles two groups
        outp = a.e.group_tab[gi] + (size_t)(ib - gi * a.e.group_rows) * a.e.ldo;
      }
      const bool plain = a.e.store != 0 && a.nz == 1;  // (uniform)
      for (int id = lane; id < 16 * C::WTJ; id += 64) {
        const int row = id / C::WTJ, col = id % C::WTJ;
        if (ib + row < a.I && jb + col < a.J) {
          float* dst = outp + (size_t)row * a.e.ldo + jb + col;
          if (plain) *dst = strip[row * SLD + col];
          else atomicAdd(dst, strip[row * SLD + col]);
        }
      }
      __syncthreads();
    }
  } else if constexpr (C::DBG == 4 || C::DBG == 5) {
    // ablation (tools/gemm_bench.py): keep every MFMA live but write 16 bytes per lane per tile instead of the whole tile
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int x = 0; x < C::TI; ++x)
#pragma unroll
      for (int y = 0; y < C::TJ; ++y) sum += acc[x][y];
    const int i = i0 + wi * C::WTI + c, j = j0 + wj * C::WTJ + 4 * g;
    if (i < a.I && j < a.J) store4(reinterpret_cast<typename C::TO*>(a.e.out) + (size_t)i * a.e.ldo + j, sum);
  } else if constexpr ((C::EPI == EPI_STORE || C::EPI == EPI_GELU || C::EPI == EPI_DGELU) && sizeof(T) == 2 && sizeof(typename C::TO) == 2 && C::TJ % 2 == 1 &&
                       C::TI % 2 == 0 && C::DBG != 9) {
    // bf16 outputs straight from the registers: pairs of 16 x 16 accumulator tiles are exchanged between the 16-lane groups (v_permlane16_swap,
    // swap_pair), after which a lane holds 8 consecutive columns of a row = 16-byte accesses for the output and for every element-wise operand;
    // no LDS strip, no LDS wait between the last MFMA and the first store.  Column-tile pairs (y, y + 1) give 64 contiguous bytes per row and
    // instruction; the odd last column tile is paired over two row strips.  Measured against the LDS-strip epilogue below on the eight
    // plain-store shapes of a block: 6-9 % less time per call (tools/gemm_bench.py cfg 29 = DBG 9 = strips; profiles/r02_gemm_direct_store.txt).
    using TO = typename C::TO;
    constexpr int NCH = C::TI * C::TJ / 2;
    const int ge = g & 1, gh = g >> 1;
    const int rowA = i0 + wi * C::WTI + c, colA = j0 + wj * C::WTJ + 16 * ge + 8 * gh;
    const int colT = j0 + wj * C::WTJ + (C::TJ - 1) * 16 + 8 * gh;
    auto pos = [&](int n, int& i, int& j) {  // chunk n of this lane: row, first column
      constexpr int NP = C::TJ / 2;          // column-tile pairs per row strip
      if (n < C::TI * NP) { i = rowA + (n / NP) * 16; j = colA + (n % NP) * 32; }
      else { i = rowA + (2 * (n - C::TI * NP) + ge) * 16; j = colT; }
    };
    auto val = [&](int n) -> f32x8 {
      constexpr int NP = C::TJ / 2;
      if (n < C::TI * NP) return swap_pair(acc[n / NP][2 * (n % NP)], acc[n / NP][2 * (n % NP) + 1]);
      return swap_pair(acc[2 * (n - C::TI * NP)][C::TJ - 1], acc[2 * (n - C::TI * NP) + 1][C::TJ - 1]);
    };
    if constexpr (C::EPI == EPI_DGELU) {  // out = acc * aux: the saved gelu' of the whole wave tile is requested up front (NCH x 4 VGPRs of raw bf16)
      bf16x8 raw[NCH];
      const TO* aux = reinterpret_cast<const TO*>(a.e.aux);
#pragma unroll
      for (int n = 0; n < NCH; ++n) {
        int i, j;
        pos(n, i, j);
        if (i < a.I && j + 8 <= a.J) raw[n] = *reinterpret_cast<const bf16x8*>(aux + (size_t)i * a.e.ld_aux + j);
      }
#pragma unroll
      for (int n = 0; n < NCH; ++n) {
        int i, j;
        pos(n, i, j);
        f32x8 v = val(n);
#pragma unroll
        for (int r = 0; r < 8; ++r) v.v[r] *= (float)raw[n][r];
        if (i < a.I && j + 8 <= a.J) store8(reinterpret_cast<TO*>(a.e.out) + (size_t)i * a.e.ldo + j, v);
      }
    } else {
#pragma unroll
      for (int n = 0; n < NCH; ++n) {
        int i, j;
        pos(n, i, j);
        f32x8 v = val(n);
        const bool ok = i < a.I && j + 8 <= a.J;
        if (a.e.bias != nullptr && j + 8 <= a.J) v = add8(v, load8(a.e.bias + j));
        if constexpr (C::EPI == EPI_STORE) {
          if (ok) store8(reinterpret_cast<TO*>(a.e.out) + (size_t)i * a.e.ldo + j, v);
        } else {  // EPI_GELU: out = gelu'(pre) (training only), out2 = gelu(pre)
          if (a.e.out) {
            f32x8 d;
            gelu8_and_grad<T>(v, d);
            if (ok) store8_saved(reinterpret_cast<TO*>(a.e.out) + (size_t)i * a.e.ldo + j, d);
          } else {
            gelu8_only<T>(v);
          }
          if (ok) store8(reinterpret_cast<TO*>(a.e.out2) + (size_t)i * a.e.ldo2 + j, v);
        }
      }
    }
  } else if constexpr (C::EPI == EPI_SLAB_F32 && sizeof(T) == 2 && C::DBG != 9) {
    // f32 split-K partials straight from the registers: a lane's 4 accumulator registers are 4 consecutive columns = one 16-byte store, 64
    // contiguous bytes per row and instruction; no exchange needed
    float* slab = reinterpret_cast<float*>(a.e.out) + (size_t)tz * a.e.slab_stride;
#pragma unroll
    for (int x = 0; x < C::TI; ++x)
#pragma unroll
      for (int y = 0; y < C::TJ; ++y) {
        const int i = i0 + wi * C::WTI + x * 16 + c, j = j0 + wj * C::WTJ + y * 16 + 4 * g;
        if (i < a.I && j + 4 <= a.J) store4(slab + (size_t)i * a.e.ldo + j, acc[x][y]);
      }
  } else if constexpr (C::DBG != 3) {
    // Every other epilogue: the MFMA accumulator layout (a lane holds 4 columns of ONE row, 16 rows per instruction) makes
    // 8-byte stores into 16 different cache lines.  Instead each wave passes its tile through a private LDS strip, 16 rows
    // at a time, and each lane then owns 8 CONSECUTIVE columns of a row: 16/32-byte accesses, WTJ*sizeof contiguous per row,
    // for the output and for every epilogue operand (bias, residual, gate, pre-activation).
    constexpr int FREE = (int)C::LDS_BYTES - (C::NSTAGE / 2) * BUF_BYTES;  // strips live in the ring half the prefetch does not touch
    constexpr int SLD = C::WTJ + ((C::NW * 16 * (C::WTJ + 4) * 4 <= FREE) ? 4 : 0), CPRW = C::WTJ / 8, NCH = 16 * CPRW, NIT = (NCH + 63) / 64;
    using Epi = Epilogue<C::EPI, T, typename C::TO>;
    EpiArgs ea = a.e;
    if constexpr (C::EPI == EPI_SLAB_F32) ea.out = reinterpret_cast<float*>(a.e.out) + (size_t)tz * a.e.slab_stride;
    static_assert(C::WTJ % 8 == 0, "wave tile width must be a multiple of 8");
    constexpr int STRIP_OFF = (C::NSTAGE / 2) * BUF_BYTES;  // buffers not targeted by the prefetch above
    static_assert(C::NW * 16 * SLD * sizeof(float) <= C::LDS_BYTES - STRIP_OFF, "epilogue staging strips must fit the free half of the ring");
    float* strip = reinterpret_cast<float*>(smem + STRIP_OFF) + wave * (16 * SLD);
    const int jb = j0 + wj * C::WTJ;
    int rrow[NIT], rcol[NIT];
    bool cok[NIT];
    f32x8 bias8[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {  // a lane's chunk columns (and their bias) are the same for every strip
      const int id = lane + it * 64;
      rrow[it] = id / CPRW;
      rcol[it] = (id % CPRW) * 8;
      cok[it] = id < NCH && jb + rcol[it] < a.J;
#pragma unroll
      for (int r = 0; r < 8; ++r) bias8[it].v[r] = 0.f;
      if constexpr (Epi::HAS_BIAS) {
        if (a.e.bias != nullptr && cok[it]) bias8[it] = load8(a.e.bias + jb + rcol[it]);
      }
    }
    // EPI_DGELU (16-bit operands): the saved gelu' of the WHOLE wave tile is requested up front, as raw 16-byte chunks (TI * NIT * 4 VGPRs):
    // one exposed round trip per tile instead of one per 16-row strip; the strips are then handled chunk by chunk (no staging arrays).
    constexpr bool HOIST = C::EPI == EPI_DGELU && sizeof(typename C::TO) == 2 && C::DBG != 8;
    if constexpr (HOIST) {
      bf16x8 raw[C::TI][NIT];
#pragma unroll
      for (int x = 0; x < C::TI; ++x)
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const int ib = i0 + wi * C::WTI + x * 16;
          if (cok[it] && ib + rrow[it] < a.I)
            raw[x][it] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(ea.aux) + (size_t)(ib + rrow[it]) * ea.ld_aux + jb + rcol[it]));
        }
#pragma unroll
      for (int x = 0; x < C::TI; ++x) {
#pragma unroll
        for (int y = 0; y < C::TJ; ++y) *reinterpret_cast<f32x4*>(strip + c * SLD + y * 16 + 4 * g) = acc[x][y];
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
        const int ib = i0 + wi * C::WTI + x * 16;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          if (cok[it] && ib + rrow[it] < a.I) {
            f32x8 v = make8(*reinterpret_cast<const f32x4*>(strip + rrow[it] * SLD + rcol[it]), *reinterpret_cast<const f32x4*>(strip + rrow[it] * SLD + rcol[it] + 4));
#pragma unroll
            for (int r = 0; r < 8; ++r) v.v[r] *= (float)raw[x][it][r];
            store8(reinterpret_cast<typename C::TO*>(ea.out) + (size_t)(ib + rrow[it]) * ea.ldo + jb + rcol[it], v);
          }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
      }
    } else if constexpr (C::EPI == EPI_GATE_RESID && sizeof(T) == 2 && C::DBG != 8) {  // (f32 mode: fragments twice the size, no registers to spare)
      // the f32 residual rows of the NEXT strip are requested before the current strip is processed (two buffers of NIT chunks)
      f32x8 res[2][NIT];
      auto request = [&](int x, f32x8* dst) {
        const int ibx = i0 + wi * C::WTI + x * 16;
#pragma unroll
        for (int it = 0; it < NIT; ++it)
          if (cok[it] && ibx + rrow[it] < a.I) dst[it] = load8(ea.resid + (size_t)(ibx + rrow[it]) * ea.ld_resid + jb + rcol[it]);
      };
      request(0, res[0]);
#pragma unroll
      for (int x = 0; x < C::TI; ++x) {
#pragma unroll
        for (int y = 0; y < C::TJ; ++y) *reinterpret_cast<f32x4*>(strip + c * SLD + y * 16 + 4 * g) = acc[x][y];
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
        if (x + 1 < C::TI) request(x + 1, res[(x + 1) & 1]);
        const int ib = i0 + wi * C::WTI + x * 16;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          if (cok[it] && ib + rrow[it] < a.I) {
            const int i = ib + rrow[it], j = jb + rcol[it];
            const f32x8 v = add8(make8(*reinterpret_cast<const f32x4*>(strip + rrow[it] * SLD + rcol[it]),
                                       *reinterpret_cast<const f32x4*>(strip + rrow[it] * SLD + rcol[it] + 4)), bias8[it]);
            const f32x8 gate = load8(ea.rowvec + (size_t)(i / ea.T) * ea.ld_rowvec + j);
            if (ea.out2) store8(reinterpret_cast<typename C::TO*>(ea.out2) + (size_t)i * ea.ldo2 + j, v);
            f32x8 xn = res[x & 1][it];
#pragma unroll
            for (int r = 0; r < 8; ++r) xn.v[r] += gate.v[r] * v.v[r];
            store8(reinterpret_cast<float*>(ea.out) + (size_t)i * ea.ldo + j, xn);
          }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
      }
    } else {
#pragma unroll
    for (int x = 0; x < C::TI; ++x) {
#pragma unroll
      for (int y = 0; y < C::TJ; ++y) *reinterpret_cast<f32x4*>(strip + c * SLD + y * 16 + 4 * g) = acc[x][y];
      __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0) only: LDS strip visible to the wave; global stores stay in flight
      __builtin_amdgcn_wave_barrier();
      const int ib = i0 + wi * C::WTI + x * 16;
      f32x8 v[NIT];
      typename Epi::Ops ops[NIT];
#pragma unroll
      for (int it = 0; it < NIT; ++it) {  // phase 1: LDS reads + every global load of the strip
        if (cok[it] && ib + rrow[it] < a.I) {
          v[it] = add8(make8(*reinterpret_cast<const f32x4*>(strip + rrow[it] * SLD + rcol[it]),
                             *reinterpret_cast<const f32x4*>(strip + rrow[it] * SLD + rcol[it] + 4)), bias8[it]);
          ops[it] = Epi::load(ea, ib + rrow[it], jb + rcol[it]);
        }
      }
#pragma unroll
      for (int it = 0; it < NIT; ++it) {  // phase 2: math + stores
        if (cok[it] && ib + rrow[it] < a.I) Epi::finish(ea, ib + rrow[it], jb + rcol[it], v[it], ops[it]);
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_wave_barrier();
    }
    }
  }
  if constexpr (C::COLSUM) {
    if (a.colsum != nullptr) {
#pragma unroll
      for (int x = 0; x < C::TI; ++x) {
        float s = cs[x];
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        const int i = i0 + wi * C::WTI + x * 16 + c;
        float* dst = a.colsum + i;
        if (C::EPI == EPI_ATOMIC_F32 && a.e.group_rows > 0 && i < a.I) dst = a.e.group_tab[V4H_GEMM_MAX_GROUPS + i / a.e.group_rows] + i % a.e.group_rows;
        if (g == 0 && i < a.I) atomicAdd(dst, s);
      }
    }
  }
  }  // persistent tile loop
}

template <class C> int v4h_gemm_launch(GemmArgs a, int splitk, hipStream_t stream, const char* name) {
  V4H_CHECK_ARG(a.I > 0 && a.J > 0 && a.K > 0, "%s: empty problem I=%d J=%d K=%d", name, a.I, a.J, a.K);
  constexpr int CH = 16 / (int)sizeof(typename C::T);
  V4H_CHECK_ARG(a.J % 8 == 0 || C::EPI == EPI_ATOMIC_F32, "%s: J=%d must be a multiple of 8", name, a.J);
  V4H_CHECK_ARG(a.ldp % CH == 0 && a.ldq % CH == 0, "%s: operand row strides (%d,%d) must be whole 16-byte chunks", name, a.ldp, a.ldq);
  V4H_CHECK_ARG(C::PKS ? (a.I % CH == 0) : (a.K % CH == 0), "%s: P extent not a whole number of 16-byte chunks", name);
  V4H_CHECK_ARG(C::QKS ? (a.J % CH == 0) : (a.K % CH == 0), "%s: Q extent not a whole number of 16-byte chunks", name);
  V4H_CHECK_ARG(((uintptr_t)a.P % 16) == 0 && ((uintptr_t)a.Q % 16) == 0, "%s: operands must be 16-byte aligned", name);
  if (splitk < 1) splitk = 1;
  if (C::EPI != EPI_ATOMIC_F32 && C::EPI != EPI_SLAB_F32) splitk = 1;
  int klen = (a.K + splitk - 1) / splitk;
  klen = (klen + C::BK - 1) / C::BK * C::BK;
  a.klen = klen;
  a.nz = (a.K + klen - 1) / klen;
  a.nti = (a.I + C::BI - 1) / C::BI;
  a.ntj = (a.J + C::BJ - 1) / C::BJ;
  long nblocks = a.nz == 1 ? (a.nti < 8 ? (long)a.nti * a.ntj : (long)((a.nti + 7) / 8) * 8 * a.ntj) : (long)a.nti * a.ntj * a.nz;
  V4H_CHECK_ARG(nblocks < (1L << 31), "%s: grid too large", name);
  if (a.nz == 1) {  // persistent workgroups: as many as are co-resident (256 CUs x workgroups per CU by LDS), a multiple of 8
    const long per_cu = (160 * 1024) / (long)C::LDS_BYTES > 0 ? (160 * 1024) / (long)C::LDS_BYTES : 1;
    const long resident = v4h_compute_units() * (per_cu > 2 ? 2 : per_cu);
    if (nblocks > resident) nblocks = resident;
  }
  dim3 grid((unsigned)nblocks);
  if constexpr (C::LDS_BYTES > 48 * 1024) {
    static DeviceOnce lds_attr;  // the attribute belongs to the function object of ONE device
    if (int rc = lds_attr.ensure([&]() -> hipError_t {
          return hipFuncSetAttribute(reinterpret_cast<const void*>(&v4h_gemm_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES);
        }, name, "reserve the LDS of the tile")) return rc;
  }
  V4H_LAUNCH(v4h_gemm_kernel<C>, grid, dim3(C::NT), C::LDS_BYTES, stream, a);
  V4H_CHECK_LAUNCH(name);
  return V4H_OK;
}
