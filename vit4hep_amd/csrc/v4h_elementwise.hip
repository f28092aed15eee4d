// Memory-bound pieces of the ViT-CFM path: patch gather/scatter, positional / timestep embeddings, LayerNorm +
// adaLN modulate (forward and backward, with the gate backward fused in), weight cast/pad, the CFM trajectory and
// loss, gradient norm, AdamW and the ODE-solver vector updates.  All HBM-bound: vectorised 16-byte accesses where
// the layout allows, one wave per LayerNorm row, wavefront-shuffle row reductions, f32 statistics.
#include <stdlib.h>

#include "v4h_ops.h"

namespace v4h {
namespace {

template <typename T> V4H_DEV void st1(T* p, float v) { *p = (T)v; }

// ------------------------------------------------------------------------------------------------ cast / pad
constexpr int MAX_ITEMS = 48;
struct CastPadTable {
  CastPadItem it[MAX_ITEMS];
  int first_block[MAX_ITEMS + 1];
  int n;
};

template <typename T> __global__ void cast_pad_kernel(const CastPadTable tb) {
  int e = 0;
  while (e + 1 < tb.n && (int)blockIdx.x >= tb.first_block[e + 1]) ++e;
  const CastPadItem it = tb.it[e];
  const long total = (long)it.Rp * it.Cp;
  T* dst = reinterpret_cast<T*>(it.dst);
  float* dstf = reinterpret_cast<float*>(it.dst);
  if (it.R == it.Rp && it.C == it.Cp && !it.dst_f32 && (total & 3) == 0) {  // plain cast (almost all weights): no index arithmetic, 16-byte loads
    for (long idx = (long)(blockIdx.x - tb.first_block[e]) * blockDim.x * 4 + threadIdx.x * 4; idx < total;
         idx += (long)(tb.first_block[e + 1] - tb.first_block[e]) * blockDim.x * 4)
      store4(dst + idx, load4(it.src + idx));
    return;
  }
  for (long idx = (long)(blockIdx.x - tb.first_block[e]) * blockDim.x * 4 + threadIdx.x * 4; idx < total;
       idx += (long)(tb.first_block[e + 1] - tb.first_block[e]) * blockDim.x * 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long k = idx + u;
      if (k < total) {
        const int r = (int)(k / it.Cp), c = (int)(k % it.Cp);
        const float val = (r < it.R && c < it.C) ? it.src[(long)r * it.C + c] : 0.0f;
        if (it.dst_f32) dstf[k] = val;
        else dst[k] = (T)val;
      }
    }
  }
}

__global__ void unpad_kernel(const float* __restrict__ src, int ld_src, float* __restrict__ dst, int R, int C) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < R * C) dst[idx] += src[(long)(idx / C) * ld_src + idx % C];
}

// ------------------------------------------------------------------------------------------------ patching
// CaloChallengeCFM.to_patches (calochallenge_cfm/model.py:54-60): token n=(li*a+ai)*r+ri, feature f=(pi*p2+pj)*p3+pk.
// One thread per voxel, voxel index fastest -> coalesced reads; writes land in the token row (<= Ppad apart).
template <typename T> __global__ void patchify_kernel(const float* __restrict__ vox, T* __restrict__ xp, int B, PatchGeom g, int P, int Ppad) {
  const long nvox = (long)g.L * g.A * g.R;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * nvox) return;
  const int b = (int)(idx / nvox);
  const int v = (int)(idx % nvox);
  const int x = v % g.R, y = (v / g.R) % g.A, z = v / (g.R * g.A);
  const int li = z / g.p1, pi = z % g.p1, ai = y / g.p2, pj = y % g.p2, ri = x / g.p3, pk = x % g.p3;
  const int n = (li * g.a + ai) * g.r + ri, f = (pi * g.p2 + pj) * g.p3 + pk;
  const int Tn = g.l * g.a * g.r;
  xp[((long)b * Tn + n) * Ppad + f] = (T)vox[idx];
}
// The same through LDS, as a regular grid allows: the p1 voxel layers that make up one row of patches are ONE contiguous block of the input
// (p1 * A * R floats of sample b, starting at layer li * p1).  A workgroup reads its block of (layer, eta, phi) voxels with coalesced 16-byte
// loads into LDS and writes the a * r token rows of that block, each lane 8 consecutive features of a token (16 bytes of bf16), zero padding
// included - so neither side of the transpose touches HBM with a stride.
template <typename T> __global__ __launch_bounds__(256) void patchify_slab_kernel(const float* __restrict__ vox, T* __restrict__ xp, PatchGeom g, int P, int Ppad) {
  extern __shared__ __attribute__((aligned(16))) float slab[];
  const int li = blockIdx.x, b = blockIdx.y;
  const int S = g.p1 * g.A * g.R;
  const float* src = vox + ((long)b * g.L + (long)li * g.p1) * g.A * g.R;
  if ((S & 3) == 0 && ((uintptr_t)src & 15) == 0) {
    for (int k = threadIdx.x; k < S / 4; k += blockDim.x) reinterpret_cast<f32x4*>(slab)[k] = reinterpret_cast<const f32x4*>(src)[k];
  } else {
    for (int k = threadIdx.x; k < S; k += blockDim.x) slab[k] = src[k];
  }
  __syncthreads();
  const int ntok = g.a * g.r, Tn = g.l * ntok, cpr = Ppad / 8;  // 8-feature chunks per token row
  T* dst = xp + ((long)b * Tn + (long)li * ntok) * Ppad;
  for (int e = threadIdx.x; e < ntok * cpr; e += blockDim.x) {
    const int tok = e / cpr, f0 = (e % cpr) * 8;
    const int ai = tok / g.r, ri = tok % g.r;
    f32x8 v;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int f = f0 + k;
      float val = 0.f;
      if (f < P) {
        const int pk = f % g.p3, pj = (f / g.p3) % g.p2, pi = f / (g.p3 * g.p2);
        val = slab[(pi * g.A + ai * g.p2 + pj) * g.R + ri * g.p3 + pk];
      }
      v.v[k] = val;
    }
    store8(dst + (long)tok * Ppad + f0, v);
  }
}
template <typename T> __global__ void zero_pad_cols_kernel(T* __restrict__ xp, long rows, int P, int Ppad) {
  const int w = Ppad - P;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < rows * w) xp[(idx / w) * Ppad + P + idx % w] = (T)0.0f;
}
__global__ void unpatchify_kernel(const float* __restrict__ tok, int ld, float* __restrict__ vox, int B, PatchGeom g, int P) {
  const long nvox = (long)g.L * g.A * g.R;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * nvox) return;
  const int b = (int)(idx / nvox);
  const int v = (int)(idx % nvox);
  const int x = v % g.R, y = (v / g.R) % g.A, z = v / (g.R * g.A);
  const int li = z / g.p1, pi = z % g.p1, ai = y / g.p2, pj = y % g.p2, ri = x / g.p3, pk = x % g.p3;
  const int n = (li * g.a + ai) * g.r + ri, f = (pi * g.p2 + pj) * g.p3 + pk;
  const int Tn = g.l * g.a * g.r;
  vox[idx] = tok[((long)b * Tn + n) * ld + f];
}

// General geometry: gather / scatter through an index table (multi-segment patching of the DS1 / CaloGAN / CaloHad wrappers:
// torch.split by list_edges + per-segment rearrange + cat, e.g. calochallenge_cfm/model.py:143-173).
template <typename TO> __global__ void patchify_map_kernel(const float* __restrict__ vox, const int* __restrict__ map, TO* __restrict__ xp, int B, long V, int Tn, int P,
                                                          int Ppad) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long per = (long)Tn * Ppad;
  if (idx >= (long)B * per) return;
  const int b = (int)(idx / per);
  const int rem = (int)(idx % per), n = rem / Ppad, f = rem % Ppad;
  float v = 0.f;
  if (f < P) {
    const int vi = map[(long)n * P + f];
    if (vi >= 0) v = vox[(long)b * V + vi];
  }
  xp[idx] = (TO)v;
}
__global__ void unpatchify_map_kernel(const float* __restrict__ tok, int ld, const int* __restrict__ map, float* __restrict__ vox, int B, long V, int Tn, int P) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long per = (long)Tn * P;
  if (idx >= (long)B * per) return;
  const int b = (int)(idx / per);
  const int rem = (int)(idx % per), n = rem / P, f = rem % P;
  const int vi = map[rem];
  if (vi >= 0) vox[(long)b * V + vi] = tok[((long)b * Tn + n) * ld + f];
}
// positional embedding from explicit position buffers pos = [pos_x | pos_y | pos_z] (T each)
__global__ void pos_embed_fwd_pos_kernel(const float* __restrict__ freqs, const float* __restrict__ pos, float* __restrict__ pe, int Tn, int D) {
  const int nf = D / 6;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= Tn * 3 * nf) return;
  const int n = idx / (3 * nf), rem = idx % (3 * nf), axis = rem / nf, j = rem % nf;
  const float arg = pos[axis * Tn + n] * (freqs[j] * 6.283185307179586f);
  pe[(long)n * D + axis * 2 * nf + j] = sinf(arg);
  pe[(long)n * D + axis * 2 * nf + nf + j] = cosf(arg);
}
__global__ void pos_embed_bwd_pos_kernel(const float* __restrict__ G, const float* __restrict__ freqs, const float* __restrict__ pos, float* __restrict__ dfreqs, int Tn,
                                         int D) {
  const int nf = D / 6;
  const int j = blockIdx.x;
  const float w = freqs[j] * 6.283185307179586f;
  float s = 0.f;
  for (int n = threadIdx.x; n < Tn; n += blockDim.x) {
#pragma unroll
    for (int axis = 0; axis < 3; ++axis) {
      const float ps = pos[axis * Tn + n];
      const float arg = ps * w;
      const float gs = G[(long)n * D + axis * 2 * nf + j], gc = G[(long)n * D + axis * 2 * nf + nf + j];
      s += ps * (gs * cosf(arg) - gc * sinf(arg));
    }
  }
  s = wave_sum(s);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
    for (int w2 = 0; w2 < (int)(blockDim.x >> 6); ++w2) tot += red[w2];
    atomicAdd(dfreqs + j, tot * 6.283185307179586f);
  }
}

// ------------------------------------------------------------------------------------------------ embeddings
// ViT.learnable_pos_embedding (nn/vit.py:156-162) with create_meshgrid buffers (nn/vit.py:137-154), single segment:
// pe[n] = [sin(px w), cos(px w), sin(py w), cos(py w), sin(pz w), cos(pz w)], w = 2 pi freqs, each D/6 wide.
__global__ void pos_embed_fwd_kernel(const float* __restrict__ freqs, float* __restrict__ pe, PatchGeom g, int D) {
  const int nf = D / 6;
  const int Tn = g.l * g.a * g.r;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= Tn * 3 * nf) return;
  const int n = idx / (3 * nf), rem = idx % (3 * nf), axis = rem / nf, j = rem % nf;
  const int ri = n % g.r, ai = (n / g.r) % g.a, li = n / (g.r * g.a);
  const float pos = axis == 0 ? (float)ri / (float)g.r : (axis == 1 ? (float)ai / (float)g.a : (float)li / (float)g.l);
  const float w = freqs[j] * 6.283185307179586f;
  const float arg = pos * w;
  pe[(long)n * D + axis * 2 * nf + j] = sinf(arg);
  pe[(long)n * D + axis * 2 * nf + nf + j] = cosf(arg);
}
// stage 1: G[n][d] += sum_b dx0[b][n][d]   (G zeroed by the caller).  A thread owns 8 consecutive features (one 16-byte load per sample in
// bf16 mode) of a chunk of the batch (gridDim.y chunks: enough workgroups to fill the chip) and adds its partial sums with f32 atomics.
// [one thread per feature looping over the whole batch with 2-byte loads: 33-36 us for 16.6 MB]
template <typename T> __global__ void sum_over_batch_kernel(const T* __restrict__ dx0, float* __restrict__ G, int B, int TD, int bchunk) {
  const int i8 = (blockIdx.x * blockDim.x + threadIdx.x) * 8;
  const int b0 = blockIdx.y * bchunk, b1 = min(B, b0 + bchunk);
  f32x8 s;
#pragma unroll
  for (int k = 0; k < 8; ++k) s.v[k] = 0.f;
  if (i8 < TD) {
#pragma unroll 4
    for (int b = b0; b < b1; ++b) s = add8(s, load8(dx0 + (long)b * TD + i8));
  }
  // float atomics run at full rate only when a wave-instruction covers consecutive addresses (a lane adding its own 8 consecutive sums makes
  // every instruction a stride-8 scatter: 61 us instead of 33 for the kernel this one replaced): re-order the block's 2048 sums through LDS
  __shared__ float sm[256 * 8];
#pragma unroll
  for (int k = 0; k < 8; ++k) sm[threadIdx.x * 8 + k] = s.v[k];
  __syncthreads();
  const int base = blockIdx.x * blockDim.x * 8;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int idx = k * 256 + threadIdx.x;
    if (base + idx < TD) atomicAdd(G + base + idx, sm[idx]);
  }
}
// any width: one thread per feature (G zeroed by the caller as well)
template <typename T> __global__ void sum_over_batch_scalar_kernel(const T* __restrict__ dx0, float* __restrict__ G, int B, int TD) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= TD) return;
  float s = 0.f;
  for (int b = 0; b < B; ++b) s += to_f32(dx0[(long)b * TD + idx]);
  G[idx] += s;
}
template <typename T> void sum_over_batch(const void* dx0, float* G, int B, int TD, hipStream_t s) {
  if (TD % 8 != 0) {
    hipLaunchKernelGGL(sum_over_batch_scalar_kernel<T>, dim3((TD + 255) / 256), dim3(256), 0, s, (const T*)dx0, G, B, TD);
    return;
  }
  const int nbx = (TD / 8 + 255) / 256;
  int chunks = (1024 + nbx - 1) / nbx;  // ~1000 workgroups
  if (chunks > B) chunks = B;
  const int bchunk = (B + chunks - 1) / chunks;
  hipLaunchKernelGGL(sum_over_batch_kernel<T>, dim3(nbx, (B + bchunk - 1) / bchunk), dim3(256), 0, s, (const T*)dx0, G, B, TD, bchunk);
}
// stage 2: dfreqs[j] += 2 pi sum_n sum_axis pos * (G_sin * cos(arg) - G_cos * sin(arg)); one block per j
__global__ void pos_embed_bwd_kernel(const float* __restrict__ G, const float* __restrict__ freqs, float* __restrict__ dfreqs, PatchGeom g, int D) {
  const int nf = D / 6;
  const int j = blockIdx.x;
  const int Tn = g.l * g.a * g.r;
  const float w = freqs[j] * 6.283185307179586f;
  float s = 0.f;
  for (int n = threadIdx.x; n < Tn; n += blockDim.x) {
    const int ri = n % g.r, ai = (n / g.r) % g.a, li = n / (g.r * g.a);
    const float pos[3] = {(float)ri / (float)g.r, (float)ai / (float)g.a, (float)li / (float)g.l};
#pragma unroll
    for (int axis = 0; axis < 3; ++axis) {
      const float arg = pos[axis] * w;
      const float gs = G[(long)n * D + axis * 2 * nf + j], gc = G[(long)n * D + axis * 2 * nf + nf + j];
      s += pos[axis] * (gs * cosf(arg) - gc * sinf(arg));
    }
  }
  s = wave_sum(s);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
    for (int w2 = 0; w2 < (int)(blockDim.x >> 6); ++w2) tot += red[w2];
    atomicAdd(dfreqs + j, tot * 6.283185307179586f);
  }
}
// TimestepEmbedder.timestep_embedding (nn/vit.py:368-389): [cos(t f_i), sin(t f_i)], f_i = exp(-ln(1e4) i / half)
template <typename T> __global__ void timestep_embed_kernel(const float* __restrict__ t, T* __restrict__ out, int B, int F) {
  const int half = F / 2;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * half) return;
  const int b = idx / half, i = idx % half;
  const float f = expf(-9.210340371976184f * (float)i / (float)half);
  const float arg = t[b] * f;
  out[(long)b * F + i] = (T)cosf(arg);
  out[(long)b * F + half + i] = (T)sinf(arg);
}

// ------------------------------------------------------------------------------------------------ LayerNorm + modulate
// nn.LayerNorm(D, elementwise_affine=False, eps=1e-6) then modulate (nn/vit.py:309,457-458).  One wave per token row;
// lane owns float4 groups lane*4 + 256*n.  Two-pass statistics in registers.
// NV = float4 groups per lane (D <= 256 * NV); the kernels are instantiated for NV = 2 (D = 480) and 4
template <typename T, int LN_MAXV> __global__ __launch_bounds__(256) void ln_modulate_fwd_kernel(const float* __restrict__ x, const float* __restrict__ shift,
                                                                                     const float* __restrict__ scale, int ld_mod, T* __restrict__ u,
                                                                                     float* __restrict__ mean, float* __restrict__ rstd, int BT, int Tn, int D) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= BT) return;
  const int b = row / Tn;
  const float* xr = x + (long)row * D;
  f32x4 v[LN_MAXV];
  float s = 0.f;
#pragma unroll
  for (int n = 0; n < LN_MAXV; ++n) {
    const int c = lane * 4 + 256 * n;
    v[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (c < D) {
      v[n] = load4(xr + c);
      s += v[n][0] + v[n][1] + v[n][2] + v[n][3];
    }
  }
  const float mu = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int n = 0; n < LN_MAXV; ++n) {
    const int c = lane * 4 + 256 * n;
    if (c < D) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = v[n][r] - mu;
        q += d * d;
      }
    }
  }
  const float rs = 1.0f / sqrtf(wave_sum(q) / (float)D + 1e-6f);
  if (lane == 0) {
    if (mean) mean[row] = mu;
    if (rstd) rstd[row] = rs;
  }
#pragma unroll
  for (int n = 0; n < LN_MAXV; ++n) {
    const int c = lane * 4 + 256 * n;
    if (c < D) {
      const f32x4 sh = load4(shift + (long)b * ld_mod + c), sc = load4(scale + (long)b * ld_mod + c);
      f32x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (v[n][r] - mu) * rs * (1.0f + sc[r]) + sh[r];
      store4(u + (long)row * D + c, o);
    }
  }
}

// Backward of LN+modulate fused with the gate backward of the branch below.  Grid (chunks, B): a workgroup owns
// ROWS_PER_WG consecutive tokens of ONE sample, so the per-sample sums (dshift, dscale, dgate) are reduced in
// registers -> LDS -> one f32 atomic per feature per workgroup.
template <typename T, int LN_MAXV, int LNB_ROWS, int R> __global__ __launch_bounds__(256) void ln_modulate_bwd_kernel(const LnBwdArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * LNB_ROWS;
  const int t1 = min(a.T, t0 + LNB_ROWS);
  const int D = a.D;
  const T* du = reinterpret_cast<const T*>(a.du);
  const T* y = reinterpret_cast<const T*>(a.y);
  f32x4 acc_sh[LN_MAXV], acc_sc[LN_MAXV], acc_g[LN_MAXV], sc[LN_MAXV], gt[LN_MAXV];
#pragma unroll
  for (int n = 0; n < LN_MAXV; ++n) {
    const int c = lane * 4 + 256 * n;
    acc_sh[n] = acc_sc[n] = acc_g[n] = sc[n] = gt[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (c < D) {
      sc[n] = load4(a.scale + (long)b * a.ld_mod + c);
      if (y) gt[n] = load4(a.gate + (long)b * a.ld_mod_gate + c);
    }
  }
  // R token rows per iteration: all loads of these rows are issued before the first reduction (memory-level parallelism)
  for (int tb = t0 + wave * R; tb < t1; tb += 4 * R) {
    long row[R];
    bool ok[R];
    float mu[R], rs[R], s1[R], s2[R];
    f32x4 gy[R][LN_MAXV], xh[R][LN_MAXV], dxi[R][LN_MAXV], yv[R][LN_MAXV];
#pragma unroll
    for (int q = 0; q < R; ++q) {
      ok[q] = tb + q < t1;
      row[q] = (long)b * a.T + (ok[q] ? tb + q : t0);
      mu[q] = a.mean[row[q]];
      rs[q] = a.rstd[row[q]];
      s1[q] = s2[q] = 0.f;
#pragma unroll
      for (int n = 0; n < LN_MAXV; ++n) {
        const int c = lane * 4 + 256 * n;
        gy[q][n] = xh[q][n] = dxi[q][n] = yv[q][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c < D && ok[q]) {
          gy[q][n] = load4(du + row[q] * D + c);  // holds du for now
          xh[q][n] = load4(reinterpret_cast<const float*>(a.x) + row[q] * D + c);  // holds x for now
          if (a.dx_in) dxi[q][n] = load4(reinterpret_cast<const float*>(a.dx_in) + row[q] * D + c);
          if (y) yv[q][n] = load4(y + row[q] * D + c);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < R; ++q) {
#pragma unroll
      for (int n = 0; n < LN_MAXV; ++n) {
        const int c = lane * 4 + 256 * n;
        if (c < D && ok[q]) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float d = gy[q][n][r];
            const float xhat = (xh[q][n][r] - mu[q]) * rs[q];
            xh[q][n][r] = xhat;
            gy[q][n][r] = d * (1.0f + sc[n][r]);
            acc_sh[n][r] += d;
            acc_sc[n][r] += d * xhat;
            s1[q] += gy[q][n][r];
            s2[q] += gy[q][n][r] * xhat;
          }
        }
      }
    }
#pragma unroll
    for (int q = 0; q < R; ++q) {
      s1[q] = wave_sum(s1[q]) / (float)D;
      s2[q] = wave_sum(s2[q]) / (float)D;
    }
#pragma unroll
    for (int q = 0; q < R; ++q) {
#pragma unroll
      for (int n = 0; n < LN_MAXV; ++n) {
        const int c = lane * 4 + 256 * n;
        if (c < D && ok[q]) {
          f32x4 dx;
#pragma unroll
          for (int r = 0; r < 4; ++r) dx[r] = rs[q] * (gy[q][n][r] - s1[q] - xh[q][n][r] * s2[q]);
          dx += dxi[q][n];
          if (a.dx_out) store4(reinterpret_cast<float*>(a.dx_out) + row[q] * D + c, dx);
          if (a.dx_out_t) store4(reinterpret_cast<T*>(a.dx_out_t) + row[q] * D + c, dx);
          if (y) {
            acc_g[n] += dx * yv[q][n];
            store4(reinterpret_cast<T*>(a.dy) + row[q] * D + c, dx * gt[n]);
          }
        }
      }
    }
  }
  // cross-wave reduction through LDS, then one atomic per feature
  __shared__ float red[3][4][LN_MAXV * 256];
#pragma unroll
  for (int n = 0; n < LN_MAXV; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = lane * 4 + 256 * n + r;
      red[0][wave][c] = acc_sh[n][r];
      red[1][wave][c] = acc_sc[n][r];
      red[2][wave][c] = acc_g[n][r];
    }
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += 256) {
    const float v0 = red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c];
    const float v1 = red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c];
    atomicAdd(a.dshift + (long)b * a.ld_dmod + c, v0);
    atomicAdd(a.dscale + (long)b * a.ld_dmod + c, v1);
    if (y) {
      const float v2 = red[2][0][c] + red[2][1][c] + red[2][2][c] + red[2][3][c];
      atomicAdd(a.dgate + (long)b * a.ld_dgate + c, v2);
    }
  }
}

// ---- wide forms (D % 8 == 0): a lane owns 8 CONSECUTIVE columns (c = 8 lane + 512 n), so every 16-bit tensor is touched with 16-byte and every
// f32 tensor with 2 x 16-byte accesses per lane - the 4-column forms above read and write bf16 in 8-byte pieces, which run at 0.5-0.7 of the
// 16-byte rate (MI355X_MICROARCH.md).  D = 480: 60 of 64 lanes active, one group per lane.
// With `y` given the kernel first applies the gated residual update of the branch above (nn/vit.py:331-332): x = x + gate[b] * y, written to
// x_out - the contraction that produced y then has a plain-store epilogue instead of reading and rewriting the f32 residual stream.
// XT: storage type of the residual stream (x, x_out): float, or - round 5, bf16 mode - bf16: the row is then read and written in 2 instead of 4 bytes
// per element (12 -> 8 bytes per element and launch).  The update x + gate * y, the statistics and u are computed from the f32 sum in registers; only
// what is handed to the next kernel is rounded.
template <typename T, typename XT, int NV8> __global__ __launch_bounds__(256) void ln_modulate_fwd8_kernel(const XT* __restrict__ x, const float* __restrict__ shift,
                                                                                  const float* __restrict__ scale, int ld_mod, T* __restrict__ u,
                                                                                  float* __restrict__ mean, float* __restrict__ rstd, int BT, int Tn, int D,
                                                                                  const T* __restrict__ y, const float* __restrict__ gate, int ld_gate,
                                                                                  XT* __restrict__ x_out) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= BT) return;
  const int b = row / Tn;
  const XT* xr = x + (long)row * D;
  f32x8 v[NV8];
  float s = 0.f;
#pragma unroll
  for (int n = 0; n < NV8; ++n) {
    const int c = lane * 8 + 512 * n;
#pragma unroll
    for (int r = 0; r < 8; ++r) v[n].v[r] = 0.f;
    if (c < D) {
      v[n] = load8(xr + c);
      if (y) {
        const f32x8 yv = load8(y + (long)row * D + c), gv = load8(gate + (long)b * ld_gate + c);
#pragma unroll
        for (int r = 0; r < 8; ++r) v[n].v[r] += gv.v[r] * yv.v[r];
        store8(x_out + (long)row * D + c, v[n]);
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) s += v[n].v[r];
    }
  }
  const float mu = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int n = 0; n < NV8; ++n) {
    const int c = lane * 8 + 512 * n;
    if (c < D) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const float d = v[n].v[r] - mu;
        q += d * d;
      }
    }
  }
  const float rs = 1.0f / sqrtf(wave_sum(q) / (float)D + 1e-6f);
  if (lane == 0) {
    if (mean) mean[row] = mu;
    if (rstd) rstd[row] = rs;
  }
#pragma unroll
  for (int n = 0; n < NV8; ++n) {
    const int c = lane * 8 + 512 * n;
    if (c < D) {
      const f32x8 sh = load8(shift + (long)b * ld_mod + c), sc = load8(scale + (long)b * ld_mod + c);
      f32x8 o;
#pragma unroll
      for (int r = 0; r < 8; ++r) o.v[r] = (v[n].v[r] - mu) * rs * (1.0f + sc.v[r]) + sh.v[r];
      store8(u + (long)row * D + c, o);
    }
  }
}

template <typename T, int NV8, int LNB_ROWS, int R> __global__ __launch_bounds__(256) void ln_modulate_bwd8_kernel(const LnBwdArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * LNB_ROWS;
  const int t1 = min(a.T, t0 + LNB_ROWS);
  const int D = a.D;
  const T* du = reinterpret_cast<const T*>(a.du);
  const T* y = reinterpret_cast<const T*>(a.y);
  f32x8 acc_sh[NV8], acc_sc[NV8], acc_g[NV8], sc[NV8], gt[NV8];
#pragma unroll
  for (int n = 0; n < NV8; ++n) {
    const int c = lane * 8 + 512 * n;
#pragma unroll
    for (int r = 0; r < 8; ++r) acc_sh[n].v[r] = acc_sc[n].v[r] = acc_g[n].v[r] = sc[n].v[r] = gt[n].v[r] = 0.f;
    if (c < D) {
      sc[n] = load8(a.scale + (long)b * a.ld_mod + c);
      if (y) gt[n] = load8(a.gate + (long)b * a.ld_mod_gate + c);
    }
  }
  for (int tb = t0 + wave * R; tb < t1; tb += 4 * R) {
    long row[R];
    bool ok[R];
    float mu[R], rs[R], s1[R], s2[R];
    f32x8 gy[R][NV8], xh[R][NV8], dxi[R][NV8], yv[R][NV8];
#pragma unroll
    for (int q = 0; q < R; ++q) {
      ok[q] = tb + q < t1;
      row[q] = (long)b * a.T + (ok[q] ? tb + q : t0);
      mu[q] = a.mean[row[q]];
      rs[q] = a.rstd[row[q]];
      s1[q] = s2[q] = 0.f;
#pragma unroll
      for (int n = 0; n < NV8; ++n) {
        const int c = lane * 8 + 512 * n;
#pragma unroll
        for (int r = 0; r < 8; ++r) gy[q][n].v[r] = xh[q][n].v[r] = dxi[q][n].v[r] = yv[q][n].v[r] = 0.f;
        if (c < D && ok[q]) {
          gy[q][n] = load8(du + row[q] * D + c);   // holds du for now
          xh[q][n] = load8(reinterpret_cast<const float*>(a.x) + row[q] * D + c);  // holds x for now
          if (a.dx_in) dxi[q][n] = load8(reinterpret_cast<const float*>(a.dx_in) + row[q] * D + c);
          if (y) yv[q][n] = load8(y + row[q] * D + c);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < R; ++q) {
#pragma unroll
      for (int n = 0; n < NV8; ++n) {
        const int c = lane * 8 + 512 * n;
        if (c < D && ok[q]) {
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            const float d = gy[q][n].v[r];
            const float xhat = (xh[q][n].v[r] - mu[q]) * rs[q];
            xh[q][n].v[r] = xhat;
            gy[q][n].v[r] = d * (1.0f + sc[n].v[r]);
            acc_sh[n].v[r] += d;
            acc_sc[n].v[r] += d * xhat;
            s1[q] += gy[q][n].v[r];
            s2[q] += gy[q][n].v[r] * xhat;
          }
        }
      }
    }
#pragma unroll
    for (int q = 0; q < R; ++q) {
      s1[q] = wave_sum(s1[q]) / (float)D;
      s2[q] = wave_sum(s2[q]) / (float)D;
    }
#pragma unroll
    for (int q = 0; q < R; ++q) {
#pragma unroll
      for (int n = 0; n < NV8; ++n) {
        const int c = lane * 8 + 512 * n;
        if (c < D && ok[q]) {
          f32x8 dx, dyv;
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            dx.v[r] = rs[q] * (gy[q][n].v[r] - s1[q] - xh[q][n].v[r] * s2[q]) + dxi[q][n].v[r];
            acc_g[n].v[r] += dx.v[r] * yv[q][n].v[r];
            dyv.v[r] = dx.v[r] * gt[n].v[r];
          }
          if (a.dx_out) store8(reinterpret_cast<float*>(a.dx_out) + row[q] * D + c, dx);
          if (a.dx_out_t) store8(reinterpret_cast<T*>(a.dx_out_t) + row[q] * D + c, dx);
          if (y) store8(reinterpret_cast<T*>(a.dy) + row[q] * D + c, dyv);
        }
      }
    }
  }
  // cross-wave reduction through LDS, then one atomic per feature
  __shared__ float red[3][4][NV8 * 512];
#pragma unroll
  for (int n = 0; n < NV8; ++n)
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int c = lane * 8 + 512 * n + r;
      red[0][wave][c] = acc_sh[n].v[r];
      red[1][wave][c] = acc_sc[n].v[r];
      red[2][wave][c] = acc_g[n].v[r];
    }
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += 256) {
    const float v0 = red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c];
    const float v1 = red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c];
    atomicAdd(a.dshift + (long)b * a.ld_dmod + c, v0);
    atomicAdd(a.dscale + (long)b * a.ld_dmod + c, v1);
    if (y) {
      const float v2 = red[2][0][c] + red[2][1][c] + red[2][2][c] + red[2][3][c];
      atomicAdd(a.dgate + (long)b * a.ld_dgate + c, v2);
    }
  }
}

// ---- round-4 forms of the wide kernels (bf16 / f32, D % 8 == 0).  What the ISA of the forms above showed (hipcc -S): their optional tensors are
// run-time branches, and every join of such a branch carries an `s_waitcnt vmcnt(0)` - the forward waited for its x_out STORES to be acknowledged before it
// started the row statistics, fetched shift / scale in a second, exposed round trip after them, and the backward drained the loads of one row before it
// requested the next ("two rows in flight" were one); the six-step wave sums went through ds_bpermute with a full wait each.  Here the options are
// template parameters, a wave requests everything it will read up front (inactive lanes and rows beyond the tile are clamped to valid addresses and masked
// in the arithmetic instead of branching around the loads), reduces with DPP adds, and stores last.
V4H_DEV float wave_sum_dpp(float v) {
  // quad (xor 1, xor 2), then the 16-lane row (rotate by 4, by 8): every lane of a row holds the row's sum; the four rows meet through SGPRs
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x124, 0xF, 0xF, true));  // row_ror:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));  // row_ror:8
  const int i = __builtin_bit_cast(int, v);
  return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 16))) +
         (__builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 48)));
}
template <typename T> struct Raw8;
template <> struct Raw8<bf16> {
  bf16x8 r;
  V4H_DEV void ld(const bf16* p) { r = *reinterpret_cast<const bf16x8*>(p); }
  V4H_DEV f32x8 cvt(float m) const {
    f32x8 x;
#pragma unroll
    for (int k = 0; k < 8; ++k) x.v[k] = (float)r[k] * m;
    return x;
  }
};
template <> struct Raw8<float> {
  f32x4 lo, hi;
  V4H_DEV void ld(const float* p) { lo = load4(p); hi = load4(p + 4); }
  V4H_DEV f32x8 cvt(float m) const {
    f32x8 x;
#pragma unroll
    for (int k = 0; k < 4; ++k) { x.v[k] = lo[k] * m; x.v[4 + k] = hi[k] * m; }
    return x;
  }
};

// Backward: NW waves per workgroup, ROWS consecutive tokens of one sample per workgroup, R rows requested at once per wave.
// XT / GT: storage types of the saved LayerNorm input x and of the residual-stream gradient (dx_in, dx_out): float, or bf16 in bf16 mode (round 5:
// 18 -> 12 bytes per element and launch).  Every sum - the row statistics, dx_in + the LayerNorm term, the per-sample reductions - is f32 in registers.
template <typename T, typename XT, typename GT, int NV8, int ROWS, int NW, int R, bool DXIN, bool HASY, bool DXOUT, bool DXOUT_T>
__global__ __launch_bounds__(64 * NW) void ln_modulate_bwd8v2_kernel(const LnBwdArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * ROWS;
  const int t1 = min(a.T, t0 + ROWS);
  const int D = a.D;
  const T* du = reinterpret_cast<const T*>(a.du);
  const T* y = reinterpret_cast<const T*>(a.y);
  f32x8 acc_sh[NV8], acc_sc[NV8], acc_g[NV8], sc[NV8], gt[NV8];
#pragma unroll
  for (int n = 0; n < NV8; ++n) {
    const int c = lane * 8 + 512 * n, cl = c < D ? c : 0;
#pragma unroll
    for (int r = 0; r < 8; ++r) acc_sh[n].v[r] = acc_sc[n].v[r] = acc_g[n].v[r] = gt[n].v[r] = 0.f;
    sc[n] = load8(a.scale + (long)b * a.ld_mod + cl);
    if (HASY) gt[n] = load8(a.gate + (long)b * a.ld_mod_gate + cl);
  }
  for (int tb = t0 + wave * R; tb < t1; tb += NW * R) {
    long row[R];
    bool ok[R];
    float mu[R], rs[R], s1[R], s2[R];
    Raw8<T> dur[R][NV8], yr[R][NV8];
    Raw8<XT> xr[R][NV8];
    Raw8<GT> dir[R][NV8];
#pragma unroll
    for (int q = 0; q < R; ++q) {  // every request of the wave's R rows, back to back (rows beyond the tile: the tile's first row, masked below)
      ok[q] = tb + q < t1;
      row[q] = (long)b * a.T + (ok[q] ? tb + q : t0);
      mu[q] = a.mean[row[q]];
      rs[q] = a.rstd[row[q]];
#pragma unroll
      for (int n = 0; n < NV8; ++n) {
        const int c = lane * 8 + 512 * n, cl = c < D ? c : 0;
        dur[q][n].ld(du + row[q] * D + cl);
        xr[q][n].ld(reinterpret_cast<const XT*>(a.x) + row[q] * D + cl);
        if (DXIN) dir[q][n].ld(reinterpret_cast<const GT*>(a.dx_in) + row[q] * D + cl);
        if (HASY) yr[q][n].ld(y + row[q] * D + cl);
      }
    }
    __builtin_amdgcn_sched_barrier(0);  // (as in the forward: keep every request ahead of the arithmetic)
    f32x8 gy[R][NV8], xh[R][NV8];
#pragma unroll
    for (int q = 0; q < R; ++q) {
      s1[q] = s2[q] = 0.f;
#pragma unroll
      for (int n = 0; n < NV8; ++n) {
        const float m = (lane * 8 + 512 * n < D && ok[q]) ? 1.0f : 0.0f;
        gy[q][n] = dur[q][n].cvt(m);  // d u, zero where masked
        xh[q][n] = xr[q][n].cvt(1.0f);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const float d = gy[q][n].v[r];
          const float xhat = (xh[q][n].v[r] - mu[q]) * rs[q];
          xh[q][n].v[r] = xhat;
          gy[q][n].v[r] = d * (1.0f + sc[n].v[r]);
          acc_sh[n].v[r] += d;
          acc_sc[n].v[r] += d * xhat;
          s1[q] += gy[q][n].v[r];
          s2[q] += gy[q][n].v[r] * xhat;
        }
      }
    }
#pragma unroll
    for (int q = 0; q < R; ++q) {
      s1[q] = wave_sum_dpp(s1[q]) / (float)D;
      s2[q] = wave_sum_dpp(s2[q]) / (float)D;
    }
#pragma unroll
    for (int q = 0; q < R; ++q) {
#pragma unroll
      for (int n = 0; n < NV8; ++n) {
        const int c = lane * 8 + 512 * n;
        const float m = (c < D && ok[q]) ? 1.0f : 0.0f;
        f32x8 dx, dyv;
        f32x8 dxi, yv;
        if (DXIN) dxi = dir[q][n].cvt(1.0f);
        if (HASY) yv = yr[q][n].cvt(m);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          dx.v[r] = rs[q] * (gy[q][n].v[r] - s1[q] - xh[q][n].v[r] * s2[q]);
          if (DXIN) dx.v[r] += dxi.v[r];
          if (HASY) {
            acc_g[n].v[r] += dx.v[r] * yv.v[r];
            dyv.v[r] = dx.v[r] * gt[n].v[r];
          }
        }
        if (c < D && ok[q]) {
          if (DXOUT) store8(reinterpret_cast<GT*>(a.dx_out) + row[q] * D + c, dx);
          if (DXOUT_T) store8(reinterpret_cast<T*>(a.dx_out_t) + row[q] * D + c, dx);
          if (HASY) store8(reinterpret_cast<T*>(a.dy) + row[q] * D + c, dyv);
        }
      }
    }
  }
  // cross-wave reduction through LDS (16-byte pieces: a lane's 8 columns are two conflict-free ds_write_b128), then one atomic per feature
  __shared__ __attribute__((aligned(16))) float red[3][NW][NV8 * 512];
#pragma unroll
  for (int n = 0; n < NV8; ++n) {
    const int c = lane * 8 + 512 * n;
    *reinterpret_cast<f32x4*>(&red[0][wave][c]) = f32x4{acc_sh[n].v[0], acc_sh[n].v[1], acc_sh[n].v[2], acc_sh[n].v[3]};
    *reinterpret_cast<f32x4*>(&red[0][wave][c + 4]) = f32x4{acc_sh[n].v[4], acc_sh[n].v[5], acc_sh[n].v[6], acc_sh[n].v[7]};
    *reinterpret_cast<f32x4*>(&red[1][wave][c]) = f32x4{acc_sc[n].v[0], acc_sc[n].v[1], acc_sc[n].v[2], acc_sc[n].v[3]};
    *reinterpret_cast<f32x4*>(&red[1][wave][c + 4]) = f32x4{acc_sc[n].v[4], acc_sc[n].v[5], acc_sc[n].v[6], acc_sc[n].v[7]};
    if (HASY) {
      *reinterpret_cast<f32x4*>(&red[2][wave][c]) = f32x4{acc_g[n].v[0], acc_g[n].v[1], acc_g[n].v[2], acc_g[n].v[3]};
      *reinterpret_cast<f32x4*>(&red[2][wave][c + 4]) = f32x4{acc_g[n].v[4], acc_g[n].v[5], acc_g[n].v[6], acc_g[n].v[7]};
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += 64 * NW) {
    float v0 = 0.f, v1 = 0.f, v2 = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      v0 += red[0][w][c];
      v1 += red[1][w][c];
      if (HASY) v2 += red[2][w][c];
    }
    atomicAdd(a.dshift + (long)b * a.ld_dmod + c, v0);
    atomicAdd(a.dscale + (long)b * a.ld_dmod + c, v1);
    if (HASY) atomicAdd(a.dgate + (long)b * a.ld_dgate + c, v2);
  }
}

template <typename T> __global__ void silu_bwd_kernel(const float* __restrict__ ds, const float* __restrict__ pre, T* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (T)(ds[i] * dsilu_f(pre[i]));
}

// ------------------------------------------------------------------------------------------------ CFM step
// linear_trajectory (models/trajectories.py:5-8) as used by CFM._batch_loss (models/base_model.py:209-215)
__global__ void cfm_prepare_kernel(const float* __restrict__ x1, const float* __restrict__ x0, const float* __restrict__ t, float* __restrict__ xt,
                                   float* __restrict__ target, int B, int per, float* zero0, float* zero1) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) {  // the step's two scalar accumulators (loss, squared gradient norm) start from zero here instead of in a fill launch each
    if (zero0) *zero0 = 0.f;
    if (zero1) *zero1 = 0.f;
  }
  if (i >= (long)B * per) return;
  const float tt = t[i / per];
  const float a = x0[i], b = x1[i];
  xt[i] = (1.0f - tt) * a + tt * b;
  target[i] = b - a;
}
// loss = mean((v - target)^2) (models/base_model.py:217-218) and dv = 2 (v - target) / n
__global__ void mse_kernel(const float* __restrict__ v, const float* __restrict__ target, float* __restrict__ loss, float* __restrict__ dv, long n, float inv_n) {
  float s = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float d = v[i] - target[i];
    s += d * d;
    if (dv) dv[i] = 2.0f * d * inv_n;
  }
  s = wave_sum(s);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(loss, (red[0] + red[1] + red[2] + red[3]) * inv_n);
}
__global__ void sq_norm_kernel(const float* __restrict__ g, long n, float* __restrict__ out) {
  float s = 0.f;
  const long n4 = n / 4;
  const long stride = (long)gridDim.x * blockDim.x;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {  // four independent 16-byte loads in flight per lane
    const f32x4 a = load4(g + 4 * i), b = load4(g + 4 * (i + stride)), c = load4(g + 4 * (i + 2 * stride)), d = load4(g + 4 * (i + 3 * stride));
    s += a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3] + b[0] * b[0] + b[1] * b[1] + b[2] * b[2] + b[3] * b[3];
    s += c[0] * c[0] + c[1] * c[1] + c[2] * c[2] + c[3] * c[3] + d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
  }
  for (; i < n4; i += stride) {
    const f32x4 v = load4(g + 4 * i);
    s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  if (blockIdx.x == 0 && threadIdx.x < n - 4 * n4) {
    const float v = g[4 * n4 + threadIdx.x];
    s += v * v;
  }
  s = wave_sum(s);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}
// clip_grad_norm_ + torch.optim.AdamW single tensor (base_experiment.py:573-592; A12 of SURVEY.md):
// coef = min(1, clip / (norm + 1e-6)); p *= 1 - lr wd; m,v update; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n,
                             const float* __restrict__ gnorm_sq, float clip, float lr, float b1, float b2, float eps, float wd, float bc1, float sqrt_bc2,
                             int* nonfinite) {
  // error_if_nonfinite (base_experiment.py:581): on a non-finite gradient norm parameters and moments stay untouched and a counter tells the host.
  // The counter is STICKY: while it is non-zero every later update is skipped (and counted) too, so when the host looks - every step or every 50 -
  // the state is exactly the one the reference's raise would have left behind, and no update was applied with a shifted step index meanwhile.
  // (Every thread of a launch takes the same decision: the counter only grows in launches in which all of them skip anyway.)
  float coef = 1.0f;
  const float nrm = gnorm_sq ? sqrtf(*gnorm_sq) : 0.0f;
  const bool stuck = nonfinite && __hip_atomic_load(nonfinite, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0;
  if (!isfinite(nrm) || stuck) {
    if (nonfinite && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(nonfinite, 1);
    return;
  }
  if (gnorm_sq) coef = fminf(1.0f, clip / (nrm + 1e-6f));  // clip = +inf (no clipping, the reference's max_norm = inf): 1
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float gi = g[i] * coef;
    float pi = p[i] * (1.0f - lr * wd);
    const float mi = m[i] * b1 + (1.0f - b1) * gi;
    const float vi = v[i] * b2 + (1.0f - b2) * gi * gi;
    const float denom = sqrtf(vi) / sqrt_bc2 + eps;
    pi -= (lr / bc1) * (mi / denom);
    p[i] = pi;
    m[i] = mi;
    v[i] = vi;
  }
}
// The same update with the optimizer's step index and the CosineAnnealingLR position kept ON THE DEVICE (reference base_experiment.py:586-597): an update
// that is skipped - non-finite norm, or gradient norm above max_grad_norm after MIN_STEP_SKIP iterations - advances neither, exactly as the reference's
// early `return` skips optimizer.step() and scheduler.step(); the host never has to know.  state_in = {applied optimizer steps, scheduler steps,
// updates skipped for max_grad_norm, -}; every thread reads state_in, thread 0 of workgroup 0 writes state_out (another 16 bytes: no race with readers).
// The update runs over up to ADAM_MAX_RANGES element ranges of the flat buffers per launch (one range = the whole buffer, or the slices of one stage of
// the pipelined update, v4h_vit_update_ahead); `leader` marks the ONE launch of a step that writes state_out and counts a skipped update.
// EMA (round 5): the shadow parameters of the reference's torch_ema.ExponentialMovingAverage (base_experiment.py:127-134,593-594: ema.update() right behind
// optimizer.step(), not for a skipped update) in the same pass - one more f32 stream read and written: shadow -= (1 - d) (shadow - p_new) with
// d = min(decay, (1 + n) / (10 + n)), n = the number of updates applied including this one (torch_ema's num_updates warm-up).
template <bool EMA>
__global__ void adamw_sched_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, const AdamRanges rg,
                                   const float* __restrict__ gnorm_sq, float clip, float lr0, float eta_min, int t_max, float b1, float b2, float log_b1,
                                   float log_b2, float eps, float wd, const int* __restrict__ state_in, int* __restrict__ state_out, float max_grad_norm,
                                   int* nonfinite, float* __restrict__ gnorm_out, int leader, float* __restrict__ ema, float ema_decay) {
  const int applied = state_in[0], sched = state_in[1], skipped = state_in[2];
  const bool lead = leader && blockIdx.x == 0 && threadIdx.x == 0;
  float coef = 1.0f;
  const float nrm = gnorm_sq ? sqrtf(*gnorm_sq) : 0.0f;
  if (lead && gnorm_out) *gnorm_out = nrm;  // the pre-clip norm clip_grad_norm_ returns (base_experiment.py:573-585), without a sqrt launch of its own
  const bool stuck = nonfinite && __hip_atomic_load(nonfinite, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0;
  if (!isfinite(nrm) || stuck) {
    if (lead) {
      if (nonfinite) atomicAdd(nonfinite, 1);
      state_out[0] = applied; state_out[1] = sched; state_out[2] = skipped; state_out[3] = 0;
    }
    return;
  }
  if (nrm > max_grad_norm) {  // (max_grad_norm = +inf: never)
    if (lead) { state_out[0] = applied; state_out[1] = sched; state_out[2] = skipped + 1; state_out[3] = 0; }
    return;
  }
  if (lead) { state_out[0] = applied + 1; state_out[1] = sched + 1; state_out[2] = skipped; state_out[3] = 0; }
  // Bias corrections 1 - beta^step and the cosine learning rate, per thread, in f32 - accurate to f32 rounding because the host passes log(beta) rounded
  // from double and expm1f keeps the relative accuracy of a small argument (1 - 0.999^1 = 1e-3 would lose four digits as 1 - powf).  [First form: one
  // thread per workgroup in double + LDS broadcast - software double transcendentals in front of every workgroup's first load made the kernel 17 us
  // slower (132.6 vs 115.0 us, profiles/r04_*).]
  const float stepf = (float)(applied + 1);
  const float bc1 = -expm1f(stepf * log_b1), sqrt_bc2 = sqrtf(-expm1f(stepf * log_b2));
  const float lr = eta_min + (lr0 - eta_min) * 0.5f * (1.0f + cospif((float)sched / (float)t_max));
  if (gnorm_sq) coef = fminf(1.0f, clip / (nrm + 1e-6f));
  const float one_minus_decay = 1.0f - fminf(ema_decay, (1.0f + stepf) / (10.0f + stepf));
  int r = 0;
  while (r + 1 < rg.count && (int)blockIdx.x >= rg.first_block[r + 1]) ++r;  // this workgroup's range (wave-uniform)
  const long lo = rg.lo[r], hi = lo + rg.n[r];
  const long nb = rg.first_block[r + 1] - rg.first_block[r];
  for (long i = lo + (long)(blockIdx.x - rg.first_block[r]) * blockDim.x + threadIdx.x; i < hi; i += nb * blockDim.x) {
    const float gi = g[i] * coef;
    float pi = p[i] * (1.0f - lr * wd);
    const float mi = m[i] * b1 + (1.0f - b1) * gi;
    const float vi = v[i] * b2 + (1.0f - b2) * gi * gi;
    const float denom = sqrtf(vi) / sqrt_bc2 + eps;
    pi -= (lr / bc1) * (mi / denom);
    p[i] = pi;
    m[i] = mi;
    v[i] = vi;
    if constexpr (EMA) {
      const float sh = ema[i];
      ema[i] = sh - (sh - pi) * one_minus_decay;
    }
  }
}
__global__ void axpby_kernel(float* __restrict__ out, const float* __restrict__ a, const float* __restrict__ b, float alpha, float beta, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = alpha * a[i] + beta * b[i];
}
// torchdiffeq 'rk4' (3/8 rule) final combination: y += h (k1 + 3 (k2 + k3) + k4) / 8
__global__ void rk4_combine_kernel(float* __restrict__ y, const float* __restrict__ k1, const float* __restrict__ k2, const float* __restrict__ k3,
                                   const float* __restrict__ k4, float h, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] += (k1[i] + 3.0f * (k2[i] + k3[i]) + k4[i]) * (h * 0.125f);
}

// out[k] += sum_z slab[z][k]: the second half of a split-K weight gradient (deterministic, unlike float atomics)
// SET: out[k] = sum_z slab[z][k] (the gradient tensor need not be initialised: v4h_plan_set_gradient_mode)
// The partials are requested eight at a time and added in their order (with a run-time trip count and one load per iteration a lane waits for each partial in
// turn: 33 dependent round trips for the 32 splits of the small embedder gradients - 20 us for 4 MB, in the tail of the step).
template <bool SET> __global__ void slab_reduce_kernel(const float* __restrict__ slab, int nz, long n4, float* __restrict__ out) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    f32x4 acc = SET ? f32x4{0.f, 0.f, 0.f, 0.f} : load4(out + 4 * i);
    for (int z0 = 0; z0 < nz; z0 += 8) {
      f32x4 part[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) part[u] = load4(slab + ((long)min(z0 + u, nz - 1) * n4 + i) * 4);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (z0 + u < nz) acc += part[u];
    }
    store4(out + 4 * i, acc);
  }
}

// several buffers zeroed by one launch (gradient tensors that are accumulated into + the workspace accumulators of a backward pass)
__global__ void zero_many_kernel(const ZeroTable tb) {
  float* p = tb.p[blockIdx.y];
  const long n = tb.n[blockIdx.y], n4 = n / 4;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) store4(p + 4 * i, z);
  if (blockIdx.x == 0 && threadIdx.x < n - 4 * n4) p[4 * n4 + threadIdx.x] = 0.f;
}

inline int nblocks(long n, int per_block, int cap = 2048) {
  long b = (n + per_block - 1) / per_block;
  if (b < 1) b = 1;
  return (int)(b > cap ? cap : b);
}

}  // namespace

// ================================================================================================ host wrappers
int cast_pad_many(Mode m, const CastPadItem* items, int n, hipStream_t s) {
  for (int base = 0; base < n; base += MAX_ITEMS) {
    CastPadTable tb;
    tb.n = (n - base) < MAX_ITEMS ? (n - base) : MAX_ITEMS;
    int nb = 0;
    for (int e = 0; e < tb.n; ++e) {
      tb.it[e] = items[base + e];
      tb.first_block[e] = nb;
      const long total = (long)tb.it[e].Rp * tb.it[e].Cp;
      V4H_CHECK_ARG(total > 0, "cast_pad: empty item %d", base + e);
      nb += nblocks(total, 1024, n == 1 ? 2048 : 256);  // (a lone item - the d-modulation table of the backward, 2.3 M elements - would walk 9 dependent rounds on 256 blocks)
    }
    tb.first_block[tb.n] = nb;
    if (m == MODE_BF16) hipLaunchKernelGGL(cast_pad_kernel<bf16>, dim3(nb), dim3(256), 0, s, tb);
    else hipLaunchKernelGGL(cast_pad_kernel<float>, dim3(nb), dim3(256), 0, s, tb);
    V4H_CHECK_LAUNCH("cast_pad");
  }
  return V4H_OK;
}
namespace {
__global__ void write_ptr_table_kernel(PtrTable t, float** dst) {
  if (threadIdx.x < 2 * V4H_GEMM_MAX_GROUPS) dst[threadIdx.x] = t.p[threadIdx.x];
}
}  // namespace
int write_ptr_table(const PtrTable& t, float** dst, hipStream_t s) {
  hipLaunchKernelGGL(write_ptr_table_kernel, dim3(1), dim3(2 * V4H_GEMM_MAX_GROUPS), 0, s, t, dst);
  V4H_CHECK_LAUNCH("write_ptr_table");
  return V4H_OK;
}
int unpad_f32(const float* src, int ld_src, float* dst, int R, int C, hipStream_t s) {
  hipLaunchKernelGGL(unpad_kernel, dim3((R * C + 255) / 256), dim3(256), 0, s, src, ld_src, dst, R, C);
  V4H_CHECK_LAUNCH("unpad");
  return V4H_OK;
}
int patchify(Mode m, const float* vox, void* xp, int B, const PatchGeom& g, int P, int Ppad, hipStream_t s) {
  const size_t slab_bytes = (size_t)g.p1 * g.A * g.R * 4;
  if (Ppad % 8 == 0 && slab_bytes <= 48 * 1024 && ((uintptr_t)xp % 16) == 0 && B <= 65535) {  // regular grid through LDS (one launch, padding included)
    if (m == MODE_BF16) hipLaunchKernelGGL(patchify_slab_kernel<bf16>, dim3(g.l, B), dim3(256), slab_bytes, s, vox, (bf16*)xp, g, P, Ppad);
    else hipLaunchKernelGGL(patchify_slab_kernel<float>, dim3(g.l, B), dim3(256), slab_bytes, s, vox, (float*)xp, g, P, Ppad);
    V4H_CHECK_LAUNCH("patchify/slab");
    return V4H_OK;
  }
  const long n = (long)B * g.L * g.A * g.R;
  const long rows = (long)B * g.l * g.a * g.r;
  const int nb = (int)((n + 255) / 256);
  if (m == MODE_BF16) hipLaunchKernelGGL(patchify_kernel<bf16>, dim3(nb), dim3(256), 0, s, vox, (bf16*)xp, B, g, P, Ppad);
  else hipLaunchKernelGGL(patchify_kernel<float>, dim3(nb), dim3(256), 0, s, vox, (float*)xp, B, g, P, Ppad);
  V4H_CHECK_LAUNCH("patchify");
  if (Ppad > P) {
    const int nbz = (int)((rows * (Ppad - P) + 255) / 256);
    if (m == MODE_BF16) hipLaunchKernelGGL(zero_pad_cols_kernel<bf16>, dim3(nbz), dim3(256), 0, s, (bf16*)xp, rows, P, Ppad);
    else hipLaunchKernelGGL(zero_pad_cols_kernel<float>, dim3(nbz), dim3(256), 0, s, (float*)xp, rows, P, Ppad);
    V4H_CHECK_LAUNCH("patchify/pad");
  }
  return V4H_OK;
}
int unpatchify_f32(const float* tok, int ld, float* vox, int B, const PatchGeom& g, int P, hipStream_t s) {
  const long n = (long)B * g.L * g.A * g.R;
  hipLaunchKernelGGL(unpatchify_kernel, dim3((int)((n + 255) / 256)), dim3(256), 0, s, tok, ld, vox, B, g, P);
  V4H_CHECK_LAUNCH("unpatchify");
  return V4H_OK;
}
int patchify_map(Mode m, bool out_f32, const float* vox, const int* map, void* xp, int B, long V, int T, int P, int Ppad, hipStream_t s) {
  const long n = (long)B * T * Ppad;
  const int nb = (int)((n + 255) / 256);
  if (m == MODE_BF16 && !out_f32) hipLaunchKernelGGL(patchify_map_kernel<bf16>, dim3(nb), dim3(256), 0, s, vox, map, (bf16*)xp, B, V, T, P, Ppad);
  else hipLaunchKernelGGL(patchify_map_kernel<float>, dim3(nb), dim3(256), 0, s, vox, map, (float*)xp, B, V, T, P, Ppad);
  V4H_CHECK_LAUNCH("patchify_map");
  return V4H_OK;
}
int unpatchify_map_f32(const float* tok, int ld, const int* map, float* vox, int B, long V, int T, int P, hipStream_t s) {
  const long n = (long)B * T * P;
  hipLaunchKernelGGL(unpatchify_map_kernel, dim3((int)((n + 255) / 256)), dim3(256), 0, s, tok, ld, map, vox, B, V, T, P);
  V4H_CHECK_LAUNCH("unpatchify_map");
  return V4H_OK;
}
int pos_embed_fwd_pos(const float* freqs, const float* pos, float* pe, int T, int D, hipStream_t s) {
  V4H_CHECK_ARG(D % 6 == 0, "pos_embed: hidden_dim %d not divisible by 6", D);
  const int n = T * 3 * (D / 6);
  hipLaunchKernelGGL(pos_embed_fwd_pos_kernel, dim3((n + 255) / 256), dim3(256), 0, s, freqs, pos, pe, T, D);
  V4H_CHECK_LAUNCH("pos_embed_fwd_pos");
  return V4H_OK;
}
int pos_embed_bwd_pos(Mode m, const void* dx0, const float* freqs, const float* pos, float* dfreqs, float* scratch, int B, int T, int D, hipStream_t s) {
  const int TD = T * D;
  if (m == MODE_BF16) sum_over_batch<bf16>(dx0, scratch, B, TD, s);  // scratch: zeroed by the caller
  else sum_over_batch<float>(dx0, scratch, B, TD, s);
  V4H_CHECK_LAUNCH("pos_embed_bwd/sum");
  hipLaunchKernelGGL(pos_embed_bwd_pos_kernel, dim3(D / 6), dim3(256), 0, s, scratch, freqs, pos, dfreqs, T, D);
  V4H_CHECK_LAUNCH("pos_embed_bwd_pos");
  return V4H_OK;
}
int pos_embed_fwd(const float* freqs, float* pe, const PatchGeom& g, int D, hipStream_t s) {
  V4H_CHECK_ARG(D % 6 == 0, "pos_embed: hidden_dim %d not divisible by 6", D);
  const int n = g.l * g.a * g.r * 3 * (D / 6);
  hipLaunchKernelGGL(pos_embed_fwd_kernel, dim3((n + 255) / 256), dim3(256), 0, s, freqs, pe, g, D);
  V4H_CHECK_LAUNCH("pos_embed_fwd");
  return V4H_OK;
}
int pos_embed_bwd(Mode m, const void* dx0, const float* freqs, float* dfreqs, float* scratch, int B, const PatchGeom& g, int D, hipStream_t s) {
  const int TD = g.l * g.a * g.r * D;
  if (m == MODE_BF16) sum_over_batch<bf16>(dx0, scratch, B, TD, s);  // scratch: zeroed by the caller
  else sum_over_batch<float>(dx0, scratch, B, TD, s);
  V4H_CHECK_LAUNCH("pos_embed_bwd/sum");
  hipLaunchKernelGGL(pos_embed_bwd_kernel, dim3(D / 6), dim3(256), 0, s, scratch, freqs, dfreqs, g, D);
  V4H_CHECK_LAUNCH("pos_embed_bwd");
  return V4H_OK;
}
int timestep_embed(Mode m, const float* t, void* out, int B, int F, hipStream_t s) {
  V4H_CHECK_ARG(F % 2 == 0, "timestep_embed: odd frequency_embedding_size %d", F);
  const int n = B * (F / 2);
  if (m == MODE_BF16) hipLaunchKernelGGL(timestep_embed_kernel<bf16>, dim3((n + 255) / 256), dim3(256), 0, s, t, (bf16*)out, B, F);
  else hipLaunchKernelGGL(timestep_embed_kernel<float>, dim3((n + 255) / 256), dim3(256), 0, s, t, (float*)out, B, F);
  V4H_CHECK_LAUNCH("timestep_embed");
  return V4H_OK;
}
bool ln_resid16_supported(Mode m, int D) { return m == MODE_BF16 && D % 8 == 0 && D <= 512; }
int ln_modulate_fwd(Mode m, const void* x, const float* shift, const float* scale, int ld_mod, void* u, float* mean, float* rstd, int BT, int T, int D,
                    hipStream_t s, bool x16) {
  V4H_CHECK_ARG(D % 4 == 0 && D <= 1024, "ln_modulate: hidden_dim %d unsupported (multiple of 4, <= 1024)", D);
  const dim3 grid((BT + 3) / 4);
  const bool al = ((uintptr_t)x % 16) == 0 && ((uintptr_t)u % 16) == 0 && ((uintptr_t)shift % 16) == 0 && ((uintptr_t)scale % 16) == 0 && ld_mod % 4 == 0;
  V4H_CHECK_ARG(!x16 || (al && ln_resid16_supported(m, D)), "ln_modulate: a 16-bit residual stream needs bf16 mode, hidden_dim %d a multiple of 8 up to 512 and 16-byte aligned tensors", D);
  if (al && D % 8 == 0) {
#define V4H_LNF8(TT, XT, NV) hipLaunchKernelGGL((ln_modulate_fwd8_kernel<TT, XT, NV>), grid, dim3(256), 0, s, (const XT*)x, shift, scale, ld_mod, (TT*)u, mean, rstd, BT, T, D, (const TT*)nullptr, (const float*)nullptr, 0, (XT*)nullptr)
    if (x16) V4H_LNF8(bf16, bf16, 1);
    else if (D <= 512) { if (m == MODE_BF16) V4H_LNF8(bf16, float, 1); else V4H_LNF8(float, float, 1); }
    else { if (m == MODE_BF16) V4H_LNF8(bf16, float, 2); else V4H_LNF8(float, float, 2); }
#undef V4H_LNF8
  } else if (D <= 512) {
    if (m == MODE_BF16) hipLaunchKernelGGL((ln_modulate_fwd_kernel<bf16, 2>), grid, dim3(256), 0, s, (const float*)x, shift, scale, ld_mod, (bf16*)u, mean, rstd, BT, T, D);
    else hipLaunchKernelGGL((ln_modulate_fwd_kernel<float, 2>), grid, dim3(256), 0, s, (const float*)x, shift, scale, ld_mod, (float*)u, mean, rstd, BT, T, D);
  } else {
    if (m == MODE_BF16) hipLaunchKernelGGL((ln_modulate_fwd_kernel<bf16, 4>), grid, dim3(256), 0, s, (const float*)x, shift, scale, ld_mod, (bf16*)u, mean, rstd, BT, T, D);
    else hipLaunchKernelGGL((ln_modulate_fwd_kernel<float, 4>), grid, dim3(256), 0, s, (const float*)x, shift, scale, ld_mod, (float*)u, mean, rstd, BT, T, D);
  }
  V4H_CHECK_LAUNCH("ln_modulate_fwd");
  return V4H_OK;
}
bool ln_resid_supported(int D) { return D % 8 == 0 && D <= 1024; }
// x_out = x + gate[b] * y, then LayerNorm + modulate of x_out (the wide kernel only: D % 8 == 0, 16-byte aligned tensors)
int ln_resid_modulate_fwd(Mode m, const void* x, const void* y, const float* gate, int ld_gate, void* x_out, const float* shift, const float* scale, int ld_mod,
                          void* u, float* mean, float* rstd, int BT, int T, int D, hipStream_t s, bool x16) {
  auto al16 = [](const void* p) { return ((uintptr_t)p % 16) == 0; };
  V4H_CHECK_ARG(ln_resid_supported(D) && y && gate && x_out, "ln_resid_modulate: hidden_dim %d unsupported or null tensor", D);
  V4H_CHECK_ARG(al16(x) && al16(y) && al16(gate) && al16(x_out) && al16(shift) && al16(scale) && al16(u) && ld_gate % 4 == 0 && ld_mod % 4 == 0,
                "ln_resid_modulate: tensors must be 16-byte aligned");
  V4H_CHECK_ARG(!x16 || ln_resid16_supported(m, D), "ln_resid_modulate: a 16-bit residual stream needs bf16 mode and hidden_dim %d a multiple of 8 up to 512", D);
  const dim3 grid((BT + 3) / 4);
#define V4H_LNR8(TT, XT, NV) hipLaunchKernelGGL((ln_modulate_fwd8_kernel<TT, XT, NV>), grid, dim3(256), 0, s, (const XT*)x, shift, scale, ld_mod, (TT*)u, mean, rstd, BT, T, D, (const TT*)y, gate, ld_gate, (XT*)x_out)
  if (x16) V4H_LNR8(bf16, bf16, 1);
  else if (D <= 512) { if (m == MODE_BF16) V4H_LNR8(bf16, float, 1); else V4H_LNR8(float, float, 1); }
  else { if (m == MODE_BF16) V4H_LNR8(bf16, float, 2); else V4H_LNR8(float, float, 2); }
#undef V4H_LNR8
  V4H_CHECK_LAUNCH("ln_resid_modulate_fwd");
  return V4H_OK;
}
int ln_modulate_bwd(Mode m, const LnBwdArgs& a, hipStream_t s) {
  V4H_CHECK_ARG(a.D % 4 == 0 && a.D <= 1024, "ln_modulate_bwd: hidden_dim %d unsupported", a.D);
  // 16 rows per workgroup, 2 rows in flight per wave: measured best (4 rows in flight or 32-48 rows per workgroup: -1...-6 % end to end)
#define V4H_LNB_LAUNCH(TT, MAXV) V4H_LAUNCH((ln_modulate_bwd_kernel<TT, MAXV, 16, 2>), dim3((a.T + 15) / 16, a.B), dim3(256), 0, s, a)
#define V4H_LNB8_LAUNCH(TT, NV) V4H_LAUNCH((ln_modulate_bwd8_kernel<TT, NV, 16, 2>), dim3((a.T + 15) / 16, a.B), dim3(256), 0, s, a)
  auto al16 = [](const void* p) { return ((uintptr_t)p % 16) == 0; };
  const bool al = al16(a.du) && al16(a.x) && al16(a.dx_in) && al16(a.dx_out) && al16(a.dx_out_t) && al16(a.y) && al16(a.dy) && al16(a.scale) && al16(a.gate) &&
                  a.ld_mod % 4 == 0 && a.ld_mod_gate % 4 == 0;
  // Round-4 form (options as template parameters, every request of a wave up front; 8 rows per workgroup of 4 waves, one row per wave at a time - the
  // other tile shapes measured slower).  V4H_LNB_V2=0: the round-2 kernel (A/B hook; f32 residual storage only).
  static const int v2 = getenv("V4H_LNB_V2") ? atoi(getenv("V4H_LNB_V2")) : 1;
  const bool r16 = a.x16 || a.g16;
  const bool dxin = a.dx_in != nullptr, hasy = a.y != nullptr, dxo = a.dx_out != nullptr, dxt = a.dx_out_t != nullptr;
  int combo = -1;  // the three combinations the backward pass uses (v4h_runtime.hip); anything else keeps the generic kernel
  if (!dxin && hasy && dxo && !dxt) combo = 0;        // final layer
  else if (dxin && hasy && dxo && !dxt) combo = 1;    // inside the stack
  else if (dxin && !hasy && !dxo && dxt) combo = 2;   // bottom of the stack
  V4H_CHECK_ARG(!r16 || (al && combo >= 0 && ln_resid16_supported(m, a.D)),
                "ln_modulate_bwd: 16-bit residual storage needs bf16 mode, hidden_dim %d a multiple of 8 up to 512, aligned tensors and one of the backward pass's option sets", a.D);
  if ((v2 > 0 || r16) && al && a.D % 8 == 0 && a.D <= 512 && combo >= 0) {
    // 16 rows per workgroup of 4 waves, one row per wave at a time.  (Round 4: 8 rows.  With the bf16 streams of round 5 the per-sample float atomics of a
    // workgroup - 3 x 480 floats whatever its row count, 12.4 MB per launch at 8 rows against 100 MB of streams - weigh more: 16 rows halve them, 68 MB less
    // per step, 256.5 vs 256.0 steps/s; 24 rows 255.3.)
#define V4H_LNB2(TT, XT, GT, A, B_, C_, D_) V4H_LAUNCH((ln_modulate_bwd8v2_kernel<TT, XT, GT, 1, 16, 4, 1, A, B_, C_, D_>), dim3((a.T + 15) / 16, a.B), dim3(256), 0, s, a)
#define V4H_LNB2_COMBO(TT, XT, GT)                                     \
  do {                                                                 \
    if (combo == 0) V4H_LNB2(TT, XT, GT, false, true, true, false);    \
    else if (combo == 1) V4H_LNB2(TT, XT, GT, true, true, true, false); \
    else V4H_LNB2(TT, XT, GT, true, false, false, true);               \
  } while (0)
    if (m != MODE_BF16) V4H_LNB2_COMBO(float, float, float);
    else if (a.x16 && a.g16) V4H_LNB2_COMBO(bf16, bf16, bf16);
    else if (a.x16) V4H_LNB2_COMBO(bf16, bf16, float);
    else if (a.g16) V4H_LNB2_COMBO(bf16, float, bf16);
    else V4H_LNB2_COMBO(bf16, float, float);
    V4H_CHECK_LAUNCH("ln_modulate_bwd");
    return V4H_OK;
#undef V4H_LNB2_COMBO
#undef V4H_LNB2
  }
  if (al && a.D % 8 == 0) {
    if (a.D <= 512) { if (m == MODE_BF16) V4H_LNB8_LAUNCH(bf16, 1); else V4H_LNB8_LAUNCH(float, 1); }
    else { if (m == MODE_BF16) V4H_LNB8_LAUNCH(bf16, 2); else V4H_LNB8_LAUNCH(float, 2); }
  } else if (a.D <= 512) {
    if (m == MODE_BF16) V4H_LNB_LAUNCH(bf16, 2);
    else V4H_LNB_LAUNCH(float, 2);
  } else {
    if (m == MODE_BF16) V4H_LNB_LAUNCH(bf16, 4);
    else V4H_LNB_LAUNCH(float, 4);
  }
#undef V4H_LNB_LAUNCH
#undef V4H_LNB8_LAUNCH
  V4H_CHECK_LAUNCH("ln_modulate_bwd");
  return V4H_OK;
}
int silu_bwd(Mode m, const float* dsilu, const float* pre, void* out, int n, hipStream_t s) {
  if (m == MODE_BF16) hipLaunchKernelGGL(silu_bwd_kernel<bf16>, dim3((n + 255) / 256), dim3(256), 0, s, dsilu, pre, (bf16*)out, n);
  else hipLaunchKernelGGL(silu_bwd_kernel<float>, dim3((n + 255) / 256), dim3(256), 0, s, dsilu, pre, (float*)out, n);
  V4H_CHECK_LAUNCH("silu_bwd");
  return V4H_OK;
}
int cfm_prepare(const float* x1, const float* x0, const float* t, float* xt, float* target, int B, int per, hipStream_t s, float* zero0, float* zero1) {
  const long n = (long)B * per;
  hipLaunchKernelGGL(cfm_prepare_kernel, dim3((int)((n + 255) / 256)), dim3(256), 0, s, x1, x0, t, xt, target, B, per, zero0, zero1);
  V4H_CHECK_LAUNCH("cfm_prepare");
  return V4H_OK;
}
int mse_fwd_bwd(const float* v, const float* target, float* loss, float* dv, long n, hipStream_t s, bool zero_first) {
  if (zero_first) {
    hipError_t e = hipMemsetAsync(loss, 0, sizeof(float), s);
    if (e != hipSuccess) { v4h_set_error("mse: memset failed: %s", hipGetErrorString(e)); return V4H_ERR_HIP; }
  }
  hipLaunchKernelGGL(mse_kernel, dim3(nblocks(n, 1024, 256)), dim3(256), 0, s, v, target, loss, dv, n, 1.0f / (float)n);  // (same-address atomic per workgroup)
  V4H_CHECK_LAUNCH("mse");
  return V4H_OK;
}
int sq_norm_accum(const float* g, long n, float* out, hipStream_t s) {
  V4H_CHECK_ARG(((uintptr_t)g % 16) == 0, "sq_norm: gradient buffer must be 16-byte aligned");
  // One workgroup per CU: every workgroup ends with an atomic on the SAME address, ~9 ns each at the L2 - with 2048 workgroups that was half
  // of the kernel (26 M floats: 35.1 us; 1024: 26.0; 512: 20.1; 256: 18.0 us = 5.8 TB/s; tools/experiments/sqnorm_time.py).
  hipLaunchKernelGGL(sq_norm_kernel, dim3(nblocks(n, 4096, 256)), dim3(256), 0, s, g, n, out);
  V4H_CHECK_LAUNCH("sq_norm");
  return V4H_OK;
}
int adamw_step(float* p, const float* g, float* m, float* v, long n, const float* gnorm_sq, float clip, float lr, float b1, float b2, float eps, float wd,
               float bc1, float bc2, int* nonfinite, hipStream_t s) {
  hipLaunchKernelGGL(adamw_kernel, dim3(nblocks(n, 1024)), dim3(256), 0, s, p, g, m, v, n, gnorm_sq, clip, lr, b1, b2, eps, wd, bc1, sqrtf(bc2), nonfinite);
  V4H_CHECK_LAUNCH("adamw");
  return V4H_OK;
}
int adamw_step_ranges(float* p, const float* g, float* m, float* v, const long* lo, const long* n, int count, const AdamwHyper& h, const float* gnorm_sq,
                      const int* state_in, int* state_out, int* nonfinite, float* gnorm_out, bool leader, hipStream_t s) {
  V4H_CHECK_ARG(count >= 1 && count <= ADAM_MAX_RANGES, "adamw: %d ranges (1 .. %d per launch)", count, ADAM_MAX_RANGES);
  AdamRanges rg;
  long total = 0;
  for (int r = 0; r < count; ++r) total += n[r];
  const int grid = nblocks(total, 1024);  // workgroups dealt to the ranges in proportion to their length, at least one each
  int used = 0;
  for (int r = 0; r < count; ++r) {
    rg.lo[r] = lo[r]; rg.n[r] = n[r]; rg.first_block[r] = used;
    int nb = (int)((double)grid * (double)n[r] / (double)(total > 0 ? total : 1) + 0.5);
    used += nb < 1 ? 1 : nb;
  }
  rg.first_block[count] = used;
  rg.count = count;
  if (h.ema)
    hipLaunchKernelGGL(adamw_sched_kernel<true>, dim3(used), dim3(256), 0, s, p, g, m, v, rg, gnorm_sq, h.clip, h.lr0, h.eta_min, h.t_max, h.b1, h.b2,
                       (float)log((double)h.b1), (float)log((double)h.b2), h.eps, h.wd, state_in, state_out, h.max_grad_norm, nonfinite, gnorm_out, leader ? 1 : 0,
                       h.ema, h.ema_decay);
  else
    hipLaunchKernelGGL(adamw_sched_kernel<false>, dim3(used), dim3(256), 0, s, p, g, m, v, rg, gnorm_sq, h.clip, h.lr0, h.eta_min, h.t_max, h.b1, h.b2,
                       (float)log((double)h.b1), (float)log((double)h.b2), h.eps, h.wd, state_in, state_out, h.max_grad_norm, nonfinite, gnorm_out, leader ? 1 : 0,
                       (float*)nullptr, 0.0f);
  V4H_CHECK_LAUNCH("adamw_sched");
  return V4H_OK;
}
int adamw_step_sched(float* p, const float* g, float* m, float* v, long n, const float* gnorm_sq, float clip, float lr0, float eta_min, int t_max, float b1, float b2,
                     float eps, float wd, const int* state_in, int* state_out, float max_grad_norm, int* nonfinite, float* gnorm_out, hipStream_t s, float* ema,
                     float ema_decay) {
  const long lo = 0;
  const AdamwHyper h{clip, lr0, eta_min, t_max, b1, b2, eps, wd, max_grad_norm, ema, ema_decay};
  return adamw_step_ranges(p, g, m, v, &lo, &n, 1, h, gnorm_sq, state_in, state_out, nonfinite, gnorm_out, true, s);
}
int slab_reduce(const float* slab, int nz, long n, float* out, hipStream_t s, bool set) {
  V4H_CHECK_ARG(n % 4 == 0 && ((uintptr_t)out % 16) == 0 && ((uintptr_t)slab % 16) == 0, "slab_reduce: size / alignment");
  const int bs = n / 4 >= 65536 ? 256 : 64;  // small results (embedder gradients: 7680 float4 of 32 partials each) spread over more CUs
  if (set) hipLaunchKernelGGL(slab_reduce_kernel<true>, dim3(nblocks(n / 4, bs)), dim3(bs), 0, s, slab, nz, n / 4, out);
  else hipLaunchKernelGGL(slab_reduce_kernel<false>, dim3(nblocks(n / 4, bs)), dim3(bs), 0, s, slab, nz, n / 4, out);
  V4H_CHECK_LAUNCH("slab_reduce");
  return V4H_OK;
}
int zero_many(const std::pair<float*, long>* items, int n, hipStream_t s) {
  for (int base = 0; base < n; base += ZERO_MAX_ITEMS) {
    ZeroTable tb;
    const int cnt = (n - base) < ZERO_MAX_ITEMS ? (n - base) : ZERO_MAX_ITEMS;
    for (int e = 0; e < ZERO_MAX_ITEMS; ++e) {
      const bool live = e < cnt;
      tb.p[e] = live ? items[base + e].first : nullptr;
      tb.n[e] = live ? items[base + e].second : 0;
      V4H_CHECK_ARG(!live || (tb.p[e] != nullptr && tb.n[e] >= 0 && ((uintptr_t)tb.p[e] % 16) == 0), "zero_many: item %d null, negative or not 16-byte aligned", base + e);
    }
    hipLaunchKernelGGL(zero_many_kernel, dim3(32, cnt), dim3(256), 0, s, tb);
    V4H_CHECK_LAUNCH("zero_many");
  }
  return V4H_OK;
}
int axpby(float* out, const float* a, const float* b, float alpha, float beta, long n, hipStream_t s) {
  hipLaunchKernelGGL(axpby_kernel, dim3(nblocks(n, 1024)), dim3(256), 0, s, out, a, b, alpha, beta, n);
  V4H_CHECK_LAUNCH("axpby");
  return V4H_OK;
}
int rk4_combine(float* y, const float* k1, const float* k2, const float* k3, const float* k4, float h, long n, hipStream_t s) {
  hipLaunchKernelGGL(rk4_combine_kernel, dim3(nblocks(n, 1024)), dim3(256), 0, s, y, k1, k2, k3, k4, h, n);
  V4H_CHECK_LAUNCH("rk4_combine");
  return V4H_OK;
}

}  // namespace v4h
