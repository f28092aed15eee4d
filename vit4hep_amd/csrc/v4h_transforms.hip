// Pre-/post-processing chain of the shape models as two fused kernels.  The reference applies nine transform
// objects one after the other (experiments/calochallenge/transforms.py; configs/calochallenge/cfm/calochallenge_ds2.yaml:15-28), each
// a handful of full-tensor PyTorch ops and, in NormalizeByElayer, Python loops over the 45 layers.  Per shower the whole chain is:
// one segmented sum over the layers, a 45-step scalar recurrence, and an element-wise map - one read and one write of the shower.
//   forward : NormalizeByElayer -> ScaleTotalEnergy -> CutValues (identity) -> ExclusiveLogitTransform(rescale) ->
//             GlobalStandardizeFromFile -> LogEnergy -> ScaleEnergy -> AddFeaturesToCond -> Reshape
//   reverse : the same list backwards with rev=True (what `sample_n` does to the shape model's output, experiment.py:190-223)
// HBM-bound: 4 * (n_voxels + n_layers + 1) bytes read and 4 * n_voxels (+ conditions) written per shower.  f32 throughout, libm-accurate
// exp / log (these are data transforms, not a throughput mode).
#include <stdio.h>

#include "../../include/vit4hep_hip.h"
#include "v4h_ops.h"

namespace {
constexpr int MAX_LAYERS = 128;  // <= 256 threads: one lane per u

// GlobalStandardize rev + ExclusiveLogit rev (rescale); inv_scale = 1 / (1 - 2 delta), hoisted (a multiply instead of a division per voxel: <= 1 ulp)
__device__ __forceinline__ float sigmoid_rescaled(float x, float std, float mean, float delta, float inv_scale) {
  const float z = 1.0f / (1.0f + expf(-(x * std + mean)));
  return (z - delta) * inv_scale;
}
__device__ __forceinline__ float logit_standardized(float v, float std, float mean, float delta) {  // ExclusiveLogit fwd (rescale) + GlobalStandardize fwd
  const float z = v * (1.0f - 2.0f * delta) + delta;
  return (logf(z / (1.0f - z)) - mean) / std;
}

// A group of G lanes (16 for short layers such as ds2's 144 voxels, else a whole wave) owns whole layers.  Layers of up to G * MAXE voxels
// are read ONCE into registers (up to MAXE independent loads in flight per lane), summed inside the group, transformed and written: one
// read and one write of the shower.  Longer layers fall back to two passes over the layer (second one from cache).
constexpr int MAXE = 16;
template <int G> __device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <int G, typename Load, typename Store> __device__ __forceinline__ void per_layer_g(const int* __restrict__ bounds, int n_layers, float* sums, Load load, Store store) {
  const int gl = threadIdx.x % G, grp = threadIdx.x / G, ng = blockDim.x / G;
  for (int L = grp; L < n_layers; L += ng) {
    const int lo = bounds[L], hi = bounds[L + 1];
    float s = 0.f;
    if (hi - lo <= G * MAXE) {
      float v[MAXE];
#pragma unroll
      for (int k = 0; k < MAXE; ++k) {
        const int i = lo + gl + G * k;
        v[k] = i < hi ? load(i) : 0.0f;
      }
#pragma unroll
      for (int k = 0; k < MAXE; ++k) s += v[k];
      s = group_sum<G>(s);
      if (gl == 0 && sums) sums[L] = s;
#pragma unroll
      for (int k = 0; k < MAXE; ++k) {
        const int i = lo + gl + G * k;
        if (i < hi) store(i, v[k], s, L);
      }
    } else {
      for (int i = lo + gl; i < hi; i += G) s += load(i);
      s = group_sum<G>(s);
      if (gl == 0 && sums) sums[L] = s;
      for (int i = lo + gl; i < hi; i += G) store(i, load(i), s, L);
    }
  }
}
template <typename Load, typename Store> __device__ __forceinline__ void per_layer(const v4h_chain_spec& sp, const int* __restrict__ bounds, float* sums, Load load, Store store) {
  if (sp.n_voxels <= (long)sp.n_layers * 16 * MAXE) per_layer_g<16>(bounds, sp.n_layers, sums, load, store);  // uniform branch on the mean layer length
  else per_layer_g<64>(bounds, sp.n_layers, sums, load, store);
}

__global__ __launch_bounds__(256) void shape_preprocess_kernel(v4h_chain_spec sp, const int* __restrict__ bounds, const float* __restrict__ showers,
                                                               const float* __restrict__ energy, float* __restrict__ x, float* __restrict__ cond) {
  __shared__ float layer_E[MAX_LAYERS];
  const int b = blockIdx.x, nl = sp.n_layers;
  const float* sh = showers + (long)b * sp.n_voxels;
  float* xo = x + (long)b * sp.n_voxels;
  // NormalizeByElayer forward (transforms.py:379-383) + logit + standardisation, one pass per layer
  per_layer(sp, bounds, layer_E, [&](int i) { return sh[i]; },
            [&](int i, float v, float layer_sum, int) { xo[i] = logit_standardized(v / (layer_sum + sp.eps), sp.std, sp.mean, sp.delta); });
  __syncthreads();
  float* c = cond + (long)b * (nl + 1);
  const float E = energy[b];
  if (threadIdx.x < nl) {  // the u's, one lane each (transforms.py:385-392): u_0 = E_tot / E_inc (scaled), u_{L+1} = E_L / (sum_{k >= L} E_k + eps)
    const int j = threadIdx.x;
    const int L = j == 0 ? 0 : j - 1;
    float rem = 0.f;
    for (int k = L; k < nl; ++k) rem += layer_E[k];
    const float u = j == 0 ? rem / E * sp.factor : layer_E[L] / (rem + sp.eps);
    c[j] = logit_standardized(u, sp.std, sp.mean, sp.delta);
  }
  if (threadIdx.x == 0) c[nl] = (logf(E + sp.alpha) - sp.e_min) / (sp.e_max - sp.e_min);  // LogEnergy, ScaleEnergy
}

__global__ __launch_bounds__(256) void shape_postprocess_kernel(v4h_chain_spec sp, const int* __restrict__ bounds, const float* __restrict__ samples,
                                                                const float* __restrict__ cond, float* __restrict__ showers, float* __restrict__ energy_out) {
  __shared__ float us[MAX_LAYERS];
  __shared__ float layer_E[MAX_LAYERS];
  const int b = blockIdx.x, nl = sp.n_layers;
  const float* sm = samples + (long)b * sp.n_voxels;
  float* out = showers + (long)b * sp.n_voxels;
  const float* c = cond + (long)b * (nl + 1);
  const float inv_scale = 1.0f / (1.0f - 2.0f * sp.delta);
  // the u's in parallel (standardisation rev, logit rev; CutValues leaves them alone), then the cheap sequential part by one lane
  if (threadIdx.x < nl) us[threadIdx.x] = sigmoid_rescaled(c[threadIdx.x], sp.std, sp.mean, sp.delta, inv_scale);
  __syncthreads();
  if (threadIdx.x == 0) {
    const float E = expf(c[nl] * (sp.e_max - sp.e_min) + sp.e_min) - sp.alpha;  // ScaleEnergy rev, LogEnergy rev
    energy_out[b] = E;
    // NormalizeByElayer rev: layer energies from the u's (transforms.py:345-364); u_0 un-scaled (ScaleTotalEnergy rev), u_{i>0} clipped to [0, 1]
    const float total = E * (us[0] / sp.factor);
    float cum = 0.f;
    for (int i = 0; i < nl - 1; ++i) {
      const float e = (total - cum) * fminf(fmaxf(us[i + 1], 0.0f), 1.0f);
      layer_E[i] = e;
      cum += e;
    }
    layer_E[nl - 1] = total - cum;
  }
  __syncthreads();
  per_layer(sp, bounds, nullptr,
            [&](int i) {  // standardisation rev, logit rev, CutValues rev (voxels only; transforms.py:303-308)
              const float v = sigmoid_rescaled(sm[i], sp.std, sp.mean, sp.delta, inv_scale);
              return (sp.cut != 0.0f && v <= sp.cut) ? 0.0f : v;
            },
            [&](int i, float v, float layer_sum, int L) {
              const float n = v * (1.0f / (layer_sum + sp.eps));         // normalise the layer to unity (the reciprocal is hoisted out of the per-voxel loop by the compiler)
              out[i] = (n <= sp.norm_cut ? 0.0f : n) * layer_E[L];       // normalised cut, scale to the layer energy (transforms.py:368-372)
            });
}

int check_spec(const v4h_chain_spec* sp, const void* bounds, int B, const char* who) {
  V4H_CHECK_ARG(sp && bounds, "%s: null argument", who);
  V4H_CHECK_ARG(B > 0, "%s: empty batch (B=%d)", who, B);
  V4H_CHECK_ARG(sp->n_layers >= 1 && sp->n_layers <= MAX_LAYERS, "%s: n_layers %d outside 1..%d", who, sp->n_layers, MAX_LAYERS);
  V4H_CHECK_ARG(sp->n_voxels >= sp->n_layers && sp->n_voxels < (1LL << 31), "%s: n_voxels %lld", who, (long long)sp->n_voxels);
  V4H_CHECK_ARG(sp->std != 0.0f && sp->factor != 0.0f && sp->e_max != sp->e_min && sp->delta >= 0.0f && sp->delta < 0.5f, "%s: degenerate chain constants", who);
  return V4H_OK;
}
}  // namespace

extern "C" int32_t v4h_shape_preprocess(const v4h_chain_spec* sp, const int32_t* d_bounds, const float* d_showers, const float* d_energy, float* d_x, float* d_cond,
                                        int32_t B, void* stream) {
  int rc = check_spec(sp, d_bounds, B, "shape_preprocess");
  if (rc) return rc;
  V4H_CHECK_ARG(d_showers && d_energy && d_x && d_cond, "shape_preprocess: null tensor");
  hipLaunchKernelGGL(shape_preprocess_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, *sp, d_bounds, d_showers, d_energy, d_x, d_cond);
  V4H_CHECK_LAUNCH("shape_preprocess");
  return V4H_OK;
}
extern "C" int32_t v4h_shape_postprocess(const v4h_chain_spec* sp, const int32_t* d_bounds, const float* d_samples, const float* d_cond, float* d_showers,
                                         float* d_energy, int32_t B, void* stream) {
  int rc = check_spec(sp, d_bounds, B, "shape_postprocess");
  if (rc) return rc;
  V4H_CHECK_ARG(d_samples && d_cond && d_showers && d_energy, "shape_postprocess: null tensor");
  hipLaunchKernelGGL(shape_postprocess_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, *sp, d_bounds, d_samples, d_cond, d_showers, d_energy);
  V4H_CHECK_LAUNCH("shape_postprocess");
  return V4H_OK;
}
