// Instantiations + dispatch of the contractions for the three layouts of a Linear: the two-workgroup kernel (v4h_gemm.h) and the 256 x 160 ring kernel
// on its ping-pong schedule (v4h_gemm2.h).  A product build contains only kernels that compute the contraction exactly as specified; every tuning
// table, ablation build and measured loser lives in v4h_gemm_ablations.inc behind -DV4H_ABLATIONS.
#include <stdio.h>
#include <stdlib.h>

#include "v4h_ops.h"
#include "v4h_gemm2.h"
#include "v4h_gemm3.h"
#include "v4h_gemm_small.h"

namespace v4h {
namespace {

// K-step: 64 for bf16 (128-byte rows = whole cache lines per DMA row, half the barriers), 32 for f32 (also 128-byte rows)
template <typename T> constexpr int bk_of() { return sizeof(T) == 2 ? 64 : 32; }
static int env_flag(const char* n, int dflt) { const char* e = getenv(n); return e ? atoi(e) : dflt; }

// Which kernel serves the token-sized bf16 contractions (v4h_select_contraction_kernel / V4H_GEMM2, both choose among CORRECT kernels only):
//   KERNEL_AUTO  the measured winners per contraction class (g_pp below),
//   KERNEL_TWO_WG  the 128 x 160 two-workgroup kernel everywhere (what f32 mode and every small contraction use anyway),
//   KERNEL_RING  the ring kernel wherever the shape is eligible.
//   KERNEL_WS  like KERNEL_AUTO (the weight-stationary kernel of v4h_gemm3.h wherever eligible; KERNEL_TWO_WG and KERNEL_RING never use it).
enum { KERNEL_AUTO = 0, KERNEL_TWO_WG = 1, KERNEL_RING = 2, KERNEL_WS = 3 };
int kernel_from_env() {
  const char* e = getenv("V4H_GEMM2");
  if (!e) return KERNEL_AUTO;
  const int v = atoi(e);  // historical numbering of the switch: -1 automatic, 0 never the ring kernel, 8 ping-pong wherever eligible
  if (v == -1) return KERNEL_AUTO;
  if (v == 0) return KERNEL_TWO_WG;
  if (v == 8) return KERNEL_RING;
#ifndef V4H_ABLATIONS
  fprintf(stderr, "vit4hep_amd: V4H_GEMM2=%s selects an ablation build that this library does not contain (rebuild with -DV4H_ABLATIONS); ignored\n", e);
#endif
  return KERNEL_AUTO;
}
int g_kernel = kernel_from_env();
// Contraction classes on the ring kernel under KERNEL_AUTO (bits: 1 forward plain store, 2 forward GELU of the update step - two outputs, 4 dgrad
// plain store, 8 dgrad DGELU, 16 split-K weight-gradient slabs, 32 forward GELU without the saved derivative - inference), wherever the shape is eligible.
int g_pp = env_flag("V4H_GEMM2_PP", 53) & 63;
// Tile shape of the two N = mlp_hidden contractions with the fused GELU / GELU' epilogues (two-workgroup kernel, 512 workgroup slots on the chip).  All
// tiles of a call take the same time and the persistent grid deals them out statically, so a call costs ceil(tiles / slots) ROUNDS: ds2's fc1 is 1620
// tiles of 128 x 160 = 3.16 rounds paid as 4, but 2025 tiles of 128 x 128 = 3.96 rounds of tiles 0.8 times the size - +2.2 % on the whole step
// (243.5 / 242.3 vs 238.3 / 237.0 steps/s, interleaved, same box; 96 x 160 and 128 x 96: 0 / -0.6 %).  -1 (default): per call, the shape with the
// fewest rounds x tile area, smaller tiles charged a few per cent for their lower FLOP per staged byte.  V4H_MLP_TILE = 0 .. 3 pins 128 x 160, 128 x 128,
// 96 x 160, 128 x 96 (A/B hook).
int g_mlp_tile = env_flag("V4H_MLP_TILE", -1);
inline int pick_mlp_tile(const GemmArgs& a) {
  if (g_mlp_tile >= 0) return g_mlp_tile;
  const long slots = 2L * v4h_compute_units();
  auto cost = [&](int bi, int bj, double penalty) -> double {
    const long tiles = (long)((a.I + bi - 1) / bi) * ((a.J + bj - 1) / bj);
    return (double)((tiles + slots - 1) / slots) * bi * bj * penalty;
  };
  int best = 0;
  double c = cost(128, 160, 1.0);
  if (a.J % 128 == 0 && cost(128, 128, 1.04) < c) { best = 1; c = cost(128, 128, 1.04); }
  if (cost(96, 160, 1.08) < c) { best = 2; c = cost(96, 160, 1.08); }
  return best;
}
int g_small = env_flag("V4H_GEMM_SMALL", 1);  // A/B hook: 0 = the tiled kernel also for the batch-row contractions

#ifdef V4H_ABLATIONS
#include "v4h_gemm_ablations.inc"
#endif

inline bool v2_eligible(const GemmArgs& a, int klen) {
  return a.I >= 2048 && a.J % 160 == 0 && klen >= 192 && a.e.ldo % 8 == 0 && ((uintptr_t)a.e.out % 16) == 0 && (long)a.I * a.e.ldo * 4 < 0x7FFFFFF0L;
}
#ifdef V4H_ABLATIONS
inline bool abl_v2_ok(const GemmArgs& a, int klen) { return v2_eligible(a, klen); }
#endif
inline bool on_ring(const GemmArgs& a, int klen, int bit) {
  if (g_kernel == KERNEL_TWO_WG) return false;
  return (g_kernel == KERNEL_RING || (g_pp & bit)) && v2_eligible(a, klen);
}

// The K = hidden_dim contractions (qkv, attn.proj, fc1 + GELU forward; attn.proj and fc2 x GELU' input gradients) on the weight-stationary kernel:
// the weight slice of a workgroup lives in registers, only activation rows stream (v4h_gemm3.h).  V4H_GEMM3=0: off (A/B hook).
// Contraction classes on it under KERNEL_AUTO (bits: 1 forward plain store with >= 960 output columns - qkv -, 2 forward plain store below that - attn.proj -,
// 4 forward GELU of the update step - two outputs -, 8 dgrad plain store, 16 dgrad DGELU, 32 forward GELU without the saved derivative - inference).
// V4H_GEMM3 overrides (A/B hook); KERNEL_WS takes every eligible class.  Default: every class but the plain dgrad (attn.proj's input gradient: 257.8 / 262.0
// against 263.8 / 265.4 steps/s with it on the ring kernel).  The GELU / DGELU classes joined once their auxiliary slots were cut down - GELU written on
// pairs of elements (v_pk_* f32), the saved derivative of the DGELU form through the DMA ring instead of buffer loads whose wait drained it, the lane part
// of every DMA address hoisted: 259.7 / 260.4 (qkv + proj forward only, previous build) -> 267.2 / 267.3 steps/s, sampler 1528 -> 1569 showers/s, same box.
int g_ws = env_flag("V4H_GEMM3", 55) & 63;
constexpr int WS_K = 480;
inline bool on_ws(const GemmArgs& a, int bit) {
  if (g_kernel == KERNEL_TWO_WG || g_kernel == KERNEL_RING) return false;
  return ((g_ws & bit) != 0 || g_kernel == KERNEL_WS) && v4h_gemm3_eligible(a, WS_K);
}
// Column tiles per wave: 3 where the output width allows (fewer LDS fragment reads per MFMA, fewer column slices re-reading the activations), except for
// the GELU form (value + derivative: fits at 252 VGPRs with 3 tiles since the packed epilogue, but measured 255.0 against 262.0 steps/s) and the DGELU form
// (its strips of the saved derivative fit the LDS beside the ring with 2 tiles only), which run with 2.  V4H_GEMM3_NT pins it (A/B hook).
template <bool QKS, int EPI> int run_ws(const GemmArgs& a, hipStream_t s, const char* name) {
  static const int pin = env_flag("V4H_GEMM3_NT", 0);
  int nt = v4h_gemm3_pick_nt(a.J);
  if ((EPI == EPI_GELU || EPI == EPI_DGELU) && a.J % 32 == 0) nt = 2;
  if (pin == 2 && a.J % 32 == 0) nt = 2;
  if (pin == 3 && a.J % 48 == 0) nt = 3;
  if constexpr (EPI != EPI_DGELU) {  // (the DGELU form's strips of the saved derivative fit the LDS beside the ring with 2 column tiles per wave only)
    if (nt == 3) return v4h_gemm3_launch<Gemm3Cfg<QKS, EPI, 3>>(a, s, name);
  }
  V4H_CHECK_ARG(a.J % 32 == 0, "%s: J=%d must be a multiple of 32", name, a.J);
  return v4h_gemm3_launch<Gemm3Cfg<QKS, EPI, 2>>(a, s, name);
}

template <typename T, typename TO, bool PKS, bool QKS, int BI, int BJ, int EPI, bool CS = false>
int run(const GemmArgs& a, int splitk, hipStream_t s, const char* name) {
#ifdef V4H_ABLATIONS
  if constexpr (sizeof(T) == 2) {
    if (g_ring) return v4h_gemm_launch<GemmCfg<T, TO, PKS, QKS, BI, BJ, 32, 2, 2, EPI, CS, 0, 4>>(a, splitk, s, name);
  }
#endif
  return v4h_gemm_launch<GemmCfg<T, TO, PKS, QKS, BI, BJ, bk_of<T>(), 2, 2, EPI, CS>>(a, splitk, s, name);
}
// bf16 plain store / GELU / DGELU / slab epilogues of the two-workgroup kernel go through the wave-private LDS strips (GemmCfg variant 9): whole-row
// runs per store instruction.  (The register form - variant 0 - is 6-9 % faster on re-used buffers and 4 % slower inside the step: ablations.)
template <typename T, bool QKS> int run_store(const GemmArgs& a, hipStream_t s, const char* name) {
  if constexpr (sizeof(T) == 2) return v4h_gemm_launch<GemmCfg<T, T, false, QKS, 128, 160, 64, 2, 2, EPI_STORE, false, 9>>(a, 1, s, name);
  else return v4h_gemm_launch<GemmCfg<T, T, false, QKS, 128, 160, bk_of<T>(), 2, 2, EPI_STORE, false>>(a, 1, s, name);
}

template <typename T> int fwd_t(int epi, const GemmArgs& a, hipStream_t s) {
#ifdef V4H_ABLATIONS
  { int rc; if (ablation_fwd<T>(epi, a, s, rc)) return rc; }
#endif
  if constexpr (sizeof(T) == 2) {
    if (g_small && v4h_small::smallm_eligible(a)) {  // batch-row contractions (conditioning MLPs): the whole K extent in one round trip
      if (epi == EPI_SILU) return v4h_small::smallm_launch<false, EPI_SILU>(a, s, "gemm_small/silu");
      if (epi == EPI_COND_SUM) return v4h_small::smallm_launch<false, EPI_COND_SUM>(a, s, "gemm_small/cond_sum");
    }
    if (epi == EPI_STORE && on_ws(a, a.J >= 960 ? 1 : 2)) return run_ws<false, EPI_STORE>(a, s, "gemm3_fwd/store");
    if (epi == EPI_GELU && on_ws(a, a.e.out != nullptr ? 4 : 32) && a.e.out2 != nullptr && a.e.ldo2 % 8 == 0 && ((uintptr_t)a.e.out2 % 16) == 0 && (long)a.I * a.e.ldo2 * 2 < 0x7FFFFFF0L)
      return run_ws<false, EPI_GELU>(a, s, "gemm3_fwd/gelu");
    if (epi == EPI_STORE && on_ring(a, a.K, 1)) return v4h_gemm2_launch<Gemm2Cfg<false, false, EPI_STORE, false>>(a, 1, s, "gemm2_fwd/store");
    if (epi == EPI_GELU && on_ring(a, a.K, a.e.out != nullptr ? 2 : 32) && a.e.ldo2 % 8 == 0) return v4h_gemm2_launch<Gemm2Cfg<false, false, EPI_GELU, false>>(a, 1, s, "gemm2_fwd/gelu");
  }
  switch (epi) {
    case EPI_STORE: {
#ifdef V4H_ABLATIONS
      int rc; if (ablation_store<T, false>(a, s, "gemm_fwd/store", rc)) return rc;
#endif
      return run_store<T, false>(a, s, "gemm_fwd/store");
    }
    case EPI_STORE_F32: return run<T, T, false, false, 128, 160, EPI_STORE_F32>(a, 1, s, "gemm_fwd/store_f32");
    case EPI_SILU: return run<T, T, false, false, 128, 160, EPI_SILU>(a, 1, s, "gemm_fwd/silu");
    case EPI_COND_SUM: return run<T, T, false, false, 128, 160, EPI_COND_SUM>(a, 1, s, "gemm_fwd/cond_sum");
    case EPI_EMBED: return run<T, T, false, false, 128, 160, EPI_EMBED>(a, 1, s, "gemm_fwd/embed");
    case EPI_GATE_RESID: return run<T, T, false, false, 128, 160, EPI_GATE_RESID>(a, 1, s, "gemm_fwd/gate_resid");
    case EPI_GELU:
      if constexpr (sizeof(T) == 2) {
        const int mt = pick_mlp_tile(a);
        if (mt == 1 && a.J % 128 == 0) return v4h_gemm_launch<GemmCfg<T, T, false, false, 128, 128, 64, 2, 2, EPI_GELU, false, 9>>(a, 1, s, "gemm_fwd/gelu128");
        if (mt == 2) return v4h_gemm_launch<GemmCfg<T, T, false, false, 96, 160, 64, 2, 2, EPI_GELU, false, 9>>(a, 1, s, "gemm_fwd/gelu96");
        if (mt == 3 && a.J % 96 == 0) return v4h_gemm_launch<GemmCfg<T, T, false, false, 128, 96, 64, 2, 2, EPI_GELU, false, 9>>(a, 1, s, "gemm_fwd/gelu_96c");
        return v4h_gemm_launch<GemmCfg<T, T, false, false, 128, 160, 64, 2, 2, EPI_GELU, false, 9>>(a, 1, s, "gemm_fwd/gelu");
      }
      else return run<T, T, false, false, 128, 160, EPI_GELU>(a, 1, s, "gemm_fwd/gelu");
    case EPI_UNPATCH: return run<T, T, false, false, 128, 96, EPI_UNPATCH>(a, 1, s, "gemm_fwd/unpatch");
    case EPI_RELU: return run<T, T, false, false, 128, 160, EPI_RELU>(a, 1, s, "gemm_fwd/relu");
    case EPI_ROWADD_SILU: return run<T, T, false, false, 128, 160, EPI_ROWADD_SILU>(a, 1, s, "gemm_fwd/rowadd_silu");
  }
  v4h_set_error("gemm_fwd: epilogue %d not built", epi);
  return V4H_ERR_UNSUPPORTED;
}

template <typename T> int dgrad_t(int epi, const GemmArgs& a, int splitk, hipStream_t s) {
#ifdef V4H_ABLATIONS
  { int rc; if (ablation_dgrad<T>(epi, a, s, rc)) return rc; }
#endif
  if constexpr (sizeof(T) == 2) {
    if (g_small && epi == EPI_DSILU && v4h_small::smallm_eligible(a)) return v4h_small::smallm_launch<true, EPI_DSILU>(a, s, "gemm_small/dsilu");
    if (epi == EPI_STORE && on_ws(a, 8)) return run_ws<true, EPI_STORE>(a, s, "gemm3_dgrad/store");
    if (epi == EPI_DGELU && on_ws(a, 16) && a.J % 32 == 0 && a.e.aux != nullptr && a.e.ld_aux % 8 == 0 && ((uintptr_t)a.e.aux % 16) == 0 && (long)a.I * a.e.ld_aux * 2 < 0x7FFFFFF0L)
      return run_ws<true, EPI_DGELU>(a, s, "gemm3_dgrad/dgelu");
    if (epi == EPI_STORE && on_ring(a, a.K, 4)) return v4h_gemm2_launch<Gemm2Cfg<false, true, EPI_STORE, false>>(a, 1, s, "gemm2_dgrad/store");
    if (epi == EPI_DGELU && on_ring(a, a.K, 8) && a.e.ld_aux % 8 == 0) return v4h_gemm2_launch<Gemm2Cfg<false, true, EPI_DGELU, false>>(a, 1, s, "gemm2_dgrad/dgelu");
  }
  switch (epi) {
    case EPI_STORE: {
#ifdef V4H_ABLATIONS
      int rc; if (ablation_store<T, true>(a, s, "gemm_dgrad/store", rc)) return rc;
#endif
      return run_store<T, true>(a, s, "gemm_dgrad/store");
    }
    case EPI_DGELU:
      if constexpr (sizeof(T) == 2) {
        const int mt = pick_mlp_tile(a);
        if (mt == 1 && a.J % 128 == 0) return v4h_gemm_launch<GemmCfg<T, T, false, true, 128, 128, 64, 2, 2, EPI_DGELU, false, 9>>(a, 1, s, "gemm_dgrad/dgelu128");
        if (mt == 2) return v4h_gemm_launch<GemmCfg<T, T, false, true, 96, 160, 64, 2, 2, EPI_DGELU, false, 9>>(a, 1, s, "gemm_dgrad/dgelu96");
        if (mt == 3 && a.J % 96 == 0) return v4h_gemm_launch<GemmCfg<T, T, false, true, 128, 96, 64, 2, 2, EPI_DGELU, false, 9>>(a, 1, s, "gemm_dgrad/dgelu_96c");
        return v4h_gemm_launch<GemmCfg<T, T, false, true, 128, 160, 64, 2, 2, EPI_DGELU, false, 9>>(a, 1, s, "gemm_dgrad/dgelu");
      }
      else return run<T, T, false, true, 128, 160, EPI_DGELU>(a, 1, s, "gemm_dgrad/dgelu");
    case EPI_DSILU: return run<T, T, false, true, 128, 160, EPI_DSILU>(a, 1, s, "gemm_dgrad/dsilu");
    case EPI_ACCUM_F32: return run<T, T, false, true, 128, 160, EPI_ACCUM_F32>(a, 1, s, "gemm_dgrad/accum");
    case EPI_STORE_F32: return run<T, T, false, true, 128, 160, EPI_STORE_F32>(a, 1, s, "gemm_dgrad/store_f32");
    case EPI_ATOMIC_F32: return run<T, T, false, true, 128, 160, EPI_ATOMIC_F32>(a, splitk, s, "gemm_dgrad/atomic");
  }
  v4h_set_error("gemm_dgrad: epilogue %d not built", epi);
  return V4H_ERR_UNSUPPORTED;
}

template <typename T> int wgrad_t(const GemmArgs& a, int splitk, hipStream_t s) {
#ifdef V4H_ABLATIONS
  { int rc; if (ablation_wgrad<T>(a, splitk, s, rc)) return rc; }
#endif
  if constexpr (sizeof(T) == 2) {  // over B tokens only (embedder MLPs): whole K in one round trip, plain read-modify-write instead of atomics
    if (g_small && splitk <= 1 && v4h_small::smallk_wgrad_eligible(a)) return v4h_small::smallk_wgrad_launch(a, s);
  }
  // tile shape: a plateau (profiles/r02_wgrad_tile_sweep.txt: 160x96, 128x160, 160x160, 96x160 within 4 % of each other at the split the runtime uses)
  return v4h_gemm_launch<GemmCfg<T, T, true, true, 160, 96, bk_of<T>(), 2, 2, EPI_ATOMIC_F32, true>>(a, splitk, s, "gemm_wgrad");
}

}  // namespace

int gemm_fwd(Mode m, int epi, const GemmArgs& a, hipStream_t s) { return m == MODE_BF16 ? fwd_t<bf16>(epi, a, s) : fwd_t<float>(epi, a, s); }
int gemm_dgrad(Mode m, int epi, const GemmArgs& a, hipStream_t s, int splitk) { return m == MODE_BF16 ? dgrad_t<bf16>(epi, a, splitk, s) : dgrad_t<float>(epi, a, splitk, s); }
int gemm_wgrad(Mode m, const GemmArgs& a, int splitk, hipStream_t s) { return m == MODE_BF16 ? wgrad_t<bf16>(a, splitk, s) : wgrad_t<float>(a, splitk, s); }

// Does the split-K weight gradient of this shape take the ring kernel?  256 x 160 tiles, one workgroup per CU: it needs at least a dozen tiles to be
// worth it (attn.proj, 480 x 480 = 6 tiles, stays on the two-workgroup kernel: 32.8 vs 34.5 us at 16 splits alone; inside the step the ring kernel at 8 splits =
// 48 workgroups measured 225.0 vs 228.5 steps/s, round 3, same box).
static bool wgrad_ring_shape(Mode m, int I, int J) {
  const int tiles = ((I + 255) / 256) * (J / 160);
  if (m != MODE_BF16 || I < 160 || J % 160 != 0 || I % 8 != 0 || g_kernel == KERNEL_TWO_WG) return false;
  return g_kernel == KERNEL_RING || ((g_pp & 16) && tiles >= 12);
}
// K splits of a weight gradient: a multiple of 8 (one or more K slices per XCD).  Ring kernel (256 x 160 tiles): 8 - 144 to 192 workgroups for the block's
// shapes.  Alone, one workgroup per CU is faster (18 tiles x 14 splits: 45.2 vs 52.9 us for attn.qkv, 24 x 10: 49.8 vs 53.8 for mlp.fc1, reduction included -
// tools/gemm2_bench.py); inside the backward pass it LOSES (213.2 vs 217.1 steps/s, interleaved on one box; 192 workgroups 214.8, 160: 216.2): the weight
// gradients run on the side stream beside the dgrad chain, a 160 KB-LDS workgroup owns its CU, and the CUs it leaves free are where dgrad runs meanwhile.
// V4H_WGRAD_WGS = n > 0: as many splits as give about n workgroups; -n: n splits.
int gemm_wgrad_splitk(Mode m, int I, int J, int K) {
  int sk;
  if (wgrad_ring_shape(m, I, J) && g_kernel == KERNEL_AUTO) {
    static const int target = env_flag("V4H_WGRAD_WGS", -8);
    sk = target / (((I + 255) / 256) * (J / 160));
    if (sk > 16) sk = 16;
    if (target < 0) sk = -target;
  } else {
    const int tiles = ((I + 95) / 96) * ((J + 159) / 160);
    sk = tiles >= 40 ? 8 : 16;
    if (tiles < 8) sk = 32;
    // (Round 4, measured neutral: 24 / 32 / 34 splits for the 15-tile attn.proj gradient - 480 instead of 240 workgroups on the 512 slots - and the ring
    //  kernel with 16 / 32 / 40 splits for it: 243.0 ... 244.6 against 243.3 steps/s.)
  }
  const int maxk = K / 128;  // at least two K-steps of 64 per split
  if (sk > maxk) sk = maxk;
  return sk < 1 ? 1 : sk;
}
// split-K partials into a slab [nz][I][J] with plain stores; returns the number of splits actually used in *nz_out
int gemm_wgrad_slab(Mode m, const GemmArgs& a0, int splitk, float* slab, int* nz_out, hipStream_t s) {
  GemmArgs a = a0;
  a.e.out = slab; a.e.ldo = a.J; a.e.slab_stride = (long)a.I * a.J;
  const int bk = m == MODE_BF16 ? 64 : 32;
  int klen = (a.K + splitk - 1) / splitk;
  klen = (klen + bk - 1) / bk * bk;
  *nz_out = (a.K + klen - 1) / klen;
  const bool ring_ok = wgrad_ring_shape(m, a.I, a.J) && klen >= 192 && a.K - (*nz_out - 1) * klen >= 192;
#ifdef V4H_ABLATIONS
  { int rc; if (ablation_wgrad_slab(m, a, splitk, ring_ok, s, rc)) return rc; }
#endif
  if (ring_ok) return v4h_gemm2_launch<Gemm2Cfg<true, true, EPI_SLAB_F32, true>>(a, splitk, s, "gemm2_wgrad/slab");
  if (m == MODE_BF16) return v4h_gemm_launch<GemmCfg<bf16, bf16, true, true, 96, 160, 64, 2, 2, EPI_SLAB_F32, true, 9>>(a, splitk, s, "gemm_wgrad/slab");
  return v4h_gemm_launch<GemmCfg<float, float, true, true, 160, 96, 32, 2, 2, EPI_SLAB_F32, true>>(a, splitk, s, "gemm_wgrad/slab");
}

int select_contraction_kernel(int which) {
  if (which != KERNEL_AUTO && which != KERNEL_TWO_WG && which != KERNEL_RING && which != KERNEL_WS) {
    v4h_set_error("select_contraction_kernel: %d is not one of 0 (automatic), 1 (two-workgroup kernel), 2 (ring kernel wherever eligible), 3 (weight-stationary kernel wherever eligible)", which);
    return V4H_ERR_ARG;
  }
  g_kernel = which;
  return V4H_OK;
}
int selected_contraction_kernel() { return g_kernel; }

#ifdef V4H_GEMM3_STAMPS
extern "C" int v4h_debug_gemm3_stamps(void* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(v4h_gemm3_stamp_buf), sizeof(v4h_gemm3_stamp_buf)) == hipSuccess ? 0 : 1;
}
#endif
#ifdef V4H_ABLATIONS
#ifdef V4H_GEMM2_STAMPS
extern "C" int v4h_debug_gemm2_stamps(void* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(v4h_gemm2_stamp_buf), sizeof(v4h_gemm2_stamp_buf)) == hipSuccess ? 0 : 1;
}
#endif
// tuning hook of the ablation build (tools/gemm_bench.py, tools/gemm2_bench.py, tools/wgrad_tile_bench.py): not part of the C ABI
extern "C" void v4h_debug_set_gemm_cfg(int cfg, int cfg_wgrad) {
  if (cfg_wgrad < 0) { g_v2 = -1; g_kernel = KERNEL_AUTO; cfg_wgrad = 0; }  // back to the default choice
  if (cfg_wgrad >= 1000) {
    g_v2 = cfg_wgrad / 1000 - 1;
    g_kernel = g_v2 == 0 ? KERNEL_TWO_WG : g_v2 == 8 ? KERNEL_RING : KERNEL_AUTO;
    cfg_wgrad %= 1000;
  }
  g_cfg = cfg % 100; g_stagger = cfg / 100; g_cfg_wgrad = cfg_wgrad % 100; g_big = cfg_wgrad / 100 ? 0 : 1;
}
#endif

}  // namespace v4h
