// Instantiations + dispatch of the generic contraction (v4h_gemm.h) for the three layouts of a Linear.
#include <stdlib.h>

#include "v4h_ops.h"
#include "v4h_gemm2.h"

namespace v4h {
namespace {

// K-step: 64 for bf16 (128-byte rows = whole cache lines per DMA row, half the barriers), 32 for f32 (also 128-byte rows)
template <typename T> constexpr int bk_of() { return sizeof(T) == 2 ? 64 : 32; }

int g_ring = 0;  // tuning hook: 0 = two buffers x BK 64, 1 = four-deep ring x BK 32 (bf16)
template <typename T, typename TO, bool PKS, bool QKS, int BI, int BJ, int EPI, bool CS = false>
int run(const GemmArgs& a, int splitk, hipStream_t s, const char* name) {
  if constexpr (sizeof(T) == 2) {
    if (g_ring) return v4h_gemm_launch<GemmCfg<T, TO, PKS, QKS, BI, BJ, 32, 2, 2, EPI, CS, 0, 4>>(a, splitk, s, name);
  }
  return v4h_gemm_launch<GemmCfg<T, TO, PKS, QKS, BI, BJ, bk_of<T>(), 2, 2, EPI, CS>>(a, splitk, s, name);
}

// 256 x 192 tile, 12 waves (4 x 3, wave tile 64 x 64), one workgroup per CU: 110 FLOP per staged byte instead of 71.  Pays when
// J is a multiple of 192 (N = 1920) or K is long (fc2 forward); loses for a 192-wide K-strided operand (2-way tr-read conflicts).
static int env_flag(const char* n, int dflt) { const char* e = getenv(n); return e ? atoi(e) : dflt; }
int g_big = env_flag("V4H_GEMM_BIG", 0);  // tuning hook / env: 0 disables
template <typename T, typename TO, bool QKS, int EPI> int run_big(const GemmArgs& a, hipStream_t s, const char* name) {
  return v4h_gemm_launch<GemmCfg<T, TO, false, QKS, 256, 192, 64, 4, 3, EPI, false>>(a, 1, s, name);
}
inline bool big_fwd(const GemmArgs& a) { return g_big && a.I >= 4096 && (a.J % 192 == 0 || a.K >= 1920); }
inline bool big_dgrad(const GemmArgs& a) { return g_big && a.I >= 4096 && a.J % 192 == 0 && a.J >= 960; }

// ---- tile-shape tuning hook (tools/gemm_bench.py): selects the configuration used for EPI_STORE fwd/dgrad and wgrad ----
int g_cfg = env_flag("V4H_GEMM_CFG", 0), g_cfg_wgrad = env_flag("V4H_GEMM_WCFG", 0), g_stagger = 0;  // (env: A/B runs of whole steps)
// Which epilogues go through the LDS strips (default: all) instead of straight from the registers (bit 0 plain store, 1 GELU, 2 DGELU, 3 split-K slab).
// The register form (v_permlane16_swap pairs, 16-byte stores, 64-byte row segments) wins 6-9 % per call when the SAME buffers are re-used in a
// loop (tools/gemm_bench.py: outputs stay cache-resident) and LOSES inside the update step (201.6 vs 209.9 steps/s with all four on it; GELU with
// its two outputs 84.8 vs 69.1 us per call): cold output lines written in 64-byte pieces cost more than whole-row strips.  Kept as a measured
// negative result and as the A/B hook that found it (profiles/r02_gemm_direct_store.txt).
int g_strips = env_flag("V4H_GEMM_STRIPS", 15);
template <typename T, bool QKS> int run_store_cfg(const GemmArgs& a0, hipStream_t s, const char* name) {
  GemmArgs a = a0;
  a.stagger_sleeps = g_stagger;
  if constexpr (sizeof(T) == 2) {
    if (g_cfg == 0 && (g_strips & 1)) return v4h_gemm_launch<GemmCfg<T, T, false, QKS, 128, 160, 64, 2, 2, EPI_STORE, false, 9>>(a, 1, s, name);
    if (g_cfg == 0 && g_ring) return v4h_gemm_launch<GemmCfg<T, T, false, QKS, 128, 160, 32, 2, 2, EPI_STORE, false, 0, 4>>(a, 1, s, name);
    if (g_cfg == 20) return v4h_gemm_launch<GemmCfg<T, T, false, QKS, 128, 160, 64, 2, 2, EPI_STORE, false>>(a, 1, s, name);
    if (g_cfg == 21) return v4h_gemm_launch<GemmCfg<T, T, false, QKS, 128, 160, 32, 2, 2, EPI_STORE, false, 4, 4>>(a, 1, s, name);
    switch (g_cfg) {
      case 2: return v4h_gemm_launch<GemmCfg<T, T, false, QKS, 256, 160, 64, 4, 2, EPI_STORE, false>>(a, 1, s, name);
      case 4: return v4h_gemm_launch<GemmCfg<T, T, false, QKS, 256, 96, 64, 4, 2, EPI_STORE, false>>(a, 1, s, name);
      case 5: return v4h_gemm_launch<GemmCfg<T, T, false, QKS, 128, 96, 64, 2, 2, EPI_STORE, false>>(a, 1, s, name);
      case 6: return v4h_gemm_launch<GemmCfg<T, T, false, QKS, 256, 160, 64, 2, 2, EPI_STORE, false>>(a, 1, s, name);
      case 8: return v4h_gemm_launch<GemmCfg<T, T, false, QKS, 128, 160, 64, 2, 2, EPI_STORE, false, 1>>(a, 1, s, name);
      case 9: return v4h_gemm_launch<GemmCfg<T, T, false, QKS, 128, 160, 64, 2, 2, EPI_STORE, false, 2>>(a, 1, s, name);
      case 13: return v4h_gemm_launch<GemmCfg<T, T, false, QKS, 128, 128, 64, 2, 2, EPI_STORE, false>>(a, 1, s, name);
      case 14: return v4h_gemm_launch<GemmCfg<T, T, false, QKS, 128, 128, 64, 2, 2, EPI_STORE, false, 3>>(a, 1, s, name);
      case 18: if constexpr (!QKS) return v4h_gemm_launch<GemmCfg<T, T, false, false, 256, 240, 64, 4, 3, EPI_STORE, false>>(a, 1, s, name); else break;
      case 19: return v4h_gemm_launch<GemmCfg<T, T, false, QKS, 256, 192, 64, 4, 3, EPI_STORE, false>>(a, 1, s, name);
      case 22: return v4h_gemm_launch<GemmCfg<T, T, false, QKS, 256, 192, 64, 4, 2, EPI_STORE, false>>(a, 1, s, name);
      case 28: return v4h_gemm_launch<GemmCfg<T, T, false, QKS, 128, 160, 64, 2, 2, EPI_STORE, false, 7>>(a, 1, s, name);
      case 29: return v4h_gemm_launch<GemmCfg<T, T, false, QKS, 128, 160, 64, 2, 2, EPI_STORE, false, 9>>(a, 1, s, name);  // LDS-strip epilogue (A/B of the register store)
      // (measured and removed: 112-row tiles - 465 instead of 405 tiles on the 512 slots for J = 480 - as 1 x 5 or 1 x 2 waves: 13-35 % slower)
      // (measured and removed: one 8-wave workgroup per CU with a 4-deep ring of BK = 64 slabs, 108 KB in flight: 25-40 % slower
      //  than two 4-wave workgroups with two slabs each - DESIGN.md section 5)
      case 17: return v4h_gemm_launch<GemmCfg<T, T, false, QKS, 128, 160, 64, 2, 2, EPI_STORE, false, 6>>(a, 1, s, name);
      case 16: return v4h_gemm_launch<GemmCfg<T, T, false, QKS, 128, 160, 64, 2, 2, EPI_STORE, false, 5>>(a, 1, s, name);
      case 15: return v4h_gemm_launch<GemmCfg<T, T, false, QKS, 128, 160, 64, 2, 2, EPI_STORE, false, 4>>(a, 1, s, name);
      case 12: return v4h_gemm_launch<GemmCfg<T, T, false, QKS, 128, 160, 64, 2, 2, EPI_STORE, false, 3>>(a, 1, s, name);
      case 10: return v4h_gemm_launch<GemmCfg<T, T, false, QKS, 256, 160, 64, 4, 2, EPI_STORE, false, 1>>(a, 1, s, name);
      case 11: return v4h_gemm_launch<GemmCfg<T, T, false, QKS, 256, 160, 64, 4, 2, EPI_STORE, false, 2>>(a, 1, s, name);
      default: break;
    }
  }
  if constexpr (sizeof(T) == 2) {
  }
  return v4h_gemm_launch<GemmCfg<T, T, false, QKS, 128, 160, bk_of<T>(), 2, 2, EPI_STORE, false>>(a, 1, s, name);
}
template <typename T> int run_wgrad_cfg(const GemmArgs& a, int splitk, hipStream_t s) {
  if constexpr (sizeof(T) == 2) {
    switch (g_cfg_wgrad) {
      case 1: return v4h_gemm_launch<GemmCfg<T, T, true, true, 160, 160, 64, 2, 2, EPI_ATOMIC_F32, true>>(a, splitk, s, "gemm_wgrad");
      case 2: return v4h_gemm_launch<GemmCfg<T, T, true, true, 320, 160, 32, 4, 2, EPI_ATOMIC_F32, true>>(a, splitk, s, "gemm_wgrad");
      case 3: return v4h_gemm_launch<GemmCfg<T, T, true, true, 320, 160, 64, 4, 2, EPI_ATOMIC_F32, true>>(a, splitk, s, "gemm_wgrad");
      default: break;
    }
  }
  if constexpr (sizeof(T) == 2) {
    if (g_cfg_wgrad == 7) return v4h_gemm_launch<GemmCfg<T, T, true, true, 160, 160, 32, 2, 2, EPI_ATOMIC_F32, true, 0, 4>>(a, splitk, s, "gemm_wgrad");
    if (g_cfg_wgrad == 8) return v4h_gemm_launch<GemmCfg<T, T, true, true, 160, 96, 32, 2, 2, EPI_ATOMIC_F32, true, 0, 4>>(a, splitk, s, "gemm_wgrad");
    return v4h_gemm_launch<GemmCfg<T, T, true, true, 160, 96, 64, 2, 2, EPI_ATOMIC_F32, true>>(a, splitk, s, "gemm_wgrad");
  }
  return v4h_gemm_launch<GemmCfg<T, T, true, true, 160, 96, 32, 2, 2, EPI_ATOMIC_F32, true>>(a, splitk, s, "gemm_wgrad");
}

// ---- 256 x 160 / 8-wave / three-stage kernel (v4h_gemm2.h): bf16, token-sized contractions ----
// V4H_GEMM2: -1 (default) = where it was measured to win INSIDE the path: forward contractions whose token count is a whole number of 256-row tiles
// (the sampler at the reference's batch of 256: 34560 rows = 135 tiles; 1210 -> 1270 showers/s, profiles/r02_ab_in_context.txt); 0 = never;
// 1 = wherever eligible (the update step at bs = 128, 67.5 row tiles: 195 vs 208 steps/s - not the default); 2.. = ablation builds (tools/gemm2_bench.py).
int g_v2 = env_flag("V4H_GEMM2", -1);
// Which contractions take the ring kernel's ping-pong schedule when V4H_GEMM2 is -1 (bits: 1 forward plain store, 2 forward GELU of the update step - two
// outputs, 4 dgrad plain store, 8 dgrad DGELU, 16 split-K weight-gradient slabs, 32 forward GELU without the saved derivative - inference), wherever the shape
// is eligible.  Default: everything but the two heavy epilogues of the update step, which have nothing to hide behind in that schedule.
int g_pp = env_flag("V4H_GEMM2_PP", 53);
inline bool v2_eligible(const GemmArgs& a, int klen) {
  return a.I >= 2048 && a.J % 160 == 0 && klen >= 192 && a.e.ldo % 8 == 0 && ((uintptr_t)a.e.out % 16) == 0 && (long)a.I * a.e.ldo * 4 < 0x7FFFFFF0L;
}
inline bool v2_ok(const GemmArgs& a, int klen) { return g_v2 > 0 && v2_eligible(a, klen); }
int g_fwd_lockstep = env_flag("V4H_FWD_LOCKSTEP", 0);  // 1: forward at whole 256-row tiles (the sampler at its batch of 256) on the LOCK-STEP schedule of the ring kernel
                                                       // (the default until the ping-pong schedule dropped its mid-stage barrier: 1292 vs 1342 showers/s)
inline bool v2_auto_fwd(const GemmArgs& a) { return g_v2 < 0 && g_fwd_lockstep && a.I % 256 == 0 && a.I >= 8192 && v2_eligible(a, a.K); }
inline bool pp_auto(const GemmArgs& a, int klen, int bit) { return g_v2 < 0 && (g_pp & bit) && v2_eligible(a, klen); }

template <typename T> int fwd_t(int epi, const GemmArgs& a, hipStream_t s) {
  if constexpr (sizeof(T) == 2) {
    if (epi == EPI_STORE && pp_auto(a, a.K, 1) && !v2_auto_fwd(a)) return v4h_gemm2_launch<Gemm2Cfg<false, false, EPI_STORE, false, 0, true>>(a, 1, s, "gemm2pp_fwd/store");
    if (epi == EPI_GELU && pp_auto(a, a.K, a.e.out != nullptr ? 2 : 32) && !v2_auto_fwd(a) && a.e.ldo2 % 8 == 0) return v4h_gemm2_launch<Gemm2Cfg<false, false, EPI_GELU, false, 0, true>>(a, 1, s, "gemm2pp_fwd/gelu");
    if (v2_ok(a, a.K) || v2_auto_fwd(a)) {
      if (epi == EPI_STORE && g_v2 == 2) return v4h_gemm2_launch<Gemm2Cfg<false, false, EPI_STORE, false, 1>>(a, 1, s, "gemm2_fwd/store/dbg1");
      if (epi == EPI_STORE && g_v2 == 3) return v4h_gemm2_launch<Gemm2Cfg<false, false, EPI_STORE, false, 2>>(a, 1, s, "gemm2_fwd/store/dbg2");
      if (epi == EPI_STORE && g_v2 == 4) return v4h_gemm2_launch<Gemm2Cfg<false, false, EPI_STORE, false, 3>>(a, 1, s, "gemm2_fwd/store/dbg3");
      if (epi == EPI_STORE && g_v2 == 5) return v4h_gemm2_launch<Gemm2Cfg<false, false, EPI_STORE, false, 7>>(a, 1, s, "gemm2_fwd/store/dbg7");
      if (epi == EPI_STORE && g_v2 == 6) return v4h_gemm2_launch<Gemm2Cfg<false, false, EPI_STORE, false, 11>>(a, 1, s, "gemm2_fwd/store/dbg11");
      if (epi == EPI_STORE && g_v2 == 7) return v4h_gemm2_launch<Gemm2Cfg<false, false, EPI_STORE, false, 15>>(a, 1, s, "gemm2_fwd/store/dbg15");
      if (epi == EPI_STORE && g_v2 == 9) return v4h_gemm2_launch<Gemm2Cfg<false, false, EPI_STORE, false, 1, true>>(a, 1, s, "gemm2pp_fwd/store/dbg1");
      if (epi == EPI_STORE && g_v2 == 10) return v4h_gemm2_launch<Gemm2Cfg<false, false, EPI_STORE, false, 2, true>>(a, 1, s, "gemm2pp_fwd/store/dbg2");
      if (epi == EPI_STORE && g_v2 == 11) return v4h_gemm2_launch<Gemm2Cfg<false, false, EPI_STORE, false, 3, true>>(a, 1, s, "gemm2pp_fwd/store/dbg3");
      if (epi == EPI_STORE && g_v2 == 8) return v4h_gemm2_launch<Gemm2Cfg<false, false, EPI_STORE, false, 0, true>>(a, 1, s, "gemm2pp_fwd/store");
      if (epi == EPI_GELU && g_v2 == 8 && a.e.ldo2 % 8 == 0) return v4h_gemm2_launch<Gemm2Cfg<false, false, EPI_GELU, false, 0, true>>(a, 1, s, "gemm2pp_fwd/gelu");
      if (epi == EPI_STORE) return v4h_gemm2_launch<Gemm2Cfg<false, false, EPI_STORE, false>>(a, 1, s, "gemm2_fwd/store");
      if (epi == EPI_GELU && a.e.ldo2 % 8 == 0) return v4h_gemm2_launch<Gemm2Cfg<false, false, EPI_GELU, false>>(a, 1, s, "gemm2_fwd/gelu");
    }
  }
  switch (epi) {
    case EPI_STORE: return run_store_cfg<T, false>(a, s, "gemm_fwd/store");
    case EPI_STORE_F32: return run<T, T, false, false, 128, 160, EPI_STORE_F32>(a, 1, s, "gemm_fwd/store_f32");
    case EPI_SILU: return run<T, T, false, false, 128, 160, EPI_SILU>(a, 1, s, "gemm_fwd/silu");
    case EPI_COND_SUM: return run<T, T, false, false, 128, 160, EPI_COND_SUM>(a, 1, s, "gemm_fwd/cond_sum");
    case EPI_EMBED: return run<T, T, false, false, 128, 160, EPI_EMBED>(a, 1, s, "gemm_fwd/embed");
    case EPI_GATE_RESID:
      if constexpr (sizeof(T) == 2) { if (big_fwd(a)) return run_big<T, T, false, EPI_GATE_RESID>(a, s, "gemm_fwd/gate_resid/big"); }
      return run<T, T, false, false, 128, 160, EPI_GATE_RESID>(a, 1, s, "gemm_fwd/gate_resid");
    case EPI_GELU:
      if constexpr (sizeof(T) == 2) { if (g_cfg == 29 || (g_strips & 2)) return v4h_gemm_launch<GemmCfg<T, T, false, false, 128, 160, 64, 2, 2, EPI_GELU, false, 9>>(a, 1, s, "gemm_fwd/gelu/strips"); }
      if constexpr (sizeof(T) == 2) { if (big_fwd(a)) return run_big<T, T, false, EPI_GELU>(a, s, "gemm_fwd/gelu/big"); }
      return run<T, T, false, false, 128, 160, EPI_GELU>(a, 1, s, "gemm_fwd/gelu");
    case EPI_UNPATCH: return run<T, T, false, false, 128, 96, EPI_UNPATCH>(a, 1, s, "gemm_fwd/unpatch");
    case EPI_RELU: return run<T, T, false, false, 128, 160, EPI_RELU>(a, 1, s, "gemm_fwd/relu");
    case EPI_ROWADD_SILU: return run<T, T, false, false, 128, 160, EPI_ROWADD_SILU>(a, 1, s, "gemm_fwd/rowadd_silu");
  }
  v4h_set_error("gemm_fwd: epilogue %d not built", epi);
  return V4H_ERR_UNSUPPORTED;
}

template <typename T> int dgrad_t(int epi, const GemmArgs& a, int splitk, hipStream_t s) {
  if constexpr (sizeof(T) == 2) {
    if (epi == EPI_STORE && pp_auto(a, a.K, 4)) return v4h_gemm2_launch<Gemm2Cfg<false, true, EPI_STORE, false, 0, true>>(a, 1, s, "gemm2pp_dgrad/store");
    if (epi == EPI_DGELU && pp_auto(a, a.K, 8) && a.e.ld_aux % 8 == 0) return v4h_gemm2_launch<Gemm2Cfg<false, true, EPI_DGELU, false, 0, true>>(a, 1, s, "gemm2pp_dgrad/dgelu");
    if (v2_ok(a, a.K)) {
      if (epi == EPI_STORE && g_v2 == 8) return v4h_gemm2_launch<Gemm2Cfg<false, true, EPI_STORE, false, 0, true>>(a, 1, s, "gemm2pp_dgrad/store");
      if (epi == EPI_DGELU && g_v2 == 8 && a.e.ld_aux % 8 == 0) return v4h_gemm2_launch<Gemm2Cfg<false, true, EPI_DGELU, false, 0, true>>(a, 1, s, "gemm2pp_dgrad/dgelu");
      if (epi == EPI_STORE) return v4h_gemm2_launch<Gemm2Cfg<false, true, EPI_STORE, false>>(a, 1, s, "gemm2_dgrad/store");
      if (epi == EPI_DGELU && a.e.ld_aux % 8 == 0) return v4h_gemm2_launch<Gemm2Cfg<false, true, EPI_DGELU, false>>(a, 1, s, "gemm2_dgrad/dgelu");
    }
  }
  switch (epi) {
    case EPI_STORE: return run_store_cfg<T, true>(a, s, "gemm_dgrad/store");
    case EPI_DGELU:
      if constexpr (sizeof(T) == 2) { if (g_cfg == 29 || (g_strips & 4)) return v4h_gemm_launch<GemmCfg<T, T, false, true, 128, 160, 64, 2, 2, EPI_DGELU, false, 9>>(a, 1, s, "gemm_dgrad/dgelu/strips"); }
      if constexpr (sizeof(T) == 2) { if (big_dgrad(a)) return run_big<T, T, true, EPI_DGELU>(a, s, "gemm_dgrad/dgelu/big"); }
      return run<T, T, false, true, 128, 160, EPI_DGELU>(a, 1, s, "gemm_dgrad/dgelu");
    case EPI_DSILU: return run<T, T, false, true, 128, 160, EPI_DSILU>(a, 1, s, "gemm_dgrad/dsilu");
    case EPI_ACCUM_F32: return run<T, T, false, true, 128, 160, EPI_ACCUM_F32>(a, 1, s, "gemm_dgrad/accum");
    case EPI_STORE_F32: return run<T, T, false, true, 128, 160, EPI_STORE_F32>(a, 1, s, "gemm_dgrad/store_f32");
    case EPI_ATOMIC_F32: return run<T, T, false, true, 128, 160, EPI_ATOMIC_F32>(a, splitk, s, "gemm_dgrad/atomic");
  }
  v4h_set_error("gemm_dgrad: epilogue %d not built", epi);
  return V4H_ERR_UNSUPPORTED;
}

}  // namespace

int gemm_fwd(Mode m, int epi, const GemmArgs& a, hipStream_t s) { return m == MODE_BF16 ? fwd_t<bf16>(epi, a, s) : fwd_t<float>(epi, a, s); }
int gemm_dgrad(Mode m, int epi, const GemmArgs& a, hipStream_t s, int splitk) { return m == MODE_BF16 ? dgrad_t<bf16>(epi, a, splitk, s) : dgrad_t<float>(epi, a, splitk, s); }
int gemm_wgrad(Mode m, const GemmArgs& a, int splitk, hipStream_t s) {
  return m == MODE_BF16 ? run_wgrad_cfg<bf16>(a, splitk, s) : run_wgrad_cfg<float>(a, splitk, s);
}
// Does the split-K weight gradient of this shape take the ring kernel (ping-pong schedule)?  256 x 160 tiles, one workgroup per CU: it needs at least a
// dozen tiles to be worth it (attn.proj, 480 x 480 = 6 tiles, stays on the two-workgroup kernel: 32.8 vs 34.5 us at 16 splits).
static bool wgrad_ring_shape(Mode m, int I, int J) {
  const int tiles = ((I + 255) / 256) * (J / 160);
  return m == MODE_BF16 && I >= 160 && J % 160 == 0 && I % 8 == 0 && (g_v2 > 0 || (g_v2 < 0 && (g_pp & 16) && tiles >= 12));
}
// K splits of a weight gradient: a multiple of 8 (one or more K slices per XCD).  Ring kernel (256 x 160 tiles): 8 - 144 to 192 workgroups for the block's
// shapes.  Alone, one workgroup per CU is faster (18 tiles x 14 splits: 45.2 vs 52.9 us for attn.qkv, 24 x 10: 49.8 vs 53.8 for mlp.fc1, reduction included -
// tools/gemm2_bench.py); inside the backward pass it LOSES (213.2 vs 217.1 steps/s, interleaved on one box; 192 workgroups 214.8, 160: 216.2): the weight
// gradients run on the side stream beside the dgrad chain, a 160 KB-LDS workgroup owns its CU, and the CUs it leaves free are where dgrad runs meanwhile.
// V4H_WGRAD_WGS = n > 0: as many splits as give about n workgroups; -n: n splits.
int gemm_wgrad_splitk(Mode m, int I, int J, int K) {
  int sk;
  if (wgrad_ring_shape(m, I, J) && g_v2 < 0) {
    static const int target = env_flag("V4H_WGRAD_WGS", -8);
    sk = target / (((I + 255) / 256) * (J / 160));
    if (sk > 16) sk = 16;
    if (target < 0) sk = -target;
  } else {
    const int tiles = ((I + 95) / 96) * ((J + 159) / 160);
    sk = tiles >= 40 ? 8 : 16;
    if (tiles < 8) sk = 32;
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
  if (wgrad_ring_shape(m, a.I, a.J) && klen >= 192 && a.K - (*nz_out - 1) * klen >= 192)
    return (g_v2 == 8 || g_v2 < 0) ? v4h_gemm2_launch<Gemm2Cfg<true, true, EPI_SLAB_F32, true, 0, true>>(a, splitk, s, "gemm2pp_wgrad/slab")
                     : v4h_gemm2_launch<Gemm2Cfg<true, true, EPI_SLAB_F32, true>>(a, splitk, s, "gemm2_wgrad/slab");
  // tile shape: a plateau (tools/wgrad_tile_bench.py, profiles/r02_wgrad_tile_sweep.txt: 160x96, 128x160, 160x160, 96x160 within 4 % of each other at the
  // split the runtime uses); 96 x 160 is 2-4 % ahead on three of the four block shapes
  if (m == MODE_BF16 && (g_cfg_wgrad == 13 || (g_strips & 8))) return v4h_gemm_launch<GemmCfg<bf16, bf16, true, true, 96, 160, 64, 2, 2, EPI_SLAB_F32, true, 9>>(a, splitk, s, "gemm_wgrad/slab/strips");
  if (m == MODE_BF16 && g_cfg_wgrad == 12) return v4h_gemm_launch<GemmCfg<bf16, bf16, true, true, 160, 96, 64, 2, 2, EPI_SLAB_F32, true>>(a, splitk, s, "gemm_wgrad/slab160x96");
  if (m == MODE_BF16) return v4h_gemm_launch<GemmCfg<bf16, bf16, true, true, 96, 160, 64, 2, 2, EPI_SLAB_F32, true>>(a, splitk, s, "gemm_wgrad/slab");
  return v4h_gemm_launch<GemmCfg<float, float, true, true, 160, 96, 32, 2, 2, EPI_SLAB_F32, true>>(a, splitk, s, "gemm_wgrad/slab");
}

#ifdef V4H_GEMM2_STAMPS
extern "C" int v4h_debug_gemm2_stamps(void* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(v4h_gemm2_stamp_buf), sizeof(v4h_gemm2_stamp_buf)) == hipSuccess ? 0 : 1;
}
#endif

void debug_set_gemm_cfg(int cfg, int cfg_wgrad) {
  if (cfg_wgrad < 0) { g_v2 = -1; cfg_wgrad = 0; }  // back to the default choice
  if (cfg_wgrad >= 1000) { g_v2 = cfg_wgrad / 1000 - 1; cfg_wgrad %= 1000; }
  g_cfg = cfg % 100; g_stagger = cfg / 100; g_cfg_wgrad = cfg_wgrad % 100; g_big = cfg_wgrad / 100 ? 0 : 1; }

}  // namespace v4h
