// Instantiations + dispatch of the generic contraction (v4h_gemm.h) for the three layouts of a Linear.
#include "v4h_ops.h"

namespace v4h {
namespace {

template <typename T, typename TO, bool PKS, bool QKS, int BI, int BJ, int EPI, bool CS = false>
int run(const GemmArgs& a, int splitk, hipStream_t s, const char* name) {
  return v4h_gemm_launch<GemmCfg<T, TO, PKS, QKS, BI, BJ, 32, EPI, CS>>(a, splitk, s, name);
}

template <typename T> int fwd_t(int epi, const GemmArgs& a, hipStream_t s) {
  switch (epi) {
    case EPI_STORE: return run<T, T, false, false, 128, 160, EPI_STORE>(a, 1, s, "gemm_fwd/store");
    case EPI_STORE_F32: return run<T, T, false, false, 128, 160, EPI_STORE_F32>(a, 1, s, "gemm_fwd/store_f32");
    case EPI_SILU: return run<T, T, false, false, 128, 160, EPI_SILU>(a, 1, s, "gemm_fwd/silu");
    case EPI_COND_SUM: return run<T, T, false, false, 128, 160, EPI_COND_SUM>(a, 1, s, "gemm_fwd/cond_sum");
    case EPI_EMBED: return run<T, T, false, false, 128, 160, EPI_EMBED>(a, 1, s, "gemm_fwd/embed");
    case EPI_GATE_RESID: return run<T, T, false, false, 128, 160, EPI_GATE_RESID>(a, 1, s, "gemm_fwd/gate_resid");
    case EPI_GELU: return run<T, T, false, false, 128, 160, EPI_GELU>(a, 1, s, "gemm_fwd/gelu");
    case EPI_UNPATCH: return run<T, T, false, false, 128, 96, EPI_UNPATCH>(a, 1, s, "gemm_fwd/unpatch");
  }
  v4h_set_error("gemm_fwd: epilogue %d not built", epi);
  return V4H_ERR_UNSUPPORTED;
}

template <typename T> int dgrad_t(int epi, const GemmArgs& a, hipStream_t s) {
  switch (epi) {
    case EPI_STORE: return run<T, T, false, true, 128, 160, EPI_STORE>(a, 1, s, "gemm_dgrad/store");
    case EPI_DGELU: return run<T, T, false, true, 128, 160, EPI_DGELU>(a, 1, s, "gemm_dgrad/dgelu");
    case EPI_DSILU: return run<T, T, false, true, 128, 160, EPI_DSILU>(a, 1, s, "gemm_dgrad/dsilu");
    case EPI_ACCUM_F32: return run<T, T, false, true, 128, 160, EPI_ACCUM_F32>(a, 1, s, "gemm_dgrad/accum");
    case EPI_STORE_F32: return run<T, T, false, true, 128, 160, EPI_STORE_F32>(a, 1, s, "gemm_dgrad/store_f32");
  }
  v4h_set_error("gemm_dgrad: epilogue %d not built", epi);
  return V4H_ERR_UNSUPPORTED;
}

}  // namespace

int gemm_fwd(Mode m, int epi, const GemmArgs& a, hipStream_t s) { return m == MODE_BF16 ? fwd_t<bf16>(epi, a, s) : fwd_t<float>(epi, a, s); }
int gemm_dgrad(Mode m, int epi, const GemmArgs& a, hipStream_t s) { return m == MODE_BF16 ? dgrad_t<bf16>(epi, a, s) : dgrad_t<float>(epi, a, s); }
int gemm_wgrad(Mode m, const GemmArgs& a, int splitk, hipStream_t s) {
  if (m == MODE_BF16) return run<bf16, bf16, true, true, 160, 160, EPI_ATOMIC_F32, true>(a, splitk, s, "gemm_wgrad");
  return run<float, float, true, true, 160, 160, EPI_ATOMIC_F32, true>(a, splitk, s, "gemm_wgrad");
}

}  // namespace v4h
