// Internal C++ interface between the kernel translation units and the runtime (v4h_runtime.hip).
// Everything here takes raw device pointers + a hipStream_t; the public C ABI is include/vit4hep_hip.h.
#pragma once
#include <utility>
#include "v4h_gemm.h"

namespace v4h {

enum Mode : int { MODE_F32 = 0, MODE_BF16 = 1 };
inline size_t esize(Mode m) { return m == MODE_BF16 ? 2 : 4; }

// ---- contractions (v4h_gemm.hip) ----
int gemm_fwd(Mode m, int epi, const GemmArgs& a, hipStream_t s);              // P contig, Q contig
int gemm_dgrad(Mode m, int epi, const GemmArgs& a, hipStream_t s, int splitk = 1);  // P contig, Q K-strided (split-K only with EPI_ATOMIC_F32)
int gemm_wgrad(Mode m, const GemmArgs& a, int splitk, hipStream_t s);         // both K-strided, f32 atomics (+ colsum)
int gemm_wgrad_splitk(Mode m, int I, int J, int K);  // the split count the runtime uses for a weight gradient of this shape
int gemm_wgrad_slab(Mode m, const GemmArgs& a, int splitk, float* slab, int* nz_out, hipStream_t s);  // partials [nz][I][J], plain stores
int slab_reduce(const float* slab, int nz, long n, float* out, hipStream_t s, bool set = false);  // out[k] (+)= sum_z slab[z*n + k]; set: plain store
constexpr int ZERO_MAX_ITEMS = 64;
struct ZeroTable { float* p[ZERO_MAX_ITEMS]; long n[ZERO_MAX_ITEMS]; };
int zero_many(const std::pair<float*, long>* items, int n, hipStream_t s);  // items[i] = (16-byte aligned buffer, floats): all zeroed by one launch per 64

int select_contraction_kernel(int which);  // 0 automatic, 1 two-workgroup kernel everywhere, 2 ring kernel wherever eligible (all exact)
int selected_contraction_kernel();

// ---- attention (v4h_attention.hip) ----
int attention_fwd(Mode m, const void* qkv, void* o, float* lse, int B, int T, int H, int DH, hipStream_t s);
int attention_bwd(Mode m, const void* qkv, const void* o, const void* dout, const float* lse, float* delta, void* dqkv, int B, int T, int H, int DH,
                  hipStream_t s);

// ---- element-wise / reductions (v4h_elementwise.hip) ----
struct CastPadItem { const float* src; void* dst; int R, C, Rp, Cp; int dst_f32; };  // dst[Rp][Cp] (mode type, or f32 if dst_f32) <- zero-padded src[R][C]
int cast_pad_many(Mode m, const CastPadItem* items, int n, hipStream_t s);
struct PtrTable { float* p[2 * V4H_GEMM_MAX_GROUPS]; };
int write_ptr_table(const PtrTable& t, float** dst, hipStream_t s);                      // dst[k] = t.p[k]: a small device table filled from kernel arguments
int unpad_f32(const float* src, int ld_src, float* dst, int R, int C, hipStream_t s);   // dst[R][C] += src[r][c]

int patchify(Mode m, const float* vox, void* xp, int B, const PatchGeom& g, int P, int Ppad, hipStream_t s);
int unpatchify_f32(const float* tok, int ld, float* vox, int B, const PatchGeom& g, int P, hipStream_t s);
// general geometry (index map [T*P], V voxels per sample)
int patchify_map(Mode m, bool out_f32, const float* vox, const int* map, void* xp, int B, long V, int T, int P, int Ppad, hipStream_t s);
int unpatchify_map_f32(const float* tok, int ld, const int* map, float* vox, int B, long V, int T, int P, hipStream_t s);
int pos_embed_fwd_pos(const float* freqs, const float* pos, float* pe, int T, int D, hipStream_t s);
int pos_embed_bwd_pos(Mode m, const void* dx0, const float* freqs, const float* pos, float* dfreqs, float* scratch, int B, int T, int D, hipStream_t s);
int pos_embed_fwd(const float* freqs, float* pe, const PatchGeom& g, int D, hipStream_t s);
int pos_embed_bwd(Mode m, const void* dx0, const float* freqs, float* dfreqs, float* scratch, int B, const PatchGeom& g, int D, hipStream_t s);
int timestep_embed(Mode m, const float* t, void* out, int B, int F, hipStream_t s);

// Storage of the residual stream x (and of its gradient d x): f32, or - bf16 mode only, round 5 - the mode type.  The LayerNorm kernels are the only
// readers / writers of both besides the embedding epilogue; their statistics, the gated update and every sum stay f32 in registers.
//   x16:  x, x_out (forward), LnBwdArgs::x are mode-typed     g16: LnBwdArgs::dx_in, dx_out are mode-typed
bool ln_resid16_supported(Mode m, int D);  // can the 16-bit residual forms serve this width?  (bf16 mode, D % 8 == 0, D <= 512)
int ln_modulate_fwd(Mode m, const void* x, const float* shift, const float* scale, int ld_mod, void* u, float* mean, float* rstd, int BT, int T, int D,
                    hipStream_t s, bool x16 = false);
bool ln_resid_supported(int D);
// x_out = x + gate[b] * y (gated residual of the branch above, nn/vit.py:331-332), then LayerNorm + modulate of x_out
int ln_resid_modulate_fwd(Mode m, const void* x, const void* y, const float* gate, int ld_gate, void* x_out, const float* shift, const float* scale, int ld_mod,
                          void* u, float* mean, float* rstd, int BT, int T, int D, hipStream_t s, bool x16 = false);
struct LnBwdArgs {
  // LayerNorm+modulate backward (reference nn/vit.py:309-311,331-332,457-458)
  const void* du;       // [BT][D] mode type: grad wrt the modulated output
  const void* x;        // [BT][D] LayerNorm input (f32, or mode type with x16)
  const float* mean; const float* rstd;
  const float* scale;   // mod chunk, row stride ld_mod
  int ld_mod;
  const void* dx_in;    // residual-stream grad to add (may be null); f32, or mode type with g16
  void* dx_out;         // f32 / mode type with g16 (may be null when only dx_out_t is wanted)
  void* dx_out_t;       // optional copy in mode type
  float* dshift; float* dscale; int ld_dmod;   // f32 atomics, [B][ld_dmod] chunks
  // fused gate backward of the branch below (nn/vit.py:331-332): dy = gate * dx_out ; dgate += sum_t dx_out * y
  const void* y;        // [BT][D] mode type (null -> skip)
  const float* gate;    // mod chunk (row stride ld_mod_gate)
  int ld_mod_gate;
  void* dy;             // [BT][D] mode type
  float* dgate; int ld_dgate;
  int B, T, D;
  int x16, g16;         // storage of x / of dx_in, dx_out: 0 = f32, 1 = mode type (ln_resid16_supported)
};
int ln_modulate_bwd(Mode m, const LnBwdArgs& a, hipStream_t s);
int silu_bwd(Mode m, const float* dsilu, const float* pre, void* out, int n, hipStream_t s);  // out = dsilu * silu'(pre)

// ---- CFM step pieces ----
int cfm_prepare(const float* x1, const float* x0, const float* t, float* xt, float* target, int B, int per_sample, hipStream_t s, float* zero0 = nullptr,
                float* zero1 = nullptr);  // zero0 / zero1: optional scalars set to 0 by the same launch
int mse_fwd_bwd(const float* v, const float* target, float* loss, float* dv, long n, hipStream_t s, bool zero_first = true);
int sq_norm_accum(const float* g, long n, float* out, hipStream_t s);  // out[0] += sum g^2 (atomic)
int adamw_step(float* p, const float* g, float* m, float* v, long n, const float* gnorm_sq, float clip, float lr, float b1, float b2, float eps, float wd,
               float bc1, float bc2, int* nonfinite, hipStream_t s);
constexpr int ADAM_MAX_RANGES = 12;
struct AdamRanges { long lo[ADAM_MAX_RANGES], n[ADAM_MAX_RANGES]; int first_block[ADAM_MAX_RANGES + 1]; int count; };
struct AdamwHyper { float clip, lr0, eta_min; int t_max; float b1, b2, eps, wd, max_grad_norm; float* ema = nullptr; float ema_decay = 0.f; };  // ema: optional shadow parameters (flat, like p)
// over `count` element ranges [lo, lo + n) of the flat buffers in one launch; leader: the launch of a step that writes state_out / counts a skip
int adamw_step_ranges(float* p, const float* g, float* m, float* v, const long* lo, const long* n, int count, const AdamwHyper& h, const float* gnorm_sq,
                      const int* state_in, int* state_out, int* nonfinite, float* gnorm_out, bool leader, hipStream_t s);
// the same with the step index and the cosine-schedule position on the device (state_in / state_out: 4 ints each, distinct)
int adamw_step_sched(float* p, const float* g, float* m, float* v, long n, const float* gnorm_sq, float clip, float lr0, float eta_min, int t_max, float b1, float b2,
                     float eps, float wd, const int* state_in, int* state_out, float max_grad_norm, int* nonfinite, float* gnorm_out, hipStream_t s,
                     float* ema = nullptr, float ema_decay = 0.f);
int axpby(float* out, const float* a, const float* b, float alpha, float beta, long n, hipStream_t s);  // out = alpha*a + beta*b (a may alias out)
int rk4_combine(float* y, const float* k1, const float* k2, const float* k3, const float* k4, float h, long n, hipStream_t s);

// ---- box calibration for bench.py (v4h_calib.hip) ----
int calib_mfma_loop(const void* rnd_bf16, float* sink, int iters, int blocks, hipStream_t s);  // blocks x 256 lanes, each iteration 16 MFMA 16x16x32 bf16 per wave
int calib_copy(const void* src, void* dst, long bytes, hipStream_t s);

// ---- energy model, resident decoder (v4h_energy_fused.hip; bf16 mode) ----
size_t energy_fused_stream_bytes(int nd);
bool energy_fused_supported(int d, int ff, int H, int L, int nd, int te);
int energy_fused_pack(const void* const* params, char* stream, int nd, int te, int dec0, int dcount, int dec_norm, int head_w, int head_b, int out_w, int out_b,
                      hipStream_t s);
int energy_fused_decoder(const char* stream, const float* x, const float* t, const float* gfp_w, const float* te_w, const float* te_b, const float* wx, const float* bx,
                         const float* pos, const float* cv, const float* head_w, const float* head_b, float* out, int B, int L, int nd, int te, hipStream_t s);

}  // namespace v4h
