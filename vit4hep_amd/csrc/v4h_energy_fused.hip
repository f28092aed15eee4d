// Energy-model decoder + head as ONE launch (bf16 mode): a 256-thread workgroup owns one sample and keeps its whole state - 45 tokens x
// d_model 128: residual stream (f32), operand copies, q/k/v, attention output, feed-forward hidden - in LDS through all decoder layers
// and the head, while the weights stream past it from L2 as pre-packed, pre-swizzled 16 KiB LDS images (direct global -> LDS DMA, 3-deep
// ring, one barrier per image).  The composed path (v4h_energy.hip) needs 32 launches of ~8 us for the same work at the reference's
// sampling batch of 256: contractions of 11 520 rows x K = 128 are nothing but launch + DMA round-trip latency there.
//
//   per decoder layer (nn.TransformerDecoderLayer, post-norm; reference nn/cfm/transformer_cfm.py:55-64):
//     image  0      : parameter image (biases, LayerNorm weights; f32)
//     images 1..6   : self_attn.in_proj   3 row blocks of 128 outputs x 2 K tiles of 64     -> q | k | v   (LDS, bf16)
//                     attention: wave h = head h (4 heads x 32), 45 keys, softmax in registers
//     images 7..8   : self_attn.out_proj  + residual, LayerNorm1, + cross-attention vector, LayerNorm2
//     images 9..16  : linear1 (4 row blocks x 2 K tiles), relu                               -> hidden (LDS, bf16)
//     images 17..24 : linear2 (8 K tiles) + residual, LayerNorm3 (+ the stack's final LayerNorm after the last layer)
//   head: parameter image, 8 images of layer[:, t_dim:], + per-sample time term, SiLU, dot with layers.2.weight
#include <string.h>

#include "../../include/vit4hep_hip.h"
#include "v4h_ops.h"

namespace v4h {
namespace {
constexpr int FD = 128, FFF = 512, FL_MAX = 48, FH = 4;        // d_model, feed-forward, padded tokens, heads (head_dim 32)
constexpr int IMG = 16384, NST = 5;                            // bytes per image, ring depth (4 images = 64 KiB in flight)
constexpr int LDH = FD + 8, LDQ = 3 * FD + 8, LDF = FFF + 8;   // bf16 row strides (elements): +16 B keeps 16 rows on distinct banks
constexpr int IMGS_PER_LAYER = 25, IMGS_HEAD = 9;
// parameter image of a decoder layer (float offsets)
enum { PB_IN = 0, PB_OUT = 384, PB_L1 = 512, PB_L2 = 1024, PB_N1W = 1152, PB_N1B = 1280, PB_N2W = 1408, PB_N2B = 1536, PB_N3W = 1664, PB_N3B = 1792,
       PB_NFW = 1920, PB_NFB = 2048, PB_COUNT = 2176 };
enum { PH_W2 = 0, PH_B2 = 512 };

// ---- pack: f32 parameters -> the image stream (run when the weights change; cf. V4H_FWD_REUSE_OPERANDS) ------------------------
constexpr int MAX_PTRS = 4 * 18 + 8;
struct PackArgs {
  const float* params[MAX_PTRS];  // by value: decoder layers' 18 tensors each (index dec0 + dcount * layer + k, re-based to 0), then dec_norm w/b, head w/b, out w/b
  char* stream;
  int nd, te;
  // parameter indices
  int dec0, dcount, dec_norm, head_w, head_b, out_w, out_b;
};
__global__ __launch_bounds__(256) void energy_pack_kernel(PackArgs a) {
  const int img = blockIdx.x;
  char* dst = a.stream + (size_t)img * IMG;
  const int layer = img / IMGS_PER_LAYER, j = img % IMGS_PER_LAYER;
  const bool head = layer >= a.nd;
  const int jj = head ? img - a.nd * IMGS_PER_LAYER : j;
  auto P = [&](int idx) { return a.params[idx]; };
  if (jj == 0) {  // parameter image
    float* d = reinterpret_cast<float*>(dst);
    for (int i = threadIdx.x; i < IMG / 4; i += 256) {
      float v = 0.f;
      if (head) {
        if (i < 512) v = P(a.out_w)[i];
        else if (i == PH_B2) v = P(a.out_b)[0];
      } else {
        const int b = a.dec0 + a.dcount * layer;  // sa_in_w, sa_in_b, sa_out_w, sa_out_b, ca x4, l1w, l1b, l2w, l2b, n1w, n1b, n2w, n2b, n3w, n3b
        if (i < PB_OUT) v = P(b + 1)[i];
        else if (i < PB_L1) v = P(b + 3)[i - PB_OUT];
        else if (i < PB_L2) v = P(b + 9)[i - PB_L1];
        else if (i < PB_N1W) v = P(b + 11)[i - PB_L2];
        else if (i < PB_N1B) v = P(b + 12)[i - PB_N1W];
        else if (i < PB_N2W) v = P(b + 13)[i - PB_N1B];
        else if (i < PB_N2B) v = P(b + 14)[i - PB_N2W];
        else if (i < PB_N3W) v = P(b + 15)[i - PB_N2B];
        else if (i < PB_N3B) v = P(b + 16)[i - PB_N3W];
        else if (i < PB_NFW) v = P(b + 17)[i - PB_N3B];
        else if (i < PB_NFB) v = P(a.dec_norm)[i - PB_NFW];
        else if (i < PB_COUNT) v = P(a.dec_norm + 1)[i - PB_NFB];
      }
      d[i] = v;
    }
    return;
  }
  // weight image: 128 output rows x 64 K columns, in the layout ImgKContig<bf16, 128, 64>::frag reads (unit u = row * 8 + pos holds source chunk pos ^ (row & 6))
  const float* W;
  int ld, n0, k0;
  if (head) {
    W = P(a.head_w); ld = 3 * (FD / 2); n0 = ((jj - 1) / 2) * 128; k0 = a.te + ((jj - 1) % 2) * 64;  // layer[:, t_dim:] : the embedding columns
  } else {
    const int b = a.dec0 + a.dcount * layer;
    if (jj <= 6) { W = P(b + 0); ld = FD; n0 = ((jj - 1) / 2) * 128; k0 = ((jj - 1) % 2) * 64; }
    else if (jj <= 8) { W = P(b + 2); ld = FD; n0 = 0; k0 = (jj - 7) * 64; }
    else if (jj <= 16) { W = P(b + 8); ld = FD; n0 = ((jj - 9) / 2) * 128; k0 = ((jj - 9) % 2) * 64; }
    else { W = P(b + 10); ld = FFF; n0 = 0; k0 = (jj - 17) * 64; }
  }
  for (int u = threadIdx.x; u < 1024; u += 256) {
    const int row = u >> 3, pos = u & 7, kc = pos ^ (row & 6);
    const float* src = W + (size_t)(n0 + row) * ld + k0 + kc * 8;
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (bf16)src[e];
    *reinterpret_cast<bf16x8*>(dst + u * 16) = v;
  }
}

// ---- the resident decoder ---------------------------------------------------------------------------------------------------------
struct FusedArgs {
  const char* stream;   // packed images
  const float* x;       // (B, L)
  const float* t;       // (B)
  const float* gfp_w; const float* te_w; const float* te_b;  // time_embed: GaussianFourierProjection.W (te/2), Linear (te, te)
  const float* wx; const float* bx; const float* pos;  // x_embed.weight (e,1), bias, pos_embed_x (L, e)
  const float* cv;      // (nd, B, d) cross-attention vectors
  const float* head_w; const float* head_b;  // layer.weight (ff, te + d): its first te columns give the per-sample time term of the head
  float* out;           // (B, L)
  int B, L, nd, te;
};

// workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt, i.e. the prefetched weight images
V4H_DEV void wg_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

V4H_DEV Frag<bf16> wfrag(const char* img, int n0, int kk, int lane) {  // ImgKContig<bf16, 128, 64>::frag
  const int row = n0 + (lane & 15), kc = kk / 8 + (lane >> 4);
  Frag<bf16> f;
  f.v = *reinterpret_cast<const bf16x8*>(img + (row * 8 + (kc ^ (row & 6))) * 16);
  return f;
}

__global__ __launch_bounds__(256, 1) void energy_decoder_kernel(const FusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // The f32 residual stream lives in REGISTERS, in the accumulator layout of the contractions (lane (c, g) of res[x][y]: token 16 x + c,
  // features 32 wave + 16 y + 4 g + 0..3): the residual add is a register add, LayerNorm exchanges per-row partial sums through `stat`.
  // That keeps 24 KB of LDS for two more weight images in flight.
  bf16* hT = reinterpret_cast<bf16*>(smem);                                        // [48][LDH] operand copy of h; aliased by the attention output
  bf16* big = hT + FL_MAX * LDH;                                                   // q|k|v [48][LDQ], later the feed-forward hidden [48][LDF]
  char* ring = reinterpret_cast<char*>(big + FL_MAX * LDF);                        // NST images
  float* pbuf = reinterpret_cast<float*>(ring + NST * IMG);                        // parameter image of the current layer
  float* vbuf = pbuf + PB_COUNT;                                                   // cv (nd x 128) | hv (512)
  float* red = vbuf + 4 * FD + FFF;                                                // [4][48] head partial sums
  float* stat = red + 4 * FL_MAX;                                                  // [2][4 waves][48 rows][sum, sum of squares]

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int b = blockIdx.x, L = a.L;
  const int nimg = a.nd * IMGS_PER_LAYER + IMGS_HEAD;

  // ---- per-sample vectors, time embedding and the target embedding (transformer_cfm.py:39-42,84-90,153-165), before the DMA pipeline starts
  for (int i = tid; i < a.nd * FD; i += 256) vbuf[i] = a.cv[((size_t)(i / FD) * a.B + b) * FD + (i % FD)];
  for (int i = tid; i < FL_MAX * LDF / 2; i += 256) reinterpret_cast<uint32_t*>(big)[i] = 0u;  // rows 48..63 of the q|k|v view are read as (zero-weighted) keys
  float* gfp = red;             // [te] Fourier features, then [te] time embedding (red is free until the head's reduction)
  float* tembs = red + 64;
  if (tid < a.te) {             // same f32 operation order as the reference: ((t * W) * 2) * pi
    const int half = a.te / 2;
    float pr = a.t[b] * a.gfp_w[tid < half ? tid : tid - half];
    pr = pr * 2.0f;
    pr = pr * 3.14159265358979323846f;
    gfp[tid] = tid < half ? sinf(pr) : cosf(pr);
  }
  __syncthreads();
  {  // temb = W1 gfp + b1: 4 lanes per output
    const int j = tid >> 2, q = tid & 3;
    float sacc = 0.f;
    if (j < a.te)
      for (int k = q; k < a.te; k += 4) sacc += a.te_w[j * a.te + k] * gfp[k];
    sacc += __shfl_xor(sacc, 1, 64);
    sacc += __shfl_xor(sacc, 2, 64);
    if (j < a.te && q == 0) tembs[j] = sacc + a.te_b[j];
  }
  __syncthreads();
  for (int n = tid; n < FFF; n += 256) {  // head, time part: hv[n] = layer.weight[n, :te] . temb + layer.bias[n]
    const float* wr = a.head_w + (size_t)n * (a.te + FD);
    float sacc = a.head_b[n];
    for (int k = 0; k < a.te; k += 4) {
      const f32x4 wv = load4(wr + k);
      sacc += wv[0] * tembs[k] + wv[1] * tembs[k + 1] + wv[2] * tembs[k + 2] + wv[3] * tembs[k + 3];
    }
    vbuf[4 * FD + n] = sacc;
  }
  f32x4 res[3][2];
  auto put_hT = [&]() {  // operand copy of the residual stream for the next contraction
#pragma unroll
    for (int x = 0; x < 3; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y) {
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (bf16)res[x][y][r];
        *reinterpret_cast<bf16x4*>(hT + (16 * x + c) * LDH + 32 * wave + 16 * y + 4 * g) = o;
      }
  };
#pragma unroll
  for (int x = 0; x < 3; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = 16 * x + c, j = 32 * wave + 16 * y + 4 * g + r;
        float v = 0.f;
        if (n < L) v = j < a.te ? tembs[j] : a.x[(size_t)b * L + n] * a.wx[j - a.te] + a.bx[j - a.te] + a.pos[(size_t)n * (FD - a.te) + j - a.te];
        res[x][y][r] = v;
      }
  put_hT();
  __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0): the counted waits below see only the image DMAs
  __syncthreads();

  // ---- weight stream: each wave moves 4 KiB of every image
  int issued = 0;
  auto issue = [&]() {
    if (issued < nimg) {
      const char* src = a.stream + (size_t)issued * IMG + (wave * 4) * 1024 + lane * 16;
      char* dst = ring + (issued % NST) * IMG + (wave * 4) * 1024;
#pragma unroll
      for (int q = 0; q < 4; ++q) dma16(src + q * 1024, dst + q * 1024);
    }
    ++issued;  // counted even past the end so that the wait arithmetic stays uniform
  };
  for (int q = 0; q < NST - 1; ++q) issue();
  int cur = 0;  // next image to consume
  // image `cur` has landed everywhere and the slot of image cur - 1 is free; then keep the ring full
  auto acquire = [&]() -> const char* {
    const int inflight_after = (issued < nimg ? issued : nimg) - 1 - cur;  // younger images allowed to stay in flight
    wait_vmcnt(4 * (inflight_after > 0 ? inflight_after : 0));
    wg_barrier();
    issue();
    return ring + (cur++ % NST) * IMG;
  };

  f32x4 acc[3][2];
  auto zero_acc = [&]() {
#pragma unroll
    for (int x = 0; x < 3; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y) acc[x][y] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  // acc[x][y] += A[16x .. ][k0 + 0..63] . W_image[32 wave + 16 y ..][0..63]^T     (A: bf16 rows of stride lda in LDS)
  auto gemm_image = [&](const bf16* A, int lda, int k0, const char* img) {
#pragma unroll
    for (int kk = 0; kk < 64; kk += 32) {
      Frag<bf16> af[3], wf[2];
#pragma unroll
      for (int x = 0; x < 3; ++x) af[x] = frag_kcontig(A, lda, 16 * x, k0 + kk, lane);
#pragma unroll
      for (int y = 0; y < 2; ++y) wf[y] = wfrag(img, 32 * wave + 16 * y, kk, lane);
#pragma unroll
      for (int x = 0; x < 3; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = mma(wf[y], af[x], acc[x][y]);
    }
  };
  // lane (c, g) of acc[x][y] holds token 16 x + c, features n0 + 16 y + 4 g + 0..3 with n0 = 32 wave
  auto for_acc = [&](auto&& f) {
#pragma unroll
    for (int x = 0; x < 3; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y) f(16 * x + c, 32 * wave + 16 * y + 4 * g, acc[x][y]);
  };
  // Up to two chained LayerNorms of the register-resident rows (the second optionally after adding a per-sample vector).  Per stage: every
  // lane sums its 8 features per row, two shuffles finish the wave's 32 features, the four waves exchange (sum, sum of squares) through
  // `stat` (double-buffered by stage parity: one barrier per stage).
  int ln_parity = 0;
  auto layer_norms = [&](const float* g1, const float* b1, const float* add, const float* g2, const float* b2) {
    for (int st = 0; st < (g2 ? 2 : 1); ++st) {
      const float* gg = st ? g2 : g1;
      const float* bb = st ? b2 : b1;
      float* sb = stat + ln_parity * (4 * FL_MAX * 2);
      ln_parity ^= 1;
#pragma unroll
      for (int x = 0; x < 3; ++x) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (st && add) res[x][y][r] += add[32 * wave + 16 * y + 4 * g + r];
            s1 += res[x][y][r];
            s2 += res[x][y][r] * res[x][y][r];
          }
        s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
        if (g == 0) { sb[(wave * FL_MAX + 16 * x + c) * 2] = s1; sb[(wave * FL_MAX + 16 * x + c) * 2 + 1] = s2; }
      }
      wg_barrier();
#pragma unroll
      for (int x = 0; x < 3; ++x) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < 4; ++w2) { s1 += sb[(w2 * FL_MAX + 16 * x + c) * 2]; s2 += sb[(w2 * FL_MAX + 16 * x + c) * 2 + 1]; }
        const float mu = s1 * (1.0f / FD);
        const float rs = 1.0f / sqrtf(fmaxf(s2 * (1.0f / FD) - mu * mu, 0.0f) + 1e-5f);
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int n = 32 * wave + 16 * y + 4 * g + r;
            res[x][y][r] = (res[x][y][r] - mu) * rs * gg[n] + bb[n];
          }
      }
    }
    put_hT();
  };

  for (int layer = 0; layer < a.nd; ++layer) {
    {  // parameter image -> pbuf
      const float* pimg = reinterpret_cast<const float*>(acquire());
      for (int i = tid; i < PB_COUNT; i += 256) pbuf[i] = pimg[i];
    }
    // ---- self-attention in_proj: q | k | v
    for (int nb = 0; nb < 3; ++nb) {
      zero_acc();
      for (int kt = 0; kt < 2; ++kt) gemm_image(hT, LDH, 64 * kt, acquire());
      for_acc([&](int m, int n, f32x4 v) {
        const float* bias = pbuf + PB_IN + nb * FD + n;
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (bf16)(v[r] + bias[r]);
        *reinterpret_cast<bf16x4*>(big + m * LDQ + nb * FD + n) = o;
      });
    }
    wg_barrier();  // q, k, v complete (and pbuf visible)
    // ---- attention: wave = head.  S = q k^T / sqrt(32) on accumulator rows (lane (c, g): query 16 qt + c, keys 16 kt + 4 g + r)
    {
      const int h = wave;
      const bf16* q = big + h * 32;
      const bf16* k = big + FD + h * 32;
      const bf16* vv = big + 2 * FD + h * 32;
      const float scale = 0.17677669529663687f;  // 32^-0.5
#pragma unroll
      for (int qt = 0; qt < 3; ++qt) {
        const Frag<bf16> qf = frag_kcontig(q, LDQ, 16 * qt, 0, lane);
        f32x4 s[4];
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) {
          s[kt] = mma(frag_kcontig(k, LDQ, 16 * kt, 0, lane), qf, f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            s[kt][r] = (16 * kt + 4 * g + r) < L ? s[kt][r] * scale : -INFINITY;
            mx = fmaxf(mx, s[kt][r]);
          }
        }
        s[3] = f32x4{0.f, 0.f, 0.f, 0.f};  // keys 48..63: padding of the second 32-key slab
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float rs = 0.f;
#pragma unroll
        for (int kt = 0; kt < 3; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            s[kt][r] = __expf(s[kt][r] - mx);
            rs += s[kt][r];
          }
        rs += __shfl_xor(rs, 16, 64);
        rs += __shfl_xor(rs, 32, 64);
        const float inv = 1.0f / rs;
        // O = P V: P re-used from the accumulators as the lane-side operand (keys = contraction), V read transposed
        f32x4 o[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const Frag<bf16> pf = frag_from_acc(s[2 * ks], s[2 * ks + 1], (bf16)0.f);
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) o[dt] = mma(frag_kstrided2(vv, LDQ, 32 * ks, 32 * ks + 16, 16 * dt, lane), pf, o[dt]);
        }
        // attention output (aliases hT: the layer's operand copy of h is dead once q, k, v exist)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          bf16x4 ov;
#pragma unroll
          for (int r = 0; r < 4; ++r) ov[r] = (bf16)(o[dt][r] * inv);
          *reinterpret_cast<bf16x4*>(hT + (16 * qt + c) * LDH + h * 32 + 16 * dt + 4 * g) = ov;
        }
      }
    }
    // ---- out_proj + residual, LayerNorm1, + cross-attention vector, LayerNorm2
    zero_acc();
    for (int kt = 0; kt < 2; ++kt) gemm_image(hT, LDH, 64 * kt, acquire());  // the barrier inside acquire() publishes the attention output
#pragma unroll
    for (int x = 0; x < 3; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int r = 0; r < 4; ++r) res[x][y][r] += acc[x][y][r] + pbuf[PB_OUT + 32 * wave + 16 * y + 4 * g + r];
    wg_barrier();  // every wave is done reading the attention output (aliased by hT) before the LayerNorms rewrite hT
    layer_norms(pbuf + PB_N1W, pbuf + PB_N1B, vbuf + layer * FD, pbuf + PB_N2W, pbuf + PB_N2B);
    // ---- feed-forward: relu(linear1) -> hidden (aliases q|k|v), linear2 + residual, LayerNorm3
    for (int nb = 0; nb < 4; ++nb) {
      zero_acc();
      for (int kt = 0; kt < 2; ++kt) gemm_image(hT, LDH, 64 * kt, acquire());  // first acquire(): barrier after the LayerNorms
      for_acc([&](int m, int n, f32x4 v) {
        const float* bias = pbuf + PB_L1 + nb * FD + n;
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (bf16)fmaxf(v[r] + bias[r], 0.0f);
        *reinterpret_cast<bf16x4*>(big + m * LDF + nb * FD + n) = o;
      });
    }
    zero_acc();
    for (int kt = 0; kt < 8; ++kt) gemm_image(big, LDF, 64 * kt, acquire());  // first acquire(): barrier after the hidden activations
#pragma unroll
    for (int x = 0; x < 3; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int r = 0; r < 4; ++r) res[x][y][r] += acc[x][y][r] + pbuf[PB_L2 + 32 * wave + 16 * y + 4 * g + r];
    const bool last = layer == a.nd - 1;
    layer_norms(pbuf + PB_N3W, pbuf + PB_N3B, nullptr, last ? pbuf + PB_NFW : nullptr, pbuf + PB_NFB);
  }

  // ---- head: silu(W_head[:, t_dim:] h + hv) . w_out + b_out      (transformer_cfm.py:66-70,114-119)
  {
    const float* pimg = reinterpret_cast<const float*>(acquire());
    for (int i = tid; i < 513; i += 256) pbuf[i] = pimg[i];
  }
  float part[3] = {0.f, 0.f, 0.f};  // lane (c, g): partial dot product of token 16 x + c over this lane's features
  for (int nb = 0; nb < 4; ++nb) {
    zero_acc();
    for (int kt = 0; kt < 2; ++kt) gemm_image(hT, LDH, 64 * kt, acquire());
#pragma unroll
    for (int x = 0; x < 3; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y) {
        const int n = nb * FD + 32 * wave + 16 * y + 4 * g;
#pragma unroll
        for (int r = 0; r < 4; ++r) part[x] += silu_f(acc[x][y][r] + vbuf[4 * FD + n + r]) * pbuf[PH_W2 + n + r];
      }
  }
#pragma unroll
  for (int x = 0; x < 3; ++x) {
    float s = part[x];
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (g == 0) red[wave * FL_MAX + 16 * x + c] = s;
  }
  wg_barrier();
  if (tid < L) a.out[(size_t)b * L + tid] = red[tid] + red[FL_MAX + tid] + red[2 * FL_MAX + tid] + red[3 * FL_MAX + tid] + pbuf[PH_B2];
}

constexpr size_t FUSED_LDS = (size_t)FL_MAX * LDH * 2 + (size_t)FL_MAX * LDF * 2 + (size_t)NST * IMG + (size_t)PB_COUNT * 4 + (size_t)(4 * FD + FFF) * 4 +
                             (size_t)4 * FL_MAX * 4 + (size_t)2 * 4 * FL_MAX * 2 * 4;
static_assert(FUSED_LDS <= 160 * 1024, "resident decoder must fit the 160 KiB of LDS");
}  // namespace

size_t energy_fused_stream_bytes(int nd) { return (size_t)(nd * IMGS_PER_LAYER + IMGS_HEAD) * IMG; }
bool energy_fused_supported(int d, int ff, int H, int L, int nd, int te) { return d == FD && ff == FFF && H == FH && L <= FL_MAX && nd >= 1 && nd <= 4 && te == FD / 2; }

// params: host table of the plan's parameter pointers (device memory); the tensors the decoder and head need travel by value
int energy_fused_pack(const void* const* params, char* stream, int nd, int te, int dec0, int dcount, int dec_norm, int head_w, int head_b, int out_w, int out_b,
                      hipStream_t s) {
  PackArgs a;
  memset(&a, 0, sizeof(a));
  for (int l = 0; l < nd; ++l)
    for (int k = 0; k < 18; ++k) a.params[18 * l + k] = (const float*)params[dec0 + dcount * l + k];
  const int base = 18 * nd;
  a.params[base + 0] = (const float*)params[dec_norm]; a.params[base + 1] = (const float*)params[dec_norm + 1];
  a.params[base + 2] = (const float*)params[head_w];   a.params[base + 3] = (const float*)params[head_b];
  a.params[base + 4] = (const float*)params[out_w];    a.params[base + 5] = (const float*)params[out_b];
  a.stream = stream; a.nd = nd; a.te = te;
  a.dec0 = 0; a.dcount = 18; a.dec_norm = base; a.head_w = base + 2; a.head_b = base + 3; a.out_w = base + 4; a.out_b = base + 5;
  hipLaunchKernelGGL(energy_pack_kernel, dim3(nd * IMGS_PER_LAYER + IMGS_HEAD), dim3(256), 0, s, a);
  V4H_CHECK_LAUNCH("energy_pack");
  return V4H_OK;
}
int energy_fused_decoder(const char* stream, const float* x, const float* t, const float* gfp_w, const float* te_w, const float* te_b, const float* wx, const float* bx,
                         const float* pos, const float* cv, const float* head_w, const float* head_b, float* out, int B, int L, int nd, int te, hipStream_t s) {
  static DeviceOnce lds_attr;
  if (int rc = lds_attr.ensure([&]() -> hipError_t {
        return hipFuncSetAttribute((const void*)energy_decoder_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FUSED_LDS);
      }, "energy_decoder", "reserve the decoder's LDS")) return rc;
  FusedArgs a{stream, x, t, gfp_w, te_w, te_b, wx, bx, pos, cv, head_w, head_b, out, B, L, nd, te};
  hipLaunchKernelGGL(energy_decoder_kernel, dim3(B), dim3(256), FUSED_LDS, s, a);
  V4H_CHECK_LAUNCH("energy_decoder");
  return V4H_OK;
}
}  // namespace v4h
