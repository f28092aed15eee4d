// Single-chunk attention (sequences of 81 .. 160 tokens = 6 .. 10 key tiles: ds2 / LEMURS T = 135, ds1 T = 88 / 125, CaloGAN T = 84), bf16, head_dim 80 - the
// instruction-lean forms of round 3.
// Reference: nn/vit.py:425-451 (softmax(q k^T / sqrt(dh)) v per (batch, head), token-major qkv as the qkv Linear writes it).
//
// What round 2's counters said about the persistent kernels of v4h_attention.hip (profiles/r03_attn_counters.md): a wave executes ~710 vector + ~430 scalar
// instructions per (batch, head) item around 55 MFMAs, the vector-issue port of the busiest SIMD (3 of the 9 waves) is ~70 % busy and every wave is parked
// half of its life - the kernel is bound by instruction issue at 2.25 waves per SIMD, not by the matrix pipe (20 %), LDS (10 %) or memory.  So this form
//   * spends ~190 vector instructions per item: scores stay raw (the softmax scale is folded into the exponent: p = exp2(s c - m c)), only the one key
//     tile that can hold rows >= T is masked, the row maximum is 3-input maxima + two lane swaps, the row SUM comes out of the P V product itself (a sixth
//     output tile against an all-ones operand: every lane then holds its row's sum, no lane exchange), the head_dim tail (80 = 2.5 x 32) is a K = 32 MFMA
//     whose query-side operand is zero beyond head_dim (those lanes' loads lie outside the buffer descriptor), so the key side needs no mask - it reads on
//     into the next image row, finite data times zero; the key tail of an odd tile count likewise has a zero half.  (A K = 16 MFMA for the two tails,
//     v_mfma_f32_16x16x16_bf16 accumulating straight onto a 16x16x32 result, came out wrong in two of four accumulator registers whenever the compiler
//     scheduled the pair back to back - found with tools/experiments/attn_debug2.py - so no K = 16 instruction is used.)
//   * addresses everything per item through ONE buffer descriptor built from scalars (item base in SGPRs, per-lane offsets fixed for the whole kernel): no
//     per-item 64-bit vector address arithmetic, rows >= T are dropped / zero-filled by the descriptor's bounds check instead of by predicates;
//   * comes in two forms of the same body: NBUF = 2 (the product form: one workgroup per CU, the next item's images and query rows are requested before the
//     current item is computed) and NBUF = 1 (one pair of images per workgroup, 2 x 23 KB for 9 tiles, so that two or three workgroups share a CU and one's
//     DMA wait is another's compute time: measured equal or slower, not instantiated).
#pragma once

namespace v4h_dense {  // (a named namespace: the host stubs of two of these kernel instantiations were not emitted from an anonymous one)

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_;

// exchange between the 16-lane rows (0<->1, 2<->3) / the 32-lane halves, then reduce: both copies hold the other side's value afterwards
V4H_DEV float max_xor16(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return fmaxf(a, b);
}
V4H_DEV float max_xor32(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return fmaxf(a, b);
}
V4H_DEV unsigned pack_bf16(float a, float b) {
  typedef __bf16 b2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, b2{(bf16)a, (bf16)b});
}
// accumulator tiles (lane (c, g): row c, columns 4 g .. 4 g + 3 of a 16-column tile) -> lane-side operand of the next product
V4H_DEV Frag<bf16> acc_pair_frag(f32x4 a0, f32x4 a1) {  // K = 32 step: k = 4 g + j (tile 0), 16 + 4 g + j (tile 1)   (= frag_from_acc)
  const u32x4_ w{pack_bf16(a0[0], a0[1]), pack_bf16(a0[2], a0[3]), pack_bf16(a1[0], a1[1]), pack_bf16(a1[2], a1[3])};
  Frag<bf16> f;
  f.v = __builtin_bit_cast(bf16x8, w);
  return f;
}
V4H_DEV Frag<bf16> acc_half_frag(f32x4 a0) {  // the same with a zero second tile (odd tile count: 16 keys in the last K = 32 step)
  const u32x4_ w{pack_bf16(a0[0], a0[1]), pack_bf16(a0[2], a0[3]), 0u, 0u};
  Frag<bf16> f;
  f.v = __builtin_bit_cast(bf16x8, w);
  return f;
}

constexpr int AD_DH = 80, AD_ROWB = 160;  // head_dim, bytes per image row (dense)

// Fragment reads from a dense image through per-lane base pointers fixed for the whole kernel: the row / column of a tile is then a constant byte offset
// of the DS instruction (the generic helpers rebuild base + row + lane terms per call, which the compiler hoists out of the unit loop as one address
// register per tile and then spills).  kc = image + c * 80 + 8 g (k-contiguous reads), ks = image + (4 g + q) * 80 + 4 p (transposed reads).
V4H_DEV Frag<bf16> lane_kcontig(const bf16* kc, int row0, int k0) {
  Frag<bf16> f;
  f.v = *reinterpret_cast<const bf16x8*>(kc + row0 * AD_DH + k0);
  return f;
}
V4H_DEV Frag<bf16> lane_kstrided(const bf16* ks, int row0, int col0) {  // rows row0 + 4 g + 0..3 and row0 + 16 + 4 g + 0..3, columns col0 + c
  const bf16x4 lo = lds_tr_read(ks + row0 * AD_DH + col0), hi = lds_tr_read(ks + (row0 + 16) * AD_DH + col0);
  Frag<bf16> f;
  f.v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return f;
}


// Image of NT * 16 rows x 80 bf16 (dense 160-byte rows) filled by buffer_load ... lds: instruction `inst` writes bytes [inst * 1024, + 1024), lane `l` its 16-byte
// unit u = inst * 64 + l = (row u / 10, chunk u % 10).  The source offset of a unit is fixed for the whole kernel (row * ld_bytes + chunk * 16, relative to the
// item's q/k/v base in the descriptor); units of rows >= T lie beyond the descriptor's range and read as zero.
template <int NT, int NWV = NT> struct DenseImage {  // NWV: waves that share the DMA instructions
  static constexpr int ROWS = NT * 16, UNITS = ROWS * 10, NI = (UNITS + 63) / 64, BYTES = NI * 1024, NPW = (NI + NWV - 1) / NWV;
  unsigned voff[NPW];  // per DMA instruction of this wave (inst = wave + k * NWV): the lane's source byte offset
  V4H_DEV void init(int wave, int lane, int ld_bytes) {
#pragma unroll
    for (int k = 0; k < NPW; ++k) {
      const int u = (wave + k * NWV) * 64 + lane;
      voff[k] = u < UNITS ? (unsigned)((u / 10) * ld_bytes + (u % 10) * 16) : 0x7FFFFF00u;
    }
  }
  // (`add` = byte offset of the tensor's slice inside the descriptor, added to the vector offset: the scalar offset of a buffer instruction is not
  //  part of the range check on every generation, and the range check is what zero-fills the rows >= T)
  V4H_DEV void stage(__amdgpu_buffer_rsrc_t rsrc, char* img, unsigned add, int wave) const {
#pragma unroll
    for (int k = 0; k < NPW; ++k) {
      const int inst = wave + k * NWV;
      if (inst < NI) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (V4H_LDS void*)(img + inst * 1024), 16, voff[k] + add, 0, 0, 0);
    }
  }
};

// Persistent (batch, head) walk, XCD-aware (attn_item, v4h_attention.hip).  NT = query tiles = key tiles = waves (9: T <= 144, 10: T <= 160).
template <int NT, int WPE, int NBUF> __global__ __launch_bounds__(64 * NT, WPE) void attn_fwd_dense_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ o,
                                                                                              float* __restrict__ lse, int Tn, int H, int nitems, float scale) {
  using IMG = DenseImage<NT>;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // NBUF x [K image | V image]
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int D = H * AD_DH, ldb = 3 * D * 2;  // bytes between consecutive tokens of qkv
  const int Bn = nitems / H;
  // softmax in base 2 with the scale folded into the exponent
  const float c2 = scale * 1.4426950408889634f;
  constexpr bool ODD = (NT & 1) != 0;  // odd tile count: the last 16 keys are a K = 16 step of the P V product
  constexpr int NKS = NT / 2;
  // rows >= Tn of the last key tile: masked before the maximum (their scores are 0, from the zero rows of the image, not -inf)
  bool dead[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) dead[r] = (NT - 1) * 16 + 4 * g + r >= Tn;

  IMG img;
  img.init(wave, lane, ldb);
  // (head_dim tail, d = 64 .. 95: lanes g >= 2 would hold d >= 80 - their offset lies outside every descriptor, so they load zeros)
  const unsigned q_off = (unsigned)((wave * 16 + c) * ldb + 16 * g), q_tail = g < 2 ? (unsigned)((wave * 16 + c) * ldb + 128 + 16 * g) : 0x7FFFFF00u;
  const unsigned o_off = (unsigned)((wave * 16 + c) * D * 2);
  // images start from zeros: a unit beyond the sequence is never written by the DMA if the hardware drops (rather than zero-fills) out-of-range LDS loads
  for (int i = tid * 16; i < NBUF * 2 * IMG::BYTES; i += 64 * NT * 16) *reinterpret_cast<u32x4_*>(smem + i) = u32x4_{0u, 0u, 0u, 0u};
  __syncthreads();
  // one descriptor per item for its q, k, v: rows [0, T) of sample b, starting at head h's q slice; k and v are D and 2 D elements further
  auto item_rsrc = [&](int it) {
    const int b = it / H, h = it - b * H;
    const bf16* ibase = qkv + ((size_t)b * Tn * 3 * D + h * AD_DH);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(ibase), 0, (Tn - 1) * ldb + 4 * D + AD_ROWB, 0x00020000);
  };
  auto stage_item = [&](__amdgpu_buffer_rsrc_t r, int buf) {
    img.stage(r, smem + buf * 2 * IMG::BYTES, 2u * D, wave);
    img.stage(r, smem + buf * 2 * IMG::BYTES + IMG::BYTES, 4u * D, wave);
  };

  Frag<bf16> ones;
#pragma unroll
  for (int r = 0; r < 8; ++r) ones.v[r] = (bf16)1.0f;

  // NBUF == 1: two workgroups per CU, each waits for its own images (the other one computes meanwhile).  NBUF == 2: one workgroup per CU, the next
  // item's images and query rows are requested before the current item is computed (the memory system sees a continuous stream).
  int it = __builtin_amdgcn_readfirstlane(attn_item(blockIdx.x, gridDim.x, 0, Bn, H));
  u32x4_ q0n = {0u, 0u, 0u, 0u}, q1n = q0n, q2n = q0n;
  if (NBUF == 2 && it >= 0) {
    const __amdgpu_buffer_rsrc_t r0 = item_rsrc(it);
    stage_item(r0, 0);
    q0n = __builtin_amdgcn_raw_buffer_load_b128(r0, q_off, 0, 0);
    q1n = __builtin_amdgcn_raw_buffer_load_b128(r0, q_off + 64, 0, 0);
    q2n = __builtin_amdgcn_raw_buffer_load_b128(r0, q_tail, 0, 0);
  }
  for (int n = 0; it >= 0; ++n) {
    const int b = it / H, h = it - b * H;
    const int it_next = __builtin_amdgcn_readfirstlane(attn_item(blockIdx.x, gridDim.x, n + 1, Bn, H));
    const int buf = NBUF == 2 ? (n & 1) : 0;
    const bf16* sK = reinterpret_cast<const bf16*>(smem + buf * 2 * IMG::BYTES);
    const bf16* sV = reinterpret_cast<const bf16*>(smem + buf * 2 * IMG::BYTES + IMG::BYTES);
    Frag<bf16> xq0, xq1, xq2;
    if constexpr (NBUF == 1) {
      const __amdgpu_buffer_rsrc_t rq = item_rsrc(it);
      if (n > 0) __syncthreads();  // every wave is done with the previous item's images
      stage_item(rq, 0);
      // this wave's 16 query rows (lane side): two K = 32 slabs and the tail of head_dim
      const u32x4_ q0 = __builtin_amdgcn_raw_buffer_load_b128(rq, q_off, 0, 0), q1 = __builtin_amdgcn_raw_buffer_load_b128(rq, q_off + 64, 0, 0);
      const u32x4_ q2 = __builtin_amdgcn_raw_buffer_load_b128(rq, q_tail, 0, 0);
      xq0.v = __builtin_bit_cast(bf16x8, q0);
      xq1.v = __builtin_bit_cast(bf16x8, q1);
      xq2.v = __builtin_bit_cast(bf16x8, q2);
      __syncthreads();  // (vmcnt(0) + barrier) the images have landed for every wave
    } else {
      xq0.v = __builtin_bit_cast(bf16x8, q0n);
      xq1.v = __builtin_bit_cast(bf16x8, q1n);
      xq2.v = __builtin_bit_cast(bf16x8, q2n);
      __syncthreads();  // this item's images have landed (vmcnt(0)); everyone is done with the buffer the next request overwrites
      if (it_next >= 0) {
        const __amdgpu_buffer_rsrc_t rn = item_rsrc(it_next);
        stage_item(rn, buf ^ 1);
        q0n = __builtin_amdgcn_raw_buffer_load_b128(rn, q_off, 0, 0);
        q1n = __builtin_amdgcn_raw_buffer_load_b128(rn, q_off + 64, 0, 0);
        q2n = __builtin_amdgcn_raw_buffer_load_b128(rn, q_tail, 0, 0);
      }
    }

    // ---- scores: p[jt][r] = q_c . k_(16 jt + 4 g + r), raw
    f32x4 p[NT];
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
      a = mma(frag_kcontig(sK, AD_DH, jt * 16, 0, lane), xq0, a);
      a = mma(frag_kcontig(sK, AD_DH, jt * 16, 32, lane), xq1, a);
      a = mma(frag_kcontig(sK, AD_DH, jt * 16, 64, lane), xq2, a);  // (lanes g >= 2 read the next row's first columns: multiplied by the zeros of xq2)
      p[jt] = a;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) p[NT - 1][r] = dead[r] ? -INFINITY : p[NT - 1][r];
    float mx = p[0][0];
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
      mx = fmaxf(fmaxf(mx, p[jt][0]), p[jt][1]);  // (v_max3_f32)
      mx = fmaxf(fmaxf(mx, p[jt][2]), p[jt][3]);
    }
    mx = max_xor32(max_xor16(mx));
    const float m2 = mx * c2;
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) p[jt][r] = __builtin_amdgcn_exp2f(fmaf(p[jt][r], c2, -m2));
    // ---- o = P V, and the row sum of the bf16-rounded P as a sixth output tile
    f32x4 oacc[6];
#pragma unroll
    for (int dt = 0; dt < 6; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const Frag<bf16> wf = acc_pair_frag(p[2 * ks], p[2 * ks + 1]);
#pragma unroll
      for (int dt = 0; dt < 5; ++dt) oacc[dt] = mma(frag_kstrided2(sV, AD_DH, 32 * ks, 32 * ks + 16, dt * 16, lane), wf, oacc[dt]);
      oacc[5] = mma(ones, wf, oacc[5]);
    }
    if constexpr (ODD) {
      const Frag<bf16> wf = acc_half_frag(p[NT - 1]);
      const int q4 = (lane >> 2) & 3, p4 = lane & 3;
#pragma unroll
      for (int dt = 0; dt < 5; ++dt) {
        const bf16x4 lo = lds_tr_read(sV + ((NT - 1) * 16 + 4 * g + q4) * AD_DH + dt * 16 + 4 * p4);
        Frag<bf16> zf;
        zf.v = __builtin_shufflevector(lo, bf16x4{(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f}, 0, 1, 2, 3, 4, 5, 6, 7);
        oacc[dt] = mma(zf, wf, oacc[dt]);
      }
      oacc[5] = mma(ones, wf, oacc[5]);
    }
    const float l = oacc[5][0];
    const float inv = __builtin_amdgcn_rcpf(l);
    // ---- store: 8 consecutive columns per lane (tile pairs exchanged between the 16-lane rows), rows >= T dropped by the descriptor
    bf16* obase = o + ((size_t)b * Tn * D + h * AD_DH);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(obase, 0, (Tn - 1) * D * 2 + AD_ROWB, 0x00020000);
    const int ge = g & 1, gh = g >> 1;
    // (normalise BEFORE the lane exchange: the exchange is inline assembly, and the wait states an MFMA result needs before a vector instruction
    //  reads it are only inserted for instructions the compiler can see - read straight after the last MFMA, element 0 of a tile came out wrong)
#pragma unroll
    for (int dt = 0; dt < 5; ++dt) oacc[dt] *= inv;
#pragma unroll
    for (int d = 0; d < 4; d += 2)
      __builtin_amdgcn_raw_buffer_store_b128(pack_bf16x8(swap_pair(oacc[d], oacc[d + 1])), ro, o_off + (unsigned)(((d + ge) * 16 + 8 * gh) * 2), 0, 0);
    {
      const u32x2_ w{pack_bf16(oacc[4][0], oacc[4][1]), pack_bf16(oacc[4][2], oacc[4][3])};
      __builtin_amdgcn_raw_buffer_store_b64(w, ro, o_off + (unsigned)((64 + 4 * g) * 2), 0, 0);
    }
    const int q = wave * 16 + c;
    if (lse != nullptr && g == 0 && q < Tn) lse[((size_t)b * H + h) * Tn + q] = mx * scale + __logf(l);
    it = it_next;
  }
}

// (explicit instantiations: used only through a launcher template, the host stubs of the NBUF = 1 forms were not emitted by this hipcc)
#define V4H_DENSE_FWD(NT, WPE, NBUF) \
  template __global__ void attn_fwd_dense_kernel<NT, WPE, NBUF>(const bf16* __restrict__, bf16* __restrict__, float* __restrict__, int, int, int, float);
V4H_DENSE_FWD(6, 3, 2) V4H_DENSE_FWD(7, 3, 2) V4H_DENSE_FWD(8, 3, 2) V4H_DENSE_FWD(9, 3, 2) V4H_DENSE_FWD(10, 3, 2)
#undef V4H_DENSE_FWD

// ---------------------------------------------------------------------------------------------------------------------------------------------------------
// Sequences of 369 .. 480 tokens (CaloChallenge ds3: T = 450), bf16, head_dim 80.  The K and V images of a WHOLE (batch, head) item fit the CU's LDS
// (2 x 30 x 16 rows x 160 bytes = 150 KB), so the single-chunk form above carries over: one 8-wave workgroup per CU, images filled by
// buffer_load ... lds through one descriptor per item, a wave owns 16 query rows at a time and keeps the whole score row (30 tiles) in registers - no
// online-softmax rescaling, no chunk loop, no second pass over K.  A work unit is HALF of an item's query tiles (QSPLIT = 2): B * H = 384 items on 256
// CUs are 1.5 rounds, 768 halves are exactly 3, and the two halves of an item sit next to each other on one XCD, so the second fill comes from L2.
// Replaces attn_fwd_kernel's key chunks of 160 rows staged through registers (99.5 us per call at ds3 B = 64).
constexpr int AL_NT = 30, AL_NW = 8;
// Per unit: fill both images (150 KB), wait, two rounds of query tiles per wave.  Measured at ds3 B = 64: 63.8 us per call (390 TFLOP/s).  A pipelined form
// (V fill under the first score phase, the next unit's K fill under the last P V phase, DMA requests hidden from the compiler in inline assembly because
// its wait-count pass puts vmcnt(0) in front of every transposed LDS read that follows a visible LDS-DMA) was built, verified and measured at 63.4 us: the
// unit is bound by LDS reads, not by the fill - every query tile reads both images again, 2.2 MB per unit at 128 bytes per clock = 8 of the unit's 21 us -
// so the simple form stays.
template <int QSPLIT> __global__ __launch_bounds__(64 * AL_NW, 1) void attn_fwd_long_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ o, float* __restrict__ lse,
                                                                                           int Tn, int H, int nitems, float scale) {
  constexpr int NT = AL_NT;
  using IMG = DenseImage<NT, AL_NW>;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [K image | V image]
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int D = H * AD_DH, ldb = 3 * D * 2;
  const int Bn = nitems / H;
  const float c2 = scale * 1.4426950408889634f;
  const int ntiles = (Tn + 15) >> 4;                      // 24 .. 30 (launcher)
  const int tph = (ntiles + QSPLIT - 1) / QSPLIT;         // query tiles per unit
  bool dead[4];                                           // rows >= Tn of the last key tile
#pragma unroll
  for (int r = 0; r < 4; ++r) dead[r] = (ntiles - 1) * 16 + 4 * g + r >= Tn;
  IMG img;
  img.init(wave, lane, ldb);
  for (int i = tid * 16; i < 2 * IMG::BYTES; i += 64 * AL_NW * 16) *reinterpret_cast<u32x4_*>(smem + i) = u32x4_{0u, 0u, 0u, 0u};
  const bf16* sK = reinterpret_cast<const bf16*>(smem);
  const bf16* sV = reinterpret_cast<const bf16*>(smem + IMG::BYTES);
  const bf16* kKc = sK + c * AD_DH + 8 * g;                                     // per-lane bases of the fragment reads (lane_kcontig / lane_kstrided)
  const bf16* kVs = sV + (4 * g + ((lane >> 2) & 3)) * AD_DH + 4 * (lane & 3);
  Frag<bf16> ones;
#pragma unroll
  for (int r = 0; r < 8; ++r) ones.v[r] = (bf16)1.0f;
  const int ge = g & 1, gh = g >> 1;

  int u = __builtin_amdgcn_readfirstlane(attn_item(blockIdx.x, gridDim.x, 0, Bn, H * QSPLIT));
  for (int n = 0; u >= 0; ++n) {
    const int it = u / QSPLIT, half = u - it * QSPLIT;
    const int b = it / H, h = it - b * H;
    const bf16* ibase = qkv + ((size_t)b * Tn * 3 * D + h * AD_DH);
    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(ibase), 0, (Tn - 1) * ldb + 4 * D + AD_ROWB, 0x00020000);
    bf16* obase = o + ((size_t)b * Tn * D + h * AD_DH);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(obase, 0, (Tn - 1) * D * 2 + AD_ROWB, 0x00020000);
    __syncthreads();  // every wave is done with the previous unit's images (first unit: the zero fill is complete)
    img.stage(rq, smem, 2u * D, wave);
    img.stage(rq, smem + IMG::BYTES, 4u * D, wave);
    const int q_lo = half * tph, q_hi = min(ntiles, q_lo + tph);
    int qt = q_lo + wave;
    // this wave's 16 query rows (lane side): two K = 32 slabs and the tail of head_dim (lanes g >= 2 load zeros: outside the descriptor)
    auto q_load = [&](int t, u32x4_& a0, u32x4_& a1, u32x4_& a2) {
      const unsigned off = (unsigned)((t * 16 + c) * ldb + 16 * g);
      a0 = __builtin_amdgcn_raw_buffer_load_b128(rq, off, 0, 0);
      a1 = __builtin_amdgcn_raw_buffer_load_b128(rq, off + 64, 0, 0);
      a2 = __builtin_amdgcn_raw_buffer_load_b128(rq, g < 2 ? off + 128 : 0x7FFFFF00u, 0, 0);
    };
    u32x4_ q0n = {0u, 0u, 0u, 0u}, q1n = q0n, q2n = q0n;
    if (qt < q_hi) q_load(qt, q0n, q1n, q2n);
    __syncthreads();  // (vmcnt(0) + barrier) the images have landed for every wave
    for (; qt < q_hi; qt += AL_NW) {
      Frag<bf16> xq0, xq1, xq2;
      xq0.v = __builtin_bit_cast(bf16x8, q0n);
      xq1.v = __builtin_bit_cast(bf16x8, q1n);
      xq2.v = __builtin_bit_cast(bf16x8, q2n);
      if (qt + AL_NW < q_hi) q_load(qt + AL_NW, q0n, q1n, q2n);  // the next round's rows, under this round's products
      // ---- scores, raw: p[jt][r] = q_c . k_(16 jt + 4 g + r); tiles beyond the sequence are skipped (wave-uniform) and count as -inf
      f32x4 p[NT];
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        if (jt < 24 || jt < ntiles) {
          a = mma(lane_kcontig(kKc, jt * 16, 0), xq0, a);
          a = mma(lane_kcontig(kKc, jt * 16, 32), xq1, a);
          a = mma(lane_kcontig(kKc, jt * 16, 64), xq2, a);
        }
        p[jt] = a;
      }
#pragma unroll
      for (int jt = 23; jt < NT; ++jt) {
        if (jt == ntiles - 1) {
#pragma unroll
          for (int r = 0; r < 4; ++r) p[jt][r] = dead[r] ? -INFINITY : p[jt][r];
        } else if (jt >= ntiles) {
          p[jt] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        }
      }
      float mx = p[0][0];
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) {
        mx = fmaxf(fmaxf(mx, p[jt][0]), p[jt][1]);
        mx = fmaxf(fmaxf(mx, p[jt][2]), p[jt][3]);
      }
      mx = max_xor32(max_xor16(mx));
      const float m2 = mx * c2;
#pragma unroll
      for (int jt = 0; jt < NT; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) p[jt][r] = __builtin_amdgcn_exp2f(fmaf(p[jt][r], c2, -m2));
      // ---- o = P V and the row sum of the bf16-rounded P (sixth output tile against an all-ones operand)
      f32x4 oacc[6];
#pragma unroll
      for (int dt = 0; dt < 6; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < NT / 2; ++ks) {
        if (ks < 12 || 2 * ks < ntiles) {  // (a skipped tile of an odd count has p = 0 and meets zero rows of the V image)
          const Frag<bf16> wf = acc_pair_frag(p[2 * ks], p[2 * ks + 1]);
#pragma unroll
          for (int dt = 0; dt < 5; ++dt) oacc[dt] = mma(lane_kstrided(kVs, 32 * ks, dt * 16), wf, oacc[dt]);
          oacc[5] = mma(ones, wf, oacc[5]);
        }
      }
      const float l = oacc[5][0];
      const float inv = __builtin_amdgcn_rcpf(l);
      const unsigned o_off = (unsigned)((qt * 16 + c) * D * 2);
#pragma unroll
      for (int dt = 0; dt < 5; ++dt) oacc[dt] *= inv;  // (before the lane exchange: see the note in the single-chunk kernel)
#pragma unroll
      for (int d = 0; d < 4; d += 2)
        __builtin_amdgcn_raw_buffer_store_b128(pack_bf16x8(swap_pair(oacc[d], oacc[d + 1])), ro, o_off + (unsigned)(((d + ge) * 16 + 8 * gh) * 2), 0, 0);
      {
        const u32x2_ w{pack_bf16(oacc[4][0], oacc[4][1]), pack_bf16(oacc[4][2], oacc[4][3])};
        __builtin_amdgcn_raw_buffer_store_b64(w, ro, o_off + (unsigned)((64 + 4 * g) * 2), 0, 0);
      }
      const int q = qt * 16 + c;
      if (lse != nullptr && g == 0 && q < Tn) lse[((size_t)b * H + h) * Tn + q] = mx * scale + __logf(l);
    }
    u = __builtin_amdgcn_readfirstlane(attn_item(blockIdx.x, gridDim.x, n + 1, Bn, H * QSPLIT));
  }
}
template __global__ void attn_fwd_long_kernel<2>(const bf16* __restrict__, bf16* __restrict__, float* __restrict__, int, int, int, float);

// ---------------------------------------------------------------------------------------------------------------------------------------------------------
// Backward for the same sequences (369 .. 480 tokens), as two kernels of the same build as attn_fwd_long_kernel (P is recomputed from the saved
// log-sum-exp, as everywhere in this file's family; no atomics, deterministic):
//   dq   K and V images of the item in LDS; a wave owns 16 query rows: s = q k^T and dp = dO v^T tile by tile, ds = p (dp - delta) scale packed to bf16
//        at once (the whole ds row of 480 keys is 60 registers), then dQ = ds K from the K image read k-strided.  Also writes delta = scale sum_d dO O.
//   dkv  Q and dO images in LDS, log-sum-exp and delta of the item's rows beside them; a wave owns 16 keys and walks the query tiles in pairs (one K = 32
//        step of the two output products): s^T, dp^T, p^T, ds^T for the pair, dV += p^T dO, dK += ds^T Q - no score row is kept at all.
// Replace attn_bwd_dq_kernel / attn_bwd_dkv_kernel (key / query chunks of 160 rows staged through registers, 99.5 + 157.6 us per call at ds3 B = 64).
constexpr int AL_PAD = 64;  // zero bytes behind the last image: the head_dim tail of an image's last row reads 32 bytes past it (times zero - but not NaN)
V4H_DEV void store_row5(__amdgpu_buffer_rsrc_t r, unsigned row_off, f32x4* t, int g) {  // 5 output tiles of 16 columns -> 80 bf16 of this lane's row
  const int ge = g & 1, gh = g >> 1;
  // the lane exchange below is inline assembly: the wait states an MFMA result needs before a vector instruction reads it are inserted only for
  // instructions the compiler can see, so they are spent here explicitly (tied to the five tiles, so that their MFMAs are issued before)
  asm volatile("s_nop 7\n\ts_nop 7" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]));
#pragma unroll
  for (int d = 0; d < 4; d += 2)
    __builtin_amdgcn_raw_buffer_store_b128(pack_bf16x8(swap_pair(t[d], t[d + 1])), r, row_off + (unsigned)(((d + ge) * 16 + 8 * gh) * 2), 0, 0);
  const u32x2_ w{pack_bf16(t[4][0], t[4][1]), pack_bf16(t[4][2], t[4][3])};
  __builtin_amdgcn_raw_buffer_store_b64(w, r, row_off + (unsigned)((64 + 4 * g) * 2), 0, 0);
}
V4H_DEV void load_row3(__amdgpu_buffer_rsrc_t r, unsigned off, int g, Frag<bf16>& a0, Frag<bf16>& a1, Frag<bf16>& a2) {  // a lane-side row: d = 8 g + 0..7, + 32, + 64 (zero from 80)
  a0.v = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
  a1.v = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, off + 64, 0, 0));
  a2.v = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, g < 2 ? off + 128 : 0x7FFFFF00u, 0, 0));
}

template <int NT, int NW, int QSPLIT, int MINB> __global__ __launch_bounds__(64 * NW, MINB) void attn_bwd_dq_img_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ o,
                                                                                              const bf16* __restrict__ dout, const float* __restrict__ lse,
                                                                                              float* __restrict__ delta, bf16* __restrict__ dqkv, int Tn, int H,
                                                                                              int nitems, float scale) {
  constexpr int NRD = (NT / QSPLIT + NW - 1) / NW;  // rounds of tiles per wave and unit
  using IMG = DenseImage<NT, NW>;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [K image | V image | pad]
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int D = H * AD_DH, ldb = 3 * D * 2, ldo = D * 2;
  const int Bn = nitems / H;
  const float c2 = scale * 1.4426950408889634f;
  const int ntiles = (Tn + 15) >> 4;
  const int tph = (ntiles + QSPLIT - 1) / QSPLIT;
  bool dead[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) dead[r] = (ntiles - 1) * 16 + 4 * g + r >= Tn;
  IMG img;
  img.init(wave, lane, ldb);
  for (int i = tid * 16; i < 2 * IMG::BYTES + AL_PAD; i += 64 * NW * 16) *reinterpret_cast<u32x4_*>(smem + i) = u32x4_{0u, 0u, 0u, 0u};
  const bf16* sK = reinterpret_cast<const bf16*>(smem);
  const bf16* sV = reinterpret_cast<const bf16*>(smem + IMG::BYTES);
  const bf16* kKc = sK + c * AD_DH + 8 * g;
  const bf16* kVc = sV + c * AD_DH + 8 * g;
  const bf16* kKs = sK + (4 * g + ((lane >> 2) & 3)) * AD_DH + 4 * (lane & 3);

  int u = __builtin_amdgcn_readfirstlane(attn_item(blockIdx.x, gridDim.x, 0, Bn, H * QSPLIT));
  for (int n = 0; u >= 0; ++n) {
    const int it = u / QSPLIT, half = u - it * QSPLIT;
    const int b = it / H, h = it - b * H;
    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(qkv + ((size_t)b * Tn * 3 * D + h * AD_DH)), 0,
                                                                         (Tn - 1) * ldb + 4 * D + AD_ROWB, 0x00020000);
    const size_t orow = (size_t)b * Tn * D + h * AD_DH;
    const __amdgpu_buffer_rsrc_t rdo = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(dout + orow), 0, (Tn - 1) * ldo + AD_ROWB, 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(o + orow), 0, (Tn - 1) * ldo + AD_ROWB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rdq = __builtin_amdgcn_make_buffer_rsrc(dqkv + ((size_t)b * Tn * 3 * D + h * AD_DH), 0, (Tn - 1) * ldb + AD_ROWB, 0x00020000);
    __syncthreads();  // every wave is done with the previous unit's images
    img.stage(rq, smem, 2u * D, wave);
    img.stage(rq, smem + IMG::BYTES, 4u * D, wave);
    const int q_hi = min(ntiles, (half + 1) * tph);
    // Both rounds' rows (q, dO of 16 queries each, lane side), their delta = sum_d dO O and log-sum-exp are fetched here, under the fill of the images:
    // nothing is loaded inside the product loops.
    Frag<bf16> rq0[NRD], rq1[NRD], rq2[NRD], rd0[NRD], rd1[NRD], rd2[NRD];
    float rdelta[NRD], rlse[NRD];
#pragma unroll
    for (int rd = 0; rd < NRD; ++rd) {
      const int qq = (half * tph + wave + NW * rd) * 16 + c;  // (rows of a tile beyond this unit's range are loaded and not used)
      load_row3(rq, (unsigned)(qq * ldb + 16 * g), g, rq0[rd], rq1[rd], rq2[rd]);
      load_row3(rdo, (unsigned)(qq * ldo + 16 * g), g, rd0[rd], rd1[rd], rd2[rd]);
      Frag<bf16> xo0, xo1, xo2;
      load_row3(ro, (unsigned)(qq * ldo + 16 * g), g, xo0, xo1, xo2);
      float dq_ = 0.f;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj)
        dq_ += (float)xo0.v[jj] * (float)rd0[rd].v[jj] + (float)xo1.v[jj] * (float)rd1[rd].v[jj] + (float)xo2.v[jj] * (float)rd2[rd].v[jj];
      dq_ += __shfl_xor(dq_, 16, 64);
      dq_ += __shfl_xor(dq_, 32, 64);
      rdelta[rd] = dq_;
      rlse[rd] = qq < Tn ? lse[((size_t)b * H + h) * Tn + qq] * 1.4426950408889634f : 0.f;
    }
    __syncthreads();  // the images have landed for every wave
#pragma unroll
    for (int rd = 0; rd < NRD; ++rd) {
      const int qt = half * tph + wave + NW * rd;
      if (qt >= q_hi) break;  // wave-uniform
      const int q = qt * 16 + c;
      const Frag<bf16> xq0 = rq0[rd], xq1 = rq1[rd], xq2 = rq2[rd], xd0 = rd0[rd], xd1 = rd1[rd], xd2 = rd2[rd];
      const float lse2 = rlse[rd], delta_q = rdelta[rd];
      const float delta_s = delta_q * scale;
      if (g == 0 && q < Tn) delta[((size_t)b * H + h) * Tn + q] = delta_s;
      // key tiles in pairs (one K = 32 step of the dQ product): s, dp, p, ds of the pair, then dQ += ds K at once - no score row is kept.  Keys beyond the
      // sequence need no mask here: their ds is finite and meets zero rows of the K image.
      f32x4 dq[5];
#pragma unroll
      for (int dt = 0; dt < 5; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
      for (int ks = 0; 2 * ks < ntiles; ++ks) {
        unsigned dw[4];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int j0 = (2 * ks + e) * 16;
          f32x4 a = {0.f, 0.f, 0.f, 0.f}, d = {0.f, 0.f, 0.f, 0.f};
          a = mma(lane_kcontig(kKc, j0, 0), xq0, a);
          a = mma(lane_kcontig(kKc, j0, 32), xq1, a);
          a = mma(lane_kcontig(kKc, j0, 64), xq2, a);
          d = mma(lane_kcontig(kVc, j0, 0), xd0, d);
          d = mma(lane_kcontig(kVc, j0, 32), xd1, d);
          d = mma(lane_kcontig(kVc, j0, 64), xd2, d);
          float ds[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) ds[r] = __builtin_amdgcn_exp2f(fmaf(a[r], c2, -lse2)) * fmaf(d[r], scale, -delta_s);
          dw[2 * e] = pack_bf16(ds[0], ds[1]);
          dw[2 * e + 1] = pack_bf16(ds[2], ds[3]);
        }
        Frag<bf16> wf;
        wf.v = __builtin_bit_cast(bf16x8, u32x4_{dw[0], dw[1], dw[2], dw[3]});
#pragma unroll
        for (int dt = 0; dt < 5; ++dt) dq[dt] = mma(lane_kstrided(kKs, 32 * ks, dt * 16), wf, dq[dt]);
      }
      store_row5(rdq, (unsigned)(q * ldb), dq, g);
    }
    u = __builtin_amdgcn_readfirstlane(attn_item(blockIdx.x, gridDim.x, n + 1, Bn, H * QSPLIT));
  }
}

template <int NT, int NW, int QSPLIT, int MINB> __global__ __launch_bounds__(64 * NW, MINB) void attn_bwd_dkv_img_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ dout,
                                                                                               const float* __restrict__ lse, const float* __restrict__ delta,
                                                                                               bf16* __restrict__ dqkv, int Tn, int H, int nitems, float scale) {
  constexpr int NRD = (NT / QSPLIT + NW - 1) / NW;  // rounds of tiles per wave and unit
  using IMG = DenseImage<NT, NW>;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [Q image | dO image | pad | lse (log-2 units) | delta * scale]
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int D = H * AD_DH, ldb = 3 * D * 2, ldo = D * 2;
  const int Bn = nitems / H;
  const float c2 = scale * 1.4426950408889634f;
  const int ntiles = (Tn + 15) >> 4;
  const int tph = (ntiles + QSPLIT - 1) / QSPLIT;
  IMG imq, imd;
  imq.init(wave, lane, ldb);
  imd.init(wave, lane, ldo);
  for (int i = tid * 16; i < 2 * IMG::BYTES + AL_PAD; i += 64 * NW * 16) *reinterpret_cast<u32x4_*>(smem + i) = u32x4_{0u, 0u, 0u, 0u};
  const bf16* sQ = reinterpret_cast<const bf16*>(smem);
  const bf16* sDO = reinterpret_cast<const bf16*>(smem + IMG::BYTES);
  float* sLse = reinterpret_cast<float*>(smem + 2 * IMG::BYTES + AL_PAD);
  float* sDelta = sLse + NT * 16;
  const bf16* kQc = sQ + c * AD_DH + 8 * g;
  const bf16* kDc = sDO + c * AD_DH + 8 * g;
  const bf16* kQs = sQ + (4 * g + ((lane >> 2) & 3)) * AD_DH + 4 * (lane & 3);
  const bf16* kDs = sDO + (4 * g + ((lane >> 2) & 3)) * AD_DH + 4 * (lane & 3);

  int u = __builtin_amdgcn_readfirstlane(attn_item(blockIdx.x, gridDim.x, 0, Bn, H * QSPLIT));
  for (int n = 0; u >= 0; ++n) {
    const int it = u / QSPLIT, half = u - it * QSPLIT;
    const int b = it / H, h = it - b * H;
    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(qkv + ((size_t)b * Tn * 3 * D + h * AD_DH)), 0,
                                                                         (Tn - 1) * ldb + 4 * D + AD_ROWB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rdo = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(dout + ((size_t)b * Tn * D + h * AD_DH)), 0, (Tn - 1) * ldo + AD_ROWB,
                                                                          0x00020000);
    const __amdgpu_buffer_rsrc_t rdx = __builtin_amdgcn_make_buffer_rsrc(dqkv + ((size_t)b * Tn * 3 * D + h * AD_DH), 0, (Tn - 1) * ldb + 4 * D + AD_ROWB, 0x00020000);
    __syncthreads();  // every wave is done with the previous unit's images and rows
    imq.stage(rq, smem, 0u, wave);
    imd.stage(rdo, smem + IMG::BYTES, 0u, wave);
    for (int r = tid; r < NT * 16; r += 64 * NW) {  // a query row beyond the sequence gets p = exp2(s - huge) = 0
      sLse[r] = r < Tn ? lse[((size_t)b * H + h) * Tn + r] * 1.4426950408889634f : 1e30f;
      sDelta[r] = r < Tn ? delta[((size_t)b * H + h) * Tn + r] : 0.f;
    }
    const int k_hi = min(ntiles, (half + 1) * tph);
    __syncthreads();  // images and rows have landed for every wave
    for (int kt = half * tph + wave; kt < k_hi; kt += NW) {
      const int key = kt * 16 + c;
      Frag<bf16> xk0, xk1, xk2, xv0, xv1, xv2;  // this wave's 16 keys, lane side (fetching both rounds' rows under the fill measured slower: 105.8 vs 89.7 us)
      load_row3(rq, (unsigned)(key * ldb + 2 * D + 16 * g), g, xk0, xk1, xk2);
      load_row3(rq, (unsigned)(key * ldb + 4 * D + 16 * g), g, xv0, xv1, xv2);
      f32x4 dk[5], dv[5];
#pragma unroll
      for (int dt = 0; dt < 5; ++dt) dk[dt] = dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
      for (int ks = 0; 2 * ks < ntiles; ++ks) {
        Frag<bf16> pf, df;
        unsigned pw[4], dw[4];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int i0 = (2 * ks + e) * 16;  // (an odd count's last pair has a second tile of zero rows: lse = huge there, p = 0)
          f32x4 a = {0.f, 0.f, 0.f, 0.f}, d = {0.f, 0.f, 0.f, 0.f};
          a = mma(lane_kcontig(kQc, i0, 0), xk0, a);
          a = mma(lane_kcontig(kQc, i0, 32), xk1, a);
          a = mma(lane_kcontig(kQc, i0, 64), xk2, a);
          d = mma(lane_kcontig(kDc, i0, 0), xv0, d);
          d = mma(lane_kcontig(kDc, i0, 32), xv1, d);
          d = mma(lane_kcontig(kDc, i0, 64), xv2, d);
          const f32x4 l4 = *reinterpret_cast<const f32x4*>(sLse + i0 + 4 * g), e4 = *reinterpret_cast<const f32x4*>(sDelta + i0 + 4 * g);
          float pr[4], ds[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            pr[r] = __builtin_amdgcn_exp2f(fmaf(a[r], c2, -l4[r]));
            ds[r] = pr[r] * fmaf(d[r], scale, -e4[r]);
          }
          pw[2 * e] = pack_bf16(pr[0], pr[1]); pw[2 * e + 1] = pack_bf16(pr[2], pr[3]);
          dw[2 * e] = pack_bf16(ds[0], ds[1]); dw[2 * e + 1] = pack_bf16(ds[2], ds[3]);
        }
        pf.v = __builtin_bit_cast(bf16x8, u32x4_{pw[0], pw[1], pw[2], pw[3]});
        df.v = __builtin_bit_cast(bf16x8, u32x4_{dw[0], dw[1], dw[2], dw[3]});
#pragma unroll
        for (int dt = 0; dt < 5; ++dt) {
          dv[dt] = mma(lane_kstrided(kDs, 32 * ks, dt * 16), pf, dv[dt]);
          dk[dt] = mma(lane_kstrided(kQs, 32 * ks, dt * 16), df, dk[dt]);
        }
      }
      store_row5(rdx, (unsigned)(key * ldb + 2 * D), dk, g);
      store_row5(rdx, (unsigned)(key * ldb + 4 * D), dv, g);
    }
    u = __builtin_amdgcn_readfirstlane(attn_item(blockIdx.x, gridDim.x, n + 1, Bn, H * QSPLIT));
  }
}
#define V4H_BWD_IMG(NT, NW, QS, MB)                                                                                                                       \
  template __global__ void attn_bwd_dq_img_kernel<NT, NW, QS, MB>(const bf16* __restrict__, const bf16* __restrict__, const bf16* __restrict__,          \
                                                                  const float* __restrict__, float* __restrict__, bf16* __restrict__, int, int, int, float); \
  template __global__ void attn_bwd_dkv_img_kernel<NT, NW, QS, MB>(const bf16* __restrict__, const bf16* __restrict__, const float* __restrict__,         \
                                                                   const float* __restrict__, bf16* __restrict__, int, int, int, float);
V4H_BWD_IMG(AL_NT, AL_NW, 2, 1)  // 369 .. 480 tokens: half an item per unit, one workgroup per CU
// (NT = 10, NW = 10, an item per unit, two workgroups per CU was measured for 129 .. 160 tokens against attn_bwd_fused_kernel: 37.1 + 32.2 us per call
//  against 55 for the fused kernel - every operand read twice, two fills per item - and is not instantiated)
#undef V4H_BWD_IMG

// (A single-chunk backward built from these inner loops - the schedule of attn_bwd_fused_kernel with immediate-offset fragment reads, zero rows instead of
// per-element masks, pair loops and query rows fetched a phase ahead: 168 VGPRs, every test green - measured 58.7 us per call against the fused kernel's
// 54.3-55.6 on the same box and was removed: at 2-3 waves per SIMD that kernel waits on its own dependency chains and barriers, not on its vector
// instruction count.  profiles/r03_attn_counters.md.)

}  // namespace v4h_dense
