// Second-generation bf16 contraction for the token-sized Linears of the ViT path (same operand conventions and the same
// dense, swizzled LDS images as v4h_gemm.h; reference nn/vit.py: qkv :416, proj :420, timm Mlp fc1/fc2 :317-322 and their
// dgrad / wgrad).  What changed, and why (docs/history_r01-r04.md section 5, round 2):
//
//   * ONE 8-wave workgroup per CU on a 256 x 160 tile (wave tile 64 x 80 as before): 98 instead of 71 FLOP per staged byte -
//     the CU's global->LDS intake (about 70 GB/s per CU) is what bounded the 128 x 160 / two-workgroup structure.
//   * a THREE-slot ring of K = 64 stages filled by global->LDS DMA that never drains: a stage is requested two K-steps before
//     it is read, counted `s_waitcnt vmcnt(n)` only (the exact number of younger vector-memory instructions this wave has issued
//     is tracked in scalar registers), one raw `s_barrier` per K-step, and the request stream runs straight on into the
//     workgroup's NEXT output tile, so neither the pipeline fill nor the epilogue stalls the matrix pipe.
//   * (lock-step schedule) fragments are read one K = 32 slab ahead of the MFMAs that use them (two fragment sets), so the LDS latency sits
//     behind 20 MFMAs instead of in front of them; the barrier is placed in the middle of a K-step, between the two slabs.
//   * the epilogue never touches LDS: `v_permlane16_swap` turns two 16 x 16 accumulator tiles into 8 consecutive columns per lane,
//     i.e. 16-byte bf16 / 32-byte f32 row segments, written with buffer stores whose bounds check replaces every branch
//     (a skipped store would falsify the vmcnt bookkeeping).  The bias is DMA-ed into LDS with the tile's first stage and the
//     accumulators START from it.
//   * bias gradients of the wgrad form: an extra MFMA against an all-ones fragment, the duty rotating over the waves that hold
//     the same P fragments (one slab in ntj * WJ each).
//
// Two schedules share all of the above (Gemm2Cfg<..., PP>):
//   * lock-step (the first one; now the host of the ablation builds): all eight waves run the same phase - fragments one K = 32 slab ahead
//     of the MFMAs, the request in the middle of the second MFMA group, one barrier per K-step.  Every overhead that was removed from it by
//     ablation came back 1 : 1 as time: both waves of a SIMD push their DMA instructions (about 100 clocks each for the issuing wave) at the
//     same moment, and a wave issues in order, so the MFMAs behind them wait.
//   * ping-pong (the default wherever the kernel is used): per SIMD one wave runs a LOAD slot (all 18 fragment reads of a stage, its share of
//     the DMA of the stage two ahead, at a seam the previous tile's epilogue) while its partner runs a MATRIX slot (the 40 MFMAs of the stage
//     it loaded one slot earlier) - half 0 (waves 0..3) load, matrix, barrier; half 1 matrix of the previous stage, load, barrier: one barrier
//     per stage, the halves in anti-phase by construction.  docs/history_r01-r04.md section 5 "What round 2 found" 8, profiles/r02_gemm2_pingpong_timeline.txt.
#pragma once
#include "v4h_gemm.h"

// cache policy of the saved GELU derivative (buffer instruction aux bits on gfx950: 1 = sc0, 2 = nt, 16 = sc1): non-temporal, see store8_saved (v4h_gemm.h)
#define V4H_SAVED_AUX 2


// Fragment addressing with a minimum of registers.  The images are the dense swizzled ones of v4h_gemm.h; what is new is that a wave keeps only
// the byte offsets that really differ per lane (2 to 4 integers per operand) and reaches every other fragment through the instruction's
// immediate offset - the generic ImgK*::frag forms made the compiler keep 18 to 36 hoisted address registers, which did not fit beside two
// fragment sets.  `t` = 16-wide sub-tile of the wave (x or y), `h` = K = 32 slab of the stage.
struct FragKC {  // K-contiguous image, rows of 8 chunks (BK = 64, bf16): position = kc ^ (row & 6), and 16 t rows do not change row & 6
  int b[2];
  V4H_DEV void init(int idx0_wave, int lane) {
    const int row = idx0_wave + (lane & 15), g = lane >> 4, s = row & 6;
    b[0] = (row * 8 + (g ^ s)) * 16;
    b[1] = (row * 8 + ((4 + g) ^ s)) * 16;
  }
  template <int T, int H> V4H_DEV Frag<bf16> frag(const char* stage, int slot_off) const {
    Frag<bf16> f;
    f.v = *reinterpret_cast<const bf16x8*>(stage + (b[H] + slot_off) + T * 16 * 128);
    return f;
  }
};
template <int COLS> struct FragKS {  // K-strided image [k][COLS], transposed reads; the two swizzles of sw_kstrided
  static constexpr int CPR = COLS / 8, NB = (CPR % 16 == 0) ? 4 : 2;
  int b[NB];
  V4H_DEV void init(int idx0_wave, int lane) {
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3, r0 = 8 * g + q;
#pragma unroll
    for (int t = 0; t < NB; ++t) {
      const int ch = (idx0_wave + 16 * t + 4 * p) / 8;
      b[t] = (r0 * CPR + (ch ^ sw_kstrided<bf16, CPR>(r0))) * 16 + 8 * (p & 1);
    }
  }
  // CPR % 16 == 0: the swizzle touches chunk bits 1..3, sub-tile t sits in bits 1..2: one base per t (t < 4).  Otherwise the swizzle is 0 or 2
  // and only flips bit 1, whose value alternates with t: one base per parity of t, two sub-tiles further = 64 bytes further.
  template <int T, int H> V4H_DEV Frag<bf16> frag(const char* stage, int slot_off) const {
    static_assert(NB == 2 || T < 4, "one base per sub-tile");
    constexpr int BI_ = NB == 4 ? T : (T & 1), EXTRA = NB == 4 ? 0 : (T >> 1) * 64;
    const char* a0 = stage + (b[BI_] + slot_off) + EXTRA + H * 32 * CPR * 16;
    const bf16x4 lo = lds_tr_read(reinterpret_cast<const bf16*>(a0));
    const bf16x4 hi = lds_tr_read(reinterpret_cast<const bf16*>(a0 + 4 * CPR * 16));
    Frag<bf16> f;
    f.v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return f;
  }
};

template <bool PKS_, bool QKS_, int EPI_, bool COLSUM_, int DBG_ = 0, bool PP_ = true> struct Gemm2Cfg {
#ifndef V4H_ABLATIONS
  static_assert(DBG_ == 0 && PP_, "ablation builds (DBG != 0, lock-step schedule) need -DV4H_ABLATIONS: several of them are wrong by construction");
#endif
  using T = bf16;
  static constexpr bool PKS = PKS_, QKS = QKS_, COLSUM = COLSUM_;
  static constexpr int EPI = EPI_, DBG = DBG_;
  static constexpr bool PP = PP_;  // ping-pong schedule: the two waves of a SIMD alternate a load slot and a matrix slot
  static constexpr int BI = 256, BJ = 160, BK = 64, WI = 4, WJ = 2, NW = 8, NT = 512, WTI = 64, WTJ = 80, TI = 4, TJ = 5, NS = 3;
  using ImgP = typename std::conditional<PKS, ImgKStrided<bf16, BI, BK, NW>, ImgKContig<bf16, BI, BK, NW>>::type;
  using ImgQ = typename std::conditional<QKS, ImgKStrided<bf16, BJ, BK, NW>, ImgKContig<bf16, BJ, BK, NW>>::type;
  using AddrP = typename std::conditional<PKS, FragKS<BI>, FragKC>::type;
  using AddrQ = typename std::conditional<QKS, FragKS<BJ>, FragKC>::type;
  static constexpr int P_BYTES = ImgP::BYTES, Q_BYTES = ImgQ::BYTES, STAGE = P_BYTES + Q_BYTES;
  static constexpr int BIAS_OFF = NS * STAGE, BIAS_SLOT = 1024;  // two slots of one DMA instruction each (BJ floats used)
  static constexpr size_t LDS_BYTES = (size_t)NS * STAGE + 2 * BIAS_SLOT;
  static_assert(LDS_BYTES <= 160 * 1024, "ring must fit the CU's LDS");
  // vector-memory instructions of one epilogue per wave (all unconditional: bounds are checked by the buffer hardware)
  static constexpr int NCHUNK = TI * TJ / 2;  // 8-column chunks per lane
  static constexpr int EPI_OPS = EPI == EPI_STORE ? NCHUNK : EPI == EPI_GELU ? 2 * NCHUNK : EPI == EPI_DGELU ? 2 * NCHUNK : EPI == EPI_SLAB_F32 ? 2 * NCHUNK : -1;
  static_assert(EPI_OPS > 0, "epilogue not built for the 256 x 160 kernel");
};

// (wait_vmcnt64: the run-time counted `s_waitcnt vmcnt(n)` lives in v4h_common.h)

#ifdef V4H_GEMM2_STAMPS
// Diagnostic build only (V4H_EXTRA_FLAGS=-DV4H_GEMM2_STAMPS): every wave of the ping-pong schedule stamps the shader clock at the boundaries of its slots for
// the stages 3..10 of its workgroup (a tile seam in the middle), first into the last 2 KB of the CU's LDS - no vector-memory instruction, so the counted
// waits are untouched - and copies them out at the end (tools/experiments/gemm2_stamps.py).
constexpr int G2_ST_FIRST = 3, G2_ST_N = 8, G2_ST_K = 8;
__device__ unsigned v4h_gemm2_stamp_buf[256 * 8 * G2_ST_N * G2_ST_K];
#define V4H_G2_STAMP(k)                                                                                                             \
  do {                                                                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                                              \
    unsigned long long now_;                                                                                                        \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_) :: "memory");                                                  \
    if (lane == 0 && st_n >= G2_ST_FIRST && st_n < G2_ST_FIRST + G2_ST_N)                                                            \
      reinterpret_cast<unsigned*>(smem + C::LDS_BYTES)[(wave * G2_ST_N + st_n - G2_ST_FIRST) * G2_ST_K + (k)] = (unsigned)now_;             \
    __builtin_amdgcn_sched_barrier(0);                                                                                              \
  } while (0)
#else
#define V4H_G2_STAMP(k) do { } while (0)
#endif

template <class C> __global__ __launch_bounds__(C::NT, 2) void v4h_gemm2_kernel(const GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  __builtin_assume(wave >= 0 && wave < C::NW);
  const int wi = wave / C::WJ, wj = wave % C::WJ;
  const bf16* gP = reinterpret_cast<const bf16*>(a.P);
  const bf16* gQ = reinterpret_cast<const bf16*>(a.Q);

  // virtual tile id -> (i-tile, j-tile, k-split): the XCD-aware maps of v4h_gemm.h (speed only)
  const bool few_rows = a.nz == 1 && a.nti < 8;
  const int nvirt = few_rows ? a.nti * a.ntj : (a.nz == 1 ? ((a.nti + 7) / 8) * 8 * a.ntj : a.nti * a.ntj * a.nz);
  auto sgpr = [](int x) { return __builtin_amdgcn_readfirstlane(x); };  // integer division runs on the vector ALU: pin the results back to scalars
  auto decode = [&](int& v, int& ti, int& tj, int& tz) -> bool {  // skips the holes of the padded no-split map
    for (; v < nvirt; v += gridDim.x) {
      if (few_rows) { ti = sgpr(v % a.nti); tj = sgpr(v / a.nti); tz = 0; return true; }
      if (a.nz == 1) {
        const int xcd = v & 7, slot = v >> 3, grp = sgpr(slot / a.ntj);
        tj = slot - grp * a.ntj; ti = grp * 8 + xcd; tz = 0;
        if (ti < a.nti) return true;
        continue;
      }
      const int tile = sgpr(v / a.nz);
      tz = v - tile * a.nz;
      tj = sgpr(tile / a.nti);
      ti = tile - tj * a.nti;
      return true;
    }
    return false;
  };

  // ------------------------------------------------------------------ request side: stages of this workgroup's tile list, in order
  typename C::ImgP stP;
  typename C::ImgQ stQ;
  int rv = blockIdx.x, r_ti = 0, r_tj = 0, r_tz = 0;
  bool r_valid = decode(rv, r_ti, r_tj, r_tz);
  int r_t = 0, r_nt = 0, r_kb = 0, r_ke = 0, r_slot = 0, r_par = 0, r_count = 0;
  // issue(): the DMA instructions of the next stage (returns how many this wave issued); advance(): the walk to the stage after it - at a tile
  // change the decode of the next tile (integer divisions) and both sets of per-lane source pointers.  The lock-step schedule runs them back to
  // back; the ping-pong schedule issues in the load slot and advances in the matrix slot, where scalar and vector ALU work hides between MFMAs.
  auto tile_setup = [&]() {
    r_kb = r_tz * a.klen;
    r_ke = min(a.K, r_kb + a.klen);
    r_nt = (r_ke - r_kb + C::BK - 1) >> 6;
    static_assert(C::BK == 64, "shift");
    stP.init(gP, a.ldp, r_ti * C::BI, r_kb, a.I, wave, lane);
    stQ.init(gQ, a.ldq, r_tj * C::BJ, r_kb, a.J, wave, lane);
  };
  if (r_valid) tile_setup();
  auto issue = [&]() -> int {
    if (!r_valid) return 0;
    int n = 0;
    if (r_t == 0 && C::EPI != EPI_SLAB_F32 && a.e.bias != nullptr && wave == 0) {  // the tile's bias slice travels with its first stage
      const int j = r_tj * C::BJ + 4 * lane;
      const void* src = (4 * lane < C::BJ && j + 4 <= a.J) ? (const void*)(a.e.bias + j) : (const void*)v4h_zero_page;
      dma16(src, smem + C::BIAS_OFF + r_par * C::BIAS_SLOT);
      ++n;
    }
    if (!(C::DBG & 1) || r_count < C::NS) {  // DBG 1 (ablation): only the first ring fill is really staged
      const int k0 = r_kb + r_t * C::BK;
      stP.stage(smem + r_slot * C::STAGE, k0, r_ke, a.ldp, wave);
      stQ.stage(smem + r_slot * C::STAGE + C::P_BYTES, k0, r_ke, a.ldq, wave);
      n += C::ImgP::count(wave) + C::ImgQ::count(wave);
    }
    return sgpr(n);
  };
  auto advance_walk = [&]() -> bool {  // true: a new tile starts, tile_setup() is due
    if (!r_valid) return false;
    ++r_count;
    r_slot = r_slot == C::NS - 1 ? 0 : r_slot + 1;
    if (++r_t == r_nt) {
      r_t = 0;
      r_par ^= 1;
      rv += gridDim.x;
      r_valid = decode(rv, r_ti, r_tj, r_tz);
      return r_valid;
    }
    return false;
  };
  auto request = [&]() -> int {
    const int n = issue();
    if (advance_walk()) tile_setup();
    return n;
  };

  // ------------------------------------------------------------------ compute side
  int cv = blockIdx.x, ti = 0, tj = 0, tz = 0;
  if (!decode(cv, ti, tj, tz)) return;
  int c_slot = 0, c_par = 0;
  f32x4 acc[C::TI][C::TJ];
  Frag<bf16> pA[C::TI], qA[C::TJ], pB[C::TI], qB[C::TJ];
  const int c = lane & 15, g = lane >> 4;

  typename C::AddrP adP;
  typename C::AddrQ adQ;
  adP.init(wi * C::WTI, lane);
  adQ.init(wj * C::WTJ, lane);
  auto read_frags = [&](Frag<bf16>* pf, Frag<bf16>* qf, int slot, auto hsel) {
    constexpr int H = decltype(hsel)::value;
    const int so = slot * C::STAGE;
    pf[0] = adP.template frag<0, H>(smem, so); pf[1] = adP.template frag<1, H>(smem, so);
    pf[2] = adP.template frag<2, H>(smem, so); pf[3] = adP.template frag<3, H>(smem, so);
    const char* sq = smem + C::P_BYTES;
    qf[0] = adQ.template frag<0, H>(sq, so); qf[1] = adQ.template frag<1, H>(sq, so); qf[2] = adQ.template frag<2, H>(sq, so);
    qf[3] = adQ.template frag<3, H>(sq, so); qf[4] = adQ.template frag<4, H>(sq, so);
    static_assert(C::TI == 4 && C::TJ == 5, "fragment lists are written out");
  };
  auto mfmas = [&](const Frag<bf16>* pf, const Frag<bf16>* qf, int x0, int x1) {
#pragma unroll
    for (int x = x0; x < x1; ++x)
#pragma unroll
      for (int y = 0; y < C::TJ; ++y) acc[x][y] = mma(qf[y], pf[x], acc[x][y]);
  };
  // Column sums of P over k (bias gradient of the wgrad form): one MFMA per row strip against an all-ones fragment (all 16 result rows equal),
  // added to memory right away by the lanes g == 0 - a few KB of float atomics per tile, and no register lives longer than the slab.  Each
  // (row strip, slab) is owned by exactly one of the ntj * WJ waves that hold the same P fragments (duty rotation over the slabs).
  auto colsum_slab = [&](const Frag<bf16>* pf, int row0) {
    Frag<bf16> ones;
#pragma unroll
    for (int r = 0; r < 8; ++r) ones.v[r] = (bf16)1.0f;
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(a.colsum, 0, a.I * 4, 0x00020000);
#pragma unroll
    for (int x = 0; x < C::TI; ++x) {
      const f32x4 t = mma(ones, pf[x], f32x4{0.f, 0.f, 0.f, 0.f});
      __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(t[0], rc, g == 0 ? (unsigned)(row0 + x * 16 + c) * 4u : 0x7FFFFFF0u, 0, 0);
    }
  };
  auto init_acc = [&](int par) {
    if (C::EPI != EPI_SLAB_F32 && a.e.bias != nullptr) {
      const float* bl = reinterpret_cast<const float*>(smem + C::BIAS_OFF + par * C::BIAS_SLOT) + wj * C::WTJ + 4 * g;
#pragma unroll
      for (int y = 0; y < C::TJ; ++y) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(bl + y * 16);
#pragma unroll
        for (int x = 0; x < C::TI; ++x) acc[x][y] = b;
      }
    } else {
#pragma unroll
      for (int x = 0; x < C::TI; ++x)
#pragma unroll
        for (int y = 0; y < C::TJ; ++y) acc[x][y] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };

  // ---- epilogue: registers -> memory, no LDS.  Chunk n < 8: tiles (x, 2q), (x, 2q + 1), x = n >> 1, q = n & 1 - 64 contiguous bytes of bf16
  // per row and instruction; n >= 8: the odd fifth column tile, paired over the row strips x = 2 (n - 8) and x + 1.  J is a whole number of
  // 160-column tiles (launcher), rows beyond I fall outside the buffer and are dropped by its bounds check: no predicate anywhere.
  auto epilogue = [&](int ti, int tj, int tz) -> int {
    // (the lane coordinates are laundered through an empty asm: the epilogue has three call sites in the ping-pong schedule and runs once per tile, so
    //  its per-lane offsets are to be recomputed at each, not hoisted out of the K loop into ten registers that live across it)
    int c = lane & 15, g = lane >> 4;
    asm volatile("" : "+v"(c), "+v"(g));
    const int ge = g & 1, gh = g >> 1;
    const int rowA = ti * C::BI + wi * C::WTI + c, colA = tj * C::BJ + wj * C::WTJ + 16 * ge + 8 * gh;
    const int rowB = rowA + 16 * ge, colB = tj * C::BJ + wj * C::WTJ + 64 + 8 * gh;
    auto chunk_val = [&](int n) -> f32x8 {
      if (n < 8) return swap_pair(acc[n >> 1][2 * (n & 1)], acc[n >> 1][2 * (n & 1) + 1]);
      return swap_pair(acc[2 * (n - 8)][4], acc[2 * (n - 8) + 1][4]);
    };
    auto chunk_tiles = [&](int n, f32x4& ta, f32x4& tb) {  // the two accumulator tiles of chunk n
      if (n < 8) { ta = acc[n >> 1][2 * (n & 1)]; tb = acc[n >> 1][2 * (n & 1) + 1]; }
      else { ta = acc[2 * (n - 8)][4]; tb = acc[2 * (n - 8) + 1][4]; }
    };
    // element offset of chunk n in a row-major tensor of row stride ld, from the lane's two base offsets
    auto chunk_off = [&](int n, unsigned offA, unsigned offB, int ld) -> unsigned {
      return n < 8 ? offA + (unsigned)((n >> 1) * 16 * ld + (n & 1) * 32) : offB + (unsigned)(2 * (n - 8) * 16 * ld);
    };
    if constexpr ((C::DBG & 2) != 0) {  // ablation: every MFMA live, one 16-byte store per lane and tile
      f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int x = 0; x < C::TI; ++x)
#pragma unroll
        for (int y = 0; y < C::TJ; ++y) sum += acc[x][y];
      const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(a.e.out, 0, (int)min((long)a.I * a.e.ldo * 2, 0x7FFFFFF0L), 0x00020000);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, sum), ro, (unsigned)(rowA * a.e.ldo + colA) * 2u, 0, 0);
    } else if constexpr (C::EPI == EPI_STORE || C::EPI == EPI_GELU || C::EPI == EPI_DGELU) {
      const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(a.e.out, 0, (int)min((long)a.I * a.e.ldo * 2, 0x7FFFFFF0L), 0x00020000);
      const unsigned oA = (unsigned)(rowA * a.e.ldo + colA), oB = (unsigned)(rowB * a.e.ldo + colB);
      if constexpr (C::EPI == EPI_STORE) {
#pragma unroll
        for (int n = 0; n < C::NCHUNK; ++n)
          __builtin_amdgcn_raw_buffer_store_b128(pack_bf16x8(chunk_val(n)), ro, chunk_off(n, oA, oB, a.e.ldo) * 2u, 0, 0);
      } else if constexpr (C::EPI == EPI_GELU) {  // out = gelu'(pre) (training only), out2 = gelu(pre)
        const bool train = a.e.out != nullptr;
        const __amdgpu_buffer_rsrc_t ro2 = __builtin_amdgcn_make_buffer_rsrc(a.e.out2, 0, (int)min((long)a.I * a.e.ldo2 * 2, 0x7FFFFFF0L), 0x00020000);
        const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(a.e.out, 0, train ? (int)min((long)a.I * a.e.ldo * 2, 0x7FFFFFF0L) : 0, 0x00020000);
        const unsigned pA2 = (unsigned)(rowA * a.e.ldo2 + colA), pB2 = (unsigned)(rowB * a.e.ldo2 + colB);
#pragma unroll
        for (int n = 0; n < C::NCHUNK; ++n) {
          f32x8 v = chunk_val(n), d;
          if (train) {
            gelu8_and_grad<bf16>(v, d);
          } else {
            gelu8_only<bf16>(v);
#pragma unroll
            for (int r = 0; r < 8; ++r) d.v[r] = 0.f;
          }
          __builtin_amdgcn_raw_buffer_store_b128(pack_bf16x8(d), rd, chunk_off(n, oA, oB, a.e.ldo) * 2u, 0, V4H_SAVED_AUX);  // (inference: zero-sized buffer, dropped)
          __builtin_amdgcn_raw_buffer_store_b128(pack_bf16x8(v), ro2, chunk_off(n, pA2, pB2, a.e.ldo2) * 2u, 0, 0);
        }
      } else {  // EPI_DGELU: out = acc * aux
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.e.aux), 0, (int)min((long)a.I * a.e.ld_aux * 2, 0x7FFFFFF0L), 0x00020000);
        const unsigned xA = (unsigned)(rowA * a.e.ld_aux + colA), xB = (unsigned)(rowB * a.e.ld_aux + colB);
        // (requested here, five chunks at a time; the compiler's wait for them also retires the two stages in flight, which are older anyway)
#pragma unroll
        for (int h = 0; h < C::NCHUNK; h += 5) {
          u32x4 raw[5];
#pragma unroll
          for (int n = 0; n < 5; ++n) raw[n] = __builtin_amdgcn_raw_buffer_load_b128(rx, chunk_off(h + n, xA, xB, a.e.ld_aux) * 2u, 0, V4H_SAVED_AUX);
#pragma unroll
          for (int n = 0; n < 5; ++n) {
            f32x8 v = chunk_val(h + n);
            const bf16x8 x = __builtin_bit_cast(bf16x8, raw[n]);
#pragma unroll
            for (int r = 0; r < 8; ++r) v.v[r] *= (float)x[r];
            __builtin_amdgcn_raw_buffer_store_b128(pack_bf16x8(v), ro, chunk_off(h + n, oA, oB, a.e.ldo) * 2u, 0, 0);
          }
        }
      }
    } else {  // EPI_SLAB_F32: split-K partial of the wgrad form, f32
      float* slab = reinterpret_cast<float*>(a.e.out) + (size_t)tz * a.e.slab_stride;
      const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(slab, 0, (int)min((long)a.I * a.e.ldo * 4, 0x7FFFFFF0L), 0x00020000);
      const unsigned oA = (unsigned)(rowA * a.e.ldo + colA), oB = (unsigned)(rowB * a.e.ldo + colB);
#pragma unroll
      for (int n = 0; n < C::NCHUNK; ++n) {
        const f32x8 v = chunk_val(n);
        const unsigned off = chunk_off(n, oA, oB, a.e.ldo) * 4u;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{v.v[0], v.v[1], v.v[2], v.v[3]}), ro, off, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{v.v[4], v.v[5], v.v[6], v.v[7]}), ro, off + 16u, 0, 0);
      }
    }
    return (C::DBG & 2) ? 1 : C::EPI_OPS;
  };

  // ------------------------------------------------------------------ ping-pong schedule (C::PP)
  // The waves w and w + 4 of a workgroup share a SIMD.  Per stage every wave runs a LOAD slot (both K = 32 fragment sets of the stage from LDS into
  // registers, its share of the DMA of the stage two ahead, at a tile seam the previous tile's epilogue) and a MATRIX slot (the 40 MFMAs on those
  // fragments), and the two waves of a SIMD run them in anti-phase: half 0 (waves 0..3) load(s), matrix(s), barrier; half 1 matrix(s - 1), load(s),
  // barrier.  The matrix pipe of the SIMD (almost) always has a wave with nothing but MFMAs to issue, and the slow instructions (a global->LDS DMA
  // instruction occupies the issuing wave for about 100 clocks) sit in the partner's shadow.  Ring discipline: stage s + 2 goes into the ring slot of
  // stage s - 1, which every wave finished reading before the barrier that ended stage s - 1; before the barrier that ends stage s a wave makes sure its
  // share of stage s + 1 has landed (counted vmcnt: everything it issued after that share may stay in flight).  One barrier per stage is all the data
  // needs; the first form of the schedule had one after every slot: it only pinned the phases, and cost 3-7 % (so did letting both halves write a
  // finished tile in the same slot of that form: neutral to -0.8 % in the step - both removed in round 3).
  if constexpr (C::PP) {
    static_assert(!(C::DBG & ~3), "the other ablation builds belong to the lock-step schedule");
    const int half = wave >> 2;
    auto slot_barrier = [&]() {
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_barrier" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    };
#ifdef V4H_GEMM2_STAMPS
    int st_n = 0;
#endif
    // Counted waits: yo = vector-memory instructions this wave issued after the share of the stage that has to land next, yn = after its newest
    // share.  (Measured and lost - fc1 forward 51.8 -> 54.5 us, dgrad fc1 38.5 -> 43.5: the Q part of a stage requested from inside the matrix
    // slot, between the two MFMA groups, so that every slot carries 4 + 2..3 DMA instructions per SIMD: a DMA instruction costs its wave about
    // 100 clocks, and in the matrix slot those are clocks in which the SIMD's only MFMA stream does not issue.)
    int yo = 0, yn = 0;
    {
      request();
      const int n1 = request();
      wait_vmcnt64(n1);
      slot_barrier();  // stage 0 is in LDS
    }
    int p_ti = 0, p_tj = 0, p_tz = 0;
    bool have_prev = false;
    for (;;) {
      const int kb = tz * a.klen, ke = min(a.K, kb + a.klen);
      const int nt = (ke - kb + C::BK - 1) >> 6;
      const int cs_period = a.ntj * C::WJ, cs_duty = tj * C::WJ + wj;
      int cs_phase = 0;
      int nv = cv + gridDim.x, n_ti = 0, n_tj = 0, n_tz = 0;
      const bool more = decode(nv, n_ti, n_tj, n_tz);
      for (int t = 0; t < nt; ++t) {
        // ---- load slot
        V4H_G2_STAMP(0);
        // (Measured and dropped: the fragment reads between the P and the Q part of the request, so that they issue while the memory path digests the
        //  first DMA instructions - K-contiguous operands +-1 %, K-strided ones 12-25 % slower (dgrad fc1 36.4 -> 41.9 us, wgrad fc2 50.5 -> 63.1); and the
        //  bias-gradient column sums accumulated in registers over the tile with one set of atomics per tile instead of one per duty slab - wgrad fc2
        //  50.5 -> 56.0 us, the others +-2 %.)
        read_frags(pA, qA, c_slot, std::integral_constant<int, 0>{});
        read_frags(pB, qB, c_slot, std::integral_constant<int, 1>{});
        auto p_part = [&]() {
          yo += issue();
          yn = 0;
        };
        if (C::EPI != EPI_DGELU) p_part();  // (the DGELU epilogue loads: its waits would also wait for a DMA issued in front of it)
        if (t == 0) {
          if (have_prev) {
            const int n = epilogue(p_ti, p_tj, p_tz);
            yo += n;
            yn += n;
          }
          init_acc(c_par);
        }
        if (C::EPI == EPI_DGELU) p_part();
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the fragments are in registers, this wave is done with the stage
        V4H_G2_STAMP(1);
        if (half == 1) {
          wait_vmcnt64(sgpr(yo));
          yo = yn;
        }
        V4H_G2_STAMP(2);
        if (half == 1) slot_barrier();
        __builtin_amdgcn_sched_barrier(0);
        V4H_G2_STAMP(3);
        // ---- matrix slot
        if constexpr (C::COLSUM) {
          if (a.colsum != nullptr && cs_phase == cs_duty) {
            colsum_slab(pA, ti * C::BI + wi * C::WTI);
            yo += C::TI;
            yn += C::TI;
          }
          if (++cs_phase == cs_period) cs_phase = 0;
          if (a.colsum != nullptr && cs_phase == cs_duty) {
            colsum_slab(pB, ti * C::BI + wi * C::WTI);
            yo += C::TI;
            yn += C::TI;
          }
          if (++cs_phase == cs_period) cs_phase = 0;
        }
        mfmas(pA, qA, 0, C::TI);
        if (advance_walk()) tile_setup();  // (ALU work only.  Weaving it between the MFMAs of the second group with sched_group_barrier changed nothing
                                           //  measurable and cost 45 registers.)
        mfmas(pB, qB, 0, C::TI);
        V4H_G2_STAMP(4);
        if (half == 0) {
          wait_vmcnt64(sgpr(yo));
          yo = yn;
        }
        V4H_G2_STAMP(5);
        if (half == 0) slot_barrier();
#ifdef V4H_GEMM2_STAMPS
        ++st_n;
#endif
        c_slot = c_slot == C::NS - 1 ? 0 : c_slot + 1;
      }
      p_ti = ti; p_tj = tj; p_tz = tz;
      have_prev = true;
      if (!more) break;
      cv = nv; ti = n_ti; tj = n_tj; tz = n_tz;
      c_par ^= 1;
    }
    epilogue(p_ti, p_tj, p_tz);
#ifdef V4H_GEMM2_STAMPS
    __builtin_amdgcn_s_waitcnt(0xC07F);
    v4h_gemm2_stamp_buf[(blockIdx.x * 8 + wave) * G2_ST_N * G2_ST_K + lane] = reinterpret_cast<unsigned*>(smem + C::LDS_BYTES)[wave * G2_ST_N * G2_ST_K + lane];
#endif
    return;
  }

#ifdef V4H_ABLATIONS
  // ------------------------------------------------------------------ lock-step schedule (ablation builds only)
  // prologue: three stages in flight, wait for the first
  int last_cnt, e1 = 0, e2 = 0;
  {
    request();
    const int n1 = request();
    const int n2 = request();
    wait_vmcnt64(n1 + n2);
    asm volatile("s_barrier" ::: "memory");
    last_cnt = n2;
  }
  read_frags(pA, qA, 0, std::integral_constant<int, 0>{});
  if (C::DBG & 4) read_frags(pB, qB, 0, std::integral_constant<int, 1>{});
  init_acc(0);

  for (;;) {  // tiles of this workgroup
    const int kb = tz * a.klen, ke = min(a.K, kb + a.klen);
    const int nt = (ke - kb + C::BK - 1) >> 6;
    const int cs_period = a.ntj * C::WJ, cs_duty = tj * C::WJ + wj;
    int cs_phase = 0;
    int nv = cv + gridDim.x, n_ti = 0, n_tj = 0, n_tz = 0;
    const bool more = decode(nv, n_ti, n_tj, n_tz);  // another tile after this one?

    // Both K = 32 slabs of every stage are computed (a K tail is zero-filled by the staging): no branch stands between a fragment
    // read and the MFMAs before it, so the compiler can count lgkmcnt instead of draining it.
    for (int t = 0; t < nt; ++t) {
      // ---- phase A: MFMAs of slab 2t, fragments of slab 2t + 1 on their way
      if (!(C::DBG & 4)) read_frags(pB, qB, c_slot, std::integral_constant<int, 1>{});
      if constexpr (C::COLSUM) {
        if (a.colsum != nullptr && cs_phase == cs_duty) {
          colsum_slab(pA, ti * C::BI + wi * C::WTI);
          e1 += C::TI;
        }
        if (++cs_phase == cs_period) cs_phase = 0;
      }
      mfmas(pA, qA, 0, C::TI);
      __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave has read everything it needs from the current stage
      // ---- barrier in the middle of the K-step: the next stage has landed for every wave, the current one is free
      if (!(C::DBG & 8)) {
        wait_vmcnt64(sgpr(e2 + last_cnt + e1));
        asm volatile("s_barrier" ::: "memory");
      }
      e2 = e1;
      e1 = 0;
      // ---- phase B: MFMAs of slab 2t + 1 (the first half of them BEFORE the scalar-heavy request code, so the matrix pipe has work
      // while both waves of a SIMD issue their DMA and fragment reads), first fragments of the next stage (possibly of the next tile)
      mfmas(pB, qB, 0, C::TI / 2);
      last_cnt = request();  // into the slot just freed, two K-steps ahead of its use
      const int n_slot = c_slot == C::NS - 1 ? 0 : c_slot + 1;
      if (!(C::DBG & 4)) read_frags(pA, qA, n_slot, std::integral_constant<int, 0>{});
      if constexpr (C::COLSUM) {
        if (a.colsum != nullptr && cs_phase == cs_duty) {
          colsum_slab(pB, ti * C::BI + wi * C::WTI);
          e1 += C::TI;
        }
        if (++cs_phase == cs_period) cs_phase = 0;
      }
      mfmas(pB, qB, C::TI / 2, C::TI);
      c_slot = n_slot;
    }

    e1 += epilogue(ti, tj, tz);
    if (!more) break;
    cv = nv; ti = n_ti; tj = n_tj; tz = n_tz;
    c_par ^= 1;
    init_acc(c_par);
  }
#endif  // V4H_ABLATIONS
}

template <class C> int v4h_gemm2_launch(GemmArgs a, int splitk, hipStream_t stream, const char* name) {
  V4H_CHECK_ARG(a.I > 0 && a.J > 0 && a.K > 0, "%s: empty problem I=%d J=%d K=%d", name, a.I, a.J, a.K);
  V4H_CHECK_ARG(a.J % C::BJ == 0, "%s: J=%d must be a multiple of %d", name, a.J, C::BJ);
  V4H_CHECK_ARG(a.ldp % 8 == 0 && a.ldq % 8 == 0, "%s: operand row strides (%d,%d) must be whole 16-byte chunks", name, a.ldp, a.ldq);
  V4H_CHECK_ARG(C::PKS ? (a.I % 8 == 0) : (a.K % 8 == 0), "%s: P extent not a whole number of 16-byte chunks", name);
  V4H_CHECK_ARG(C::QKS ? (a.J % 8 == 0) : (a.K % 8 == 0), "%s: Q extent not a whole number of 16-byte chunks", name);
  V4H_CHECK_ARG(((uintptr_t)a.P % 16) == 0 && ((uintptr_t)a.Q % 16) == 0, "%s: operands must be 16-byte aligned", name);
  V4H_CHECK_ARG(((uintptr_t)a.e.out % 16) == 0 && a.e.ldo % 8 == 0, "%s: output must be 16-byte aligned with a row stride of whole chunks", name);
  V4H_CHECK_ARG((long)a.I * a.e.ldo * 4 < 0x7FFFFFF0L, "%s: output too large for 32-bit buffer offsets", name);
  if (splitk < 1) splitk = 1;
  if (C::EPI != EPI_SLAB_F32) splitk = 1;
  int klen = (a.K + splitk - 1) / splitk;
  klen = (klen + C::BK - 1) / C::BK * C::BK;
  V4H_CHECK_ARG(klen >= 3 * C::BK, "%s: K range per split (%d) shorter than the stage ring", name, klen);
  a.klen = klen;
  a.nz = (a.K + klen - 1) / klen;
  V4H_CHECK_ARG(a.K - (a.nz - 1) * klen >= 3 * C::BK || a.nz == 1, "%s: last K split shorter than the stage ring", name);
  V4H_CHECK_ARG(a.K >= 3 * C::BK - 32, "%s: K=%d shorter than the stage ring", name, a.K);
  a.nti = (a.I + C::BI - 1) / C::BI;
  a.ntj = (a.J + C::BJ - 1) / C::BJ;
  long nblocks = a.nz == 1 ? (a.nti < 8 ? (long)a.nti * a.ntj : (long)((a.nti + 7) / 8) * 8 * a.ntj) : (long)a.nti * a.ntj * a.nz;
  if (nblocks > v4h_compute_units()) nblocks = v4h_compute_units();  // one persistent workgroup per CU (minus the CUs left to a communication kernel)
  static DeviceOnce lds_attr;  // the attribute belongs to the function object of ONE device
  if (int rc = lds_attr.ensure([&]() -> hipError_t {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&v4h_gemm2_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES + 2048);
      }, name, "reserve the ring's LDS")) return rc;
#ifdef V4H_GEMM2_STAMPS
  V4H_LAUNCH(v4h_gemm2_kernel<C>, dim3((unsigned)nblocks), dim3(C::NT), C::LDS_BYTES + 2048, stream, a);
#else
  V4H_LAUNCH(v4h_gemm2_kernel<C>, dim3((unsigned)nblocks), dim3(C::NT), C::LDS_BYTES, stream, a);
#endif
  V4H_CHECK_LAUNCH(name);
  return V4H_OK;
}
