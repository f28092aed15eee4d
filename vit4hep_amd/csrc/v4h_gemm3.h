// Third-generation bf16 contraction: WEIGHT-STATIONARY, for the Linears whose contraction length is the hidden width (K = 480):
// qkv, attn.proj and mlp.fc1 (+ GELU) forward, the input gradients of attn.proj and of mlp.fc2 (x GELU')
// (reference nn/vit.py:416,420,425-454 and timm Mlp :312-322).  Same operand conventions as v4h_gemm.h:
//
//     Out[i][j] = sum_k P[i][k] * Q[j][k]      P = activations [tokens][K] (K-contiguous), Q = weight, K-contiguous [J][K] (forward) or K-strided [K][J] (dgrad)
//
// Why another kernel.  The ring kernel (v4h_gemm2.h) stages BOTH operands through the CU's global->LDS path: 53 KB per K = 64 step, which that path takes
// in about the 1280 clocks the matrix pipe needs for the step - and a K = 480 tile is only 7.5 steps long, so fill, drain and whole-tile rounds (qkv: 2.39
// rounds paid as 3; N = 480: 204 tiles on 256 CUs) are never amortised: 0.20-0.22 of peak where K = 1920 reaches 0.31.  With K = 480 a column slice of the
// weight is small enough to live in REGISTERS for the whole launch:
//
//   * a wave owns NT = 3 (or 2) column tiles of 16 and keeps their fragments for every K = 32 slab in VGPRs: 15 x 3 x 4 = 180 registers (120), loaded once
//     (forward: straight from global memory in fragment layout; dgrad: the K-strided weight goes through LDS once and is read transposed);
//     8 waves = one workgroup per CU = 384 (256) columns;
//   * only ACTIVATION rows stream: 16-row tiles of 15 KB through a 6-slot LDS ring filled by global->LDS DMA five tiles ahead - 15 KB per 16 x 384 x 480
//     MACs instead of 53 KB per 256 x 160 x 64, i.e. 390 instead of 98 FLOP per staged byte; every wave reads every A fragment (one ds_read_b128 feeds its 3 MFMAs);
//   * the grid is cut into EQUAL shares: workgroup = (column slice, contiguous range of 16-row tiles), ranges differing by at most one tile - no tile rounds;
//   * ping-pong without extra registers: the two waves of a SIMD run matrix slot (45 MFMAs + 15 fragment reads) and auxiliary slot (epilogue of the tile
//     just finished, DMA of the tile five ahead) in opposite order between two barriers, so a SIMD's matrix pipe nearly always has a wave with MFMAs to issue;
//   * A image [octet of chunks][16 rows][8 chunks, XOR-swizzled as in v4h_gemm.h]: a DMA piece is 8 rows x 128 contiguous bytes with the lanes of a quad on
//     consecutive chunks of one row, and the 16 lanes of every ds_read_b128 service group fall on 16 different 16-byte bank groups (tools/lds_model.py).
//
// Where it stands (round 5, profiles/r05_notes.md sections 2 and 7 with the slot timelines): the streaming part runs at 82 % of the matrix pipe and takes half
// the ring kernel's time per row, but no MFMA can start before the 368 KB (245 KB) of a workgroup's weight slice are in its registers - 9-10 us at the 23-26
// bytes / clock a CU gets out of the L2s when all 256 pull the same 1.4 MB (direct fragment loads and whole-line DMA through the LDS alike).  Default classes:
// every K = 480 contraction of the block except attn.proj's input gradient (csrc/v4h_gemm.hip: g_ws) - qkv 33-35 us instead of 47-48 cold; the GELU / DGELU
// forms joined when their auxiliary slots (in series with the matrix slot of the same wave, beside the partner's MFMAs) were cut from 3400 / 2200 to
// 2170 / 1150 clocks: +2.7 % on the update step, +2.7 % on the sampler over the qkv + proj forward classes alone.
#pragma once
#include "v4h_gemm2.h"

template <bool QKS_, int EPI_, int NT_, int KS_ = 15> struct Gemm3Cfg {
  static constexpr bool QKS = QKS_;
  static constexpr int EPI = EPI_, NT = NT_, KS = KS_, K = 32 * KS_;
  static constexpr int NW = 8, NTHR = 512, WJ = 16 * NT_, BJ = NW * WJ;  // columns per wave / per workgroup
  static constexpr int TILE_BYTES = 16 * K * 2;                          // LDS image of one 16-row tile
  static constexpr int NI = TILE_BYTES / 1024;                           // DMA instructions (1 KiB each) per tile = KS
  static constexpr int NREG = 2 * (KS / 2);                              // ... of which regular (8 rows x 8 chunks); KS odd: one more of 16 rows x 4 chunks
  static constexpr int NPW = (NI + NW - 1) / NW;                         // per wave, at most
  static constexpr int R = 6;                                            // ring slots (tiles)
  static constexpr int BIAS_OFF = R * TILE_BYTES;
  static constexpr int EPI_ST = NT == 3 ? 2 : 1;                         // 16-byte accesses per lane, tile and tensor
  // EPI_DGELU: the saved gelu' of a tile (16 rows x BJ columns) rides the ring too - every wave requests the 1 KiB it will multiply by with its DMA share of
  // the tile, five tiles ahead, lane-linear into a private strip (lane l's 16 bytes are the 8 columns lane l owns after the epilogue's tile exchange), and reads
  // it back with one ds_read_b128.  R + 1 strips: half 1 writes tile t - 1 in interval t, behind the request that reuses the strip of tile t - 2.
  static constexpr bool AUX_DMA = EPI == EPI_DGELU;
  static constexpr int AUX_SLOTS = R + 1, AUX_WAVE = 1024 * EPI_ST, AUX_N = AUX_DMA ? EPI_ST : 0;
  static constexpr int AUX_OFF = BIAS_OFF + BJ * 4;
  static constexpr int LDS_BYTES = AUX_OFF + (AUX_DMA ? AUX_SLOTS * NW * AUX_WAVE : 0);
  using ImgQ = ImgKStrided<bf16, BJ, 32, NW>;                             // dgrad prologue: one K = 32 slab of the weight slice, [32][BJ]
  static constexpr int Q_ROUND = (R * TILE_BYTES) / ImgQ::BYTES;         // slabs staged per prologue round
  static constexpr int EPI_OPS = EPI == EPI_GELU ? 2 * EPI_ST : EPI_ST; // vector-memory instructions of one epilogue (GELU: two outputs)
  static_assert(EPI == EPI_STORE || EPI == EPI_GELU || EPI == EPI_DGELU, "epilogue not built for the weight-stationary kernel");
  static_assert(NT == 2 || NT == 3, "column tiles per wave");
  static_assert(NI == KS && LDS_BYTES <= 160 * 1024 && Q_ROUND >= 1, "ring shape");
};

// One fragment read of the tile image (slab S) and the chain of matrix steps, as templates: every LDS offset, wait count and register index is a constant.
template <int KS, int S> V4H_DEV void g3_read(Frag<bf16>& f, unsigned a0, unsigned a0h, unsigned a1) {
  if constexpr ((KS & 1) && S == KS - 1) asm volatile("ds_read_b128 %0, %1" : "=v"(f.v) : "v"(a1));
  else if constexpr (S & 1) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f.v) : "v"(a0h), "n"((S >> 1) * 2048));
  else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f.v) : "v"(a0), "n"((S >> 1) * 2048));
}
template <int KS, int NT, int S> struct G3Step {
  static V4H_DEV void run(f32x4 (&acc)[NT], const Frag<bf16> (&bq)[KS][NT], Frag<bf16> (&p)[3], unsigned a0, unsigned a0h, unsigned a1) {
    if constexpr (S + 2 < KS) g3_read<KS, S + 2>(p[(S + 2) % 3], a0, a0h, a1);
    constexpr int younger = (KS - 1 - S) < 2 ? (KS - 1 - S) : 2;  // reads issued behind the one this step needs
    if constexpr (S == 0) {  // the bias reads (into the accumulators) are older still: the same wait covers them
      if constexpr (NT == 3) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(p[0].v), "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]) : "n"(younger));
      else asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(p[0].v), "+v"(acc[0]), "+v"(acc[1]) : "n"(younger));
    } else {
      asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(p[S % 3].v) : "n"(younger));
    }
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) acc[ct] = mma(bq[S][ct], p[S % 3], acc[ct]);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (S + 1 < KS) G3Step<KS, NT, S + 1>::run(acc, bq, p, a0, a0h, a1);
  }
};

#ifdef V4H_GEMM3_STAMPS
// Diagnostic build only (V4H_BUILD_TAG=st3 V4H_EXTRA_FLAGS=-DV4H_GEMM3_STAMPS): every wave stamps the shader clock at the boundaries of its slots - into 4 KB of
// LDS behind the ring (no vector-memory instruction, the counted waits are untouched) - and copies them out at the end (tools/experiments/gemm3_stamps.py).
constexpr int G3_ST_N = 128;
__device__ unsigned v4h_gemm3_stamp_buf[256 * 8 * G3_ST_N];
#define V4H_G3_STAMP()                                                                                                  \
  do {                                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                                  \
    unsigned long long now_;                                                                                            \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");                                        \
    if (lane == 0 && st_n < G3_ST_N) reinterpret_cast<unsigned*>(smem + C::LDS_BYTES)[wave * G3_ST_N + st_n] = (unsigned)now_; \
    ++st_n;                                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                                  \
  } while (0)
#else
#define V4H_G3_STAMP() do { } while (0)
#endif

// ncs column slices x nrg row groups; wpx = shares per XCD; rcp_ncs = ceil(2^32 / ncs) (share / ncs without a division sequence); row group rg walks
// tiles [rg * tq + min(rg, tr), ...) with tq = tiles / nrg, tr = tiles % nrg: ranges that differ by at most one tile.
template <class C> __global__ __launch_bounds__(C::NTHR, 2) void v4h_gemm3_kernel(const GemmArgs a, int ncs, int nrg, int wpx, unsigned rcp_ncs, int tq, int tr) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  __builtin_assume(wave >= 0 && wave < C::NW);
  const int half = wave >> 2;
#ifdef V4H_GEMM3_STAMPS
  int st_n = 0;
#endif
  V4H_G3_STAMP();  // 0: start
  auto sgpr = [](int x) { return __builtin_amdgcn_readfirstlane(x); };
  const bf16* gP = reinterpret_cast<const bf16*>(a.P);
  const bf16* gQ = reinterpret_cast<const bf16*>(a.Q);

  // workgroup -> (column slice, row group).  Blocks are dealt round-robin over the 8 XCDs: block b runs share (b % 8) * wpx + b / 8 of the row-group-major
  // list, so the workgroups of one XCD are consecutive row groups with ALL their column slices - an activation tile is fetched into that L2 once (speed only).
  const int share = (int)(blockIdx.x & 7) * wpx + (int)(blockIdx.x >> 3);
  if (share >= ncs * nrg) return;
  const int rg = sgpr((int)__umulhi((unsigned)share, rcp_ncs)), cs = share - rg * ncs;
  const int t_begin = rg * tq + min(rg, tr), t_end = t_begin + tq + (rg < tr ? 1 : 0);
  const int jw0 = cs * C::BJ + wave * C::WJ;  // this wave's first column
  const bool active = jw0 < a.J;              // (wave-uniform; J is a whole number of wave slices)
  const int c = lane & 15, g = lane >> 4;

  // ------------------------------------------------------------------ A ring: request side
  int t_issue = t_begin, q_issue = 0, qa_issue = 0, qa_read = 0;
  // Lane part of every DMA address, once: (row of the tile, 16-byte chunk) of this lane in DMA instruction k of this wave, as a byte offset from the tile's
  // first row - the request of a tile is then a scalar base (64-bit, scalar unit) plus a 32-bit lane offset per instruction instead of a 64-bit multiply
  // chain per instruction and tile (three quarter-rate instructions each, in the auxiliary slot that the partner's matrix slot has to share the issue port
  // with).  A lane whose row lies beyond the operand's last row takes the tile's first row instead (a select between two offsets, no multiply).
  // (consecutive lanes read consecutive 16-byte chunks of ONE row: the address unit coalesces a quad of lanes into one 64-byte access.  The first
  //  form of the image had the rows running fastest - every lane of a quad in another row, i.e. four lookups per quad.)
  int dma_r[C::NPW];
  unsigned dma_off[C::NPW], dma_off0[C::NPW];
#pragma unroll
  for (int k = 0; k < C::NPW; ++k) {
    const int inst = wave + k * C::NW;  // (scalar)
    int r, chunk;
    if (inst < C::NREG) {
      r = (inst & 1) * 8 + (lane >> 3);  // row of the tile; the image keeps its 8 chunks of an octet at position kc ^ (r & 6)
      chunk = (inst >> 1) * 8 + ((lane & 7) ^ (r & 6));
    } else {
      r = lane >> 2;                     // last half octet (KS odd): 16 rows x 4 chunks at position kc ^ ((r & 4) >> 1)
      chunk = (C::KS / 2) * 8 + ((lane & 3) ^ ((r & 4) >> 1));
    }
    dma_r[k] = r;
    dma_off[k] = (unsigned)(r * a.ldp + chunk * 8) * 2u;
    dma_off0[k] = (unsigned)(chunk * 8) * 2u;
  }
  const unsigned aux_off0 = C::AUX_DMA ? (unsigned)(jw0 + (g & 1) * 16 + (g >> 1) * 8) * 2u : 0u;
  const unsigned aux_off = C::AUX_DMA ? aux_off0 + (unsigned)(c * a.e.ld_aux) * 2u : 0u;
  auto issue = [&]() -> int {  // this wave's share of the DMA of tile t_issue (into ring slot q_issue); returns the number of instructions issued
    int n = 0;
    if (t_issue < t_end) {
      const int row0 = t_issue * 16;
      char* dst = smem + q_issue * C::TILE_BYTES;
      const char* tbase = reinterpret_cast<const char*>(gP) + (size_t)row0 * a.ldp * 2;
#pragma unroll
      for (int k = 0; k < C::NPW; ++k) {
        const int inst = wave + k * C::NW;  // (scalar)
        if (inst < C::NI) {
          // rows beyond the operand: a valid row (their products only reach rows the buffer stores drop)
          dma16(tbase + (row0 + dma_r[k] < a.I ? dma_off[k] : dma_off0[k]), dst + inst * 1024);
          ++n;
        }
      }
      if constexpr (C::AUX_DMA) {
        if (active) {  // the wave's own 16 x WJ piece of the saved derivative: lane (c, g) fetches the 8 columns it will hold after the tile exchange
          const char* abase = reinterpret_cast<const char*>(a.e.aux) + (size_t)row0 * a.e.ld_aux * 2;
          const char* ap = abase + (row0 + c < a.I ? aux_off : aux_off0);
          char* adst = smem + C::AUX_OFF + (qa_issue * C::NW + wave) * C::AUX_WAVE;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ap, (__attribute__((address_space(3))) void*)adst, 16, 0, V4H_SAVED_AUX);
          ++n;
          if constexpr (C::NT == 3) {  // third column tile: the lanes with even g hold its halves (the others fetch a valid address and never read it)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ap + (32 - (g & 1) * 16) * 2), (__attribute__((address_space(3))) void*)(adst + 1024), 16, 0, V4H_SAVED_AUX);
            ++n;
          }
        }
      }
    }
    ++t_issue;
    q_issue = q_issue == C::R - 1 ? 0 : q_issue + 1;
    qa_issue = qa_issue == C::AUX_SLOTS - 1 ? 0 : qa_issue + 1;
    return sgpr(n);
  };

  // ------------------------------------------------------------------ weight slice -> registers (once)
  Frag<bf16> bq[C::KS][C::NT];
  // bias slice of the workgroup -> LDS (accumulators start from it): requested first, stored behind everything else the prologue requests, so that
  // nothing waits for its round trip
  float bias_v = 0.0f;
  {
    const int j = cs * C::BJ + tid;
    if (tid < C::BJ && a.e.bias != nullptr && j < a.J) bias_v = a.e.bias[j];
  }
  int ops = 0;            // vector-memory instructions this wave has issued since the ring started (exact: every access below is unconditional)
  int mk[C::R - 1];       // mk[k] = value of `ops` right after this wave's DMA share of tile (current + 1 + k)
  int mark_first = 0;
  if constexpr (C::QKS) {
    // K-strided weight W[k][j]: slabs of 32 k rows through LDS (the ring's memory, before the ring starts), fragments by transposed reads
    typename C::ImgQ st;
#pragma unroll
    for (int s0 = 0; s0 < C::KS; s0 += C::Q_ROUND) {
#pragma unroll
      for (int k = 0; k < C::Q_ROUND; ++k)
        if (s0 + k < C::KS) {
          st.init(gQ, a.ldq, cs * C::BJ, (s0 + k) * 32, a.J, wave, lane);
          st.stage(smem + k * C::ImgQ::BYTES, (s0 + k) * 32, C::K, a.ldq, wave);
        }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < C::Q_ROUND; ++k)
        if (s0 + k < C::KS) {
#pragma unroll
          for (int ct = 0; ct < C::NT; ++ct) bq[s0 + k][ct] = C::ImgQ::frag(smem + k * C::ImgQ::BYTES, wave * C::WJ + ct * 16, 0, lane);
        }
      __syncthreads();
    }
  }
  V4H_G3_STAMP();  // 1: (dgrad form: weight fragments in registers)
  // ring prologue: R - 1 tiles in flight (behind the bias store / the weight prologue, whose barriers are done)
  {
    issue();
    mark_first = 0;  // (ops counted from here: the first tile's share is the oldest entry)
#pragma unroll
    for (int k = 0; k < C::R - 2; ++k) {
      ops += issue();
      mk[k] = ops;
    }
    mk[C::R - 2] = ops;
  }
  if constexpr (!C::QKS) {
    // K-contiguous weight W[j][k]: a fragment is 16 bytes per lane at W[j0 + c][32 s + 8 g ...] - loaded in place, behind the ring's first requests
    const bf16* qrow = gQ + (size_t)(active ? jw0 + c : 0) * a.ldq + 8 * g;
    // (column tile by column tile, slabs in order: consecutive requests of a wave touch the two 64-byte halves of the same 128-byte lines.  Odd row groups
    //  walk the slabs downwards: two request fronts over the weight instead of one)
    if (rg & 1) {
#pragma unroll
      for (int ct = 0; ct < C::NT; ++ct)
#pragma unroll
        for (int s = C::KS - 1; s >= 0; --s) bq[s][ct].v = *reinterpret_cast<const bf16x8*>(qrow + (size_t)ct * 16 * a.ldq + s * 32);
    } else {
#pragma unroll
      for (int ct = 0; ct < C::NT; ++ct)
#pragma unroll
        for (int s = 0; s < C::KS; ++s) bq[s][ct].v = *reinterpret_cast<const bf16x8*>(qrow + (size_t)ct * 16 * a.ldq + s * 32);
    }
    __builtin_amdgcn_sched_barrier(0);
  }

  if (tid < C::BJ) reinterpret_cast<float*>(smem + C::BIAS_OFF)[tid] = bias_v;
  V4H_G3_STAMP();  // 2: ring requested, weight loads issued
#ifdef V4H_GEMM3_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  V4H_G3_STAMP();  // 3: (stamped build only) everything requested so far has arrived
  // ------------------------------------------------------------------ matrix slot
  f32x4 acc[C::NT];
  // tile image: per octet of chunks (two K = 32 slabs) a 16-row x 8-chunk block in the layout of v4h_gemm.h's K-contiguous images (chunk kc of row r at
  // position kc ^ (r & 6): conflict-free ds_read_b128 by tools/lds_model.py), the last half octet (KS odd) as 16 rows x 4 chunks at kc ^ ((r & 4) >> 1)
  const int fb0 = (c * 8 + (g ^ (c & 6))) * 16;                          // lane's byte offset of the even slab of octet 0
  const int fb0h = (c * 8 + ((4 + g) ^ (c & 6))) * 16;                   // ... of the odd slab
  const int fb1 = ((C::KS / 2) * 128 + c * 4 + (g ^ ((c & 4) >> 1))) * 16;  // ... of the last slab when KS is odd
  // 45 (30) MFMAs on fragments requested two slabs ahead, three fragment buffers rotating.  The LDS reads and their waits are written in assembly: with
  // plain loads the compiler's waitcnt pass drains lgkmcnt(0) every third slab - right behind a read it has just issued - and the slot takes 1460 clocks for
  // 720 clocks of matrix pipe (its own first order - two reads, a full wait, six MFMAs - 1610; profiles/r05_notes.md).  Here every wait is counted: before
  // the MFMAs of slab s only the reads up to slab s have to be back (LDS returns in order), the two younger ones stay in flight.  The accumulators are the
  // destination of the bias reads, which are the oldest of the slot.
  const unsigned lds0 = (unsigned)(uintptr_t)((V4H_LDS char*)smem);
  const unsigned bias_addr = lds0 + C::BIAS_OFF + (wave * C::WJ + 4 * g) * 4;
  auto matrix = [&](int slot) {
    const unsigned a0 = lds0 + slot * C::TILE_BYTES + fb0, a0h = lds0 + slot * C::TILE_BYTES + fb0h, a1 = lds0 + slot * C::TILE_BYTES + fb1;
    Frag<bf16> p[3];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ct = 0; ct < C::NT; ++ct) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(acc[ct]) : "v"(bias_addr), "n"(ct * 64));
    g3_read<C::KS, 0>(p[0], a0, a0h, a1);
    g3_read<C::KS, 1>(p[1], a0, a0h, a1);
    G3Step<C::KS, C::NT, 0>::run(acc, bq, p, a0, a0h, a1);
  };

  // ------------------------------------------------------------------ epilogue: registers -> memory (buffer accesses, rows beyond I dropped by the bounds check)
  // Pair (column tiles 0, 1): v_permlane16_swap gives every lane 8 consecutive columns - even g: tile 0, odd g: tile 1 -, one 16-byte access per lane.
  // Tile 2 (NT = 3): the same swap against itself, the lanes with even g hold its 8-column halves, the others point outside the buffer.
  constexpr unsigned OOB = 0x7FFFFFF0u;
  // (Round 5, first form of EPI_DGELU: the saved derivative by buffer loads in front of the DMA share.  The compiler's wait for them was vmcnt(0) - the share
  //  is requested under a condition, so the only count that is right on every path is zero - and drained the whole ring every interval: 3750-3980 clocks
  //  per interval against 2540 of the plain store.  Now the derivative arrives through the ring: Gemm3Cfg::AUX_DMA.)
  //  The strip is read at the head of the auxiliary slot and waited for behind the DMA requests: its LDS latency is not on the slot's critical path.
  u32x4 aux01, aux2;
  auto epi_pre = [&](int) -> int {
    if constexpr (C::AUX_DMA) {
      if (!active) return 0;
      const unsigned aaddr = lds0 + C::AUX_OFF + (qa_read * C::NW + wave) * C::AUX_WAVE + lane * 16;
      qa_read = qa_read == C::AUX_SLOTS - 1 ? 0 : qa_read + 1;
      if constexpr (C::NT == 3) asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024" : "=&v"(aux01), "=&v"(aux2) : "v"(aaddr));
      else asm volatile("ds_read_b128 %0, %1" : "=&v"(aux01) : "v"(aaddr));
    }
    return 0;
  };
  auto epi_post = [&](int t) -> int {
    if (!active) return 0;
    const int row = t * 16 + c;
    const int col01 = jw0 + (g & 1) * 16 + (g >> 1) * 8, col2 = jw0 + 32 + (g >> 1) * 8;
    if constexpr (C::EPI == EPI_STORE) {
      const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(a.e.out, 0, (int)min((long)a.I * a.e.ldo * 2, 0x7FFFFFF0L), 0x00020000);
      __builtin_amdgcn_raw_buffer_store_b128(swap_pair_bf16(acc[0], acc[1]), ro, (unsigned)(row * a.e.ldo + col01) * 2u, 0, 0);
      if constexpr (C::NT == 3)
        __builtin_amdgcn_raw_buffer_store_b128(swap_pair_bf16(acc[2], acc[2]), ro, (g & 1) ? OOB : (unsigned)(row * a.e.ldo + col2) * 2u, 0, 0);
      return C::EPI_ST;
    } else if constexpr (C::EPI == EPI_GELU) {  // out = gelu'(pre) (training only), out2 = gelu(pre)
      const bool train = a.e.out != nullptr;
      const __amdgpu_buffer_rsrc_t ro2 = __builtin_amdgcn_make_buffer_rsrc(a.e.out2, 0, (int)min((long)a.I * a.e.ldo2 * 2, 0x7FFFFFF0L), 0x00020000);
      const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(a.e.out, 0, train ? (int)min((long)a.I * a.e.ldo * 2, 0x7FFFFFF0L) : 0, 0x00020000);
      f32x4 y[C::NT], d[C::NT];
      {  // column tiles 0 and 1 on the packed form (v4h_common.h)
        f32x8 yv, dv;
#pragma unroll
        for (int r = 0; r < 4; ++r) { yv.v[r] = acc[0][r]; yv.v[4 + r] = acc[1][r]; dv.v[r] = 0.f; dv.v[4 + r] = 0.f; }
        if (train) gelu8_and_grad<bf16>(yv, dv);
        else gelu8_only<bf16>(yv);
#pragma unroll
        for (int r = 0; r < 4; ++r) { y[0][r] = yv.v[r]; y[1][r] = yv.v[4 + r]; d[0][r] = dv.v[r]; d[1][r] = dv.v[4 + r]; }
      }
#pragma unroll
      for (int ct = 2; ct < C::NT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (train) {
            float yy, dd;
            gelu_and_grad<bf16>(acc[ct][r], yy, dd);
            y[ct][r] = yy;
            d[ct][r] = dd;
          } else {
            y[ct][r] = gelu_only<bf16>(acc[ct][r]);
            d[ct][r] = 0.f;
          }
        }
      __builtin_amdgcn_raw_buffer_store_b128(swap_pair_bf16(d[0], d[1]), rd, (unsigned)(row * a.e.ldo + col01) * 2u, 0, V4H_SAVED_AUX);  // (inference: zero-sized buffer, dropped)
      __builtin_amdgcn_raw_buffer_store_b128(swap_pair_bf16(y[0], y[1]), ro2, (unsigned)(row * a.e.ldo2 + col01) * 2u, 0, 0);
      if constexpr (C::NT == 3) {
        __builtin_amdgcn_raw_buffer_store_b128(swap_pair_bf16(d[2], d[2]), rd, (g & 1) ? OOB : (unsigned)(row * a.e.ldo + col2) * 2u, 0, V4H_SAVED_AUX);
        __builtin_amdgcn_raw_buffer_store_b128(swap_pair_bf16(y[2], y[2]), ro2, (g & 1) ? OOB : (unsigned)(row * a.e.ldo2 + col2) * 2u, 0, 0);
      }
      return C::EPI_OPS;
    } else {  // EPI_DGELU: out = acc * aux
      const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(a.e.out, 0, (int)min((long)a.I * a.e.ldo * 2, 0x7FFFFFF0L), 0x00020000);
      if constexpr (C::NT == 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(aux01), "+v"(aux2));
      else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(aux01));
      {
        f32x8 v = swap_pair(acc[0], acc[1]);
        const bf16x8 x = __builtin_bit_cast(bf16x8, aux01);
#pragma unroll
        for (int r = 0; r < 8; ++r) v.v[r] *= (float)x[r];
        __builtin_amdgcn_raw_buffer_store_b128(pack_bf16x8(v), ro, (unsigned)(row * a.e.ldo + col01) * 2u, 0, 0);
      }
      if constexpr (C::NT == 3) {
        f32x8 v = swap_pair(acc[2], acc[2]);
        const bf16x8 x = __builtin_bit_cast(bf16x8, aux2);
#pragma unroll
        for (int r = 0; r < 8; ++r) v.v[r] *= (float)x[r];
        __builtin_amdgcn_raw_buffer_store_b128(pack_bf16x8(v), ro, (g & 1) ? OOB : (unsigned)(row * a.e.ldo + col2) * 2u, 0, 0);
      }
      return C::EPI_ST;
    }
  };

  // ------------------------------------------------------------------ the walk over this workgroup's tiles
  auto slot_barrier = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  // Counted wait.  In the steady state the count is one of two constants - (R - 1) epilogues + (R - 2) DMA shares of 1 or 2 instructions - and takes an
  // immediate; anything else (head and tail of the walk, waves without columns) goes through the computed jump (176 clocks per call, measured).
  auto wait_counted = [&](int n) {
    constexpr int S2 = (C::R - 1) * C::EPI_OPS + (C::R - 2) * (2 + C::AUX_N), S1 = (C::R - 1) * C::EPI_OPS + (C::R - 2) * (1 + C::AUX_N);
    static_assert(S2 <= 63, "vmcnt immediate");
    if (n == S2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(S2) : "memory");
    else if (n == S1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(S1) : "memory");
    else wait_vmcnt64(n);
  };
  // tile t_begin has landed for every wave (this wave: everything older than its newer shares), bias and weight fragments are in place
  wait_vmcnt64(sgpr(ops - mark_first));
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the bias slice is written (the raw barrier below does not wait for LDS stores)
  slot_barrier();
  V4H_G3_STAMP();  // 4: first barrier passed
  int slot = 0;
  for (int t = t_begin; t < t_end; ++t) {
    // Interval t (between two barriers): every wave computes tile t from ring slot `slot`, requests its share of tile t + R - 1 into the slot tile t - 1
    // left at the last barrier, and writes one finished tile: half 0 the one it has just computed (matrix slot first), half 1 the previous one (auxiliary
    // slot first, its accumulators survive the barrier) - the two waves of a SIMD run the two slots in opposite order.
    if (half == 0) {
      if (active) matrix(slot);
      V4H_G3_STAMP();  // 5 + 4 i: first slot done
      ops += epi_pre(t);
      ops += issue();
      mk[C::R - 2] = ops;
      ops += epi_post(t);
    } else {
      // The SIMD's issue arbiter serves its older wave first - the half-0 wave - and a half-1 wave in its auxiliary slot (dependent vector instructions,
      // first beside the partner's MFMAs, then beside the partner's own auxiliary slot) fell behind: 2170 clocks for the GELU form's slot against 890 for the
      // same work in half 0, and half 0 then waited 1300 clocks at the barrier.  Raised priority for exactly this slot: 1570 / 600, interval 3400 -> 2920
      // (sampler +1.9 %, update step +0.3 %; the same priority for both halves' auxiliary slots, or for half 1 throughout: neutral / -0.4 %).
      __builtin_amdgcn_s_setprio(1);
      if (t > t_begin) ops += epi_pre(t - 1);
      ops += issue();
      mk[C::R - 2] = ops;
      if (t > t_begin) ops += epi_post(t - 1);
      __builtin_amdgcn_s_setprio(0);
      V4H_G3_STAMP();
      if (active) matrix(slot);
    }
    V4H_G3_STAMP();  // 6 + 4 i: second slot done
    // before the barrier: this wave's share of tile t + 1 has landed (everything it issued afterwards may stay in flight)
    wait_counted(sgpr(ops - mk[0]));
    V4H_G3_STAMP();  // 7 + 4 i: counted wait done
#pragma unroll
    for (int k = 0; k < C::R - 2; ++k) mk[k] = mk[k + 1];
    slot_barrier();
    V4H_G3_STAMP();  // 8 + 4 i: barrier passed
    slot = slot == C::R - 1 ? 0 : slot + 1;
  }
  if (half == 1 && t_end > t_begin) {
    epi_pre(t_end - 1);
    epi_post(t_end - 1);
  }
#ifdef V4H_GEMM3_STAMPS
  __builtin_amdgcn_s_waitcnt(0xC07F);
  for (int k = lane; k < G3_ST_N; k += 64)
    v4h_gemm3_stamp_buf[((blockIdx.x & 255) * 8 + wave) * G3_ST_N + k] = reinterpret_cast<unsigned*>(smem + C::LDS_BYTES)[wave * G3_ST_N + k];
#endif
}

// Column tiles per wave for a J-wide output: the shape that wastes the fewest wave slices of the last column slice (3 tiles = 48 columns per wave where
// that is at least as full - fewer LDS fragment reads per MFMA).
inline int v4h_gemm3_pick_nt(int J) {
  auto waste = [&](int wj) {
    const int ws = (J + wj - 1) / wj, slices = (ws + 7) / 8;  // wave slices, column slices of 8 waves
    return (double)(slices * 8 - ws) / (slices * 8);
  };
  if (J % 48 != 0) return J % 32 == 0 ? 2 : 0;
  if (J % 32 != 0) return 3;
  return waste(48) <= waste(32) + 1e-9 ? 3 : 2;
}
inline bool v4h_gemm3_eligible(const GemmArgs& a, int K) {
  return a.K == K && a.I >= 2048 && v4h_gemm3_pick_nt(a.J) != 0 && a.ldp % 8 == 0 && a.ldq % 8 == 0 && a.e.ldo % 8 == 0 && ((uintptr_t)a.P % 16) == 0 &&
         ((uintptr_t)a.Q % 16) == 0 && ((uintptr_t)a.e.out % 16) == 0 && (long)a.I * a.e.ldo * 2 < 0x7FFFFFF0L && (long)a.I * a.ldp * 2 < (1L << 40);
}

template <class C> int v4h_gemm3_launch(const GemmArgs& a, hipStream_t stream, const char* name) {
  V4H_CHECK_ARG(a.I > 0 && a.J > 0 && a.K == C::K, "%s: K=%d is not the %d this kernel keeps in registers (I=%d J=%d)", name, a.K, C::K, a.I, a.J);
  V4H_CHECK_ARG(a.J % C::WJ == 0, "%s: J=%d must be a multiple of %d", name, a.J, C::WJ);
  V4H_CHECK_ARG(a.ldp % 8 == 0 && a.ldq % 8 == 0 && ((uintptr_t)a.P % 16) == 0 && ((uintptr_t)a.Q % 16) == 0, "%s: operands must be 16-byte aligned with row strides of whole chunks", name);
  V4H_CHECK_ARG(((uintptr_t)a.e.out % 16) == 0 && a.e.ldo % 8 == 0 && (long)a.I * a.e.ldo * 2 < 0x7FFFFFF0L, "%s: output must be 16-byte aligned, row stride of whole chunks, below 2 GB", name);
  if (C::EPI == EPI_GELU) V4H_CHECK_ARG(a.e.out2 != nullptr && ((uintptr_t)a.e.out2 % 16) == 0 && a.e.ldo2 % 8 == 0 && (long)a.I * a.e.ldo2 * 2 < 0x7FFFFFF0L, "%s: second output", name);
  if (C::EPI == EPI_DGELU) V4H_CHECK_ARG(a.e.aux != nullptr && ((uintptr_t)a.e.aux % 16) == 0 && a.e.ld_aux % 8 == 0 && (long)a.I * a.e.ld_aux * 2 < 0x7FFFFFF0L, "%s: auxiliary operand", name);
  const int ncs = (a.J + C::BJ - 1) / C::BJ, nrt = (a.I + 15) / 16;
  int nrg = v4h_compute_units() / ncs;
  if (nrg > nrt) nrg = nrt;
  V4H_CHECK_ARG(nrg >= 1, "%s: %d column slices do not fit the %d compute units", name, ncs, v4h_compute_units());
  const int wpx = (ncs * nrg + 7) / 8;
  const unsigned rcp_ncs = (unsigned)((0x100000000ULL + ncs - 1) / ncs);  // exact quotient for share < 2^16
  const int tq = nrt / nrg, tr = nrt % nrg;
  static DeviceOnce lds_attr;
  if (int rc = lds_attr.ensure([&]() -> hipError_t {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&v4h_gemm3_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES + 4096);
      }, name, "reserve the ring's LDS")) return rc;
#ifdef V4H_GEMM3_STAMPS
  V4H_LAUNCH(v4h_gemm3_kernel<C>, dim3((unsigned)(8 * wpx)), dim3(C::NTHR), C::LDS_BYTES + 4096, stream, a, ncs, nrg, wpx, rcp_ncs, tq, tr);
#else
  V4H_LAUNCH(v4h_gemm3_kernel<C>, dim3((unsigned)(8 * wpx)), dim3(C::NTHR), C::LDS_BYTES, stream, a, ncs, nrg, wpx, rcp_ncs, tq, tr);
#endif
  V4H_CHECK_LAUNCH(name);
  return V4H_OK;
}
