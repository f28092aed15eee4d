// vit4hep_amd - MI355X (gfx950 / CDNA4) device helpers shared by every kernel.
//
// One fragment convention for both arithmetic modes, so GEMM / attention kernels are written once:
//
//   frag[jj] = X[idx = lane & 15][k = k0 + 8 * (lane >> 4) + jj],   jj = 0..7      (a 16 x 32 slab)
//
//   bf16 mode : 8 bf16 in 4 VGPRs, ONE  v_mfma_f32_16x16x32_bf16 per slab pair
//   f32  mode : 8 f32  in 8 VGPRs, EIGHT v_mfma_f32_16x16x4_f32 (MFMA jj uses element jj of both
//               operands: k-slot (lane>>4) of MFMA jj is physical k0 + 8*(lane>>4) + jj for A and B
//               alike, so the contraction is a permutation of the same 32 products; exact f32 FMAs).
//
// D = A x B with A = "regs-side" operand (its idx becomes 4 consecutive rows held in the 4
// accumulator registers) and B = "lane-side" operand (its idx becomes lane & 15):
//   acc[r] = Out[lane_side_idx = lane & 15][regs_side_idx = 4 * (lane >> 4) + r]
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include <atomic>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define V4H_DEV __device__ __forceinline__
#define V4H_LDS __attribute__((address_space(3)))

template <typename T> struct Frag;
template <> struct Frag<float> { float v[8]; };
template <> struct Frag<bf16> { bf16x8 v; };

template <typename T> V4H_DEV Frag<T> frag_zero() {
  Frag<T> f;
#pragma unroll
  for (int j = 0; j < 8; ++j) f.v[j] = (T)0.0f;
  return f;
}

// acc += regs_side (A) x lane_side (B)
V4H_DEV f32x4 mma(const Frag<bf16>& regs_side, const Frag<bf16>& lane_side, f32x4 acc) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(regs_side.v, lane_side.v, acc, 0, 0, 0);
}
V4H_DEV f32x4 mma(const Frag<float>& regs_side, const Frag<float>& lane_side, f32x4 acc) {
#pragma unroll
  for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(regs_side.v[j], lane_side.v[j], acc, 0, 0, 0);
  return acc;
}

// ---- fragment loads from LDS ---------------------------------------------------------------------
// K-contiguous image  tile[idx][k]  (row stride ld elements, ld*sizeof(T) % 16 == 0)
V4H_DEV Frag<bf16> frag_kcontig(const bf16* tile, int ld, int idx0, int k0, int lane) {
  Frag<bf16> f;
  f.v = *reinterpret_cast<const bf16x8*>(tile + (idx0 + (lane & 15)) * ld + k0 + 8 * (lane >> 4));
  return f;
}
V4H_DEV Frag<float> frag_kcontig(const float* tile, int ld, int idx0, int k0, int lane) {
  Frag<float> f;
  const float4* p = reinterpret_cast<const float4*>(tile + (idx0 + (lane & 15)) * ld + k0 + 8 * (lane >> 4));
  float4 a = p[0], b = p[1];
  f.v[0] = a.x; f.v[1] = a.y; f.v[2] = a.z; f.v[3] = a.w;
  f.v[4] = b.x; f.v[5] = b.y; f.v[6] = b.z; f.v[7] = b.w;
  return f;
}

// ds_read_b64_tr_b16: per 16-lane group, lane 4q+p supplies the address of row q, columns 4p..4p+3 of a
// 4 x 16 block of 16-bit elements; lane i receives column i, row q in element q.  EXEC must be all ones.
V4H_DEV bf16x4 lds_tr_read(const bf16* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 V4H_LDS*)(p));
}

// K-strided image  tile[k][idx]  (row stride ld elements).  The two 4-row groups of the slab may come
// from different row bases (ka for jj 0..3, kb for jj 4..7): this lets a k-permutation chosen by the
// OTHER operand (e.g. an accumulator re-used as operand) be followed exactly.
//   frag[jj]   = tile[ka + 4*(lane>>4) + jj    ][idx0 + (lane&15)]   jj = 0..3
//   frag[4+jj] = tile[kb + 4*(lane>>4) + jj    ][idx0 + (lane&15)]
V4H_DEV Frag<bf16> frag_kstrided2(const bf16* tile, int ld, int ka, int kb, int idx0, int lane) {
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  bf16x4 lo = lds_tr_read(tile + (ka + 4 * g + q) * ld + idx0 + 4 * p);
  bf16x4 hi = lds_tr_read(tile + (kb + 4 * g + q) * ld + idx0 + 4 * p);
  Frag<bf16> f;
  f.v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return f;
}
V4H_DEV Frag<float> frag_kstrided2(const float* tile, int ld, int ka, int kb, int idx0, int lane) {
  const int g = lane >> 4, c = lane & 15;
  Frag<float> f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f.v[j] = tile[(ka + 4 * g + j) * ld + idx0 + c];
    f.v[4 + j] = tile[(kb + 4 * g + j) * ld + idx0 + c];
  }
  return f;
}
// canonical slab k0..k0+31: rows k0 + 8g + jj  ==  groups (k0 + 4g' ...) with ka = k0 + 4g, kb = k0 + 4g + 4:
// expressed through the two-base form by folding the extra 4*g into the bases.
template <typename T> V4H_DEV Frag<T> frag_kstrided(const T* tile, int ld, int k0, int idx0, int lane) {
  const int g = lane >> 4;
  return frag_kstrided2(tile, ld, k0 + 4 * g, k0 + 4 * g + 4, idx0, lane);
}

// accumulators of two 16-wide tiles (same lane-side idx, regs-side = contraction index of the next product)
// -> lane-side fragment of the next product.  Physical k of element jj: jj<4 -> tile0 row 4g+jj, else tile1 row 4g+jj-4,
// which is exactly what frag_kstrided2(ka = base0, kb = base1) reads for the other operand.
V4H_DEV Frag<float> frag_from_acc(f32x4 a0, f32x4 a1, float) {
  Frag<float> f;
#pragma unroll
  for (int j = 0; j < 4; ++j) { f.v[j] = a0[j]; f.v[4 + j] = a1[j]; }
  return f;
}
V4H_DEV Frag<bf16> frag_from_acc(f32x4 a0, f32x4 a1, bf16) {
  Frag<bf16> f;
#pragma unroll
  for (int j = 0; j < 4; ++j) { f.v[j] = (bf16)a0[j]; f.v[4 + j] = (bf16)a1[j]; }
  return f;
}

// ---- scalar helpers -------------------------------------------------------------------------------
V4H_DEV float to_f32(float x) { return x; }
V4H_DEV float to_f32(bf16 x) { return (float)x; }

V4H_DEV void store4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
V4H_DEV void store4(bf16* p, f32x4 v) {
  bf16x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = (bf16)v[j];
  *reinterpret_cast<bf16x4*>(p) = o;
}
V4H_DEV f32x4 load4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
V4H_DEV f32x4 load4(const bf16* p) {
  bf16x4 o = *reinterpret_cast<const bf16x4*>(p);
  f32x4 v;
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = (float)o[j];
  return v;
}

V4H_DEV float silu_f(float x) { return x / (1.0f + __expf(-x)); }
V4H_DEV float dsilu_f(float x) {
  const float s = 1.0f / (1.0f + __expf(-x));
  return s * (1.0f + x * (1.0f - s));
}
// tanh: accurate libm form in f32 (parity) mode, exp-based form in bf16 mode
template <typename T> V4H_DEV float tanh_m(float u);
template <> V4H_DEV float tanh_m<float>(float u) { return tanhf(u); }
template <> V4H_DEV float tanh_m<bf16>(float u) {  // v_exp_f32 + v_rcp_f32 (an IEEE division is a ~10-instruction sequence)
  const float e = __builtin_amdgcn_exp2f(2.8853900818f * u);
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}
// nn.GELU(approximate="tanh")  (reference nn/vit.py:314-315)
template <typename T> V4H_DEV float gelu_tanh_f(float x) {
  const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
  return 0.5f * x * (1.0f + tanh_m<T>(u));
}
template <typename T> V4H_DEV float dgelu_tanh_f(float x) {
  const float x2 = x * x;
  const float u = 0.7978845608028654f * (x + 0.044715f * x * x2);
  const float t = tanh_m<T>(u);
  return 0.5f * (1.0f + t) + 0.5f * x * (1.0f - t * t) * 0.7978845608028654f * (1.0f + 3.0f * 0.044715f * x2);
}

// gelu_tanh(x) and its derivative from ONE tanh (forward epilogue stores both; the backward is then a plain multiply)
template <typename T> V4H_DEV void gelu_and_grad(float x, float& y, float& dy) {
  const float x2 = x * x;
  const float t = tanh_m<T>(0.7978845608028654f * (x + 0.044715f * x * x2));
  y = 0.5f * x * (1.0f + t);
  dy = 0.5f * (1.0f + t) + 0.5f * x * (1.0f - t * t) * 0.7978845608028654f * (1.0f + 3.0f * 0.044715f * x2);
}
// throughput mode: 0.5 x (1 + tanh u) = x * sigmoid(2u); one v_exp_f32, one v_rcp_f32 and a handful of FMAs per element
//   2u = x (a + b x^2), a = 2*sqrt(2/pi), b = a * 0.044715 ;  d/dx [x s] = s + x s (1 - s) (a + 3 b x^2)
V4H_DEV float gelu_sigmoid_arg(float x, float x2) { return x * (1.5957691216f + 0.0713548163f * x2); }
template <> V4H_DEV void gelu_and_grad<bf16>(float x, float& y, float& dy) {
  const float x2 = x * x;
  const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950409f * gelu_sigmoid_arg(x, x2)));
  y = x * s;
  dy = s + y * (1.0f - s) * (1.5957691216f + 0.2140644489f * x2);
}
template <typename T> V4H_DEV float gelu_only(float x) { return gelu_tanh_f<T>(x); }
template <> V4H_DEV float gelu_only<bf16>(float x) {
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950409f * gelu_sigmoid_arg(x, x * x)));
}

V4H_DEV float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
V4H_DEV float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// `s_waitcnt vmcnt(n)` for a wave-uniform n that is only known at run time (gfx9 has no register form of the instruction): a computed jump into a table of
// 49 two-instruction entries - eight scalar instructions in all.  (Written as a C++ switch the compiler lowers it to a chain of some 200 scalar compares and
// branches, about 450 clocks on the critical path of every K-step: tools/experiments/gemm2_stamps.py.)  n above 48 waits for 48: only ever conservative.
// The jump arithmetic assumes 8 bytes per table entry and takes the distance from the s_getpc result to the table from the assembler (label
// difference), and the .if below fails the BUILD if an assembler ever encodes an entry in another size.  s[92:93] (the jump target needs an aligned
// pair, which an asm operand cannot be split into) is declared clobbered; the other scratch register is the compiler's choice.
V4H_DEV void wait_vmcnt64(int n) {
#define V4H_VM_ROW(k) "s_waitcnt vmcnt(" #k ")\n\ts_branch 1f\n\t"
  unsigned t32;
  asm volatile(
      "s_min_u32 %0, %1, 48\n\t"
      "s_lshl_b32 %0, %0, 3\n\t"
      "s_getpc_b64 s[92:93]\n"
      "3:\n\t"
      "s_add_u32 %0, %0, 2f-3b\n\t"
      "s_add_u32 s92, s92, %0\n\t"
      "s_addc_u32 s93, s93, 0\n\t"
      "s_setpc_b64 s[92:93]\n"
      "2:\n\t"
      V4H_VM_ROW(0) V4H_VM_ROW(1) V4H_VM_ROW(2) V4H_VM_ROW(3) V4H_VM_ROW(4) V4H_VM_ROW(5) V4H_VM_ROW(6) V4H_VM_ROW(7) V4H_VM_ROW(8) V4H_VM_ROW(9)
      V4H_VM_ROW(10) V4H_VM_ROW(11) V4H_VM_ROW(12) V4H_VM_ROW(13) V4H_VM_ROW(14) V4H_VM_ROW(15) V4H_VM_ROW(16) V4H_VM_ROW(17) V4H_VM_ROW(18) V4H_VM_ROW(19)
      V4H_VM_ROW(20) V4H_VM_ROW(21) V4H_VM_ROW(22) V4H_VM_ROW(23) V4H_VM_ROW(24) V4H_VM_ROW(25) V4H_VM_ROW(26) V4H_VM_ROW(27) V4H_VM_ROW(28) V4H_VM_ROW(29)
      V4H_VM_ROW(30) V4H_VM_ROW(31) V4H_VM_ROW(32) V4H_VM_ROW(33) V4H_VM_ROW(34) V4H_VM_ROW(35) V4H_VM_ROW(36) V4H_VM_ROW(37) V4H_VM_ROW(38) V4H_VM_ROW(39)
      V4H_VM_ROW(40) V4H_VM_ROW(41) V4H_VM_ROW(42) V4H_VM_ROW(43) V4H_VM_ROW(44) V4H_VM_ROW(45) V4H_VM_ROW(46) V4H_VM_ROW(47) V4H_VM_ROW(48)
      "\n1:\n\t"
      ".if (1b - 2b) != 49 * 8\n\t"
      ".error \"wait_vmcnt64: a table entry is not 8 bytes\"\n\t"
      ".endif"
      : "=&s"(t32)
      : "s"(n)
      : "s92", "s93", "scc", "memory");
#undef V4H_VM_ROW
}


// ---- host side ------------------------------------------------------------------------------------
#define V4H_OK 0
#define V4H_ERR_ARG 1
#define V4H_ERR_HIP 2
#define V4H_ERR_UNSUPPORTED 3

void v4h_set_error(const char* fmt, ...);
// Completion signal of ONE launch as an event (round 5).  The backward pass hands a kernel's output to the weight-gradient stream the moment the kernel ends; an
// event RECORDED behind the launch is a marker packet of its own in the main queue (1.1-1.4 us per record, ~30 per update step:
// tools/experiments/ext_launch_event.hip), the stop event of hipExtLaunchKernelGGL is the dispatch packet's own completion signal and costs nothing.  The runtime
// arms the event (v4h_tls_stop_event), the LAST launch of the producing operator takes it (V4H_LAUNCH; the other launch sites never look), and the runtime falls
// back to a record if nobody did (v4h_runtime.hip: arm_fork / complete_fork).
extern thread_local hipEvent_t v4h_tls_stop_event;
#define V4H_LAUNCH(kernel, grid, block, lds, stream, ...)                                        \
  do {                                                                                           \
    if (v4h_tls_stop_event != nullptr) {                                                         \
      hipEvent_t stop_ = v4h_tls_stop_event;                                                     \
      v4h_tls_stop_event = nullptr;                                                              \
      hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, nullptr, stop_, 0, __VA_ARGS__);   \
    } else {                                                                                     \
      hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                         \
    }                                                                                            \
  } while (0)
#define V4H_CHECK_ARG(cond, ...)                 \
  do {                                           \
    if (!(cond)) {                               \
      v4h_set_error(__VA_ARGS__);                \
      return V4H_ERR_ARG;                        \
    }                                            \
  } while (0)
#define V4H_CHECK_LAUNCH(name)                                            \
  do {                                                                    \
    hipError_t e_ = hipGetLastError();                                    \
    if (e_ != hipSuccess) {                                               \
      v4h_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
      return V4H_ERR_HIP;                                                 \
    }                                                                     \
  } while (0)

// Compute units the persistent kernels size their grids for.  The contraction kernels and the single-chunk attention kernels launch one (or two)
// workgroups per CU that own the CU's whole LDS and walk their tiles; a communication kernel (RCCL's ring, 8-16 workgroups with LDS of their own) that
// sits on some CUs while such a grid is launched keeps the workgroups meant for those CUs waiting until another workgroup of the SAME grid retires, i.e.
// until the end - the kernel then takes up to twice as long.  Under a process group the host therefore reserves CUs (v4h_reserve_compute_units) and the
// grids shrink to 256 - reserved: the tile walk redistributes, nothing waits.  A multiple of 8 keeps workgroup id % 8 == XCD.  Measured in
// profiles/r03_comm_interference.md.
extern std::atomic<int> v4h_reserved_cus;
inline int v4h_compute_units() { return 256 - v4h_reserved_cus.load(std::memory_order_relaxed); }

// A per-function attribute (hipFuncSetAttribute: maximum dynamic LDS) belongs to the function object of the CURRENT device, so "done once" is
// remembered per device ordinal, not per process: one process may drive several GPUs (_lib.on_device).  Setting it twice is harmless, so two
// threads that race to the first launch both set it and both record it.
struct DeviceOnce {
  std::atomic<unsigned long long> done[4] = {};
  template <class F> int ensure(F&& set, const char* name, const char* what) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 256) {
      v4h_set_error("%s: no current device (hipGetDevice)", name);
      return V4H_ERR_HIP;
    }
    const unsigned long long bit = 1ull << (dev & 63);
    if (done[dev >> 6].load(std::memory_order_acquire) & bit) return V4H_OK;
    const hipError_t e = set();
    if (e != hipSuccess) {
      v4h_set_error("%s: cannot %s on device %d: %s", name, what, dev, hipGetErrorString(e));
      return V4H_ERR_HIP;
    }
    done[dev >> 6].fetch_or(bit, std::memory_order_release);
    return V4H_OK;
  }
};
