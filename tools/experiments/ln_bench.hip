// Stand-alone timing of the LayerNorm + modulate kernels (forward with the gated residual update, backward with the gate backward) at the ds2 shape
// (B = 128, T = 135, D = 480; bf16 mode), cold: every launch works on another of NSETS buffer sets (> 2 x the Infinity Cache between two uses).
// Includes the product translation unit, so the kernels timed are the library's own templates.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -o tools/experiments/ln_bench tools/experiments/ln_bench.hip && tools/experiments/ln_bench
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../vit4hep_amd/csrc/v4h_elementwise.hip"

void v4h_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vfprintf(stderr, fmt, ap);
  va_end(ap);
  fputc('\n', stderr);
}
std::atomic<int> v4h_reserved_cus{0};

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } \
  } while (0)

using namespace v4h;
constexpr int B = 128, T = 135, D = 480, BT = B * T, LDM = 18240, NSETS = 6;

constexpr double fwd_bytes_c() { return (double)BT * D * (4 + 2 + 4 + 2); }
struct Set {
  float *x, *xo, *mean, *rstd, *dxin, *dxout, *mod, *dmod;
  bf16 *y, *u, *du, *dy;
};

template <typename F> float time_us(F&& launch, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int k = 0; k < NSETS; ++k) launch(k);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int k = 0; k < reps; ++k) launch(k % NSETS);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / reps;
}

static void fill(float* p, size_t n, unsigned seed, float scale) {
  std::vector<float> h(n);
  unsigned s = seed * 2654435761u + 12345u;
  for (size_t i = 0; i < n; ++i) {
    s = s * 1664525u + 1013904223u;
    h[i] = ((int)(s >> 9) % 2001 - 1000) * 1e-3f * scale;
  }
  CK(hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice));
}
__global__ void to_bf16_kernel(const float* s, bf16* d, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) d[i] = (bf16)s[i];
}
__global__ void maxdiff_kernel(const float* a, const float* b, long n, float* out) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) atomicMax((int*)out, __float_as_int(fabsf(a[i] - b[i])));
}
__global__ void maxdiff_bf_kernel(const bf16* a, const bf16* b, long n, float* out) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) atomicMax((int*)out, __float_as_int(fabsf((float)a[i] - (float)b[i])));
}

// ---- streaming-rate probes: what does a kernel that only moves bytes reach on this card, by access shape and read : write mix?
__global__ __launch_bounds__(256) void cp_f4_kernel(const f32x4* __restrict__ s, f32x4* __restrict__ d, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) d[i] = s[i];
}
__global__ __launch_bounds__(256) void cp_f4nt_kernel(const f32x4* __restrict__ s, f32x4* __restrict__ d, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) __builtin_nontemporal_store(__builtin_nontemporal_load(s + i), d + i);
}
__global__ __launch_bounds__(256) void cp_f4nts_kernel(const f32x4* __restrict__ s, f32x4* __restrict__ d, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) __builtin_nontemporal_store(s[i], d + i);
}
__global__ __launch_bounds__(256) void cp_f1_kernel(const float* __restrict__ s, float* __restrict__ d, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) d[i] = s[i];
}
__global__ __launch_bounds__(256) void rd_f4_kernel(const f32x4* __restrict__ s, float* __restrict__ out, long n4) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) acc += s[i];
  if (acc[0] + acc[1] + acc[2] + acc[3] == 1234.5f) out[0] = acc[0];
}
__global__ __launch_bounds__(256) void wr_f4_kernel(f32x4* __restrict__ d, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) d[i] = f32x4{1.f, 2.f, 3.f, 4.f};
}
// the LayerNorm forward's access shape without its arithmetic: one wave per 480-column row (60 of 64 lanes), x 2 x 16 B + y 16 B in, x_out 2 x 16 B + u 16 B out
template <bool PERSIST> __global__ __launch_bounds__(256) void rowcopy_kernel(const float* __restrict__ x, const bf16* __restrict__ y, float* __restrict__ xo,
                                                                               bf16* __restrict__ u, int BTn, int Dn) {
  const int lane = threadIdx.x & 63;
  const int cl = lane * 8 < Dn ? lane * 8 : 0;
  for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < BTn; row += PERSIST ? gridDim.x * 4 : BTn) {
    const f32x4 a = load4(x + (long)row * Dn + cl), b = load4(x + (long)row * Dn + cl + 4);
    const bf16x8 c = *reinterpret_cast<const bf16x8*>(y + (long)row * Dn + cl);
    if (lane * 8 < Dn) {
      store4(xo + (long)row * Dn + cl, a + b);
      store4(xo + (long)row * Dn + cl + 4, b);
      *reinterpret_cast<bf16x8*>(u + (long)row * Dn + cl) = c;
    }
  }
}

int main() {
  std::vector<Set> sets(NSETS);
  float* tmp;
  CK(hipMalloc(&tmp, (size_t)BT * D * 4));
  for (int k = 0; k < NSETS; ++k) {
    Set& s = sets[k];
    CK(hipMalloc(&s.x, (size_t)BT * D * 4)); CK(hipMalloc(&s.xo, (size_t)BT * D * 4)); CK(hipMalloc(&s.dxin, (size_t)BT * D * 4)); CK(hipMalloc(&s.dxout, (size_t)BT * D * 4));
    CK(hipMalloc(&s.mean, BT * 4)); CK(hipMalloc(&s.rstd, BT * 4)); CK(hipMalloc(&s.mod, (size_t)B * LDM * 4)); CK(hipMalloc(&s.dmod, (size_t)B * LDM * 4));
    CK(hipMalloc(&s.y, (size_t)BT * D * 2)); CK(hipMalloc(&s.u, (size_t)BT * D * 2)); CK(hipMalloc(&s.du, (size_t)BT * D * 2)); CK(hipMalloc(&s.dy, (size_t)BT * D * 2));
    fill(s.x, (size_t)BT * D, 11 + k, 2.0f); fill(s.dxin, (size_t)BT * D, 21 + k, 1.0f); fill(s.mod, (size_t)B * LDM, 31 + k, 0.5f);
    fill(tmp, (size_t)BT * D, 41 + k, 1.0f);
    to_bf16_kernel<<<(BT * D + 255) / 256, 256>>>(tmp, s.y, (long)BT * D);
    fill(tmp, (size_t)BT * D, 51 + k, 1.0f);
    to_bf16_kernel<<<(BT * D + 255) / 256, 256>>>(tmp, s.du, (long)BT * D);
    CK(hipMemset(s.dmod, 0, (size_t)B * LDM * 4));
  }
  CK(hipDeviceSynchronize());
  const int reps = 60;
  const double fwd_bytes = (double)BT * D * (4 + 2 + 4 + 2), bwd_bytes = (double)BT * D * (2 + 4 + 4 + 2 + 4 + 2);
  {  // streaming probes on 2 x 512 MiB (and the row-shaped copy on the LayerNorm buffers)
    const long nb = 1L << 29;
    char *ca, *cb;
    CK(hipMalloc(&ca, nb)); CK(hipMalloc(&cb, nb));
    CK(hipMemset(ca, 1, nb)); CK(hipMemset(cb, 2, nb));
    float* sink;
    CK(hipMalloc(&sink, 64));
    auto t = [&](const char* name, double bytes, auto&& f) {
      for (int r = 0; r < 2; ++r) f();
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0, 0));
      for (int r = 0; r < 10; ++r) f();
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("%-44s %8.1f us  %.2f TB/s\n", name, ms * 100.f, bytes / (ms * 1e-4) * 1e-12 * 1e-6 * 1e6);
    };
    for (int grid : {1024, 2048, 4096, 8192}) {
      char nm[64];
      snprintf(nm, 64, "copy float4, grid %d", grid);
      t(nm, 2.0 * nb, [&] { hipLaunchKernelGGL(cp_f4_kernel, dim3(grid), dim3(256), 0, 0, (const f32x4*)ca, (f32x4*)cb, nb / 16); });
    }
    t("copy float4 nt load + nt store, grid 2048", 2.0 * nb, [&] { hipLaunchKernelGGL(cp_f4nt_kernel, dim3(2048), dim3(256), 0, 0, (const f32x4*)ca, (f32x4*)cb, nb / 16); });
    t("copy float4 nt store only, grid 2048", 2.0 * nb, [&] { hipLaunchKernelGGL(cp_f4nts_kernel, dim3(2048), dim3(256), 0, 0, (const f32x4*)ca, (f32x4*)cb, nb / 16); });
    t("copy float (4 B / lane), grid 2048", 2.0 * nb, [&] { hipLaunchKernelGGL(cp_f1_kernel, dim3(2048), dim3(256), 0, 0, (const float*)ca, (float*)cb, nb / 4); });
    t("copy float (4 B / lane), grid 8192", 2.0 * nb, [&] { hipLaunchKernelGGL(cp_f1_kernel, dim3(8192), dim3(256), 0, 0, (const float*)ca, (float*)cb, nb / 4); });
    t("read only float4, grid 2048", 1.0 * nb, [&] { hipLaunchKernelGGL(rd_f4_kernel, dim3(2048), dim3(256), 0, 0, (const f32x4*)ca, sink, nb / 16); });
    t("write only float4, grid 2048", 1.0 * nb, [&] { hipLaunchKernelGGL(wr_f4_kernel, dim3(2048), dim3(256), 0, 0, (f32x4*)cb, nb / 16); });
    {  // AdamW on 26 M parameters (the library's kernel): 4 arrays of 104 MB, reads 16 B, writes 12 B per element
      const long n = 26042528;
      float *p = (float*)ca, *g = (float*)(ca + (1L << 27)), *m = (float*)(ca + (2L << 27)), *v = (float*)(ca + (3L << 27));
      t("adamw_kernel, 26 M parameters", 28.0 * n, [&] { adamw_step(p, g, m, v, n, nullptr, 1e30f, 1e-4f, 0.9f, 0.999f, 1e-8f, 0.1f, 0.1f, 0.001f, nullptr, 0); });
    }
    int k = 0;
    t("row copy (LN forward shape), 4320 workgroups", fwd_bytes_c(), [&] { Set& s = sets[k++ % NSETS]; hipLaunchKernelGGL(rowcopy_kernel<false>, dim3((BT + 3) / 4), dim3(256), 0, 0, s.x, s.y, s.xo, s.u, BT, D); });
    t("row copy (LN forward shape), persistent 2048", fwd_bytes_c(), [&] { Set& s = sets[k++ % NSETS]; hipLaunchKernelGGL(rowcopy_kernel<true>, dim3(2048), dim3(256), 0, 0, s.x, s.y, s.xo, s.u, BT, D); });
    t("row copy (LN forward shape), persistent 1024", fwd_bytes_c(), [&] { Set& s = sets[k++ % NSETS]; hipLaunchKernelGGL(rowcopy_kernel<true>, dim3(1024), dim3(256), 0, 0, s.x, s.y, s.xo, s.u, BT, D); });
    CK(hipFree(ca)); CK(hipFree(cb));
  }
  // ---------------- forward (x_out = x + gate * y; LN; modulate)
  auto fwd_old = [&](int k) {
    Set& s = sets[k];
    hipLaunchKernelGGL((ln_modulate_fwd8_kernel<bf16, 1>), dim3((BT + 3) / 4), dim3(256), 0, 0, s.x, s.mod, s.mod + D, LDM, s.u, s.mean, s.rstd, BT, T, D, (const bf16*)s.y,
                       (const float*)(s.mod + 2 * D), LDM, s.xo);
  };
  auto fwd_new = [&](int k) {
    Set& s = sets[k];
    hipLaunchKernelGGL((ln_modulate_fwd8v2_kernel<bf16, 1, true>), dim3((BT + 3) / 4), dim3(256), 0, 0, s.x, s.mod, s.mod + D, LDM, s.u, s.mean, s.rstd, BT, T, D, (const bf16*)s.y,
                       (const float*)(s.mod + 2 * D), LDM, s.xo);
  };
  float* dmax;
  CK(hipMalloc(&dmax, 16));
  {  // agreement of the two forms on set 0 (u and x_out)
    bf16* u_ref;
    float* xo_ref;
    CK(hipMalloc(&u_ref, (size_t)BT * D * 2)); CK(hipMalloc(&xo_ref, (size_t)BT * D * 4));
    fwd_old(0);
    CK(hipMemcpy(u_ref, sets[0].u, (size_t)BT * D * 2, hipMemcpyDeviceToDevice)); CK(hipMemcpy(xo_ref, sets[0].xo, (size_t)BT * D * 4, hipMemcpyDeviceToDevice));
    CK(hipMemset(sets[0].u, 0, (size_t)BT * D * 2)); CK(hipMemset(sets[0].xo, 0, (size_t)BT * D * 4));
    fwd_new(0);
    CK(hipMemset(dmax, 0, 16));
    maxdiff_bf_kernel<<<(BT * D + 255) / 256, 256>>>(u_ref, sets[0].u, (long)BT * D, dmax);
    maxdiff_kernel<<<(BT * D + 255) / 256, 256>>>(xo_ref, sets[0].xo, (long)BT * D, dmax + 1);
    float h[2];
    CK(hipMemcpy(h, dmax, 8, hipMemcpyDeviceToHost));
    printf("forward v2 vs v1: max |du| %.3g, max |dx_out| %.3g\n", h[0], h[1]);
  }
  for (int rnd = 0; rnd < 2; ++rnd) {
    const float a = time_us(fwd_old, reps), b = time_us(fwd_new, reps);
    printf("fwd  v1 %.2f us (%.2f TB/s)   v2 %.2f us (%.2f TB/s)\n", a, fwd_bytes / a * 1e-6, b, fwd_bytes / b * 1e-6);
  }
  // ---------------- backward (block form: dx_in, y / gate, dx_out)
  auto args = [&](int k) {
    Set& s = sets[k];
    LnBwdArgs l = LnBwdArgs();
    l.du = s.du; l.x = s.x; l.mean = s.mean; l.rstd = s.rstd; l.scale = s.mod + 4 * D; l.ld_mod = LDM;
    l.dx_in = s.dxin; l.dx_out = s.dxout; l.dshift = s.dmod + 3 * D; l.dscale = s.dmod + 4 * D; l.ld_dmod = LDM;
    l.y = s.y; l.gate = s.mod + 2 * D; l.ld_mod_gate = LDM; l.dy = s.dy; l.dgate = s.dmod + 2 * D; l.ld_dgate = LDM;
    l.B = B; l.T = T; l.D = D;
    return l;
  };
  auto bwd_old = [&](int k) { hipLaunchKernelGGL((ln_modulate_bwd8_kernel<bf16, 1, 16, 2>), dim3((T + 15) / 16, B), dim3(256), 0, 0, args(k)); };
#define BWD_NEW(ROWS, NW, R) [&](int k) { hipLaunchKernelGGL((ln_modulate_bwd8v2_kernel<bf16, 1, ROWS, NW, R, true, true, true, false>), dim3((T + ROWS - 1) / ROWS, B), dim3(64 * NW), 0, 0, args(k)); }
  {
    float *dx_ref, *dm_ref;
    bf16* dy_ref;
    CK(hipMalloc(&dx_ref, (size_t)BT * D * 4)); CK(hipMalloc(&dm_ref, (size_t)B * LDM * 4)); CK(hipMalloc(&dy_ref, (size_t)BT * D * 2));
    CK(hipMemset(sets[0].dmod, 0, (size_t)B * LDM * 4));
    bwd_old(0);
    CK(hipMemcpy(dx_ref, sets[0].dxout, (size_t)BT * D * 4, hipMemcpyDeviceToDevice)); CK(hipMemcpy(dm_ref, sets[0].dmod, (size_t)B * LDM * 4, hipMemcpyDeviceToDevice));
    CK(hipMemcpy(dy_ref, sets[0].dy, (size_t)BT * D * 2, hipMemcpyDeviceToDevice));
    CK(hipMemset(sets[0].dmod, 0, (size_t)B * LDM * 4)); CK(hipMemset(sets[0].dxout, 0, (size_t)BT * D * 4)); CK(hipMemset(sets[0].dy, 0, (size_t)BT * D * 2));
    auto f = BWD_NEW(16, 4, 2);
    f(0);
    CK(hipMemset(dmax, 0, 16));
    maxdiff_kernel<<<(BT * D + 255) / 256, 256>>>(dx_ref, sets[0].dxout, (long)BT * D, dmax);
    maxdiff_kernel<<<(B * LDM + 255) / 256, 256>>>(dm_ref, sets[0].dmod, (long)B * LDM, dmax + 1);
    maxdiff_bf_kernel<<<(BT * D + 255) / 256, 256>>>(dy_ref, sets[0].dy, (long)BT * D, dmax + 2);
    float h[3];
    CK(hipMemcpy(h, dmax, 12, hipMemcpyDeviceToHost));
    printf("backward v2 vs v1: max |d dx| %.3g, max |d dmod| %.3g (sums of 135 terms of O(1)), max |d dy| %.3g\n", h[0], h[1], h[2]);
  }
  for (int rnd = 0; rnd < 2; ++rnd) {
    printf("bwd  v1<16 rows,4 waves,R2> %.2f us (%.2f TB/s)\n", time_us(bwd_old, reps), bwd_bytes / time_us(bwd_old, reps) * 1e-6);
#define RUN(ROWS, NW, R) { const float t = time_us(BWD_NEW(ROWS, NW, R), reps); printf("bwd  v2<%d rows,%d waves,R%d> %.2f us (%.2f TB/s)\n", ROWS, NW, R, t, bwd_bytes / t * 1e-6); }
    RUN(16, 4, 2) RUN(16, 4, 1) RUN(8, 4, 2) RUN(8, 4, 1) RUN(32, 4, 2) RUN(16, 8, 2) RUN(16, 8, 1) RUN(32, 8, 2) RUN(32, 8, 1) RUN(27, 8, 2) RUN(45, 8, 2) RUN(45, 8, 1) RUN(27, 4, 2) RUN(24, 4, 2)
  }
  return 0;
}
