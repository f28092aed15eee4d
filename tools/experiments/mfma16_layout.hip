// Operand lane map of v_mfma_f32_16x16x16_bf16 (the K = 16 tail of head_dim 80 in the attention kernels), checked with exact integer data:
// A[i][k] = i + 100 k (distinct), B[k][j] = one-hot rows, under the map  lane l: A[l & 15][4 (l >> 4) + e], B[4 (l >> 4) + e][l & 15], e = 0..3.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
__global__ void k(float* out, const float* A, const float* B) {  // A[16][16], B[16][16] row-major, out[16][16] = A B
  const int l = threadIdx.x, c = l & 15, g = l >> 4;
  bf16x4 a, b;
  for (int e = 0; e < 4; ++e) {
    a[e] = (__bf16)A[c * 16 + 4 * g + e];
    b[e] = (__bf16)B[(4 * g + e) * 16 + c];
  }
  f32x4 d = {0, 0, 0, 0};
  d = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), d, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[(4 * g + r) * 16 + c] = d[r];  // D: row 4g + r, col c
}
int main() {
  float hA[256], hB[256], hO[256], ref[256];
  for (int i = 0; i < 16; ++i)
    for (int kk = 0; kk < 16; ++kk) { hA[i * 16 + kk] = (float)((i * 7 + kk * 3) % 11 - 5); hB[i * 16 + kk] = (float)((i * 5 + kk * 2) % 9 - 4); }
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) { float s = 0; for (int kk = 0; kk < 16; ++kk) s += hA[i * 16 + kk] * hB[kk * 16 + j]; ref[i * 16 + j] = s; }
  float *dA, *dB, *dO;
  hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dO, 1024);
  hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dO, dA, dB);
  hipMemcpy(hO, dO, 1024, hipMemcpyDeviceToHost);
  int bad = 0, badT = 0;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { bad += hO[i * 16 + j] != ref[i * 16 + j]; badT += hO[j * 16 + i] != ref[i * 16 + j]; }
  printf("mismatches vs A*B: %d ; vs (A*B)^T: %d\n", bad, badT);
  return bad != 0;
}
