// How much does a cross-stream hand-over cost the PRODUCER's own stream?  Stream A runs K1, K2 back to back; stream B must start K3 after K1.
//   (a) no hand-over at all                                  -> the K1 -> K2 gap of an undisturbed queue
//   (b) hipEventRecord(A) after K1 + hipStreamWaitEvent(B)   -> what the library does (a barrier packet behind K1)
//   (c) K1's last workgroup writes a flag in signal memory, B waits with hipStreamWaitValue32 -> no packet in A's queue
// Each kernel stamps the constant 100 MHz clock at start and end; printed: gap K1 end -> K2 start, and K1 end -> K3 start.
//   hipcc --offload-arch=gfx950 -O3 -o tools/experiments/handover_probe tools/experiments/handover_probe.hip && tools/experiments/handover_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void work(unsigned long long* stamp, int iters, unsigned* done, unsigned* flag, unsigned seq) {
  if (blockIdx.x == 0 && threadIdx.x == 0) stamp[0] = __builtin_amdgcn_s_memrealtime();
  float x = threadIdx.x;
  for (int i = 0; i < iters; ++i) x = x * 1.0001f + 0.5f;
  if (x == 123.456f) stamp[3] = 1;
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicMax(&stamp[1], __builtin_amdgcn_s_memrealtime());
    if (done != nullptr) {  // the last workgroup to finish publishes the hand-over
      __threadfence();
      const unsigned old = atomicAdd(done, 1u);
      if (old == gridDim.x - 1) {
        *done = 0;
        __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

int main() {
  int can = 0;
  CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
  printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
  hipStream_t A, B;
  int least, greatest;
  CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
  CK(hipStreamCreateWithPriority(&A, hipStreamNonBlocking, least));
  CK(hipStreamCreateWithPriority(&B, hipStreamNonBlocking, greatest));
  unsigned long long* st;
  CK(hipMalloc(&st, 3 * 4 * sizeof(unsigned long long)));
  unsigned *done, *flag;
  CK(hipMalloc(&done, 4));
  CK(hipMemset(done, 0, 4));
  CK(hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory));
  CK(hipMemset(flag, 0, 8));
  hipEvent_t ev;
  CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  const int grid = 512, block = 256, iters = 40000;  // ~60 us per kernel
  unsigned seq = 0;
  for (int mode = 0; mode < 3; ++mode) {
    double g12 = 0, g13 = 0, host_us = 0;
    const int reps = 20;
    for (int r = 0; r < reps + 2; ++r) {
      CK(hipMemsetAsync(st, 0, 3 * 4 * sizeof(unsigned long long), A));
      CK(hipStreamSynchronize(A));
      ++seq;
      hipLaunchKernelGGL(work, dim3(grid), dim3(block), 0, A, st, iters, mode == 2 ? done : nullptr, flag, seq);
      if (mode == 1) {
        CK(hipEventRecord(ev, A));
        CK(hipStreamWaitEvent(B, ev, 0));
      } else if (mode == 2) {
        timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        CK(hipStreamWaitValue32(B, flag, seq, hipStreamWaitValueGte, 0xFFFFFFFFu));
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if (r >= 2) host_us += (t1.tv_sec - t0.tv_sec) * 1e6 + (t1.tv_nsec - t0.tv_nsec) * 1e-3;
      }
      hipLaunchKernelGGL(work, dim3(grid), dim3(block), 0, A, st + 4, iters, nullptr, flag, 0u);
      if (mode != 0) hipLaunchKernelGGL(work, dim3(64), dim3(block), 0, B, st + 8, 1000, nullptr, flag, 0u);
      CK(hipDeviceSynchronize());
      unsigned long long h[12];
      CK(hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost));
      if (r >= 2) {
        g12 += ((double)h[4] - (double)h[1]) * 0.01;
        if (mode != 0) g13 += ((double)h[8] - (double)h[1]) * 0.01;
      }
    }
    printf("%-58s K1 end -> K2 start %6.2f us   K1 end -> K3 start (other stream) %6.2f us\n",
           mode == 0 ? "no hand-over" : mode == 1 ? "event record behind K1 + stream wait" : "flag written by K1's last workgroup + hipStreamWaitValue32", g12 / 20, g13 / 20);
    if (mode == 2) printf("host time of one hipStreamWaitValue32 call while stream A is busy: %.1f us\n", host_us / 20);
  }
  return 0;
}
