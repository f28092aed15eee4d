// Experiment: cost of cross-stream signalling on the producing stream.
//  (a) kernel; hipEventRecord(e, s1); kernel        - a marker packet between the two kernels of s1
//  (b) hipExtLaunchKernelGGL(kernel, ..., stopEvent = e); kernel   - the event rides on the kernel's own dispatch packet
// In both cases s2 waits for e and runs a consumer that checks the producer's output.
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void produce(float* p, int n, float v) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = v;
}
__global__ void consume(const float* p, int n, float v, int* bad) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    if (p[i] != v) atomicAdd(bad, 1);
}
int main() {
  const int n = 1 << 24, iters = 400, per = n / 8;
  float *a, *b; int* bad;
  CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&bad, 4)); CK(hipMemset(bad, 0, 4));
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  CK(hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, hi));
  hipEvent_t t0, t1;
  CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  for (int mode = 0; mode < 4; ++mode) {
    // mode 0: no signalling; 1: record (sysfence); 2: record (no sysfence); 3: stopEvent on the kernel
    unsigned fl = mode == 1 ? hipEventDisableTiming : (hipEventDisableTiming | hipEventDisableSystemFence);
    if (mode == 3 && getenv("TIMED_EVENTS")) fl = 0;
    hipEvent_t ev[8];
    for (int i = 0; i < 8; ++i) CK(hipEventCreateWithFlags(&ev[i], fl));
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(t0, s1));
      for (int it = 0; it < iters; ++it) {
        float v = (float)(mode * 1000 + it);
        hipEvent_t e = ev[it & 7];
        float* buf = a + (size_t)(it & 7) * per;  // ring of 8 buffers: the consumer of a buffer has long finished when it is rewritten
        if (mode == 3) hipExtLaunchKernelGGL(produce, dim3(1024), dim3(256), 0, s1, nullptr, e, 0, buf, per, v);
        else hipLaunchKernelGGL(produce, dim3(1024), dim3(256), 0, s1, buf, per, v);
        if (mode == 1 || mode == 2) CK(hipEventRecord(e, s1));
        if (mode != 0) {
          CK(hipStreamWaitEvent(s2, e, 0));
          hipLaunchKernelGGL(consume, dim3(1), dim3(64), 0, s2, buf + per - 4096, 4096, v, bad);
        }
        hipLaunchKernelGGL(produce, dim3(1024), dim3(256), 0, s1, b, per, v);
      }
      CK(hipEventRecord(t1, s1));
      CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, t0, t1));
      int hbad; CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
      if (rep >= 1) printf("mode %d: %.2f us per iteration (2 producer kernels), mismatches %d\n", mode, ms * 1e3 / iters, hbad);
    }
  }
  return 0;
}
