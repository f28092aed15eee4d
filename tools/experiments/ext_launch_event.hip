// What does it cost a stream to signal another one behind a kernel?  (a) hipEventRecord behind the launch = a marker packet in the queue; (b) the event handed to
// hipExtLaunchKernelGGL as the launch's stop event = the dispatch packet's own completion signal, no extra packet.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ext_launch_event tools/experiments/ext_launch_event.hip && /tmp/ext_launch_event
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
__global__ void work(float* x, long n, float a) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) x[i] = x[i] * a + 1.0f;
}
__global__ void tiny(float* x) { if (threadIdx.x == 0) x[0] += 1.0f; }
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(r_), __LINE__); return 1; } } while (0)
int main() {
  const long n = 16L << 20;
  float *x, *y;
  CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&y, 4096));
  CK(hipMemset(x, 0, n * 4)); CK(hipMemset(y, 0, 4096));
  hipStream_t s, side;
  CK(hipStreamCreate(&s)); CK(hipStreamCreate(&side));
  const int N = 200;
  std::vector<hipEvent_t> ev(N);
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence));
  hipEvent_t t0, t1;
  CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  for (int mode = 0; mode < 5; ++mode) {
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(t0, s));
      for (int i = 0; i < N; ++i) {
        if (mode == 0) hipLaunchKernelGGL(work, dim3(1024), dim3(256), 0, s, x, n, 1.0001f);
        else if (mode == 1 || mode == 2) { hipLaunchKernelGGL(work, dim3(1024), dim3(256), 0, s, x, n, 1.0001f); CK(hipEventRecord(ev[i], s)); }
        else hipExtLaunchKernelGGL(work, dim3(1024), dim3(256), 0, s, nullptr, ev[i], 0, x, n, 1.0001f);
        if (mode == 2 || mode == 4) { CK(hipStreamWaitEvent(side, ev[i], 0)); hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, side, y); }
      }
      CK(hipEventRecord(t1, s));
      CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, t0, t1));
      if (ms < best) best = ms;
    }
    const char* names[] = {"plain launches", "hipEventRecord behind every launch", "hipEventRecord + side stream waits and runs a tiny kernel",
                           "stop event of hipExtLaunchKernelGGL", "stop event of hipExtLaunchKernelGGL + side stream waits and runs a tiny kernel"};
    printf("%-80s %.2f us per launch\n", names[mode], best * 1e3f / N);
  }
  float h; CK(hipMemcpy(&h, y, 4, hipMemcpyDeviceToHost));
  printf("side kernels run: %.0f\n", h);
  return 0;
}
