// Does a stop event attached to a kernel launch (hipExtLaunchKernelGGL) avoid the bubble hipEventRecord leaves on the
// recording stream, and does it order a dependent kernel on another stream?
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void busy(long long ticks, long long* stamp) {   // stamp[0] = start, stamp[1] = end (100 MHz)
  const long long t0 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0 && stamp) stamp[0] = t0;
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
  if (threadIdx.x == 0 && blockIdx.x == 0 && stamp) stamp[1] = wall_clock64();
}
int main() {
  hipStream_t s0, sc;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
  hipEvent_t ev, t0, t1;
  CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  long long* st;
  CK(hipMalloc(&st, 64 * sizeof(long long)));
  CK(hipMemset(st, 0, 64 * sizeof(long long)));
  const int N = 200;
  for (int mode = 0; mode < 3; ++mode) {
    std::vector<float> ms;
    long long worst = 1LL << 60;
    for (int rep = 0; rep < 7; ++rep) {
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(t0, s0));
      for (int i = 0; i < N; ++i) {
        if (mode == 2) hipExtLaunchKernelGGL(busy, dim3(256), dim3(64), 0, s0, nullptr, ev, 0, 3000LL, st);       // A: 30 us
        else hipLaunchKernelGGL(busy, dim3(256), dim3(64), 0, s0, 3000LL, st);
        if (mode == 1) CK(hipEventRecord(ev, s0));
        if (mode) { CK(hipStreamWaitEvent(sc, ev, 0)); hipLaunchKernelGGL(busy, dim3(8), dim3(64), 0, sc, 500LL, st + 4); }   // C on sc
        hipLaunchKernelGGL(busy, dim3(256), dim3(64), 0, s0, 1000LL, st + 2);                                       // B: 10 us
      }
      CK(hipEventRecord(t1, s0));
      CK(hipEventSynchronize(t1));
      CK(hipDeviceSynchronize());
      float m; CK(hipEventElapsedTime(&m, t0, t1));
      ms.push_back(m / N * 1e3f);
      long long h[6];
      CK(hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost));
      if (mode) worst = std::min(worst, h[4] - h[1]);   // last iteration: C.start - A.end
    }
    std::sort(ms.begin(), ms.end());
    printf("mode %d (%s): %.2f us per A+B pair (ideal 40.0)", mode, mode == 0 ? "no event" : mode == 1 ? "hipEventRecord after A" : "stop event on A's launch", ms[3]);
    if (mode) printf(", C.start - A.end = %.2f us (must be >= 0)", worst / 100.0);
    printf("\n");
  }
  return 0;
}
