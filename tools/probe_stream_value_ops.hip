// cross-stream hand-off through stream memory operations (hipStreamWriteValue32 / hipStreamWaitValue32)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); return 1; } } while (0)
__global__ void busy(long long ticks) { const long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(2); }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int run(int* f, const char* what) {
  hipStream_t A, B;
  int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking));
  CK(hipStreamCreateWithPriority(&B, hipStreamNonBlocking, hi));
  const int N = 300; const long long T = 1000;  // 10 us
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipDeviceSynchronize()); double t0 = now();
    for (int i = 0; i < 2 * N; ++i) busy<<<1, 64, 0, A>>>(T);
    CK(hipDeviceSynchronize()); double base = (now() - t0) / N * 1e6;
    CK(hipMemset(f, 0, 8)); CK(hipDeviceSynchronize()); t0 = now();
    for (int i = 1; i <= N; ++i) {   // ping-pong
      busy<<<1, 64, 0, A>>>(T); CK(hipStreamWriteValue32(A, f, 2 * i - 1, 0)); CK(hipStreamWaitValue32(B, f, 2 * i - 1, hipStreamWaitValueGte, 0xffffffff));
      busy<<<1, 64, 0, B>>>(T); CK(hipStreamWriteValue32(B, f, 2 * i, 0)); CK(hipStreamWaitValue32(A, f, 2 * i, hipStreamWaitValueGte, 0xffffffff));
    }
    CK(hipDeviceSynchronize()); double pp = (now() - t0) / N * 1e6;
    CK(hipMemset(f, 0, 8)); CK(hipDeviceSynchronize()); t0 = now();
    for (int i = 1; i <= N; ++i) {   // one-sided fork
      busy<<<1, 64, 0, A>>>(T); CK(hipStreamWriteValue32(A, f, i, 0)); CK(hipStreamWaitValue32(B, f, i, hipStreamWaitValueGte, 0xffffffff)); busy<<<1, 64, 0, B>>>(T / 2);
      busy<<<1, 64, 0, A>>>(T);
    }
    CK(hipDeviceSynchronize()); double fork = (now() - t0) / N * 1e6;
    printf("%s: per iteration (2 x 10 us busy): same stream %.1f us | value ping-pong %.1f | one-sided fork %.1f\n", what, base, pp, fork);
  }
  return 0;
}
int main() {
  int can = 0; CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
  printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
  int* f; CK(hipMalloc(&f, 64));
  if (run(f, "hipMalloc memory")) printf("hipMalloc memory: failed\n");
  int* s = nullptr;
  hipError_t e = hipExtMallocWithFlags((void**)&s, 8, hipMallocSignalMemory);
  if (e == hipSuccess) { if (run(s, "signal memory")) printf("signal memory: failed\n"); }
  else printf("hipMallocSignalMemory: %s\n", hipGetErrorString(e));
  return 0;
}
