// cross-stream hand-off cost: HIP events vs device-side flags (one GPU)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); return 1; } } while (0)
__global__ void busy(long long ticks) { const long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(2); }
__global__ void set_flag(int* f, int v) { __hip_atomic_store(f, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
__global__ void wait_flag(const int* f, int v, int* timeout) {
  const long long t0 = wall_clock64();
  while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < v) {
    __builtin_amdgcn_s_sleep(1);
    if (wall_clock64() - t0 > 2000000) { *timeout = 1; break; }   // 20 ms
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipStream_t A, B;
  int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking));
  CK(hipStreamCreateWithPriority(&B, hipStreamNonBlocking, hi));
  int *f, *to; CK(hipMalloc(&f, 64)); CK(hipMemset(f, 0, 64)); to = f + 8;
  hipEvent_t e1, e2; CK(hipEventCreateWithFlags(&e1, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
  const int N = 300; const long long T = 1000;  // 10 us
  for (int rep = 0; rep < 2; ++rep) {
    // baseline: 2N busy kernels on A
    CK(hipDeviceSynchronize()); double t0 = now();
    for (int i = 0; i < 2 * N; ++i) busy<<<1, 64, 0, A>>>(T);
    CK(hipDeviceSynchronize()); double base = (now() - t0) / N * 1e6;
    // events ping-pong
    t0 = now();
    for (int i = 0; i < N; ++i) {
      busy<<<1, 64, 0, A>>>(T); CK(hipEventRecord(e1, A)); CK(hipStreamWaitEvent(B, e1, 0));
      busy<<<1, 64, 0, B>>>(T); CK(hipEventRecord(e2, B)); CK(hipStreamWaitEvent(A, e2, 0));
    }
    CK(hipDeviceSynchronize()); double ev = (now() - t0) / N * 1e6;
    // flags ping-pong
    CK(hipMemset(f, 0, 64)); CK(hipDeviceSynchronize()); t0 = now();
    for (int i = 1; i <= N; ++i) {
      busy<<<1, 64, 0, A>>>(T); set_flag<<<1, 1, 0, A>>>(f, i); wait_flag<<<1, 1, 0, B>>>(f, i, to);
      busy<<<1, 64, 0, B>>>(T); set_flag<<<1, 1, 0, B>>>(f + 1, i); wait_flag<<<1, 1, 0, A>>>(f + 1, i, to);
    }
    CK(hipDeviceSynchronize()); double fl = (now() - t0) / N * 1e6;
    // one-sided fork: A: busy, record, busy ; B: wait, busy  (how much does the record delay A's chain?)
    t0 = now();
    for (int i = 0; i < N; ++i) {
      busy<<<1, 64, 0, A>>>(T); CK(hipEventRecord(e1, A)); CK(hipStreamWaitEvent(B, e1, 0)); busy<<<1, 64, 0, B>>>(T / 2);
      busy<<<1, 64, 0, A>>>(T);
    }
    CK(hipDeviceSynchronize()); double fork_ev = (now() - t0) / N * 1e6;
    CK(hipMemset(f, 0, 64)); CK(hipDeviceSynchronize()); t0 = now();
    for (int i = 1; i <= N; ++i) {
      busy<<<1, 64, 0, A>>>(T); set_flag<<<1, 1, 0, A>>>(f, i); wait_flag<<<1, 1, 0, B>>>(f, i, to); busy<<<1, 64, 0, B>>>(T / 2);
      busy<<<1, 64, 0, A>>>(T);
    }
    CK(hipDeviceSynchronize()); double fork_fl = (now() - t0) / N * 1e6;
    // wait on an already completed event, on A's chain
    CK(hipEventRecord(e2, B)); CK(hipDeviceSynchronize()); t0 = now();
    for (int i = 0; i < N; ++i) { busy<<<1, 64, 0, A>>>(T); CK(hipStreamWaitEvent(A, e2, 0)); busy<<<1, 64, 0, A>>>(T); }
    CK(hipDeviceSynchronize()); double wdone = (now() - t0) / N * 1e6;
    int h[16]; CK(hipMemcpy(h, f, 64, hipMemcpyDeviceToHost));
    printf("per iteration (2 x 10 us busy): same stream %.1f us | event ping-pong %.1f | flag ping-pong %.1f (timeouts %d) | "
           "one-sided fork: event %.1f, flag %.1f | wait on a completed event %.1f\n", base, ev, fl, h[8], fork_ev, fork_fl, wdone);
  }
  return 0;
}
