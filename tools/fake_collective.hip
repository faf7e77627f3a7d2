// A stand-in for RCCL's ncclAllReduce used by tools/ddp_occupancy_probe.py on a ONE-GPU box: instead of
// exchanging data it occupies `blocks` workgroups for `microseconds` on the given stream (a bounded wait
// on the constant-rate wall clock, so it always exits), i.e. it reproduces what a collective kernel does
// to the compute kernels that run beside it: it takes CUs away.  Same C signature as ncclAllReduce.
#include <hip/hip_runtime.h>
#include <stddef.h>

struct FakeComm { int blocks; int threads; int latency_us; int kb_per_us; int lds_bytes; };  // time = latency + bytes / rate

__global__ void k_hold(long long ticks) {
  extern __shared__ char lds[];
  if (threadIdx.x == 0) lds[0] = 1;
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

extern "C" int fake_allreduce(const void*, void*, size_t count, int dtype, int, void* comm, void* stream) {
  const FakeComm* c = (const FakeComm*)comm;
  if (c->blocks <= 0 || c->kb_per_us <= 0) return 0;
  const long long bytes = (long long)count * (dtype == 9 ? 2 : 4);   // ncclBfloat16 = 9, ncclFloat32 = 7
  const long long us = c->latency_us + bytes / 1024 / c->kb_per_us;
  // wall_clock64 ticks at 100 MHz on this part
  hipLaunchKernelGGL(k_hold, dim3(c->blocks), dim3(c->threads), c->lds_bytes, (hipStream_t)stream, us * 100);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
