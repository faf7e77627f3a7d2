#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned* o, float* o2) {
  unsigned a = threadIdx.x, b = 100 + threadIdx.x;
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  o[threadIdx.x] = r[0];
  o[64 + threadIdx.x] = r[1];
  f32x4 lo, hi;
  for (int i = 0; i < 4; ++i) { lo[i] = threadIdx.x * 10 + i; hi[i] = 1000 + threadIdx.x * 10 + i; }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const auto sw = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, lo[i]), __builtin_bit_cast(unsigned, hi[i]), false, false);
    lo[i] = __builtin_bit_cast(float, sw[0]);
    hi[i] = __builtin_bit_cast(float, sw[1]);
  }
  for (int i = 0; i < 4; ++i) { o2[threadIdx.x * 8 + i] = lo[i]; o2[threadIdx.x * 8 + 4 + i] = hi[i]; }
}
int main() {
  unsigned* d; float* d2; hipMalloc(&d, 128 * 4); hipMalloc(&d2, 512 * 4);
  k<<<1, 64>>>(d, d2);
  unsigned h[128]; float h2[512];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(h2, d2, sizeof(h2), hipMemcpyDeviceToHost);
  for (int l : {0, 1, 16, 17, 32, 48, 63}) printf("lane %2d: vdst' %3u src' %3u | lo %g %g %g %g hi %g %g %g %g\n", l, h[l], h[64 + l], h2[l*8], h2[l*8+1], h2[l*8+2], h2[l*8+3], h2[l*8+4], h2[l*8+5], h2[l*8+6], h2[l*8+7]);
  return 0;
}
