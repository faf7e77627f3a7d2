// Counter-based normal RNG shared by the elementwise kernels and the fused latent kernels (gfx950).
#pragma once
#include "common.h"

namespace rv {

// ------------------------------------------------------------------ Philox4x32-10
struct u32x4 { uint32_t x, y, z, w; };

__device__ __forceinline__ u32x4 philox4x32_10(uint64_t seed, uint64_t ctr_lo, uint64_t ctr_hi) {
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  uint32_t c0 = (uint32_t)ctr_lo, c1 = (uint32_t)(ctr_lo >> 32);
  uint32_t c2 = (uint32_t)ctr_hi, c3 = (uint32_t)(ctr_hi >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return {c0, c1, c2, c3};
}

// Same draw with the hardware transcendental units (v_log/v_sin/v_cos; abs error ~1e-6):
// used where eps is consumed immediately and only its distribution matters.
__device__ __forceinline__ void normal4_fast(uint64_t seed, uint64_t idx4, uint64_t offset, float* o) {
  const u32x4 r = philox4x32_10(seed, idx4, offset);
  const float inv32 = 2.3283064365386963e-10f;  // 2^-32
  const float u0 = ((float)r.x + 0.5f) * inv32, u1 = ((float)r.y + 0.5f) * inv32;
  const float u2 = ((float)r.z + 0.5f) * inv32, u3 = ((float)r.w + 0.5f) * inv32;
  const float ra = sqrtf(-2.0f * __logf(fminf(u0, 0.99999994f)));
  const float rb = sqrtf(-2.0f * __logf(fminf(u2, 0.99999994f)));
  const float t1 = 6.283185307179586f * u1, t3 = 6.283185307179586f * u3;
  o[0] = ra * __cosf(t1); o[1] = ra * __sinf(t1);
  o[2] = rb * __cosf(t3); o[3] = rb * __sinf(t3);
}

// Four N(0,1) draws for counter (idx4, offset): Box-Muller on two uniform pairs.
__device__ __forceinline__ void normal4(uint64_t seed, uint64_t idx4, uint64_t offset, float* o) {
  const u32x4 r = philox4x32_10(seed, idx4, offset);
  const float inv32 = 2.3283064365386963e-10f;  // 2^-32
  const float u0 = ((float)r.x + 0.5f) * inv32, u1 = ((float)r.y + 0.5f) * inv32;
  const float u2 = ((float)r.z + 0.5f) * inv32, u3 = ((float)r.w + 0.5f) * inv32;
  const float ra = sqrtf(-2.0f * logf(fminf(u0, 0.99999994f) + 1e-30f));
  const float rb = sqrtf(-2.0f * logf(fminf(u2, 0.99999994f) + 1e-30f));
  float s, c;
  sincospif(2.0f * u1, &s, &c);
  o[0] = ra * c; o[1] = ra * s;
  sincospif(2.0f * u3, &s, &c);
  o[2] = rb * c; o[3] = rb * s;
}

__device__ __forceinline__ float normal1(uint64_t seed, uint64_t idx, uint64_t offset) {
  float o[4];
  normal4(seed, idx >> 2, offset, o);
  return o[idx & 3];
}

}  // namespace rv
