// Shared device/host helpers for librawvae_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include <stdio.h>

namespace rv {

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute: a kernel is opted in once per device of this process
// (`done`: one bit per device ordinal, a static at the call site; a lost race only repeats an idempotent call).
inline void lds_opt_in(const void* kern, int bytes, std::atomic<unsigned long long>& done) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(done.load(std::memory_order_relaxed) & bit)) {
    (void)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    done.fetch_or(bit, std::memory_order_relaxed);
  }
}


typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

typedef __attribute__((address_space(3))) char lds_char;
typedef const __attribute__((address_space(1))) void* glb_cptr;

constexpr int WAVE = 64;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Sum over a 256-thread block; result valid in thread 0.  `red` = 4 floats of LDS.
__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) red[w] = v;
  __syncthreads();
  float r = 0.f;
  if (threadIdx.x == 0) r = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  return r;
}

// 16-byte WRITE-THROUGH store (global_store_dwordx4 ... sc0 sc1): the line goes to memory as the store retires instead
// of staying dirty in the XCD's L2 until the end-of-kernel release writes everything back at once.  Every kernel of
// the step ends with a burst of output stores (8-34 MB) that nothing in the same launch reads again, and the next
// kernel runs on all XCDs, so keeping the lines in the writer's L2 buys nothing while the flush at the kernel
// boundary costs bytes / ~6 TB/s with the chip idle (MI355X_MICROARCH.md price list, row "boundary"; "publish-large").
// Measured on the whole step, same box: 190.8 -> 184.9 us (profiles/r03_ab_step.txt).  16-byte stores only: narrower
// write-through stores cost 2.7-12x per byte (same guide, stores table).
// Whether a launch uses them is the CALLER's choice (rv_store_wt, set by the training plan around its launches): a chain
// of same-shaped GEMMs whose blocks land on the XCD that holds their input rows (the deep variant's H x H layers)
// loses more from dropping the lines than it gains (deep C4 step 858 -> 951 us with write-through everywhere).
extern thread_local int rv_store_wt;   // host side, gemm_launch.hip: non-zero = epilogue outputs written through

template <typename V>
__device__ __forceinline__ void store_wt16(V* dst, const V& v) {
  static_assert(sizeof(V) == 16, "write-through stores are 16 bytes per lane");
  const f32x4 r = __builtin_bit_cast(f32x4, v);
  // the s_nop: a store of more than 64 bits reads its data registers a cycle after issue, and a VALU write to them in
  // the very next slot would be seen by the store (a hazard the compiler pads for its own stores, but it cannot see
  // inside an asm: without the pad 5 % of a 256 x 256 slab came out with elements of the NEXT store's values)
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(dst), "v"(r) : "memory");
}

// 16-byte output store: write-through when `wt` (wave-uniform), a plain store otherwise
template <typename V>
__device__ __forceinline__ void store_out16(V* dst, const V& v, const int wt) {
  if (wt) store_wt16(dst, v);
  else *dst = v;
}

// Scaled value of a post-ReLU activation on its way to its fp8 (e4m3) image: a POSITIVE activation never quantises to zero
// (it is held at the smallest subnormal, 2^-9), so the image keeps the activation's ReLU mask exactly -- the fp8 fc4
// backward reads its mask from the image (a positive byte) where the bf16 copy of h3 is not written.
__device__ __forceinline__ float fp8_keep_positive(float v, float scaled) {
  return v > 0.f ? fmaxf(scaled, 0.001953125f) : scaled;
}

// tanh(y) = 1 - 2/(1+exp(2y)); abs error ~2e-7 (v_exp_f32 and v_rcp_f32, 1 ulp each), saturates cleanly.
// v_rcp_f32 directly: __frcp_rn is the correctly rounded reciprocal, a ten-instruction sequence per element.
__device__ __forceinline__ float fast_tanh(float y) {
  const float e = __expf(2.0f * y);
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + e);
}

}  // namespace rv

// ---- host-side error plumbing (C-ABI functions return int, never throw) ----
enum {
  RV_OK = 0,
  RV_ERR_SHAPE = -1,
  RV_ERR_NULL = -2,
  RV_ERR_HIP = -3,
  RV_ERR_UNSUPPORTED = -4,
  RV_ERR_STATE = -5,
};

extern thread_local char rv_err_buf[512];
int rv_fail(int code, const char* fmt, ...);

#define RV_HIP(expr)                                                                  \
  do {                                                                                \
    hipError_t _e = (expr);                                                           \
    if (_e != hipSuccess)                                                             \
      return rv_fail(RV_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #expr,         \
                     hipGetErrorString(_e));                                          \
  } while (0)

#define RV_CHECK_LAUNCH() RV_HIP(hipGetLastError())

#define RV_REQUIRE(cond, code, ...)                        \
  do {                                                     \
    if (!(cond)) return rv_fail(code, __VA_ARGS__);        \
  } while (0)
