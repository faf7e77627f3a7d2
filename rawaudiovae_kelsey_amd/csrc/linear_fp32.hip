// Exact-fp32 Linear forward for the inference surface (no-grad encode / decode and the eval
// reconstruction): y = act(x W^T + b) on v_mfma_f32_32x32x2_f32, whose result is bit-for-bit a
// k-ordered fmaf chain (one rounding per product, f32 accumulate) -- the same arithmetic class as
// the reference's fp32 `F.linear` (rawvae/model.py:19-21,28-30), so outputs agree to f32 summation
// order (~1e-6) instead of bf16 operand rounding (~4e-3).  Exact shapes, any alignment, no padding.
//
// 64x64 output tile per 256-thread block; each wave owns a 32x32 sub-tile (one f32x16 accumulator);
// K advances in steps of 16 through an LDS image [64 rows][16 k + 1 pad] of each operand.
#include "common.h"
#include "../../include/rawvae_hip.h"

namespace {
using namespace rv;

constexpr int BT = 64;   // block tile (rows of x, rows of W)
constexpr int KT = 16;   // k per LDS tile
constexpr int LDS_LD = KT + 1;

// 64 x KT tile of a row-major [rows, K] matrix -> LDS, zero-filled outside (rows, K).
__device__ __forceinline__ void stage_tile(const float* __restrict__ g, long ld, long r0, long rows, long k0, long K,
                                           float (*s)[LDS_LD], int tid, bool vec_ok) {
  const int r = tid >> 2, kq = (tid & 3) * 4;  // 256 threads x 4 consecutive k
  const long gr = r0 + r, gk = k0 + kq;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (gr < rows) {
    const float* p = g + gr * ld + gk;
    if (vec_ok && gk + 3 < K) {
      v = *reinterpret_cast<const float4*>(p);
    } else {
      if (gk + 0 < K) v.x = p[0];
      if (gk + 1 < K) v.y = p[1];
      if (gk + 2 < K) v.z = p[2];
      if (gk + 3 < K) v.w = p[3];
    }
  }
  s[r][kq + 0] = v.x;
  s[r][kq + 1] = v.y;
  s[r][kq + 2] = v.z;
  s[r][kq + 3] = v.w;
}

template <int ACT>  // 0 none, 1 relu, 2 tanh
__global__ void __launch_bounds__(256) k_linear_fp32(const float* __restrict__ x, long ldx, const float* __restrict__ w,
                                                     long ldw, const float* __restrict__ bias, long M, long N, long K,
                                                     float* __restrict__ y, long ldy, int vec_x, int vec_w) {
  __shared__ float As[BT][LDS_LD];
  __shared__ float Bs[BT][LDS_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const long m0 = (long)blockIdx.y * BT, n0 = (long)blockIdx.x * BT;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int fr = lane & 31, fk = lane >> 5;  // operand lane map: row/col = lane & 31, k = lane >> 5
  for (long k0 = 0; k0 < K; k0 += KT) {
    stage_tile(x, ldx, m0, M, k0, K, As, tid, vec_x != 0);
    stage_tile(w, ldw, n0, N, k0, K, Bs, tid, vec_w != 0);
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < KT; kk += 2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[wm * 32 + fr][kk + fk], Bs[wn * 32 + fr][kk + fk], acc, 0, 0, 0);
    __syncthreads();
  }
  // C/D map: col = lane & 31, row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5)
  const long col = n0 + wn * 32 + fr;
  if (col < N) {
    const float b = bias ? bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const long row = m0 + wm * 32 + (i & 3) + 8 * (i >> 2) + 4 * fk;
      if (row < M) {
        float v = acc[i] + b;
        if (ACT == 1) v = fmaxf(v, 0.f);
        if (ACT == 2) v = tanhf(v);
        y[row * ldy + col] = v;
      }
    }
  }
}

}  // namespace

extern "C" int rv_linear_fp32(const float* x, long ldx, const float* w, long ldw, const float* bias, long M, long N,
                              long K, int act, float* y, long ldy, void* stream) {
  RV_REQUIRE(x && w && y, RV_ERR_NULL, "rv_linear_fp32: null operand");
  RV_REQUIRE(M > 0 && N > 0 && K > 0 && ldx >= K && ldw >= K && ldy >= N, RV_ERR_SHAPE,
             "rv_linear_fp32: bad extents M=%ld N=%ld K=%ld ldx=%ld ldw=%ld ldy=%ld", M, N, K, ldx, ldw, ldy);
  RV_REQUIRE(act >= 0 && act <= 2, RV_ERR_UNSUPPORTED, "rv_linear_fp32: act %d (0 none, 1 relu, 2 tanh)", act);
  RV_REQUIRE((N + BT - 1) / BT <= 0x7fffffffL && (M + BT - 1) / BT <= 65535, RV_ERR_SHAPE,
             "rv_linear_fp32: %ld rows exceed the launch grid (65535 x 64)", M);
  const int vec_x = (((uintptr_t)x & 15) == 0 && ldx % 4 == 0) ? 1 : 0;
  const int vec_w = (((uintptr_t)w & 15) == 0 && ldw % 4 == 0) ? 1 : 0;
  dim3 grid((unsigned)((N + BT - 1) / BT), (unsigned)((M + BT - 1) / BT));
  auto st = (hipStream_t)stream;
  if (act == 0) hipLaunchKernelGGL(k_linear_fp32<0>, grid, dim3(256), 0, st, x, ldx, w, ldw, bias, M, N, K, y, ldy, vec_x, vec_w);
  if (act == 1) hipLaunchKernelGGL(k_linear_fp32<1>, grid, dim3(256), 0, st, x, ldx, w, ldw, bias, M, N, K, y, ldy, vec_x, vec_w);
  if (act == 2) hipLaunchKernelGGL(k_linear_fp32<2>, grid, dim3(256), 0, st, x, ldx, w, ldw, bias, M, N, K, y, ldy, vec_x, vec_w);
  RV_CHECK_LAUNCH();
  return RV_OK;
}
