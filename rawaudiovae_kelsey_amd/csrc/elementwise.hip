// HBM-bound kernels of the VAE training step (gfx950): operand cast/pad, on-device
// normal RNG, reparameterisation forward/backward fused with the KL term, the
// standalone single-kernel loss (MSE + KL + their gradients), and the fused
// multi-tensor Adam / gradient finaliser.  C ABI: include/rawvae_hip.h.
#include "common.h"
#include "adam.h"
#include "philox.h"
#include "../../include/rawvae_hip.h"
#include "internal.h"

#include <stdarg.h>
#include <string.h>

using namespace rv;

thread_local char rv_err_buf[512] = "";

int rv_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(rv_err_buf, sizeof(rv_err_buf), fmt, ap);
  va_end(ap);
  return code;
}

namespace {

__global__ void __launch_bounds__(256) k_randn(float* out, long n, uint64_t seed, uint64_t offset) {
  const long n4 = (n + 3) / 4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float o[4];
    normal4(seed, (uint64_t)i, offset, o);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (i * 4 + j < n) out[i * 4 + j] = o[j];
  }
}

// ------------------------------------------------------------------ cast + pad
// One thread per 8 output bf16 (16 B store).  Source rows are read as two float4
// when in range and aligned, scalar at the ragged edge.
// fp32 x 8 -> 8 fp8 (e4m3) bytes of v * sc, saturating
__device__ __forceinline__ unsigned long long q8x8(const float (&v)[8], float sc) {
  float c[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) c[e] = fminf(fmaxf(v[e] * sc, -448.f), 448.f);
  unsigned lo = 0, hi = 0;
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(c[0], c[1], lo, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(c[2], c[3], lo, true);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(c[4], c[5], hi, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(c[6], c[7], hi, true);
  return ((unsigned long long)hi << 32) | lo;
}

// fp8 state block (RV_OPT_FP8): block 0 (256 threads) of the step's first kernel latches the delayed h3 scale and moves
// the weight scales after the weights.  max|W1|, max|W4| of the last optimizer update sit in 2 x 1024 slots behind the
// 32 floats of the state block proper (k_fp8_wmax below fills them behind the optimizer); they are reduced with every
// load of a thread in flight at once and reset here.
constexpr int FP8_WSLOTS = 1024;
__device__ __forceinline__ void fp8_latch_block(float* st, const float* amax_part, int n_amax, int n_amax2) {
  __shared__ float red[4][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* w_amax = st + 32;
  float a[FP8_WSLOTS / 256], b[FP8_WSLOTS / 256];
#pragma unroll
  for (int k = 0; k < FP8_WSLOTS / 256; ++k) {
    a[k] = w_amax[tid + 256 * k];
    b[k] = w_amax[FP8_WSLOTS + tid + 256 * k];
  }
  float m = 0.f, w1 = 0.f, w4 = 0.f, g1 = 0.f;
  for (int i = tid; i < n_amax; i += 256) m = fmaxf(m, amax_part[i]);
  for (int i = tid; i < n_amax2; i += 256) g1 = fmaxf(g1, amax_part[n_amax + i]);   // max|dP1| of the previous step
#pragma unroll
  for (int k = 0; k < FP8_WSLOTS / 256; ++k) {
    w1 = fmaxf(w1, a[k]);
    w4 = fmaxf(w4, b[k]);
    w_amax[tid + 256 * k] = 0.f;
    w_amax[FP8_WSLOTS + tid + 256 * k] = 0.f;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    m = fmaxf(m, __shfl_xor(m, o, 64));
    w1 = fmaxf(w1, __shfl_xor(w1, o, 64));
    w4 = fmaxf(w4, __shfl_xor(w4, o, 64));
    g1 = fmaxf(g1, __shfl_xor(g1, o, 64));
  }
  if (lane == 0) { red[0][wave] = m; red[1][wave] = w1; red[2][wave] = w4; red[3][wave] = g1; }
  __syncthreads();
  if (tid == 0) {
    m = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
    w1 = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
    w4 = fmaxf(fmaxf(red[2][0], red[2][1]), fmaxf(red[2][2], red[2][3]));
    g1 = fmaxf(fmaxf(red[3][0], red[3][1]), fmaxf(red[3][2], red[3][3]));
    // the whole block in registers first: read entry by entry between the writes below, every access would be a round
    // trip of its own (the compiler cannot move a load of st[] over a store to st[]) -- ~4 us at the head of the step
    float s[16];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float4 v = reinterpret_cast<const float4*>(st)[k];
      s[4 * k] = v.x; s[4 * k + 1] = v.y; s[4 * k + 2] = v.z; s[4 * k + 3] = v.w;
    }
    const bool live = s[7] == 0.f;
    s[4] = m;
    if (live && m > 0.f) s[3] = 224.f / m;
    s[5] = 1.f / (s[0] * s[1]);
    s[6] = 1.f / (s[3] * s[2]);
    // fp8 backward of fc4: dP4's image is written with the fixed scale [12] (set by the caller: 112 / (2 / (B S)), so
    // that |dP4| <= 2 * 2 / (B S) lands within +-224); the dgrad multiplies it with W4's shadow, the wgrad with h3's image
    s[10] = 1.f / (s[12] * s[2]);
    s[11] = 1.f / (s[12] * s[3]);
    // fp8 weight gradient of fc1: the heads' backward writes dP1's image with [13], which follows the maximum it
    // measured in the previous step (delayed scaling, as h3's); the GEMM multiplies it with x's image
    if (n_amax2 > 0) {
      s[14] = g1;
      if (live && g1 > 0.f) s[13] = 224.f / g1;
    }
    s[15] = 1.f / (s[13] * s[0]);
    // the weight shadows read by this step were written with [1] / [2] (now inside [5] / [6]); the coming optimizer
    // update quantises with scales that follow the weights it last saw
    s[8] = w1;
    s[9] = w4;
    if (live) {
      if (w1 > 0.f) s[1] = 224.f / w1;
      if (w4 > 0.f) s[2] = 224.f / w4;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) reinterpret_cast<float4*>(st)[k] = make_float4(s[4 * k], s[4 * k + 1], s[4 * k + 2], s[4 * k + 3]);
  }
}

// max|W| of the two fp8 weight shadows as the optimizer left them, for next step's scales: 2 x 128 blocks, each a slice of
// one shadow (first half of the grid: W1q, second half: W4q), max|q| / scale of the slice into the block's slot.  An
// extra pass over 4 MB (~2 us with its launch) instead of a reduction inside the optimizer kernels, where one atomic per
// wave on 64 cache lines cost 6 us and the loads to avoid them more.
constexpr int WMAX_BLOCKS = 128;   // per tensor: 4 independent 16-byte loads per thread at C2 (2 MB shadows)
__global__ void __launch_bounds__(256) k_fp8_wmax(const unsigned char* __restrict__ w1q, const long n1,
                                                  const unsigned char* __restrict__ w4q, const long n4, float* __restrict__ st) {
  __shared__ float red[4];
  const bool second = (int)blockIdx.x >= WMAX_BLOCKS;
  const unsigned char* q = second ? w4q : w1q;
  const long n = second ? n4 : n1;
  const int b = (int)blockIdx.x - (second ? WMAX_BLOCKS : 0);
  const float inv_scale = 1.f / st[second ? 2 : 1];   // (requested with the data, not behind the reduction)
  float m = 0.f;
  constexpr long STRIDE = (long)WMAX_BLOCKS * 256 * 16;
  for (long i = ((long)b * 256 + threadIdx.x) * 16; i + 16 <= n; i += 4 * STRIDE) {
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)   // four loads in flight (a slice past the end reads the first one again: same maximum)
      v[u] = *reinterpret_cast<const uint4*>(q + (i + u * STRIDE + 16 <= n ? i + u * STRIDE : i));
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const unsigned w[4] = {v[u].x & 0x7F7F7F7Fu, v[u].y & 0x7F7F7F7Fu, v[u].z & 0x7F7F7F7Fu, v[u].w & 0x7F7F7F7Fu};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        m = fmaxf(m, __builtin_amdgcn_cvt_f32_fp8((int)w[k], 0));
        m = fmaxf(m, __builtin_amdgcn_cvt_f32_fp8((int)w[k], 1));
        m = fmaxf(m, __builtin_amdgcn_cvt_f32_fp8((int)w[k], 2));
        m = fmaxf(m, __builtin_amdgcn_cvt_f32_fp8((int)w[k], 3));
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    // block b of a tensor owns slot b of that tensor's 1024 (the rest stay zero)
    st[32 + (second ? FP8_WSLOTS : 0) + b] = m * inv_scale;
  }
}

// store policy of the cast's bf16 output: the plan's (common.h rv_store_wt; A/B: profiles/r05_ab_cast_wt.txt)
static inline int cast_wt() { return rv::rv_store_wt; }

__global__ void __launch_bounds__(256) k_cast_pad_bf16(const float* __restrict__ src, long rows,
                                                       long cols, long ld_src,
                                                       bf16_t* __restrict__ dst, long rows_p,
                                                       long cols_p, long ld_dst,
                                                       long long* step_counter,
                                                       unsigned char* __restrict__ dst_fp8, long ld_fp8,
                                                       float* fp8_state, const float* __restrict__ fp8_scale,
                                                       const float* __restrict__ amax_part, int n_amax, int n_amax2,
                                                       const long long* __restrict__ frame_idx, long first_frame,
                                                       long hop, long n_samples, const int wt) {
  if (blockIdx.x == 0) {
    if (step_counter && threadIdx.x == 0) *step_counter += 1;
    if (fp8_state) {
      // with the fp8 state block the launch has one block more than the cast needs: this one only latches (a chain of
      // dependent loads, a reduction and a write-back: ~2 us that would otherwise sit in front of a share of the cast)
      fp8_latch_block(fp8_state, amax_part, n_amax, n_amax2);
      return;
    }
  }
  const float qs = dst_fp8 ? *fp8_scale : 0.f;   // the x / weight scale is constant across the latch
  const long cpr = cols_p / 8;
  const long total = rows_p * cpr;
  const bool vec_ok = (ld_src % 4 == 0) && ((reinterpret_cast<uintptr_t>(src) & 15) == 0);
  const long blk = fp8_state ? (long)blockIdx.x - 1 : (long)blockIdx.x, nblk = fp8_state ? (long)gridDim.x - 1 : (long)gridDim.x;
  for (long i = blk * 256 + threadIdx.x; i < total; i += nblk * 256) {
    const unsigned r32 = (unsigned)i / (unsigned)cpr;   // total < 2^31 (checked by the launchers)
    const long r = r32, c = (long)((unsigned)i - r32 * (unsigned)cpr) * 8;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
    if (r < rows && hop) {
      // row r is frame f of the resident waveform `src` (AudioDataset.__getitem__, dataset.py:108-118)
      const long start = (frame_idx ? (long)frame_idx[r] : first_frame + r) * hop + c;
      if ((hop & 3) == 0 && ((reinterpret_cast<uintptr_t>(src) & 15) == 0) && c + 8 <= cols && start >= 0 && start + 8 <= n_samples) {
        const float4 a = *reinterpret_cast<const float4*>(src + start);
        const float4 b = *reinterpret_cast<const float4*>(src + start + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const long a_ = start + j;
          if (c + j < cols && a_ >= 0 && a_ < n_samples) v[j] = src[a_];
        }
      }
    } else if (r < rows) {
      const float* s = src + r * ld_src + c;
      if (vec_ok && c + 8 <= cols) {
        const float4 a = *reinterpret_cast<const float4*>(s);
        const float4 b = *reinterpret_cast<const float4*>(s + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (c + j < cols) v[j] = s[j];
      }
    }
    if (dst) {
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (bf16_t)v[j];
      // (inside a step plan the operand is written through like every other output of the step: the next launch reads it
      // on all XCDs, and left dirty its 8.4 MB are flushed at the kernel boundary with the chip idle)
      store_out16(reinterpret_cast<bf16x8*>(dst + r * ld_dst + c), o, wt);
    }
    if (dst_fp8) *reinterpret_cast<unsigned long long*>(dst_fp8 + r * ld_fp8 + c) = q8x8(v, qs);
  }
}

// ------------------------------------------------------------------ reparam forward + KL
// One thread per 4 consecutive latent columns of the padded [Bp, Lp] grid (float4 slab reads,
// four slabs in flight).
__global__ void __launch_bounds__(256)
k_reparam_fwd(const float* __restrict__ slabs, int splits, long Bp, long Lp, long B, long L,
              const float* __restrict__ eps_in, float* __restrict__ eps_out, uint64_t seed,
              const long long* __restrict__ step_counter, float* __restrict__ mulv,
              bf16_t* __restrict__ z, float* __restrict__ kl_partial) {
  __shared__ float red[4];
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long lq = Lp / 4;
  const unsigned b32 = (unsigned)i / (unsigned)lq;      // Bp * Lp / 4 < 2^31 (checked by the launcher)
  const long b = b32, l = (long)((unsigned)i - b32 * (unsigned)lq) * 4;
  const long L2p = 2 * Lp;
  float kl = 0.f;
  if (b < Bp) {
    float4 mu = make_float4(0.f, 0.f, 0.f, 0.f), lv = mu;
    float zz[4] = {0.f, 0.f, 0.f, 0.f};
    if (b < B && l < L) {
      const float* base = slabs + b * L2p + l;
      const long ss = Bp * L2p;
      int s = 0;
      for (; s + 4 <= splits; s += 4) {
        float4 m[4], v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          m[u] = *reinterpret_cast<const float4*>(base + (s + u) * ss);
          v[u] = *reinterpret_cast<const float4*>(base + (s + u) * ss + Lp);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          mu.x += m[u].x; mu.y += m[u].y; mu.z += m[u].z; mu.w += m[u].w;
          lv.x += v[u].x; lv.y += v[u].y; lv.z += v[u].z; lv.w += v[u].w;
        }
      }
      for (; s < splits; ++s) {
        const float4 m = *reinterpret_cast<const float4*>(base + s * ss);
        const float4 v = *reinterpret_cast<const float4*>(base + s * ss + Lp);
        mu.x += m.x; mu.y += m.y; mu.z += m.z; mu.w += m.w;
        lv.x += v.x; lv.y += v.y; lv.z += v.z; lv.w += v.w;
      }
      float mua[4] = {mu.x, mu.y, mu.z, mu.w}, lva[4] = {lv.x, lv.y, lv.z, lv.w};
      float ev[4];
      if (!eps_in) normal4_fast(seed, (uint64_t)i, step_counter ? (uint64_t)*step_counter : 0, ev);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (l + j < L) {
          float e;
          if (eps_in) {
            e = eps_in[b * L + l + j];
          } else {
            e = ev[j];
            eps_out[b * L + l + j] = e;
          }
          const float sd = __expf(0.5f * lva[j]);
          zz[j] = mua[j] + e * sd;
          kl += 1.f + lva[j] - mua[j] * mua[j] - sd * sd;
        } else {
          mua[j] = 0.f;
          lva[j] = 0.f;
        }
      }
      mu = make_float4(mua[0], mua[1], mua[2], mua[3]);
      lv = make_float4(lva[0], lva[1], lva[2], lva[3]);
    }
    *reinterpret_cast<float4*>(mulv + b * L2p + l) = mu;
    *reinterpret_cast<float4*>(mulv + b * L2p + Lp + l) = lv;
    const bf16x4 zb = {(bf16_t)zz[0], (bf16_t)zz[1], (bf16_t)zz[2], (bf16_t)zz[3]};
    *reinterpret_cast<bf16x4*>(z + b * Lp + l) = zb;
  }
  const float s_ = block_sum_256(kl, red);
  if (threadIdx.x == 0) kl_partial[blockIdx.x] = s_;
}

// ------------------------------------------------------------------ reparam backward (+ loss finish)
// Block = RB_ROWS batch rows x all Lp columns, one thread per 4 consecutive columns of one row
// (float4 loads, all slabs in flight); rows of the block are covered in 256*4/Lp-row passes.
// Column sums (bias grads of fc21|fc22) are reduced through LDS into one partial row per block.
// One extra grid block finishes the loss scalar.
constexpr int RB_ROWS = 16;

__global__ void __launch_bounds__(256)
k_reparam_bwd(const float* __restrict__ dz_slabs, int splits, long Bp, long Lp, long B, long L,
              long S, const float* __restrict__ mulv, const float* __restrict__ eps, float kl_beta,
              bf16_t* __restrict__ dmulv, float* __restrict__ dbh_partial,
              const float* __restrict__ mse_partial, int n_mse,
              const float* __restrict__ kl_partial, int n_kl, float* __restrict__ loss_out,
              const long long* __restrict__ step_counter, int ring,
              const float* __restrict__ dmu_ext, const float* __restrict__ dlv_ext) {
  __shared__ float sh[2 * 256 * 4];
  const int tid = threadIdx.x;
  const long L2p = 2 * Lp;
  const float inv_nk_ = 1.0f / ((float)B * (float)L);
  // ONE EXTRA block (the grid has Bp / RB_ROWS + 1) finishes the loss scalar and does nothing else: as an epilogue of
  // the last row block it added a second dependent round trip to memory to the block the whole launch waits for
  if (blockIdx.x == gridDim.x - 1) {
    if (loss_out && mse_partial && kl_partial) {
      float m = 0.f, k = 0.f;
      for (int i = tid; i < n_mse; i += 256) m += mse_partial[i];
      for (int i = tid; i < n_kl; i += 256) k += kl_partial[i];
      m = block_sum_256(m, sh);
      k = block_sum_256(k, sh);
      if (tid == 0) {
        const float mse = m / ((float)B * (float)S);
        const float kld = -0.5f * k * inv_nk_;
        if (step_counter && ring > 0) loss_out += 4 * ((*step_counter - 1) % ring);
        loss_out[0] = mse + kl_beta * kld;
        loss_out[1] = mse;
        loss_out[2] = kld;
      }
    }
    return;
  }
  const int lq = (int)(Lp / 4);             // column groups per row (16, 32 or 64)
  const int rows_par = 256 / lq;            // rows covered per pass (16, 8 or 4)
  const int cg = tid % lq, r0 = tid / lq;
  const long l = (long)cg * 4;
  const float inv_nk = 1.0f / ((float)B * (float)L);
  float cs_mu[4] = {0.f, 0.f, 0.f, 0.f}, cs_lv[4] = {0.f, 0.f, 0.f, 0.f};
  for (int r = r0; r < RB_ROWS; r += rows_par) {
    const long b = (long)blockIdx.x * RB_ROWS + r;
    float dmu[4] = {0.f, 0.f, 0.f, 0.f}, dlv[4] = {0.f, 0.f, 0.f, 0.f};
    if (b < B && l < L) {
      float4 dz = make_float4(0.f, 0.f, 0.f, 0.f);
      const float* zb = dz_slabs + b * Lp + l;
      for (int s = 0; s < splits; ++s) {
        const float4 t = *reinterpret_cast<const float4*>(zb + (long)s * Bp * Lp);
        dz.x += t.x; dz.y += t.y; dz.z += t.z; dz.w += t.w;
      }
      const float4 mu4 = *reinterpret_cast<const float4*>(mulv + b * L2p + l);
      const float4 lv4 = *reinterpret_cast<const float4*>(mulv + b * L2p + Lp + l);
      const float dza[4] = {dz.x, dz.y, dz.z, dz.w}, mua[4] = {mu4.x, mu4.y, mu4.z, mu4.w};
      const float lva[4] = {lv4.x, lv4.y, lv4.z, lv4.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (l + j < L) {
          const float e = eps[b * L + l + j];
          const float sd = __expf(0.5f * lva[j]);
          dmu[j] = dza[j] + kl_beta * mua[j] * inv_nk;
          dlv[j] = dza[j] * e * 0.5f * sd + kl_beta * 0.5f * (sd * sd - 1.f) * inv_nk;
          if (dmu_ext) dmu[j] += dmu_ext[b * L + l + j];   // gradients arriving from outside (autograd)
          if (dlv_ext) dlv[j] += dlv_ext[b * L + l + j];
        }
      }
    }
    const bf16x4 m4 = {(bf16_t)dmu[0], (bf16_t)dmu[1], (bf16_t)dmu[2], (bf16_t)dmu[3]};
    const bf16x4 v4 = {(bf16_t)dlv[0], (bf16_t)dlv[1], (bf16_t)dlv[2], (bf16_t)dlv[3]};
    *reinterpret_cast<bf16x4*>(dmulv + b * L2p + l) = m4;
    *reinterpret_cast<bf16x4*>(dmulv + b * L2p + Lp + l) = v4;
#pragma unroll
    for (int j = 0; j < 4; ++j) { cs_mu[j] += dmu[j]; cs_lv[j] += dlv[j]; }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    sh[tid * 4 + j] = cs_mu[j];
    sh[1024 + tid * 4 + j] = cs_lv[j];
  }
  __syncthreads();
  if (dbh_partial && tid < Lp) {
    // column tid = group (tid/4), element (tid%4); sum over the rows_par row-lanes in a fixed order
    const int g = tid / 4, j = tid % 4;
    float a = 0.f, c = 0.f;
    for (int q = 0; q < rows_par; ++q) {
      a += sh[(q * lq + g) * 4 + j];
      c += sh[1024 + (q * lq + g) * 4 + j];
    }
    dbh_partial[(long)blockIdx.x * L2p + tid] = a;
    dbh_partial[(long)blockIdx.x * L2p + Lp + tid] = c;
  }
}

// ------------------------------------------------------------------ standalone fused loss
// loss_function (model.py:38-47) in one launch.  Every block reduces its slice with
// wave shuffles, publishes two partials, and the last block to arrive (agent-scope
// release / ticket / acquire) sums the partials in index order, so the result does not
// depend on arrival order.  Workspace: [0] ticket (u32), [64..] partials.
constexpr int LOSS_MAX_BLOCKS = 1024;

__global__ void __launch_bounds__(256)
k_loss_fused(const float* __restrict__ recon, const float* __restrict__ x,
             const float* __restrict__ mu, const float* __restrict__ lv, long n_r, long n_k,
             float kl_beta, float* __restrict__ loss_out, float* __restrict__ d_recon,
             float* __restrict__ d_mu, float* __restrict__ d_lv, unsigned* ws) {
  __shared__ float red[4];
  __shared__ unsigned is_last;
  float* part = reinterpret_cast<float*>(ws + 64);
  const float g_r = 2.0f / (float)n_r;
  const float g_k = kl_beta / (float)n_k;
  const long stride = (long)gridDim.x * 256;
  const long t0 = (long)blockIdx.x * 256 + threadIdx.x;
  float sq = 0.f, kl = 0.f;
  const bool vec = ((n_r & 3) == 0) &&
                   (((reinterpret_cast<uintptr_t>(recon) | reinterpret_cast<uintptr_t>(x) |
                      reinterpret_cast<uintptr_t>(d_recon)) & 15) == 0);
  if (vec) {
    const long n4 = n_r >> 2;
    for (long i = t0; i < n4; i += stride) {
      const float4 r = reinterpret_cast<const float4*>(recon)[i];
      const float4 t = reinterpret_cast<const float4*>(x)[i];
      const float4 d = make_float4(r.x - t.x, r.y - t.y, r.z - t.z, r.w - t.w);
      sq += d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w;
      if (d_recon)
        reinterpret_cast<float4*>(d_recon)[i] = make_float4(g_r * d.x, g_r * d.y, g_r * d.z, g_r * d.w);
    }
  } else {
    for (long i = t0; i < n_r; i += stride) {
      const float d = recon[i] - x[i];
      sq += d * d;
      if (d_recon) d_recon[i] = g_r * d;
    }
  }
  for (long i = t0; i < n_k; i += stride) {
    const float m = mu[i], l = lv[i];
    const float e = __expf(l);
    kl += 1.f + l - m * m - e;
    if (d_mu) d_mu[i] = g_k * m;
    if (d_lv) d_lv[i] = 0.5f * g_k * (e - 1.f);
  }
  sq = block_sum_256(sq, red);
  kl = block_sum_256(kl, red);
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x] = sq;
    part[2 * blockIdx.x + 1] = kl;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned t = __hip_atomic_fetch_add(ws, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    is_last = (t == gridDim.x - 1);
    if (is_last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
  if (is_last) {
    float a = 0.f, b = 0.f;
    for (int i = threadIdx.x; i < (int)gridDim.x; i += 256) {
      a += part[2 * i];
      b += part[2 * i + 1];
    }
    a = block_sum_256(a, red);
    b = block_sum_256(b, red);
    if (threadIdx.x == 0) {
      const float mse = a / (float)n_r;
      const float kld = -0.5f * b / (float)n_k;
      loss_out[0] = mse + kl_beta * kld;
      loss_out[1] = mse;
      loss_out[2] = kld;
      __hip_atomic_store(ws, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm
    }
  }
}

__global__ void __launch_bounds__(256)
k_reparameterize(const float* __restrict__ mu, const float* __restrict__ lv, long n,
                 const float* __restrict__ eps_in, float* __restrict__ eps_out, uint64_t seed,
                 uint64_t offset, float* __restrict__ z) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float e;
    if (eps_in) {
      e = eps_in[i];
    } else {
      e = normal1(seed, (uint64_t)i, offset);
      if (eps_out) eps_out[i] = e;
    }
    z[i] = mu[i] + e * __expf(0.5f * lv[i]);
  }
}

// ------------------------------------------------------------------ API-path helpers
// dP4 = d_recon * (1 - recon^2) -> zero-padded bf16 (backward of F.tanh, model.py:30).
__global__ void __launch_bounds__(256)
k_tanh_bwd_pack(const float* __restrict__ d_recon, const float* __restrict__ recon, long B, long S,
                bf16_t* __restrict__ out, long Bp, long Sp) {
  const long total = Bp * Sp;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / Sp, c = i % Sp;
    float v = 0.f;
    if (r < B && c < S) {
      const float y = recon[r * S + c];
      v = d_recon[r * S + c] * (1.f - y * y);
    }
    out[i] = (bf16_t)v;
  }
}

// Partial column sums: block (cx, ry) sums rows [256*ry, 256*ry+256) of columns
// [64*cx, 64*cx+64) -> out[ry][col].  Deterministic; finished by rv_grad_finalize.
template <typename T>
__global__ void __launch_bounds__(256)
k_colsum_partial(const T* __restrict__ src, long rows, long cols, long ld, float* __restrict__ out,
                 long ld_out) {
  __shared__ float sh[256];
  const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
  const long col = (long)blockIdx.x * 64 + c;
  const long r0 = (long)blockIdx.y * 256;
  float s = 0.f;
  if (col < cols)
    for (long r = r0 + q; r < r0 + 256 && r < rows; r += 4) s += (float)src[r * ld + col];
  sh[threadIdx.x] = s;
  __syncthreads();
  if (q == 0 && col < cols) out[(long)blockIdx.y * ld_out + col] = sh[c] + sh[64 + c] + sh[128 + c] + sh[192 + c];
}

// Backward of z = mu + eps*exp(logvar/2) (model.py:23-26): dmu = dz, dlv = dz*eps*std/2.
__global__ void __launch_bounds__(256)
k_reparameterize_bwd(const float* __restrict__ dz, const float* __restrict__ eps,
                     const float* __restrict__ lv, long n, float* __restrict__ dmu,
                     float* __restrict__ dlv) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float g = dz[i];
    if (dmu) dmu[i] = g;
    if (dlv) dlv[i] = g * eps[i] * 0.5f * __expf(0.5f * lv[i]);
  }
}

// fp32 elementwise steps of the strict-fp32 training mode (strict.py): the activations' backward and sums,
// exactly as autograd computes them for model.py:20,29,30 (threshold_backward, tanh_backward, add).
//   op 0: out = a * (1 - b*b)      d(pre-tanh) from d(recon), recon
//   op 1: out = b > 0 ? a : 0      ReLU backward from d(out), out
//   op 2: out = a + b
__global__ void __launch_bounds__(256)
k_ew_f32(int op, const float* __restrict__ a, const float* __restrict__ b, long n, float* __restrict__ out) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float x = a[i], y = b[i];
    out[i] = op == 0 ? x * (1.f - y * y) : op == 1 ? (y > 0.f ? x : 0.f) : x + y;
  }
}

// out = a * scalar[0] (upstream gradient of the 0-dim loss applied to a saved gradient).
__global__ void __launch_bounds__(256)
k_scale_by(const float* __restrict__ a, const float* __restrict__ scalar, long n, float* __restrict__ out) {
  const float g = scalar[0];
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = a[i] * g;
}

// The same for up to three tensors in ONE launch (the three gradients loss_function's backward hands on: d_recon,
// d_mu, d_logvar): the API path is bound by host time per launch, not by these bytes.
__device__ __forceinline__ void scale_span(const float* __restrict__ a, float* __restrict__ out, long n, float g) {
  const long i0 = (long)blockIdx.x * 256 + threadIdx.x, stride = (long)gridDim.x * 256;
  if ((((uintptr_t)a | (uintptr_t)out) & 15) == 0) {
    const long n4 = n >> 2;
    for (long i = i0; i < n4; i += stride) {
      float4 v = reinterpret_cast<const float4*>(a)[i];
      v.x *= g; v.y *= g; v.z *= g; v.w *= g;
      reinterpret_cast<float4*>(out)[i] = v;
    }
    for (long i = (n4 << 2) + i0; i < n; i += stride) out[i] = a[i] * g;
  } else {
    for (long i = i0; i < n; i += stride) out[i] = a[i] * g;
  }
}
__global__ void __launch_bounds__(256)
k_scale_by3(const float* __restrict__ a0, float* __restrict__ o0, long n0, const float* __restrict__ a1,
            float* __restrict__ o1, long n1, const float* __restrict__ a2, float* __restrict__ o2, long n2,
            const float* __restrict__ scalar) {
  const float g = scalar[0];
  if (n0) scale_span(a0, o0, n0, g);
  if (n1) scale_span(a1, o1, n1, g);
  if (n2) scale_span(a2, o2, n2, g);
}

// ------------------------------------------------------------------ hop-strided framing (N1)
// frame i = audio[idx[i]*hop : idx[i]*hop + S]  (AudioDataset.__getitem__, dataset.py:108-118;
// idx == NULL -> consecutive frames first_frame + i).  The waveform stays resident in HBM;
// frames overlap S/hop-fold, so the gather is served from L2.
__global__ void __launch_bounds__(256)
k_gather_frames(const float* __restrict__ audio, long n_samples, const long long* __restrict__ idx,
                long first_frame, long n_frames, long S, long hop, float* __restrict__ out) {
  const long per = (S + 3) / 4;
  const long total = n_frames * per;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
    const long f = t / per, c = (t % per) * 4;
    const long start = (idx ? (long)idx[f] : first_frame + f) * hop + c;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long a = start + j;
      v[j] = (c + j < S && a >= 0 && a < n_samples) ? audio[a] : 0.f;
    }
    float* o = out + f * S + c;
    if (c + 4 <= S && ((S & 3) == 0)) *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
    else
      for (int j = 0; j < 4 && c + j < S; ++j) o[j] = v[j];
  }
}


__global__ void __launch_bounds__(256)
k_params_from_flat(const DescTable tab, const float* flat, const long flat_base, float* param) {
  refresh_block(tab, (long)blockIdx.x, (int)threadIdx.x, flat, flat_base, param);
}

inline unsigned grid_for(long n_threads, long cap = 2048) {
  long g = (n_threads + 255) / 256;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (unsigned)g;
}


}  // namespace

extern "C" {

int rv_version(void) { return 100; }
const char* rv_last_error(void) { return rv_err_buf; }

int rv_pad_dims(long B, long S, long H, long L, long* Bp, long* Sp, long* Hp, long* Lp) {
  RV_REQUIRE(B > 0 && S > 0 && H > 0 && L > 0, RV_ERR_SHAPE, "rv_pad_dims: non-positive extent");
  RV_REQUIRE(L <= 256, RV_ERR_UNSUPPORTED, "latent_dim %ld > 256 not supported", L);
  if (Bp) *Bp = (B + 127) / 128 * 128;
  if (Sp) *Sp = (S + 127) / 128 * 128;
  if (Hp) *Hp = (H + 127) / 128 * 128;
  if (Lp) {
    long lp = 64;
    while (lp < L) lp *= 2;  // 64, 128, 256: keeps 256 % Lp == 0 for the reparam kernels
    *Lp = lp;
  }
  return RV_OK;
}

int rv_cast_pad_bf16(const float* src, long rows, long cols, long ld_src, void* dst, long rows_p,
                     long cols_p, long ld_dst, long long* step_counter, void* stream) {
  RV_REQUIRE(src && dst, RV_ERR_NULL, "rv_cast_pad_bf16: null pointer");
  RV_REQUIRE(rows >= 0 && cols >= 0 && rows <= rows_p && cols <= cols_p && cols_p % 8 == 0 && ld_src >= cols &&
                 ld_dst >= cols_p && ld_dst % 8 == 0 && ((uintptr_t)dst & 15) == 0,
             RV_ERR_SHAPE, "rv_cast_pad_bf16: bad extents %ld %ld -> %ld %ld (ld %ld)", rows, cols, rows_p, cols_p, ld_dst);
  const long total = rows_p * (cols_p / 8);
  RV_REQUIRE(total < 0x7fffffffL, RV_ERR_SHAPE, "cast: %ld x %ld is too large for one launch", rows_p, cols_p);
  hipLaunchKernelGGL(k_cast_pad_bf16, dim3(grid_for(total, 8192)), dim3(256), 0, (hipStream_t)stream,
                     src, rows, cols, ld_src, (bf16_t*)dst, rows_p, cols_p, ld_dst, step_counter,
                     (unsigned char*)nullptr, 0L, (float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0, 0,
                     (const long long*)nullptr, 0L, 0L, 0L, cast_wt());
  RV_CHECK_LAUNCH();
  return RV_OK;
}

int rv_cast_pad_bf16_q8(const float* src, long rows, long cols, long ld_src, void* dst, long rows_p, long cols_p,
                        long ld_dst, void* dst_fp8, long ld_fp8, float* fp8_state, const float* amax_part, int n_amax,
                        int n_amax2, long long* step_counter, void* stream) {
  RV_REQUIRE(src && (dst || dst_fp8), RV_ERR_NULL, "rv_cast_pad_bf16_q8: null pointer");
  RV_REQUIRE(rows >= 0 && cols >= 0 && rows <= rows_p && cols <= cols_p && cols_p % 8 == 0 && ld_src >= cols &&
                 (!dst || (ld_dst >= cols_p && ld_dst % 8 == 0)) && ((uintptr_t)dst & 15) == 0,
             RV_ERR_SHAPE, "rv_cast_pad_bf16_q8: bad extents %ld %ld -> %ld %ld (ld %ld)", rows, cols, rows_p, cols_p, ld_dst);
  RV_REQUIRE(!dst_fp8 || (fp8_state && ld_fp8 >= cols_p && ld_fp8 % 8 == 0 && ((uintptr_t)dst_fp8 & 7) == 0), RV_ERR_SHAPE,
             "rv_cast_pad_bf16_q8: the fp8 output needs the state block and 8-byte aligned rows");
  const long total = rows_p * (cols_p / 8);
  RV_REQUIRE(total < 0x7fffffffL, RV_ERR_SHAPE, "cast: %ld x %ld is too large for one launch", rows_p, cols_p);
  hipLaunchKernelGGL(k_cast_pad_bf16, dim3(grid_for(total, 8192) + (fp8_state ? 1 : 0)), dim3(256), 0, (hipStream_t)stream,
                     src, rows, cols, ld_src, (bf16_t*)dst, rows_p, cols_p, ld_dst, step_counter,
                     (unsigned char*)dst_fp8, ld_fp8, fp8_state, (const float*)fp8_state /* [0] = scale of x */,
                     amax_part, amax_part ? n_amax : 0, amax_part ? n_amax2 : 0, (const long long*)nullptr, 0L, 0L, 0L, cast_wt());
  RV_CHECK_LAUNCH();
  return RV_OK;
}

int rv_gather_cast_frames(const float* audio, long n_samples, const long long* frame_index, long first_frame, long n_frames,
                          long S, long hop, void* dst_bf16, long rows_p, long cols_p, long ld_dst, void* dst_fp8, long ld_fp8,
                          float* fp8_state, const float* amax_part, int n_amax, int n_amax2, long long* step_counter,
                          void* stream) {
  RV_REQUIRE(audio && (dst_bf16 || dst_fp8), RV_ERR_NULL, "rv_gather_cast_frames: null pointer");
  RV_REQUIRE(n_samples > 0 && n_frames >= 0 && S > 0 && hop > 0 && n_frames <= rows_p && S <= cols_p && cols_p % 8 == 0 &&
                 (!dst_bf16 || (ld_dst >= cols_p && ld_dst % 8 == 0)) && ((uintptr_t)dst_bf16 & 15) == 0,
             RV_ERR_SHAPE, "rv_gather_cast_frames: bad extents %ld frames of %ld -> %ld x %ld", n_frames, S, rows_p, cols_p);
  RV_REQUIRE(!dst_fp8 || (fp8_state && ld_fp8 >= cols_p && ld_fp8 % 8 == 0 && ((uintptr_t)dst_fp8 & 7) == 0), RV_ERR_SHAPE,
             "rv_gather_cast_frames: the fp8 output needs the state block and 8-byte aligned rows");
  const long total = rows_p * (cols_p / 8);
  RV_REQUIRE(total < 0x7fffffffL, RV_ERR_SHAPE, "cast: %ld x %ld is too large for one launch", rows_p, cols_p);
  hipLaunchKernelGGL(k_cast_pad_bf16, dim3(grid_for(total, 8192) + (fp8_state ? 1 : 0)), dim3(256), 0, (hipStream_t)stream,
                     audio, n_frames, S, 0L, (bf16_t*)dst_bf16, rows_p, cols_p, ld_dst, step_counter,
                     (unsigned char*)dst_fp8, ld_fp8, fp8_state, (const float*)fp8_state, amax_part,
                     amax_part ? n_amax : 0, amax_part ? n_amax2 : 0, frame_index, first_frame, hop, n_samples, cast_wt());
  RV_CHECK_LAUNCH();
  return RV_OK;
}

int rv_fp8_wmax(const void* w1q, long n1, const void* w4q, long n4, float* fp8_state, void* stream) {
  RV_REQUIRE(w1q && w4q && fp8_state, RV_ERR_NULL, "rv_fp8_wmax: null pointer");
  RV_REQUIRE(n1 > 0 && n4 > 0 && n1 % 16 == 0 && n4 % 16 == 0 && (((uintptr_t)w1q | (uintptr_t)w4q) & 15) == 0, RV_ERR_SHAPE,
             "rv_fp8_wmax: shadows must be 16-byte aligned multiples of 16 bytes");
  hipLaunchKernelGGL(k_fp8_wmax, dim3(2 * WMAX_BLOCKS), dim3(256), 0, (hipStream_t)stream, (const unsigned char*)w1q, n1,
                     (const unsigned char*)w4q, n4, fp8_state);
  RV_CHECK_LAUNCH();
  return RV_OK;
}

int rv_cast_pad_fp8(const float* src, long rows, long cols, long ld_src, void* dst_fp8, long rows_p, long cols_p,
                    long ld_dst, const float* scale, void* stream) {
  RV_REQUIRE(src && dst_fp8 && scale, RV_ERR_NULL, "rv_cast_pad_fp8: null pointer");
  RV_REQUIRE(rows >= 0 && cols >= 0 && rows <= rows_p && cols <= cols_p && cols_p % 8 == 0 && ld_src >= cols &&
                 ld_dst >= cols_p && ld_dst % 8 == 0 && ((uintptr_t)dst_fp8 & 7) == 0,
             RV_ERR_SHAPE, "rv_cast_pad_fp8: bad extents %ld %ld -> %ld %ld (ld %ld)", rows, cols, rows_p, cols_p, ld_dst);
  const long total = rows_p * (cols_p / 8);
  RV_REQUIRE(total < 0x7fffffffL, RV_ERR_SHAPE, "cast: %ld x %ld is too large for one launch", rows_p, cols_p);
  hipLaunchKernelGGL(k_cast_pad_bf16, dim3(grid_for(total, 8192)), dim3(256), 0, (hipStream_t)stream,
                     src, rows, cols, ld_src, (bf16_t*)nullptr, rows_p, cols_p, ld_dst, (long long*)nullptr,
                     (unsigned char*)dst_fp8, ld_dst, (float*)nullptr, scale, (const float*)nullptr, 0, 0,
                     (const long long*)nullptr, 0L, 0L, 0L, cast_wt());
  RV_CHECK_LAUNCH();
  return RV_OK;
}

int rv_randn(float* out, long n, unsigned long long seed, unsigned long long offset, void* stream) {
  RV_REQUIRE(out && n >= 0, RV_ERR_NULL, "rv_randn: bad args");
  if (n == 0) return RV_OK;
  hipLaunchKernelGGL(k_randn, dim3(grid_for((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, out, n,
                     (uint64_t)seed, (uint64_t)offset);
  RV_CHECK_LAUNCH();
  return RV_OK;
}

int rv_reparam_fwd(const float* slabs, int splits, long Bp, long Lp, long B, long L,
                   const float* eps_in, float* eps_out, unsigned long long seed,
                   const long long* step_counter, float* mulv, void* z, float* kl_partial,
                   void* stream) {
  RV_REQUIRE(slabs && mulv && z && kl_partial, RV_ERR_NULL, "rv_reparam_fwd: null pointer");
  RV_REQUIRE(eps_in || eps_out, RV_ERR_NULL, "rv_reparam_fwd: need eps_in or eps_out");
  RV_REQUIRE(splits >= 1 && B <= Bp && L <= Lp && (Bp * Lp) % 1024 == 0 && Bp * Lp < 0x7fffffffL, RV_ERR_SHAPE, "rv_reparam_fwd: bad extents");
  hipLaunchKernelGGL(k_reparam_fwd, dim3((unsigned)(Bp * Lp / 1024)), dim3(256), 0, (hipStream_t)stream,
                     slabs, splits, Bp, Lp, B, L, eps_in, eps_out, (uint64_t)seed, step_counter, mulv,
                     (bf16_t*)z, kl_partial);
  RV_CHECK_LAUNCH();
  return RV_OK;
}

int rv_reparam_bwd(const float* dz_slabs, int splits, long Bp, long Lp, long B, long L, long S,
                       const float* mulv, const float* eps, float kl_beta, const float* dmu_ext,
                       const float* dlv_ext, void* dmulv, float* dbh_partial, const float* mse_partial, int n_mse,
                       const float* kl_partial, int n_kl, float* loss_out, const long long* step_counter, int ring,
                       void* stream) {
  RV_REQUIRE(dz_slabs && mulv && eps && dmulv, RV_ERR_NULL, "rv_reparam_bwd: null pointer");
  RV_REQUIRE(splits >= 1 && B <= Bp && L <= Lp && Bp % RB_ROWS == 0 && 256 % Lp == 0, RV_ERR_SHAPE,
             "rv_reparam_bwd: bad extents (Lp must divide 256)");
  hipLaunchKernelGGL(k_reparam_bwd, dim3((unsigned)(Bp / RB_ROWS) + 1), dim3(256), 0,
                     (hipStream_t)stream, dz_slabs, splits, Bp, Lp, B, L, S, mulv, eps, kl_beta,
                     (bf16_t*)dmulv, dbh_partial, mse_partial, n_mse, kl_partial, n_kl, loss_out,
                     step_counter, ring, dmu_ext, dlv_ext);
  RV_CHECK_LAUNCH();
  return RV_OK;
}

// The loss scalar from the partial sums a forward phase left (mse_partial of the fc4 forward's epilogue, kl_partial of the
// reparameterisation): what the extra block of k_reparam_bwd / k_latent_bwd computes during the backward, in the same
// summation order, as a launch of its own -- for a caller that wants the value before (or without) a backward.
__global__ void __launch_bounds__(256)
k_loss_from_partials(const float* __restrict__ mse_partial, int n_mse, const float* __restrict__ kl_partial, int n_kl,
                     long B, long S, long L, float kl_beta, float* __restrict__ out) {
  __shared__ float sh[4];
  const int tid = threadIdx.x;
  float m = 0.f, k = 0.f;
  for (int i = tid; i < n_mse; i += 256) m += mse_partial[i];
  for (int i = tid; i < n_kl; i += 256) k += kl_partial[i];
  m = block_sum_256(m, sh);
  k = block_sum_256(k, sh);
  if (tid == 0) {
    const float mse = m / ((float)B * (float)S);
    const float kld = -0.5f * k * (1.0f / ((float)B * (float)L));
    out[0] = mse + kl_beta * kld;
    out[1] = mse;
    out[2] = kld;
  }
}
int rv_loss_from_partials(const float* mse_partial, int n_mse, const float* kl_partial, int n_kl, long B, long S, long L,
                          float kl_beta, float* out3, void* stream) {
  RV_REQUIRE(mse_partial && kl_partial && out3 && n_mse > 0 && n_kl > 0, RV_ERR_NULL, "rv_loss_from_partials: null pointer");
  hipLaunchKernelGGL(k_loss_from_partials, dim3(1), dim3(256), 0, (hipStream_t)stream, mse_partial, n_mse, kl_partial, n_kl,
                     B, S, L, kl_beta, out3);
  RV_CHECK_LAUNCH();
  return RV_OK;
}

long rv_loss_fused_workspace_bytes(void) { return 256 + 2 * LOSS_MAX_BLOCKS * (long)sizeof(float); }

int rv_loss_fused(const float* recon, const float* x, const float* mu, const float* logvar, long B,
                  long S, long L, float kl_beta, float* loss_out, float* d_recon, float* d_mu,
                  float* d_logvar, void* workspace, void* stream) {
  RV_REQUIRE(recon && x && mu && logvar && loss_out && workspace, RV_ERR_NULL, "rv_loss_fused: null pointer");
  RV_REQUIRE(B > 0 && S > 0 && L > 0, RV_ERR_SHAPE, "rv_loss_fused: empty input");
  const long n_r = B * S, n_k = B * L;
  unsigned grid = grid_for((n_r + 3) / 4, LOSS_MAX_BLOCKS);
  hipLaunchKernelGGL(k_loss_fused, dim3(grid), dim3(256), 0, (hipStream_t)stream, recon, x, mu, logvar,
                     n_r, n_k, kl_beta, loss_out, d_recon, d_mu, d_logvar, (unsigned*)workspace);
  RV_CHECK_LAUNCH();
  return RV_OK;
}

int rv_reparameterize(const float* mu, const float* logvar, long n, const float* eps_in,
                      float* eps_out, unsigned long long seed, unsigned long long offset, float* z,
                      void* stream) {
  RV_REQUIRE(mu && logvar && z && n >= 0, RV_ERR_NULL, "rv_reparameterize: bad args");
  if (n == 0) return RV_OK;
  hipLaunchKernelGGL(k_reparameterize, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, mu, logvar,
                     n, eps_in, eps_out, (uint64_t)seed, (uint64_t)offset, z);
  RV_CHECK_LAUNCH();
  return RV_OK;
}

int rv_gather_frames(const float* audio, long n_samples, const long long* frame_index, long first_frame,
                     long n_frames, long S, long hop, float* out, void* stream) {
  RV_REQUIRE(audio && out, RV_ERR_NULL, "rv_gather_frames: null pointer");
  RV_REQUIRE(n_samples > 0 && n_frames >= 0 && S > 0 && hop > 0, RV_ERR_SHAPE, "rv_gather_frames: bad extents");
  RV_REQUIRE(((uintptr_t)out & 15) == 0, RV_ERR_SHAPE, "rv_gather_frames: output must be 16-byte aligned");
  if (n_frames == 0) return RV_OK;
  hipLaunchKernelGGL(k_gather_frames, dim3(grid_for(n_frames * ((S + 3) / 4), 4096)), dim3(256), 0,
                     (hipStream_t)stream, audio, n_samples, frame_index, first_frame, n_frames, S, hop, out);
  RV_CHECK_LAUNCH();
  return RV_OK;
}

int rv_tanh_bwd_pack(const float* d_recon, const float* recon, long B, long S, void* dP4, long Bp,
                     long Sp, void* stream) {
  RV_REQUIRE(d_recon && recon && dP4, RV_ERR_NULL, "rv_tanh_bwd_pack: null pointer");
  RV_REQUIRE(B <= Bp && S <= Sp, RV_ERR_SHAPE, "rv_tanh_bwd_pack: bad extents");
  hipLaunchKernelGGL(k_tanh_bwd_pack, dim3(grid_for(Bp * Sp, 4096)), dim3(256), 0, (hipStream_t)stream,
                     d_recon, recon, B, S, (bf16_t*)dP4, Bp, Sp);
  RV_CHECK_LAUNCH();
  return RV_OK;
}

int rv_colsum_partial(const void* src, int is_bf16, long rows, long cols, long ld, float* out,
                      long ld_out, void* stream) {
  RV_REQUIRE(src && out, RV_ERR_NULL, "rv_colsum_partial: null pointer");
  RV_REQUIRE(rows > 0 && cols > 0 && ld >= cols && ld_out >= cols, RV_ERR_SHAPE, "rv_colsum_partial: bad extents");
  dim3 grid((unsigned)((cols + 63) / 64), (unsigned)((rows + 255) / 256));
  if (is_bf16)
    hipLaunchKernelGGL(k_colsum_partial<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)src, rows, cols, ld, out, ld_out);
  else
    hipLaunchKernelGGL(k_colsum_partial<float>, grid, dim3(256), 0, (hipStream_t)stream,
                       (const float*)src, rows, cols, ld, out, ld_out);
  RV_CHECK_LAUNCH();
  return RV_OK;
}

int rv_reparameterize_bwd(const float* dz, const float* eps, const float* logvar, long n, float* dmu,
                          float* dlv, void* stream) {
  RV_REQUIRE(dz && eps && logvar, RV_ERR_NULL, "rv_reparameterize_bwd: null pointer");
  if (n == 0) return RV_OK;
  hipLaunchKernelGGL(k_reparameterize_bwd, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, dz, eps,
                     logvar, n, dmu, dlv);
  RV_CHECK_LAUNCH();
  return RV_OK;
}

int rv_ew_f32(int op, const float* a, const float* b, long n, float* out, void* stream) {
  RV_REQUIRE(a && b && out, RV_ERR_NULL, "rv_ew_f32: null pointer");
  RV_REQUIRE(op >= 0 && op <= 2, RV_ERR_UNSUPPORTED, "rv_ew_f32: op %d", op);
  if (n <= 0) return RV_OK;
  hipLaunchKernelGGL(k_ew_f32, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, op, a, b, n, out);
  RV_CHECK_LAUNCH();
  return RV_OK;
}

// ---- device-side cross-stream signalling (plan.hip: the data-parallel step) ----
// A hand-over between two streams through HIP events costs ~9 us per crossing on this runtime (record -> dependent
// kernel on the other stream; 4.6 us of bubble on the recording stream alone); through a flag in device memory it
// costs ~1.8 (measured: tools/probe_cross_stream.hip).  k_flag_set runs BEHIND the producing kernel
// in its stream (the kernel boundary in front of it is the agent-scope release of the producer's data) and publishes
// a sequence number; k_flag_wait sits in the consumer stream IN FRONT of the consuming kernel and returns when the
// number has arrived.  Deadlock-free by construction whatever the runtime's stream -> hardware-queue mapping is: every
// waiter is enqueued (host order) after its setter, so the oldest unfinished kernel over all queues never waits on
// anything unfinished ON THIS DEVICE.  The wait is bounded all the same: `max_ticks` of the 100 MHz wall clock, chosen by
// the caller per edge -- seconds for an edge whose setter follows this device's own kernels, minutes (the order of a
// collective library's own watchdog) for an edge whose setter sits behind a collective, i.e. behind the slowest PEER:
// ranks reach a step tens of milliseconds apart as a matter of course and seconds apart around a checkpoint.  A
// timeout is counted in `timeouts` and the engine raises when it sees a non-zero count (that step's results are invalid).
__global__ void __launch_bounds__(64) k_flag_set(int* flag, int value, const long long* copy_src, long long* copy_dst) {
  if (threadIdx.x == 0) {
    // (rv_flag_set_copy: a device word latched on the way -- the step number this step's deferred update will need)
    if (copy_src) *copy_dst = *copy_src;
    __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}
__global__ void __launch_bounds__(64) k_flag_wait(const int* flag, int value, int* timeouts, long long max_ticks) {
  if (threadIdx.x == 0) {
    const long long t0 = wall_clock64();   // 100 MHz
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - value < 0) {
      __builtin_amdgcn_s_sleep(2);
      if (wall_clock64() - t0 > max_ticks) {
        atomicAdd(timeouts, 1);
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
}
int rv_flag_set(int* flag, int value, void* stream) { return rv_flag_set_copy(flag, value, nullptr, nullptr, stream); }
int rv_flag_set_copy(int* flag, int value, const long long* copy_src, long long* copy_dst, void* stream) {
  hipLaunchKernelGGL(k_flag_set, dim3(1), dim3(64), 0, (hipStream_t)stream, flag, value, copy_src, copy_dst);
  RV_CHECK_LAUNCH();
  return RV_OK;
}
int rv_flag_wait(const int* flag, int value, int* timeouts, long max_ms, void* stream) {
  hipLaunchKernelGGL(k_flag_wait, dim3(1), dim3(64), 0, (hipStream_t)stream, flag, value, timeouts,
                     (long long)(max_ms < 1 ? 1 : max_ms) * 100000LL);
  RV_CHECK_LAUNCH();
  return RV_OK;
}

int rv_scale_by(const float* a, const float* scalar, long n, float* out, void* stream) {
  RV_REQUIRE(a && scalar && out, RV_ERR_NULL, "rv_scale_by: null pointer");
  if (n == 0) return RV_OK;
  hipLaunchKernelGGL(k_scale_by, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, a, scalar, n, out);
  RV_CHECK_LAUNCH();
  return RV_OK;
}

int rv_scale_by3(const float* a0, float* out0, long n0, const float* a1, float* out1, long n1, const float* a2,
                 float* out2, long n2, const float* scalar, void* stream) {
  RV_REQUIRE(scalar, RV_ERR_NULL, "rv_scale_by3: null scalar");
  RV_REQUIRE(n0 >= 0 && n1 >= 0 && n2 >= 0 && (!n0 || (a0 && out0)) && (!n1 || (a1 && out1)) && (!n2 || (a2 && out2)),
             RV_ERR_NULL, "rv_scale_by3: a tensor with a non-zero count needs both pointers");
  const long nmax = n0 > n1 ? (n0 > n2 ? n0 : n2) : (n1 > n2 ? n1 : n2);
  if (nmax == 0) return RV_OK;
  hipLaunchKernelGGL(k_scale_by3, dim3(grid_for((nmax + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a0, out0, n0, a1, out1,
                     n1, a2, out2, n2, scalar);
  RV_CHECK_LAUNCH();
  return RV_OK;
}

// rv_adam_multi whose update is withheld when `*poison` is non-zero (internal.h; plan.hip's data-parallel step).
int rv_adam_multi_guarded(const rv_param_desc* descs, int n_desc, float* param, float* exp_avg, float* exp_avg_sq,
                          float* grad_out, const void* grad_bf16, float lr, float grad_scale,
                          const long long* step_counter, const int* poison, void* stream) {
  RV_REQUIRE(param && exp_avg && exp_avg_sq && step_counter, RV_ERR_NULL, "rv_adam_multi: null pointer");
  RV_REQUIRE(!(grad_bf16 && grad_out), RV_ERR_UNSUPPORTED, "rv_adam_multi: grad_out is the sum of the slabs; not with grad_bf16");
  DescTable tab;
  int rc = adam_build_table(descs, n_desc, &tab);
  if (rc) return rc;
  hipLaunchKernelGGL(k_adam<true>, dim3((unsigned)tab.blk_start[n_desc]), dim3(256), 0,
                     (hipStream_t)stream, tab, param, exp_avg, exp_avg_sq, grad_out, lr, grad_scale,
                     step_counter, (bf16_t*)nullptr, (const bf16_t*)grad_bf16, poison, (const float*)nullptr);
  RV_CHECK_LAUNCH();
  return RV_OK;
}

int rv_adam_multi(const rv_param_desc* descs, int n_desc, float* param, float* exp_avg,
                  float* exp_avg_sq, float* grad_out, const void* grad_bf16, float lr, float grad_scale,
                  const long long* step_counter, void* stream) {
  return rv_adam_multi_guarded(descs, n_desc, param, exp_avg, exp_avg_sq, grad_out, grad_bf16, lr, grad_scale, step_counter,
                               nullptr, stream);
}

int rv_params_from_flat(const rv_param_desc* descs, int n_desc, const float* flat, long flat_base, float* param,
                        void* stream) {
  RV_REQUIRE(flat, RV_ERR_NULL, "rv_params_from_flat: null pointer");
  DescTable tab;
  int rc = adam_build_table(descs, n_desc, &tab);
  if (rc) return rc;
  for (int i = 0; i < n_desc; ++i)
    RV_REQUIRE(descs[i].offset >= flat_base, RV_ERR_SHAPE, "rv_params_from_flat: tensor %d starts before the flat source", i);
  hipLaunchKernelGGL(k_params_from_flat, dim3((unsigned)tab.blk_start[n_desc]), dim3(256), 0, (hipStream_t)stream, tab,
                     flat, flat_base, param);
  RV_CHECK_LAUNCH();
  return RV_OK;
}

// rv_grad_finalize whose sums are multiplied by the device scalar *scale_dev (NULL: 1) -- the upstream gradient of a
// loss whose backward the plan runs itself (rv_plan_set_loss_grad): the host never reads it.
int rv_grad_finalize_scaled(const rv_param_desc* descs, int n_desc, void* grad_out, int out_bf16, const float* scale_dev,
                            void* stream) {
  RV_REQUIRE(grad_out, RV_ERR_NULL, "rv_grad_finalize: null pointer");
  DescTable tab;
  int rc = adam_build_table(descs, n_desc, &tab);
  if (rc) return rc;
  hipLaunchKernelGGL(k_adam<false>, dim3((unsigned)tab.blk_start[n_desc]), dim3(256), 0,
                     (hipStream_t)stream, tab, (float*)nullptr, (float*)nullptr, (float*)nullptr,
                     out_bf16 ? (float*)nullptr : (float*)grad_out, 0.f, 1.f, (const long long*)nullptr,
                     out_bf16 ? (bf16_t*)grad_out : (bf16_t*)nullptr, (const bf16_t*)nullptr, (const int*)nullptr, scale_dev);
  RV_CHECK_LAUNCH();
  return RV_OK;
}

int rv_grad_finalize(const rv_param_desc* descs, int n_desc, void* grad_out, int out_bf16, void* stream) {
  return rv_grad_finalize_scaled(descs, n_desc, grad_out, out_bf16, nullptr, stream);
}

}  // extern "C"
