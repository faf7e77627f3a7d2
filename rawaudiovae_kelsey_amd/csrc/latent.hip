// Row-local fused kernels of the latent-sized part of the step (gfx950).  C ABI: include/rawvae_hip.h.
//
//   rv_latent_fwd : heads GEMM (fc21 | fc22) -> reparameterisation + KL partial -> fc3 + bias + ReLU
//                   (rawvae/model.py:21-29) for 16 batch rows per workgroup, ONE launch instead of three.
//
// Everything here is row-local: a 16-row block needs no other block's data, so the three steps need no
// hand-off between workgroups.  The price is that every workgroup streams the whole head weight (128 x Hp)
// and W3 (Hp x 64) through its CU, 832 KB at C2; a CU takes in ~60-70 GB/s from L2 whatever the loop looks like
// (MI355X_MICROARCH.md "Indexed rows: gather into LDS"), so the kernel is a STREAMING kernel whose MFMAs are
// noise, and its design is about keeping that port busy:
//   * the K dimension of the heads GEMM is cut over the 8 waves (wave w owns k in [w Hp/8, (w+1) Hp/8) for all
//     128 output columns), the N dimension of fc3 likewise (wave w owns 1/8 of the Hp columns): every byte a wave
//     stages is consumed by that wave alone, so each wave runs PRIVATE two-slot LDS rings (weights, activations) fed
//     by global_load_lds with counted vmcnt waits and the streaming loops contain no workgroup barrier at all;
//   * the first fc3 weight slot (and the wave's fc3 bias slice) is already in flight while the 8 partial head
//     sums are reduced through LDS and reparameterised (two barriers in the whole kernel).
// Round 2's version of this kernel fed its MFMAs from compiler-scheduled register loads and streamed 21-30 GB/s
// per CU (45 us against 21 us for the three launches); this one is what DESIGN.md section 6 said it needed.
#include "gemm_bf16.h"
#include "philox.h"
#include "../../include/rawvae_hip.h"
#include "internal.h"

using namespace rv;

namespace {

constexpr int LAT_ROWS = 16;
// LDS of one wave: two 8-KiB weight slots (64 rows x 128 B) and two 2-KiB activation slots (16 rows x 128 B), all in
// the K-major image of gemm_bf16.h (128-byte rows, 16-byte chunk c of row r at chunk c ^ ((r >> 1) & 7); fragments
// by load_frag<.., true>).  128-byte rows on purpose: a first version staged 64-byte (32-k) row pieces and streamed
// 37 GB/s per CU -- every 128-byte line of the weights was fetched twice, by the pieces of two consecutive steps.
constexpr int LW_SLOT = 8 * 1024, LX_SLOT = 2 * 1024;
constexpr int LX_OFF = 2 * LW_SLOT;
constexpr int L_RING = 2 * LW_SLOT + 2 * LX_SLOT;   // 20 KiB per wave
constexpr int L_LDS = 8 * L_RING;                   // all 160 KiB of the CU
// after the streaming loops the slots are reused: partial sums in the wave's weight slot 1, the fc3 bias slice in its
// activation slot 0, z (16 x 64 bf16) in wave 0's activation slot 1, the KL partials in wave 1's
constexpr int L_Z = 0 * L_RING + LX_OFF + LX_SLOT;
constexpr int L_RED = 1 * L_RING + LX_OFF + LX_SLOT;

__device__ __forceinline__ void dma16(const bf16_t* g, lds_char* l) {
  __builtin_amdgcn_global_load_lds((glb_cptr)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// `rows` (a multiple of 8) rows of 64 bf16 starting at g (row stride ld) -> K-major LDS image at sl
template <int ROWS>
__device__ __forceinline__ void stage_rows(const bf16_t* g, long ld, lds_char* sl, int lane) {
#pragma unroll
  for (int i = 0; i < ROWS / 8; ++i) {
    const int r = 8 * i + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
    dma16(g + r * ld + c * 8, sl + i * 1024);
  }
}

// s_waitcnt vmcnt(n) with a run-time (wave-uniform) n from the handful of counts the loops need; a SMALLER count
// than the true number of younger operations is always safe (the queue retires in order)
__device__ __forceinline__ void wait_vm(int n) {
  if (n >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if (n >= 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else if (n >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (n >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if (n >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__global__ void __launch_bounds__(512)
k_latent_fwd(const bf16_t* __restrict__ h1, const long ldh, const bf16_t* __restrict__ Wh, const long ldwh,
             const float* __restrict__ bh, const bf16_t* __restrict__ W3, const long ldw3,
             const float* __restrict__ b3, const long Hp, const long B, const long L,
             const float* __restrict__ eps_in, float* __restrict__ eps_out, const uint64_t seed,
             const long long* __restrict__ step_counter, float* __restrict__ mulv, bf16_t* __restrict__ z,
             float* __restrict__ kl_partial, bf16_t* __restrict__ h3, const long ldh3, const int wt,
             unsigned char* __restrict__ h3q, const long ldq, const float* __restrict__ q_scale,
             float* __restrict__ amax_part) {
  constexpr long Lp = 64, L2p = 128;
  extern __shared__ __attribute__((aligned(16))) char smem_dyn[];
  lds_char* smem = (lds_char*)smem_dyn;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, j = lane & 15;
  const long r0 = (long)blockIdx.x * LAT_ROWS;
  lds_char* ring = smem + wave * L_RING;
  lds_char* const W0 = ring, * const W1 = ring + LW_SLOT;
  const long kw = Hp / 8;          // width of a wave's K slice of the heads GEMM = of its column slice of fc3
  const int NU = (int)(kw / 64);   // 64-deep steps of the heads GEMM = 64-column slots of fc3 (1..4)
  // Every workgroup streams the same weights.  Workgroups b, b + 8, ... share an XCD (and its L2): the one with index
  // bi = b >> 3 among them gives wave w the slice (w + bi) % 8 and starts its walk over the slice at step (bi >> 3) % NU,
  // so that at any moment the 32 CUs of an XCD ask its L2 for 32 x 8 different pieces of the weights instead of the
  // same few lines.  The summation order of a row's dot products then depends on its block: fixed, so reproducible.
  const int bi = (int)(blockIdx.x >> 3);
  const int ks = (wave + bi) & 7;              // slice index
  const int rot = (bi >> 3) % NU;              // first step / slot of the walk
  const bool do_fc3 = W3 != nullptr;           // wave-uniform (a kernel argument)

  // ---- heads: partial[j][n] = sum over this wave's k of h1[r0 + j][k] Wh[n][k], n = 0..127.  Step s = 64 k; its
  // weight columns come as two half-steps (head-weight rows 0..63 into W0, rows 64..127 into W1).
  const bf16_t* xg = h1 + r0 * ldh + ks * kw;
  const bf16_t* wg = Wh + ks * kw;
  auto walk = [&](int s) { const int t = s + rot; return t >= NU ? t - NU : t; };   // s-th step of the rotated walk
  auto issue_x = [&](int s) { stage_rows<16>(xg + 64 * walk(s), ldh, ring + LX_OFF + (s & 1) * LX_SLOT, lane); };
  auto issue_w = [&](int s, int half) { stage_rows<64>(wg + (long)(64 * half) * ldwh + 64 * walk(s), ldwh, half ? W1 : W0, lane); };
  auto issue_fc3 = [&](int u) { stage_rows<64>(W3 + (ks * kw + 64 * walk(u)) * ldw3, ldw3, (u & 1) ? W1 : W0, lane); };
  f32x4 acc[8];
#pragma unroll
  for (int cb = 0; cb < 8; ++cb) acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
  // issue order of the whole loop:  X0 W(0,A) [X1] W(0,B) | X2 W(1,A) | W(1,B) | X3 W(2,A) | W(2,B) ...
  issue_x(0);
  issue_w(0, 0);
  if (NU > 1) issue_x(1);
  issue_w(0, 1);
  for (int s = 0; s < NU; ++s) {
    bf16x8 x[2], w[4][2];
    // half A: needs X(s) and W(s,A); the only younger pieces are W(s,B)'s 8 (at s = 0 also X1's 2: waited for too)
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) x[kk] = load_frag<16, true>(ring + LX_OFF + (s & 1) * LX_SLOT, 0, kk, lane);
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) w[cb][kk] = load_frag<64, true>(W0, cb * 16, kk, lane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // fragments are in registers: both slots may be refilled
    if (s + 2 < NU) issue_x(s + 2);
    if (s + 1 < NU) {
      issue_w(s + 1, 0);
    } else if (do_fc3) {
      // behind the last step: fc3 weight slot 0 and the wave's fc3 bias slice (kw floats <= 1 KiB) start to fly
      issue_fc3(0);
      dma16((const bf16_t*)(b3 + ks * kw + (lane * 4 < kw ? lane * 4 : 0)), ring + LX_OFF);
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[cb][kk], x[kk], acc[cb], 0, 0, 0);
    // half B: needs W(s,B); younger: [X(s+2): 2] + W(s+1,A): 8, or fc3 slot 0 + bias: 9 (nothing without fc3)
    if (s + 2 < NU) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if (s + 1 < NU || do_fc3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) w[cb][kk] = load_frag<64, true>(W1, cb * 16, kk, lane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (s + 1 < NU) issue_w(s + 1, 1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
        acc[4 + cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[cb][kk], x[kk], acc[4 + cb], 0, 0, 0);
  }
  // lane (q, j) holds partial[j][16 cb + 4 q + e]: park the wave's 16 x 128 partial sums in its weight slot 1.  Rows are
  // 512 bytes, a whole number of bank rows, so the eight lanes a 16-byte LDS store services together (same q, eight j)
  // would all hit the same four banks: the 16-byte quad index is XORed with j & 7 (and again by the reader below)
#pragma unroll
  for (int cb = 0; cb < 8; ++cb)
    *(__attribute__((address_space(3))) f32x4*)(W1 + (j * 128 + (((cb * 4 + q) ^ (j & 7)) << 2)) * 4) = acc[cb];
  // raw barriers: __syncthreads() would also drain vmcnt, i.e. wait for the fc3 weight slot that is meant to fly
  // across the reduction
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // ---- reduce the 8 partial sums (fixed order), bias, reparameterise: threads 0..255 take the 256 four-column
  // groups of the block in k_reparam_fwd's order (same eps draws, same KL partial granularity)
  float kl = 0.f;
  if (tid < 256) {
    const int rr = tid >> 4;
    const long b = r0 + rr, l = (long)(tid & 15) * 4;
    const long i = (long)blockIdx.x * 256 + tid;   // group index over the padded [Bp, Lp / 4] grid
    float mua[4] = {0.f, 0.f, 0.f, 0.f}, lva[4] = {0.f, 0.f, 0.f, 0.f}, zz[4] = {0.f, 0.f, 0.f, 0.f};
    if (b < B && l < L) {
      f32x4 pm[8], pv[8];
#pragma unroll
      for (int w = 0; w < 8; ++w) {
        const lds_char* ps = smem + w * L_RING + LW_SLOT;
        const int qm = (int)(l >> 2) ^ (rr & 7), qv = (16 + (int)(l >> 2)) ^ (rr & 7);   // the writer's swizzle
        pm[w] = *(const __attribute__((address_space(3))) f32x4*)(ps + (rr * 128 + (qm << 2)) * 4);
        pv[w] = *(const __attribute__((address_space(3))) f32x4*)(ps + (rr * 128 + (qv << 2)) * 4);
      }
      const f32x4 bm = *reinterpret_cast<const f32x4*>(bh + l), bv = *reinterpret_cast<const f32x4*>(bh + Lp + l);
      float ev[4];
      if (!eps_in) normal4_fast(seed, (uint64_t)i, step_counter ? (uint64_t)*step_counter : 0, ev);
#pragma unroll
      for (int e_ = 0; e_ < 4; ++e_) {
        if (l + e_ < L) {
          float e;
          if (eps_in) {
            e = eps_in[b * L + l + e_];
          } else {
            e = ev[e_];
            eps_out[b * L + l + e_] = e;
          }
          mua[e_] = (((pm[0][e_] + pm[1][e_]) + (pm[2][e_] + pm[3][e_])) + ((pm[4][e_] + pm[5][e_]) + (pm[6][e_] + pm[7][e_]))) + bm[e_];
          lva[e_] = (((pv[0][e_] + pv[1][e_]) + (pv[2][e_] + pv[3][e_])) + ((pv[4][e_] + pv[5][e_]) + (pv[6][e_] + pv[7][e_]))) + bv[e_];
          const float sd = __expf(0.5f * lva[e_]);
          zz[e_] = mua[e_] + e * sd;
          kl += 1.f + lva[e_] - mua[e_] * mua[e_] - sd * sd;
        }
      }
    }
    *reinterpret_cast<float4*>(mulv + b * L2p + l) = make_float4(mua[0], mua[1], mua[2], mua[3]);
    *reinterpret_cast<float4*>(mulv + b * L2p + Lp + l) = make_float4(lva[0], lva[1], lva[2], lva[3]);
    const bf16x4 zb = {(bf16_t)zz[0], (bf16_t)zz[1], (bf16_t)zz[2], (bf16_t)zz[3]};
    *reinterpret_cast<bf16x4*>(z + b * Lp + l) = zb;
    *(__attribute__((address_space(3))) bf16x4*)(smem + L_Z + (rr * 64 + l) * 2) = zb;
  }
  kl = wave_sum(kl);
  if (lane == 0) *(__attribute__((address_space(3))) float*)(smem + L_RED + wave * 4) = kl;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();   // z complete, the partial sums are consumed (their slots may be refilled), kl partials complete
  asm volatile("" ::: "memory");
  if (tid == 0) {
    const __attribute__((address_space(3))) float* red = (const __attribute__((address_space(3))) float*)(smem + L_RED);
    kl_partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
  }

  if (!do_fc3) return;   // heads + reparameterisation only: fc3 runs inside the fc4 forward (gemm_bf16.h A_GEN)
  // ---- fc3: h3[r0 + j][n] = relu(sum_k W3[n][k] z[r0 + j][k] + b3[n]) for this wave's columns, 64 per slot
  if (NU > 1) issue_fc3(1);
  const lds_char* zl = smem + L_Z;
  const bf16x8 z0 = *(const __attribute__((address_space(3))) bf16x8*)(zl + j * 128 + q * 16);
  const bf16x8 z1 = *(const __attribute__((address_space(3))) bf16x8*)(zl + j * 128 + 64 + q * 16);
  const lds_char* bl = ring + LX_OFF;   // the wave's bias slice (kw floats)
  bf16_t* out = h3 + (r0 + j) * ldh3 + ks * kw;
  // fp8 forward (RV_OPT_FP8): h3 also as fp8(h3 * *q_scale), fc4's operand, and max|h3| of this wave's outputs for the
  // next step's scale (what rv_linear_fwd_ex's epilogue does per block)
  const float qs = h3q ? *q_scale : 0.f;
  unsigned char* outq = h3q ? h3q + (r0 + j) * ldq + ks * kw : nullptr;
  float amax = 0.f;
  for (int u = 0; u < NU; ++u) {
    const lds_char* sl = (u & 1) ? W1 : W0;
    // Outstanding operations issued after slot u's pieces, in order (stores: 2 per iteration):
    //   u = 0: bias slice (must have landed too), [phase-2 loads and stores], slot 1   -> wait for <= 8 (slot 1 only)
    //   u = 1: slot 2 (8, if any), stores of iteration 0 (2)
    //   u >= 2: stores of u - 2 (2), slot u + 1 (8, if any), stores of u - 1 (2)
    wait_vm(u == 0 ? (NU > 1 ? 8 : 0) : u == 1 ? (NU > 2 ? 10 : 2) : (u + 1 < NU ? 12 : 4));
    bf16x8 w[4][2];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) w[cb][kk] = load_frag<64, true>(sl, cb * 16, kk, lane);
    // bias of the four accumulator columns a lane holds BEFORE the pairing swap below: 16 cb + 4 q ..
    f32x4 bias[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) bias[cb] = *(const __attribute__((address_space(3))) f32x4*)(bl + (walk(u) * 64 + cb * 16 + q * 4) * 4);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (u + 2 < NU) issue_fc3(u + 2);
    f32x4 a[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      a[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[cb][0], z0, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      a[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[cb][1], z1, a[cb], 0, 0, 0);
    }
    // bias + ReLU first (plain VALU on the MFMA results: the compiler pads that hazard itself, which it cannot do in
    // front of the inline-asm swap), then the accumulator-direct store of gemm_bf16.h: one permlane16 swap per
    // register between the fragments of a column-block pair gives lane (q, j) the EIGHT consecutive columns
    // (2t + (q&1)) * 16 + (q>>1) * 8 .. + 8 of row j
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int e = 0; e < 4; ++e) a[cb][e] = fmaxf(a[cb][e] + bias[cb][e], 0.f);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      f32x4 lo = a[2 * t], hi = a[2 * t + 1];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float a_ = lo[r], b_ = hi[r];
        swap_rows16(a_, b_);
        lo[r] = a_;
        hi[r] = b_;
      }
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[e] = (bf16_t)lo[e];
        o[4 + e] = (bf16_t)hi[e];
      }
      if (h3) store_out16((bf16x8*)(out + walk(u) * 64 + (2 * t + (q & 1)) * 16 + (q >> 1) * 8), o, wt);
      if (h3q || amax_part) {
        float q8[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          amax = fmaxf(amax, fmaxf(lo[e], hi[e]));   // post-ReLU: non-negative
          q8[e] = fp8_keep_positive(lo[e], lo[e] * qs);
          q8[4 + e] = fp8_keep_positive(hi[e], hi[e] * qs);
        }
        if (h3q) *(unsigned long long*)(outq + walk(u) * 64 + (2 * t + (q & 1)) * 16 + (q >> 1) * 8) = pack_fp8x8(q8);
      }
    }
    asm volatile("" ::: "memory");
  }
  if (amax_part) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    if (lane == 0) amax_part[blockIdx.x * 8 + wave] = amax;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // nothing of this wave's is in flight towards LDS when it ends
}


// `64` k-rows of 64 bf16 (row stride ld) -> MN-major LDS image at sl (gemm_bf16.h: [k][64 n], 32-byte chunks swizzled
// by swz_mn<64>(k); fragments by load_frag<64, false>, i.e. transposing LDS reads): the operand whose contraction
// index is the ROW of the matrix in memory
__device__ __forceinline__ void stage_rows_mn64(const bf16_t* g, long ld, lds_char* sl, int lane) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int kr = 8 * i + (lane >> 3), p16 = lane & 7;
    const int c32 = (p16 >> 1) ^ swz_mn<64>(kr);
    dma16(g + kr * ld + (c32 * 2 + (p16 & 1)) * 8, sl + i * 1024);
  }
}

// The backward mirror of k_latent_fwd's first two steps: dz = dP3 W3 (autograd of fc3, model.py:29) for 16 batch rows
// per workgroup over the FULL contraction (K = Hp cut over the 4 waves, private LDS-DMA rings, no barrier in the
// streaming loop), then the backward of reparameterize + KL (k_reparam_bwd's arithmetic, elementwise.hip) on the
// block's 16 x 64 dz while it is still in LDS: no dz slabs in HBM, no second launch.  Streams W3 (Hp x 64) + 16 rows
// of dP3 through every CU: 320 KB at C2, and the CU's L2 -> LDS port is the bound again.
// The launch has two more roles.  Blocks [n_rows, n_rows + n_w3): fc3's weight gradient dW3 = dP3^T z as an ordinary
// split-K GEMM on 64 x 64 tiles (gemm_bf16.h's body, same 256 threads): it needs nothing this launch produces, and with
// 80 KiB of LDS per workgroup a dz block and a dW3 block share every CU, so the second read of dP3 fills the bubbles of
// the first instead of lengthening another launch.  The last block finishes the loss scalar.
constexpr int LB_WAVES = 4;
constexpr int LB_LDS = LB_WAVES * L_RING;   // 80 KiB: two workgroups per CU

__global__ void __launch_bounds__(256)
k_latent_bwd(const bf16_t* __restrict__ dP3, const long lddp, const bf16_t* __restrict__ W3, const long ldw3,
             const long Hp, const long B, const long L, const long S, const float* __restrict__ mulv,
             const float* __restrict__ eps, const float kl_beta, const float* __restrict__ dmu_ext,
             const float* __restrict__ dlv_ext, bf16_t* __restrict__ dmulv, float* __restrict__ dbh_partial,
             const float* __restrict__ mse_partial, const int n_mse, const float* __restrict__ kl_partial,
             const int n_kl, float* __restrict__ loss_out, const long long* __restrict__ step_counter, const int ring_n,
             const int n_rows, const GemmArgs w3grad) {
  constexpr long Lp = 64, L2p = 128;
  extern __shared__ __attribute__((aligned(16))) char smem_dyn[];
  lds_char* smem = (lds_char*)smem_dyn;
  const int tid = threadIdx.x, lane = tid & 63;
  const float inv_nk = 1.0f / ((float)B * (float)L);
  if ((int)blockIdx.x >= n_rows) {
    if (blockIdx.x != gridDim.x - 1) {   // fc3's weight gradient, one 64 x 64 tile of one K split
      gemm_body<64, 64, 2, 2, false, false, EPI_F32, 2>(w3grad, (int)blockIdx.x - n_rows, smem_dyn);
      return;
    }
    if (loss_out && mse_partial && kl_partial) {   // the loss scalar (k_reparam_bwd's extra block, same summation order)
      float* red = (float*)smem_dyn;
      float m = 0.f, k = 0.f;
      for (int i = tid; i < n_mse; i += 256) m += mse_partial[i];
      for (int i = tid; i < n_kl; i += 256) k += kl_partial[i];
      m = block_sum_256(m, red);
      k = block_sum_256(k, red);
      if (tid == 0) {
        const float mse = m / ((float)B * (float)S);
        const float kld = -0.5f * k * inv_nk;
        if (step_counter && ring_n > 0) loss_out += 4 * ((*step_counter - 1) % ring_n);
        loss_out[0] = mse + kl_beta * kld;
        loss_out[1] = mse;
        loss_out[2] = kld;
      }
    }
    return;
  }
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, j = lane & 15;
  // Workgroups b, b + 8, ... share an XCD and its L2.  The launch's dW3 workgroups on XCD x read the batch rows of K
  // split x (gemm_body deals each XCD a contiguous run of (split, tile) items), so the dz workgroups on XCD x take the
  // same rows: the second reader of dP3 then finds it in that L2 (PMC: 37.6 -> see profiles/r03_traffic.txt).  Any
  // bijection of workgroups onto 16-row groups is correct; this one only places them.
  const int grp = (n_rows & 7) == 0 ? (int)(blockIdx.x & 7) * (n_rows >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const long r0 = (long)grp * LAT_ROWS;
  lds_char* ring = smem + wave * L_RING;
  lds_char* const W0 = ring, * const W1 = ring + LW_SLOT;
  const long kw = Hp / LB_WAVES;
  const int NU = (int)(kw / 64);
  const int bi = (int)(blockIdx.x >> 3);   // the forward's stagger (see k_latent_fwd), over 4 slices
  const int ks = (wave + bi) & (LB_WAVES - 1);
  const int rot = (bi >> 2) % NU;
  auto walk = [&](int s) { const int t = s + rot; return t >= NU ? t - NU : t; };
  const bf16_t* xg = dP3 + r0 * lddp + ks * kw;
  const bf16_t* wg = W3 + ks * kw * ldw3;
  auto issue = [&](int s) {   // 2 + 8 LDS-DMA instructions
    stage_rows<16>(xg + 64 * walk(s), lddp, ring + LX_OFF + (s & 1) * LX_SLOT, lane);
    stage_rows_mn64(wg + (long)(64 * walk(s)) * ldw3, ldw3, (s & 1) ? W1 : W0, lane);
  };
  // the epilogue's operands (thread: row rr, columns l..l+3) are requested before the stream starts
  const int rr = tid >> 4;
  const long b = r0 + rr, l = (long)(tid & 15) * 4;
  const bool live = b < B && l < L;
  float4 mu4 = make_float4(0.f, 0.f, 0.f, 0.f), lv4 = mu4;
  float ev[4] = {0.f, 0.f, 0.f, 0.f}, xm[4] = {0.f, 0.f, 0.f, 0.f}, xv[4] = {0.f, 0.f, 0.f, 0.f};
  if (live) {
    mu4 = *reinterpret_cast<const float4*>(mulv + b * L2p + l);
    lv4 = *reinterpret_cast<const float4*>(mulv + b * L2p + Lp + l);
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (l + e < L) {
        ev[e] = eps[b * L + l + e];
        if (dmu_ext) xm[e] = dmu_ext[b * L + l + e];
        if (dlv_ext) xv[e] = dlv_ext[b * L + l + e];
      }
  }
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  f32x4 acc[4];
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
  issue(0);
  if (NU > 1) issue(1);
  for (int s = 0; s < NU; ++s) {
    // step s needs X(s), W(s); the only younger pieces are step s + 1's ten
    if (s + 1 < NU) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const lds_char* ws = (s & 1) ? W1 : W0;
    bf16x8 x[2], w[4][2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) x[kk] = load_frag<16, true>(ring + LX_OFF + (s & 1) * LX_SLOT, 0, kk, lane);
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) w[cb][kk] = load_frag<64, false>(ws, cb * 16, kk, lane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // fragments are in registers: both slots may be refilled
    if (s + 2 < NU) issue(s + 2);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[cb][kk], x[kk], acc[cb], 0, 0, 0);
  }
  // lane (q, j) holds partial dz[j][16 cb + 4 q + e]: park the wave's 16 x 64 partial sums in its weight slot 0 (256-byte
  // rows: the 16-byte quad index XOR j & 7 spreads the eight lanes of a store group over the banks, as in k_latent_fwd)
#pragma unroll
  for (int cb = 0; cb < 4; ++cb)
    *(__attribute__((address_space(3))) f32x4*)(W0 + (j * 64 + (((cb * 4 + q) ^ (j & 7)) << 2)) * 4) = acc[cb];
  __syncthreads();

  float dmu[4] = {0.f, 0.f, 0.f, 0.f}, dlv[4] = {0.f, 0.f, 0.f, 0.f};
  if (live) {
    f32x4 pz[LB_WAVES];
#pragma unroll
    for (int w = 0; w < LB_WAVES; ++w)
      pz[w] = *(const __attribute__((address_space(3))) f32x4*)(smem + w * L_RING + (rr * 64 + ((((int)(l >> 2)) ^ (rr & 7)) << 2)) * 4);
    const float mua[4] = {mu4.x, mu4.y, mu4.z, mu4.w}, lva[4] = {lv4.x, lv4.y, lv4.z, lv4.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (l + e < L) {
        const float dz = (pz[0][e] + pz[1][e]) + (pz[2][e] + pz[3][e]);
        const float sd = __expf(0.5f * lva[e]);
        dmu[e] = dz + kl_beta * mua[e] * inv_nk;
        dlv[e] = dz * ev[e] * 0.5f * sd + kl_beta * 0.5f * (sd * sd - 1.f) * inv_nk;
        if (dmu_ext) dmu[e] += xm[e];   // gradients arriving from outside (autograd)
        if (dlv_ext) dlv[e] += xv[e];
      }
    }
  }
  const bf16x4 m4 = {(bf16_t)dmu[0], (bf16_t)dmu[1], (bf16_t)dmu[2], (bf16_t)dmu[3]};
  const bf16x4 v4 = {(bf16_t)dlv[0], (bf16_t)dlv[1], (bf16_t)dlv[2], (bf16_t)dlv[3]};
  *reinterpret_cast<bf16x4*>(dmulv + b * L2p + l) = m4;
  *reinterpret_cast<bf16x4*>(dmulv + b * L2p + Lp + l) = v4;
  // column sums (bias gradients of fc21 | fc22) through the activation slots of waves 0 and 1 (idle by now)
  *(__attribute__((address_space(3))) f32x4*)(smem + 0 * L_RING + LX_OFF + tid * 16) = f32x4{dmu[0], dmu[1], dmu[2], dmu[3]};
  *(__attribute__((address_space(3))) f32x4*)(smem + 1 * L_RING + LX_OFF + tid * 16) = f32x4{dlv[0], dlv[1], dlv[2], dlv[3]};
  __syncthreads();
  if (dbh_partial && tid < 64) {
    // column tid = group (tid / 4), element (tid % 4); rows in ascending order (k_reparam_bwd's order)
    const int g = tid >> 2, e = tid & 3;
    const __attribute__((address_space(3))) float* sm = (const __attribute__((address_space(3))) float*)(smem + 0 * L_RING + LX_OFF);
    const __attribute__((address_space(3))) float* sv = (const __attribute__((address_space(3))) float*)(smem + 1 * L_RING + LX_OFF);
    float a = 0.f, c = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      a += sm[(r * 16 + g) * 4 + e];
      c += sv[(r * 16 + g) * 4 + e];
    }
    dbh_partial[(long)grp * L2p + tid] = a;
    dbh_partial[(long)grp * L2p + Lp + tid] = c;
  }
}


// ---- the heads' backward as ONE streaming kernel that reads h1 once ------------------------------------------------
// dP1 = relu'(h1) * (dmulv Wh)  (autograd of fc21 | fc22 w.r.t. h1, masked by fc1's ReLU; K = 2 Lp = 128 only) and
// dWh = dmulv^T h1 need the same h1 tile: as the mask of the first and as the operand of the second.  The generic route
// (a dual launch of a 128 x 128-tile dgrad and a split-K wgrad) reads h1 twice and runs every tile as a block of its
// own -- load, two K tiles of MFMA, store, in lockstep over the whole chip: 15 us for 58 MB.  Here a workgroup owns
// 64 columns of h1 and 512 batch rows, keeps its slice of Wh (128 x 64) in LDS and walks the rows in tiles of 64:
// the tile of h1 and the rows of dmulv are staged once (LDS-DMA, two stages, counted vmcnt), feed both products, and
// the next tile's loads fly while this tile's dP1 is stored.  dWh accumulates in registers over the 8 tiles and leaves
// as one fp32 slab per 512-row group (the split-K partials rv_adam_multi sums), the column sums of dP1 (fc1's bias
// gradient) as one partial row per group.
constexpr int HB_TR = 64;                       // rows per tile
constexpr int HB_RG = 512;                      // rows per workgroup
constexpr int HB_AK = 2 * 8192;                 // dmulv tile, K-major (rows x 128 k as two 64-k images): dgrad's operand
constexpr int HB_AM = 16384;                    // the same rows, MN-major ([64 rows][128 l]): dWh's operand (its transpose)
constexpr int HB_HM = 8192;                     // h1 tile, MN-major ([64 rows][64 cols]): dWh's operand AND the mask
constexpr int HB_STAGE = HB_AK + HB_AM + HB_HM;   // 40 KiB
constexpr int HB_WH = 2 * 8192;                 // Wh slice, MN-major, two 64-k images
constexpr int HB_NS = 3;                        // stages: the one in use, the next, and the one being refilled
constexpr int HB_LDS = HB_WH + HB_NS * HB_STAGE;   // 136 KiB

// s_waitcnt vmcnt(n) for a run-time (wave-uniform) n in 0..15
__device__ __forceinline__ void wait_vm_exact(int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
  }
}

// Q8: the fp8 weight path's outputs (instantiations of their own: the bf16 step's kernel stays as it was).  H16: the dWh
// slabs as block-floating-point fp16 -- value * 2^e with one exponent per 32 x 32 granule and slab, 2^-e in `dwh_unscale`
// (the weight-gradient GEMMs' slab format, adam.h load_slab4: half the bytes written here and read back by the optimizer).
template <bool Q8, bool H16>
__global__ void __launch_bounds__(512)
k_heads_bwd(const bf16_t* __restrict__ dmulv, const bf16_t* __restrict__ Wh, const long ldw,
            const bf16_t* __restrict__ h1, const long ldh, bf16_t* __restrict__ dP1, const long ldp,
            float* __restrict__ db1_partial, float* __restrict__ dwh_slabs, const long lddw, const long Hp,
            const int wt, unsigned char* __restrict__ dP1q, const long ldq, const float* __restrict__ q_scale,
            float* __restrict__ amax_part, float* __restrict__ dwh_unscale) {
  extern __shared__ __attribute__((aligned(16))) char smem_dyn[];
  lds_char* smem = (lds_char*)smem_dyn;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..7
  // fp8 weight path (RV_OPT_FP8 = 1): dP1 also -- or only, when dP1 is NULL -- as fp8(dP1 * *q_scale), the MN-major
  // operand of fc1's fp8 weight gradient; max|dP1| of every wave's outputs for next step's scale (delayed scaling)
  float qs = 0.f, amax = 0.f;
  if constexpr (Q8) qs = dP1q ? *q_scale : 0.f;
  const int q = lane >> 4, j = lane & 15;
  const int nstrips = (int)(Hp / 64);
  const int g = (int)blockIdx.x / nstrips, cs_ = (int)blockIdx.x - g * nstrips;
  const long r_base = (long)g * HB_RG, c0 = (long)cs_ * 64;
  constexpr int NT = HB_RG / HB_TR;
  lds_char* const whI = smem;
  auto stage_at = [&](int s) { return smem + HB_WH + s * HB_STAGE; };
  // work split over the 8 waves: the dgrad tile (64 rows x 64 columns) as row block rb = wave & 3 and the column-block
  // PAIR cp = wave >> 2 (columns 32 cp .. 32 cp + 31: one 16-byte store per lane); dWh (128 l x 64 columns) as l block
  // `wave` and all four column blocks
  const int rb = wave & 3, cp = wave >> 2;

  // one LDS-DMA instruction (1 KiB) of a stage / of the Wh slice; the 8 waves deal them out round robin
  auto issue_stage = [&](int t, int s) {
    lds_char* st = stage_at(s);
    const bf16_t* dm = dmulv + (r_base + (long)HB_TR * t) * 128;
    const bf16_t* hh = h1 + (r_base + (long)HB_TR * t) * ldh + c0;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int idx = wave + 8 * k;   // 0..39
      if (idx < 16) {                 // K-major halves of dmulv
        const int half = idx >> 3, i = idx & 7;
        const int r = 8 * i + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
        dma16(dm + r * 128 + 64 * half + c * 8, st + half * 8192 + i * 1024);
      } else if (idx < 32) {          // MN-major image of the same rows: [row][128 l]
        const int t_ = idx - 16;
        const int kr = 4 * t_ + (lane >> 4), p16 = lane & 15;
        const int c32 = (p16 >> 1) ^ swz_mn<128>(kr);
        dma16(dm + kr * 128 + (c32 * 2 + (p16 & 1)) * 8, st + HB_AK + t_ * 1024);
      } else {                        // h1 tile, MN-major: [row][64 cols]
        const int i = idx - 32;
        const int kr = 8 * i + (lane >> 3), p16 = lane & 7;
        const int c32 = (p16 >> 1) ^ swz_mn<64>(kr);
        dma16(hh + kr * ldh + (c32 * 2 + (p16 & 1)) * 8, st + HB_AK + HB_AM + i * 1024);
      }
    }
  };
  // Wh slice: rows k (2 x 64 latent outputs), this block's 64 columns: 16 instructions, 2 per wave
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int idx = wave + 8 * k, half = idx >> 3, i = idx & 7;
    const int kr = 8 * i + (lane >> 3), p16 = lane & 7;
    const int c32 = (p16 >> 1) ^ swz_mn<64>(kr);
    dma16(Wh + (long)(64 * half + kr) * ldw + c0 + (c32 * 2 + (p16 & 1)) * 8, whI + half * 8192 + i * 1024);
  }
  issue_stage(0, 0);
  issue_stage(1, 1);

  f32x4 acc2[4];      // dWh: row l = 16 wave + j, columns 16 cb + 4 q + e
  float cs[2][4];     // column sums of dP1 over this lane's rows, column blocks 2 cp, 2 cp + 1
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) acc2[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) cs[c][e] = 0.f;
  bf16x8 wf[2][2][2];   // the Wh slice's fragments [half][kk][column block of the pair]: read from LDS once
  for (int t = 0; t < NT; ++t) {
    const int s = t % HB_NS;
    lds_char* st = stage_at(s);
    // Vector-memory operations of this wave YOUNGER than its five loads of tile t (one in-order queue of loads and
    // stores): tiles 0, 1 were requested in the prologue, tile t >= 2 right behind the barrier of tile t - 2; every
    // tile ends with 1 store
    const int younger = t == 0 ? 5 : t == 1 ? 6 : (t + 1 < NT ? 7 : 2);
    wait_vm_exact(younger);               // this wave's share of the stage (and of Wh) has landed ...
    __builtin_amdgcn_s_barrier();         // ... and so has everybody else's -- and every wave has finished tile t - 1:
    asm volatile("" ::: "memory");        // ITS stage is free, and takes tile t + 2 (ONE barrier per tile, three stages)
    if (t + 2 < NT) issue_stage(t + 2, (t + 2) % HB_NS);
    if (t == 0) {
#pragma unroll
      for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int c = 0; c < 2; ++c) wf[half][kk][c] = load_frag<64, false>(whI + half * 8192, (2 * cp + c) * 16, kk, lane);
    }
    // ---- dgrad: rows 16 rb + j, columns 32 cp .. +31, K = 128
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const bf16x8 x = load_frag<64, true>(st + half * 8192, 16 * rb, kk, lane);
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[half][kk][c], x, acc[c], 0, 0, 0);
      }
    // ---- dWh += dmulv_tile^T h1_tile: row block `wave` of l, contraction over the tile's 64 rows
    const lds_char* am = st + HB_AK;
    const lds_char* hm = st + HB_AK + HB_AM;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const bf16x8 xa = load_frag<128, false>(am, wave * 16, kk, lane);
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        const bf16x8 hb = load_frag<64, false>(hm, cb * 16, kk, lane);
        acc2[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hb, xa, acc2[cb], 0, 0, 0);
      }
    }
    // ---- ReLU mask from the same h1 tile (element (row r, column c) of the MN-major image), column sums
    const int r = 16 * rb + j;
    float v[2][4];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int cb = 2 * cp + c;
      const bf16x4 m4 = *(const __attribute__((address_space(3))) bf16x4*)(hm + r * 128 + ((cb ^ swz_mn<64>(r)) * 32) + q * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[c][e] = (float)m4[e] > 0.f ? acc[c][e] : 0.f;
        cs[c][e] += v[c][e];
      }
    }
    // ---- dP1: the accumulator-direct store of gemm_bf16.h (pair the two column blocks: 8 columns per lane)
    {
      float lo[4], hi[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a_ = v[0][e], b_ = v[1][e];
        swap_rows16(a_, b_);
        lo[e] = a_;
        hi[e] = b_;
      }
      const long orow = r_base + (long)HB_TR * t + r;
      const int ocol = (2 * cp + (q & 1)) * 16 + (q >> 1) * 8;
      if (!Q8 || dP1) {
        const bf16x8 o = {(bf16_t)lo[0], (bf16_t)lo[1], (bf16_t)lo[2], (bf16_t)lo[3],
                          (bf16_t)hi[0], (bf16_t)hi[1], (bf16_t)hi[2], (bf16_t)hi[3]};
        store_out16((bf16x8*)(dP1 + orow * ldp + c0 + ocol), o, wt);
      }
      if constexpr (Q8) {
        float q8[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          amax = fmaxf(amax, fmaxf(fabsf(lo[e]), fabsf(hi[e])));
          q8[e] = lo[e] * qs;
          q8[4 + e] = hi[e] * qs;
        }
        if (dP1q) *(unsigned long long*)(dP1q + orow * ldq + c0 + ocol) = pack_fp8x8(q8);
      }
    }
  }
  // ---- dWh slab of this row group
  if constexpr (H16) {
    // one exponent per 32-row granule = per PAIR of waves (a wave holds 16 rows) and slab, for both 32-column granules of
    // the strip: the pair's maximum through LDS (the stage of tile NT - 2 is free behind the last tile's barrier)
    float mx = 0.f;
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int e = 0; e < 4; ++e) mx = fmaxf(mx, fabsf(acc2[cb][e]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    __attribute__((address_space(3))) float* ex = (__attribute__((address_space(3))) float*)stage_at((NT - 2) % HB_NS);
    if (lane == 0) ex[wave] = mx;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    mx = fmaxf(mx, ex[wave ^ 1]);
    // exponent and scale as the GEMM epilogue's (gemm_bf16.h): 2^(14 - (e - 127)), both it and its reciprocal normal
    const int e_ = (int)((__float_as_uint(mx) >> 23) & 0xffu);
    int sb = 268 - e_;
    sb = sb < 1 ? 1 : (sb > 253 ? 253 : sb);
    const float f16s = __uint_as_float((unsigned)sb << 23);
    const long us_ld = Hp / 32;
    if ((wave & 1) == 0 && lane < 2)
      dwh_unscale[(long)g * 4 * us_ld + (wave >> 1) * us_ld + c0 / 32 + lane] = __uint_as_float((unsigned)(254 - sb) << 23);
    typedef _Float16 f16x4_ __attribute__((ext_vector_type(4)));
    _Float16* slab = (_Float16*)dwh_slabs + (long)g * 128 * lddw + c0;
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      const f16x4_ h = {(_Float16)(acc2[cb][0] * f16s), (_Float16)(acc2[cb][1] * f16s), (_Float16)(acc2[cb][2] * f16s),
                        (_Float16)(acc2[cb][3] * f16s)};
      *(f16x4_*)(slab + (long)(wave * 16 + j) * lddw + cb * 16 + q * 4) = h;
    }
  } else {
    float* slab = dwh_slabs + (long)g * 128 * lddw + c0;
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
      store_out16((f32x4*)(slab + (long)(wave * 16 + j) * lddw + cb * 16 + q * 4), acc2[cb], wt);
  }
  // ---- column sums: over the 16 row lanes, then over the 4 row-block waves of a column pair (through the Wh slice's
  // LDS, idle by now)
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float x = cs[c][e];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) x += __shfl_xor(x, o, 64);
      cs[c][e] = x;
    }
  __attribute__((address_space(3))) float* red = (__attribute__((address_space(3))) float*)whI;
  // (nothing has read LDS since the last tile's second barrier; raw barrier: __syncthreads() would also wait for the
  // dP1 / slab stores to drain)
  if (j == 0)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) red[rb * 64 + (2 * cp + c) * 16 + q * 4 + e] = cs[c][e];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (db1_partial && tid < 64) db1_partial[(long)g * Hp + c0 + tid] = (red[tid] + red[64 + tid]) + (red[128 + tid] + red[192 + tid]);
  if constexpr (Q8) {
    if (amax_part) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
      if (lane == 0) amax_part[blockIdx.x * 8 + wave] = amax;
    }
  }
}


// ---- the same two launches for padded latent widths above 64 (the reference's own latent_dim = 256: default.ini:18,
// kelsey_iterable.ini:17) ------------------------------------------------------------------------------------------------
// The row-local kernels above make every workgroup stream ALL latent-sized weights for its 16 rows: 832 KB at Lp = 64,
// 3 MB at Lp = 256 -- 770 MB through the L2 -> LDS ports per launch, 50 us at their ~60 GB/s each.  From Lp = 128 on
// the heads GEMM ([Bp, 2 Lp] <- K = Hp) and dz ([Bp, Lp] <- K = Hp) are GEMMs with hundreds of output tiles, so they run
// on gemm_bf16.h's tiled body and the elementwise steps ride in its epilogues (EPI_REPARAM, EPI_REPARAM_BWD):
//   forward : 64 x 128 tiles, one per (64 batch rows, 64 latents x both heads) = (Bp / 64) (Lp / 64) workgroups of 8 waves
//             (256 at the reference's shape); 24 KB per K tile through the port; mu | logvar, z, eps and the KL partials
//             leave the accumulators directly -- no fp32 slabs, no reparameterisation launch.  fc3 (K = Lp) follows as a
//             plain forward GEMM (rv_linear_fwd_ex): its A operand needs every latent of a row, i.e. all Lp / 64 tiles.
//   backward: dz on 64 x 128 tiles (8 waves, 24 KB per K tile) with the reparameterisation backward and the head biases'
//             column sums in the epilogue; fc3's weight gradient dW3 = dP3^T z on the SAME launch's extra workgroups (the
//             second reader of dP3, as in k_latent_bwd) on 128 x 128 tiles; two workgroups share a CU (72 KiB each); one
//             more workgroup finishes the loss scalar.  (A first version ran both on 64 x 64 tiles of 4 waves: 29.2 us at the
//             reference's shape -- 268 MB through the L2 -> LDS ports for 8.6 GFLOP; these tiles move 167 MB.)
constexpr int HG_STAGES = 6;                            // five 24 KB tiles in flight per CU (four stages: 19.5 us, 43 GB/s per CU)
constexpr int HG_LDS = HG_STAGES * (64 + 128) * 128;   // 144 KiB

__global__ void __launch_bounds__(512) k_heads_reparam_gemm(const GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem_dyn[];
  gemm_body<64, 128, 2, 4, true, true, EPI_REPARAM, HG_STAGES>(p, blockIdx.x, smem_dyn);
}

// Large batches (default.ini trains at batch_size = 131072): with more than four 64-row tiles per CU the bytes a tile pulls
// through the port per flop are what matters, not filling the chip: 256 x 128 tiles (48 KB per K tile for four times the
// rows).  The row-local kernels of Lp = 64 hand over to these forms there too: they re-stream every weight for each 16 rows,
// 0.06-0.08 of the MFMA peak at B = 131072 (profiles/r06_batch_sweep.jsonl).
constexpr int LG_BIG_TILES = 1024;                       // (Bp / 64) (Lp / 64) above which the 256-row forms run
constexpr int HGB_STAGES = 3;
constexpr int HGB_LDS = HGB_STAGES * (256 + 128) * 128;  // 144 KiB
__global__ void __launch_bounds__(512) k_heads_reparam_gemm_big(const GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem_dyn[];
  gemm_body<256, 128, 4, 2, true, true, EPI_REPARAM, HGB_STAGES>(p, blockIdx.x, smem_dyn);
}

// From a padded latent width of 128 on (the heads' N = 2 Lp >= 256) large batches run on the 256 x 256 ping-pong loop: a tile then
// covers 128 latents x both heads, the gathered B rows go through the loop's own staging (mainloop_pingpong's b_rows).  At
// default.ini's shape the 256 x 128 ring above ran this GEMM at 0.25 of the MFMA peak (1.7 us per K tile of a tile, 444 us);
// the ping-pong loop's 1.3 us per K tile covers twice the columns.
constexpr int PP_LDS = 2 * 4 * 128 * 128;                // 128 KiB: two buffers of {A0, A1, B0, B1} half tiles
__global__ void __launch_bounds__(512) k_heads_reparam_gemm_pp(const GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem_dyn[];
  gemm_body<256, 256, 2, 4, true, true, EPI_REPARAM, 8>(p, blockIdx.x, smem_dyn);
}

constexpr int DZ_STAGES = 5, DW3_STAGES = 2;
constexpr int DZ_LDS = DZ_STAGES * (64 + 64) * 128;     // 80 KiB (the dW3 blocks use 64 of them): two workgroups per CU

// FORM 0: 64 x 64 dz tiles, 128 x 128 dW3 tiles (two workgroups per CU).  1: large batches -- 256 x 128 dz tiles.  2: large
// batches at a padded latent width of 256 -- dz on ONE 256 x 256 ping-pong tile per 256 rows (dP3 read once), dW3 on 256 x 256
// ping-pong tiles as well (rv_latent_bwd_pp: the plan gives it enough K splits to fill the chip once).
template <int FORM>
__global__ void __launch_bounds__(512)
k_dz_reparam_gemm(const GemmArgs dz, const GemmArgs w3grad, const int n_dz, const int n_w3, const long B, const long L,
                  const long S, const float kl_beta, const float* __restrict__ mse_partial, const int n_mse,
                  const float* __restrict__ kl_partial, const int n_kl, float* __restrict__ loss_out,
                  const long long* __restrict__ step_counter, const int ring_n) {
  extern __shared__ __attribute__((aligned(16))) char smem_dyn[];
  const int bid = (int)blockIdx.x, tid = threadIdx.x;
  // the weight-gradient blocks come FIRST in dispatch order: each contracts over a slice of the whole batch and is the
  // launch's longest block by far at large batches (behind the dz blocks they started when those were done: 413 us at
  // B = 131072, L = 64)
  if (bid < n_w3) {
    // (a padded latent width of 64 leaves no room for 128-column tiles: 128 x 64 there, three ring slots = 72 KiB)
    if constexpr (FORM == 2) gemm_body<256, 256, 2, 4, false, false, EPI_F32, 8>(w3grad, bid, smem_dyn);
    else if (w3grad.N_valid < 128) gemm_body<128, 64, 4, 2, false, false, EPI_F32, 3>(w3grad, bid, smem_dyn);
    else gemm_body<128, 128, 2, 4, false, false, EPI_F32, DW3_STAGES>(w3grad, bid, smem_dyn);
    return;
  }
  if (bid < n_w3 + n_dz) {
    if constexpr (FORM == 2) gemm_body<256, 256, 2, 4, true, false, EPI_REPARAM_BWD, 8>(dz, bid - n_w3, smem_dyn);
    else if constexpr (FORM == 1) gemm_body<256, 128, 4, 2, true, false, EPI_REPARAM_BWD, HGB_STAGES>(dz, bid - n_w3, smem_dyn);
    else gemm_body<64, 64, 4, 2, true, false, EPI_REPARAM_BWD, DZ_STAGES>(dz, bid - n_w3, smem_dyn);
    return;
  }
  if (loss_out && mse_partial && kl_partial) {   // the loss scalar (k_reparam_bwd's extra block, same summation order)
    float* red = (float*)smem_dyn;
    float m = 0.f, k = 0.f;
    // (threads 0..255 only: k_reparam_bwd's extra block has 256 threads, and the sums must come out bit-equal to its)
    if (tid < 256) {
      for (int i = tid; i < n_mse; i += 256) m += mse_partial[i];
      for (int i = tid; i < n_kl; i += 256) k += kl_partial[i];
    }
    m = block_sum<8>(m, red);
    k = block_sum<8>(k, red);
    if (tid == 0) {
      const float mse = m / ((float)B * (float)S);
      const float inv_nk = 1.0f / ((float)B * (float)L);
      const float kld = -0.5f * k * inv_nk;
      if (step_counter && ring_n > 0) loss_out += 4 * ((*step_counter - 1) % ring_n);
      loss_out[0] = mse + kl_beta * kld;
      loss_out[1] = mse;
      loss_out[2] = kld;
    }
  }
}

// host side of the two (called by rv_latent_fwd_ex / rv_latent_bwd for Lp > 64; arguments checked there)
int heads_reparam_gemm(const void* h, long ldh, const void* wh, long ldwh, const float* bias_heads, long Bp, long Hp, long Lp,
                       long B, long L, const float* eps_in, float* eps_out, unsigned long long seed,
                       const long long* step_counter, float* mulv, void* z, float* kl_partial, hipStream_t st) {
  GemmArgs a{};
  a.A = (const bf16_t*)h; a.lda = ldh; a.B = (const bf16_t*)wh; a.ldb = ldwh;
  a.k_tiles = (int)(Hp / 64); a.M_valid = (int)B; a.N_valid = (int)(2 * Lp);
  const bool big = (Bp / 64) * (Lp / 64) > LG_BIG_TILES && Bp % 256 == 0;
  const bool pp = big && Lp >= 128;   // (Hp / 64 is even: the GEMM forms require Hp % 128 == 0)
  a.tiles_m = (int)(Bp / (big ? 256 : 64)); a.tiles_n = (int)(Lp / (pp ? 128 : 64)); a.splits = 1;
  a.bias = bias_heads; a.lat_lp = Lp; a.lat_l = L; a.eps_in = eps_in; a.eps_out = eps_out; a.seed = seed;
  a.step_counter = step_counter; a.mulv = mulv; a.z = (bf16_t*)z; a.kl_partial = kl_partial;
  a.wt = rv_store_wt;
  static std::atomic<unsigned long long> attr_done0{0};
  lds_opt_in((const void*)k_heads_reparam_gemm, HG_LDS, attr_done0);
  static std::atomic<unsigned long long> attr_done1{0};
  lds_opt_in((const void*)k_heads_reparam_gemm_big, HGB_LDS, attr_done1);
  static std::atomic<unsigned long long> attr_done2{0};
  lds_opt_in((const void*)k_heads_reparam_gemm_pp, PP_LDS, attr_done2);
  if (pp) hipLaunchKernelGGL(k_heads_reparam_gemm_pp, dim3((unsigned)(a.tiles_m * a.tiles_n)), dim3(512), PP_LDS, st, a);
  else if (big) hipLaunchKernelGGL(k_heads_reparam_gemm_big, dim3((unsigned)(a.tiles_m * a.tiles_n)), dim3(512), HGB_LDS, st, a);
  else hipLaunchKernelGGL(k_heads_reparam_gemm, dim3((unsigned)(a.tiles_m * a.tiles_n)), dim3(512), HG_LDS, st, a);
  RV_CHECK_LAUNCH();
  return RV_OK;
}

}  // namespace

extern "C" {

// Which form serves a shape: the row-local kernels (16 rows per workgroup, every weight through every CU) at a padded latent
// width of 64 while the batch is small enough that filling the chip is the issue; the GEMM forms above that width and for
// large batches.  One predicate for both directions and for the plan (internal.h).
int rv_latent_rowlocal(long Bp, long Hp, long Lp) { return Lp == 64 && Hp % 512 == 0 && Hp <= 2048 && Bp <= 8192; }

// Batch rows behind one NON-ZERO row of rv_latent_bwd's bias partial table [Bp / 16][2 Lp] (the rows between are written as
// zeros): 16 for the row-local kernel, the dz tile's rows for the GEMM forms.  The plan's optimizer descriptors skip the zeros.
int rv_latent_bwd_tile_rows(long Bp, long Hp, long Lp) {
  if (rv_latent_rowlocal(Bp, Hp, Lp)) return 16;
  return (Lp >= 128 && (Bp / 64) * (Lp / 64) > LG_BIG_TILES && Bp % 256 == 0) ? 256 : 64;
}

// The latent backward's ping-pong form (k_dz_reparam_gemm<2>): a large batch at a padded latent width of 256.  Its dW3 blocks are
// 256 x 256 tiles over Hp / 256 row tiles, so the plan splits K until they fill the chip once (even K tiles per block).
int rv_latent_bwd_pp(long Bp, long Hp, long Lp) {
  return Lp == 256 && Hp % 256 == 0 && (Bp / 64) * (Lp / 64) > LG_BIG_TILES && Bp % 256 == 0;
}

int rv_latent_fwd(const void* h_bf16, long ldh, const void* wh_bf16, long ldwh, const float* bias_heads,
                  const void* w3_bf16, long ldw3, const float* bias3, long Bp, long Hp, long Lp, long B, long L,
                  const float* eps_in, float* eps_out, unsigned long long seed, const long long* step_counter,
                  float* mulv, void* z_bf16, float* kl_partial, void* h3_bf16, long ldh3, void* stream) {
  return rv_latent_fwd_ex(h_bf16, ldh, wh_bf16, ldwh, bias_heads, w3_bf16, ldw3, bias3, Bp, Hp, Lp, B, L, eps_in, eps_out, seed,
                          step_counter, mulv, z_bf16, kl_partial, h3_bf16, ldh3, nullptr, 0, nullptr, nullptr, stream);
}

// rv_latent_fwd with the fp8 forward's extra outputs of fc3 (NULL = not wanted): h3 also as fp8(h3 * *q_scale) and
// max|h3| of every wave's outputs in amax_part[8 * (Bp / 16)] (rv_linear_fwd_ex's per-block maxima, finer).
int rv_latent_fwd_ex(const void* h_bf16, long ldh, const void* wh_bf16, long ldwh, const float* bias_heads,
                     const void* w3_bf16, long ldw3, const float* bias3, long Bp, long Hp, long Lp, long B, long L,
                     const float* eps_in, float* eps_out, unsigned long long seed, const long long* step_counter,
                     float* mulv, void* z_bf16, float* kl_partial, void* h3_bf16, long ldh3, void* h3_fp8, long ldq,
                     const float* q_scale, float* amax_part, void* stream) {
  RV_REQUIRE(!h3_fp8 || (q_scale && ldq >= Hp && ldq % 8 == 0 && ((uintptr_t)h3_fp8 & 7) == 0), RV_ERR_SHAPE,
             "rv_latent_fwd: the fp8 output needs a scale and 8-byte aligned rows");
  // w3_bf16 == NULL: heads + reparameterisation only (z, mu | logvar, the KL partials); fc3 is then the caller's
  const bool heads_only = !w3_bf16;
  // (h3_bf16 may be NULL beside h3_fp8: the fp8 step whose only reader of h3 is the fp8 fc4 backward)
  RV_REQUIRE(h_bf16 && wh_bf16 && bias_heads && mulv && z_bf16 && kl_partial && (heads_only || (bias3 && (h3_bf16 || h3_fp8))), RV_ERR_NULL,
             "rv_latent_fwd: null pointer");
  RV_REQUIRE(!heads_only || (!h3_fp8 && !amax_part), RV_ERR_UNSUPPORTED, "rv_latent_fwd: the fp8 outputs belong to fc3");
  RV_REQUIRE(eps_in || eps_out, RV_ERR_NULL, "rv_latent_fwd: need eps_in or eps_out");
  RV_REQUIRE(Lp == 64 || Lp == 128 || Lp == 256, RV_ERR_UNSUPPORTED,
             "rv_latent_fwd: serves padded latent widths of 64, 128 and 256 (got %ld)", Lp);
  if (!rv_latent_rowlocal(Bp, Hp, Lp)) {
    // GEMM forms (k_heads_reparam_gemm, then fc3 as a forward GEMM): any hidden width that is a multiple of 128
    RV_REQUIRE(Bp > 0 && Bp % 64 == 0 && Hp > 0 && Hp % 128 == 0 && B <= Bp && L <= Lp && ldh >= Hp && ldwh >= Hp && ldh % 8 == 0 &&
                   ldwh % 8 == 0 && (heads_only || (ldw3 >= Lp && ldh3 >= Hp && ldw3 % 8 == 0 && ldh3 % 8 == 0)),
               RV_ERR_SHAPE, "rv_latent_fwd: bad extents Bp %ld Hp %ld Lp %ld (Bp a multiple of 64, Hp of 128)", Bp, Hp, Lp);
    RV_REQUIRE((((uintptr_t)h_bf16 | (uintptr_t)wh_bf16 | (uintptr_t)w3_bf16 | (uintptr_t)bias_heads | (uintptr_t)bias3 |
                 (uintptr_t)mulv | (uintptr_t)z_bf16 | (uintptr_t)h3_bf16) & 15) == 0,
               RV_ERR_SHAPE, "rv_latent_fwd: operands must be 16-byte aligned");
    const int rc = heads_reparam_gemm(h_bf16, ldh, wh_bf16, ldwh, bias_heads, Bp, Hp, Lp, B, L, eps_in, eps_out, seed, step_counter,
                                      mulv, z_bf16, kl_partial, (hipStream_t)stream);
    if (rc || heads_only) return rc;
    return rv_linear_fwd_ex(z_bf16, Lp, w3_bf16, ldw3, bias3, Bp, Hp, Lp, RV_ACT_RELU, h3_bf16, ldh3, h3_fp8, ldq, q_scale, amax_part,
                            stream);
  }
  RV_REQUIRE(Bp > 0 && Bp % LAT_ROWS == 0 && Hp > 0 && B <= Bp && L <= Lp && ldh >= Hp && ldwh >= Hp &&
                 (heads_only || (ldw3 >= Lp && ldh3 >= Hp && ldw3 % 8 == 0 && ldh3 % 8 == 0)) && ldh % 8 == 0 && ldwh % 8 == 0,
             RV_ERR_SHAPE, "rv_latent_fwd: bad extents Bp %ld Hp %ld", Bp, Hp);
  RV_REQUIRE((((uintptr_t)h_bf16 | (uintptr_t)wh_bf16 | (uintptr_t)w3_bf16 | (uintptr_t)bias_heads | (uintptr_t)bias3 |
               (uintptr_t)mulv | (uintptr_t)z_bf16 | (uintptr_t)h3_bf16) & 15) == 0,
             RV_ERR_SHAPE, "rv_latent_fwd: operands must be 16-byte aligned");
  static std::atomic<unsigned long long> attr_done{0};
  lds_opt_in((const void*)k_latent_fwd, L_LDS, attr_done);
  hipLaunchKernelGGL(k_latent_fwd, dim3((unsigned)(Bp / LAT_ROWS)), dim3(512), L_LDS, (hipStream_t)stream,
                     (const bf16_t*)h_bf16, ldh, (const bf16_t*)wh_bf16, ldwh, bias_heads, (const bf16_t*)w3_bf16, ldw3, bias3,
                     Hp, B, L, eps_in, eps_out, (uint64_t)seed, step_counter, mulv, (bf16_t*)z_bf16, kl_partial,
                     (bf16_t*)h3_bf16, ldh3, rv_store_wt, (unsigned char*)h3_fp8, ldq, q_scale, amax_part);
  RV_CHECK_LAUNCH();
  return RV_OK;
}

int rv_latent_bwd(const void* dp3_bf16, long lddp, const void* w3_bf16, long ldw3, long Bp, long Hp, long Lp, long B,
                  long L, long S, const float* mulv, const float* eps, float kl_beta, const float* dmu_ext,
                  const float* dlv_ext, void* dmulv_bf16, float* dbh_partial, const float* mse_partial, int n_mse,
                  const float* kl_partial, int n_kl, float* loss_out, const long long* step_counter, int ring,
                  const void* z_bf16, long ldz, float* dw3_slabs, long lddw3, int dw3_splits, void* stream) {
  RV_REQUIRE(dp3_bf16 && w3_bf16 && mulv && eps && dmulv_bf16, RV_ERR_NULL, "rv_latent_bwd: null pointer");
  RV_REQUIRE(Lp == 64 || Lp == 128 || Lp == 256, RV_ERR_UNSUPPORTED,
             "rv_latent_bwd: serves padded latent widths of 64, 128 and 256 (got %ld)", Lp);
  const bool rowlocal = rv_latent_rowlocal(Bp, Hp, Lp);
  RV_REQUIRE(rowlocal || Hp % 128 == 0, RV_ERR_UNSUPPORTED, "rv_latent_bwd: the hidden width must be a multiple of 128 (got %ld)", Hp);
  RV_REQUIRE(Bp > 0 && Bp % 64 == 0 && B <= Bp && L <= Lp && lddp >= Hp && ldw3 >= Lp && lddp % 8 == 0 &&
                 ldw3 % 8 == 0, RV_ERR_SHAPE, "rv_latent_bwd: bad extents Bp %ld Hp %ld", Bp, Hp);
  RV_REQUIRE((((uintptr_t)dp3_bf16 | (uintptr_t)w3_bf16 | (uintptr_t)mulv | (uintptr_t)dmulv_bf16) & 15) == 0, RV_ERR_SHAPE,
             "rv_latent_bwd: operands must be 16-byte aligned");
  GemmArgs g{};
  int n_w3 = 0;
  if (z_bf16) {   // fc3's weight gradient rides along: dW3 [Hp, Lp] = dP3^T z, `dw3_splits` fp32 slabs over the batch
    RV_REQUIRE(dw3_slabs && dw3_splits >= 1 && (Bp / 64) % dw3_splits == 0 && ldz >= Lp && ldz % 8 == 0 && lddw3 >= Lp &&
                   ((uintptr_t)z_bf16 & 15) == 0 && ((uintptr_t)dw3_slabs & 15) == 0 && lddw3 % 4 == 0,
               RV_ERR_SHAPE, "rv_latent_bwd: bad weight-gradient arguments (splits %d)", dw3_splits);
    g.A = (const bf16_t*)dp3_bf16; g.lda = lddp; g.B = (const bf16_t*)z_bf16; g.ldb = ldz;
    g.k_tiles = (int)(Bp / 64 / dw3_splits); g.M_valid = (int)Hp; g.N_valid = (int)Lp;
    g.out_f32 = dw3_slabs; g.ld_f32 = lddw3; g.split_stride_f32 = Hp * lddw3;
    g.tiles_m = (int)(Hp / 64); g.tiles_n = (int)(Lp / 64); g.splits = dw3_splits;
    g.wt = rv_store_wt;
    if (!rowlocal) {   // (the GEMM form's dW3 blocks run on 128 x 128 tiles, 128 x 64 at a padded latent width of 64)
      g.tiles_m = (int)(Hp / 128); g.tiles_n = Lp >= 128 ? (int)(Lp / 128) : 1;
      // (the ping-pong loop walks K tiles in pairs: a caller's odd split keeps the 256 x 128 form for the whole launch)
      if (rv_latent_bwd_pp(Bp, Hp, Lp) && g.k_tiles % 2 == 0) { g.tiles_m = (int)(Hp / 256); g.tiles_n = 1; }
    }
    n_w3 = g.tiles_m * g.tiles_n * g.splits;
  }
  if (!rowlocal) {   // GEMM form: dz tiles with the reparameterisation backward in their epilogue (k_dz_reparam_gemm)
    const bool big = rv_latent_bwd_tile_rows(Bp, Hp, Lp) == 256;
    const bool pp = rv_latent_bwd_pp(Bp, Hp, Lp) && (!z_bf16 || g.k_tiles % 2 == 0);
    GemmArgs d{};
    d.A = (const bf16_t*)dp3_bf16; d.lda = lddp; d.B = (const bf16_t*)w3_bf16; d.ldb = ldw3;
    d.k_tiles = (int)(Hp / 64); d.M_valid = (int)B; d.N_valid = (int)Lp;
    d.tiles_m = (int)(Bp / (big ? 256 : 64)); d.tiles_n = pp ? 1 : (int)(Lp / (big ? 128 : 64)); d.splits = 1;
    d.lat_lp = Lp; d.lat_l = L; d.mulv = const_cast<float*>(mulv); d.eps = eps; d.kl_beta = kl_beta;
    d.inv_nk = 1.0f / ((float)B * (float)L); d.dmu_ext = dmu_ext; d.dlv_ext = dlv_ext;
    d.dmulv = (bf16_t*)dmulv_bf16; d.dbh_partial = dbh_partial;
    d.wt = rv_store_wt;
    const int n_dz = d.tiles_m * d.tiles_n;
    static std::atomic<unsigned long long> attr_gemm0{0};
    lds_opt_in((const void*)k_dz_reparam_gemm<0>, DZ_LDS, attr_gemm0);
    static std::atomic<unsigned long long> attr_gemm1{0};
    lds_opt_in((const void*)k_dz_reparam_gemm<1>, HGB_LDS, attr_gemm1);
    static std::atomic<unsigned long long> attr_gemm2{0};
    lds_opt_in((const void*)k_dz_reparam_gemm<2>, PP_LDS, attr_gemm2);
    if (pp)
      hipLaunchKernelGGL(k_dz_reparam_gemm<2>, dim3((unsigned)(n_dz + n_w3 + 1)), dim3(512), PP_LDS, (hipStream_t)stream, d, g,
                         n_dz, n_w3, B, L, S, kl_beta, mse_partial, n_mse, kl_partial, n_kl, loss_out, step_counter, ring);
    else if (big)
      hipLaunchKernelGGL(k_dz_reparam_gemm<1>, dim3((unsigned)(n_dz + n_w3 + 1)), dim3(512), HGB_LDS, (hipStream_t)stream, d, g,
                         n_dz, n_w3, B, L, S, kl_beta, mse_partial, n_mse, kl_partial, n_kl, loss_out, step_counter, ring);
    else
      hipLaunchKernelGGL(k_dz_reparam_gemm<0>, dim3((unsigned)(n_dz + n_w3 + 1)), dim3(512), DZ_LDS, (hipStream_t)stream, d, g,
                         n_dz, n_w3, B, L, S, kl_beta, mse_partial, n_mse, kl_partial, n_kl, loss_out, step_counter, ring);
    RV_CHECK_LAUNCH();
    return RV_OK;
  }
  static std::atomic<unsigned long long> attr_done{0};
  lds_opt_in((const void*)k_latent_bwd, LB_LDS, attr_done);
  const int n_rows = (int)(Bp / LAT_ROWS);
  hipLaunchKernelGGL(k_latent_bwd, dim3((unsigned)(n_rows + n_w3 + 1)), dim3(256), LB_LDS, (hipStream_t)stream,
                     (const bf16_t*)dp3_bf16, lddp, (const bf16_t*)w3_bf16, ldw3, Hp, B, L, S, mulv, eps, kl_beta, dmu_ext,
                     dlv_ext, (bf16_t*)dmulv_bf16, dbh_partial, mse_partial, n_mse, kl_partial, n_kl, loss_out,
                     step_counter, ring, n_rows, g);
  RV_CHECK_LAUNCH();
  return RV_OK;
}

int rv_heads_bwd(const void* dmulv_bf16, const void* wh_bf16, long ldw, const void* h1_bf16, long ldh, long Bp, long Hp,
                 long Lp, void* dp1_bf16, long ldp, float* db1_partial, float* dwh_slabs, long lddw, void* stream) {
  RV_REQUIRE(dp1_bf16, RV_ERR_NULL, "rv_heads_bwd: null pointer");
  return rv_heads_bwd_ex(dmulv_bf16, wh_bf16, ldw, h1_bf16, ldh, Bp, Hp, Lp, dp1_bf16, ldp, db1_partial, dwh_slabs, lddw, nullptr,
                         0, nullptr, nullptr, nullptr, stream);
}

// rv_heads_bwd with the fp8 weight path's extra outputs (NULL = not wanted): dP1 also as fp8(dP1 * *q_scale) in
// dp1_fp8 [Bp, ldq bytes] -- then dp1_bf16 may be NULL -- and max|dP1| of every wave's outputs in
// amax_part[8 * (Bp / 512) * (Hp / 64)].  dwh_unscale (NULL = fp32 slabs): the dWh slabs as block-floating-point fp16 --
// `dwh_slabs` then holds fp16 elements with the same element strides, and dwh_unscale[(Bp / 512) * 4 * (Hp / 32)] takes
// 2^-e per slab and 32 x 32 granule (rv_param_desc.grad_unscale: us_ld = Hp / 32, us_split_stride = 4 * us_ld).
int rv_heads_bwd_ex(const void* dmulv_bf16, const void* wh_bf16, long ldw, const void* h1_bf16, long ldh, long Bp, long Hp,
                    long Lp, void* dp1_bf16, long ldp, float* db1_partial, float* dwh_slabs, long lddw, void* dp1_fp8, long ldq,
                    const float* q_scale, float* amax_part, float* dwh_unscale, void* stream) {
  RV_REQUIRE(dmulv_bf16 && wh_bf16 && h1_bf16 && (dp1_bf16 || dp1_fp8) && dwh_slabs, RV_ERR_NULL, "rv_heads_bwd: null pointer");
  RV_REQUIRE(!dp1_fp8 || (q_scale && ldq >= Hp && ldq % 16 == 0 && ((uintptr_t)dp1_fp8 & 15) == 0), RV_ERR_SHAPE,
             "rv_heads_bwd: the fp8 output needs a scale and 16-byte aligned rows");
  RV_REQUIRE(Lp == 64, RV_ERR_UNSUPPORTED, "rv_heads_bwd: built for a padded latent width of 64 (got %ld)", Lp);
  RV_REQUIRE(Bp > 0 && Bp % HB_RG == 0 && Hp > 0 && Hp % 64 == 0, RV_ERR_UNSUPPORTED,
             "rv_heads_bwd: the padded batch must be a multiple of 512 and the padded hidden width of 64 (got %ld, %ld)", Bp, Hp);
  RV_REQUIRE(ldw >= Hp && ldh >= Hp && lddw >= Hp && ldw % 8 == 0 && ldh % 8 == 0 && lddw % 4 == 0,
             RV_ERR_SHAPE, "rv_heads_bwd: bad leading dimensions");
  RV_REQUIRE((((uintptr_t)dmulv_bf16 | (uintptr_t)wh_bf16 | (uintptr_t)h1_bf16 | (uintptr_t)dp1_bf16 | (uintptr_t)dwh_slabs) & 15) == 0,
             RV_ERR_SHAPE, "rv_heads_bwd: operands must be 16-byte aligned");
  RV_REQUIRE(!dp1_bf16 || (ldp >= Hp && ldp % 8 == 0), RV_ERR_SHAPE, "rv_heads_bwd: bad leading dimension of dP1");
  const bool q8 = dp1_fp8 || amax_part, h16 = dwh_unscale != nullptr;
  RV_REQUIRE(!h16 || Hp % 32 == 0, RV_ERR_SHAPE, "rv_heads_bwd: fp16 slabs need a padded hidden width that is a multiple of 32");
  auto kern = q8 ? (h16 ? k_heads_bwd<true, true> : k_heads_bwd<true, false>) : (h16 ? k_heads_bwd<false, true> : k_heads_bwd<false, false>);
  static std::atomic<unsigned long long> attr_done[4];
  const int which = 2 * q8 + h16;
  lds_opt_in((const void*)kern, HB_LDS, attr_done[which]);
  hipLaunchKernelGGL(kern, dim3((unsigned)((Bp / HB_RG) * (Hp / 64))), dim3(512), HB_LDS, (hipStream_t)stream,
                     (const bf16_t*)dmulv_bf16, (const bf16_t*)wh_bf16, ldw, (const bf16_t*)h1_bf16, ldh, (bf16_t*)dp1_bf16, ldp,
                     db1_partial, dwh_slabs, lddw, Hp, rv_store_wt, (unsigned char*)dp1_fp8, ldq, q_scale, amax_part, dwh_unscale);
  RV_CHECK_LAUNCH();
  return RV_OK;
}

}  // extern "C"
