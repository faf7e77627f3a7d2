// Fused multi-tensor Adam / gradient finaliser: device code shared by the stand-alone kernel
// (elementwise.hip: rv_adam_multi, rv_grad_finalize) and by the weight-gradient GEMM launch that
// carries optimizer blocks beside its GEMM blocks (gemm_launch.hip: rv_linear_wgrad_adam).
// torch.optim.Adam defaults as constructed at train.py:163 and stepped at train.py:193.
#pragma once
#include "common.h"
#include "../../include/rawvae_hip.h"

namespace rv {

constexpr int MAX_DESC = 16;
struct DescTable {
  rv_param_desc d[MAX_DESC];
  long blk_start[MAX_DESC + 1];  // first block of each tensor
  int n;
};


// 4 consecutive elements of one slab: fp32, or fp16 slabs holding value * 2^e with one exponent per 32 x 32
// granule and slab (grad_half; the table grad_unscale[s * us_split_stride + (r / 32) * us_ld + c / 32] holds 2^-e,
// which undoes it exactly -- written by the weight-gradient GEMM, gemm_bf16.h GemmArgs::out_f16).  `off` is in
// elements either way; `us` is the factor of the slab the elements belong to (1 for fp32 slabs).
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 load_slab4(const rv_param_desc& d, long off, float us) {
  if (d.grad_half) {
    const f16x4 h = *reinterpret_cast<const f16x4*>(reinterpret_cast<const _Float16*>(d.grad_slabs) + off);
    return make_float4((float)h[0] * us, (float)h[1] * us, (float)h[2] * us, (float)h[3] * us);
  }
  return *reinterpret_cast<const float4*>(d.grad_slabs + off);
}
__device__ __forceinline__ float load_slab1(const rv_param_desc& d, long off, float us) {
  if (d.grad_half) return (float)reinterpret_cast<const _Float16*>(d.grad_slabs)[off] * us;
  return d.grad_slabs[off];
}
// Pointer to slab 0's factor for the granule of element (r, c); slab s is `us_split_stride` entries further.
__device__ __forceinline__ const float* slab_unscale_ptr(const rv_param_desc& d, long r, long c) {
  return d.grad_half ? d.grad_unscale + (r >> 5) * d.us_ld + (c >> 5) : nullptr;
}
__device__ __forceinline__ float slab_unscale(const rv_param_desc& d, const float* up, int s) {
  return d.grad_half ? up[(long)s * d.us_split_stride] : 1.f;
}

// Sum of the gradient slabs for 4 consecutive elements of one row (they share a 32-column granule: c % 4 == 0).
template <bool VEC>
__device__ __forceinline__ float4 slab_sum4(const rv_param_desc& d, long r, long c, int nvalid) {
  const long base = r * d.grad_ld + c;
  const float* up = slab_unscale_ptr(d, r, c);
  float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
  int s = 0;
  if constexpr (VEC) {
    for (; s + 4 <= d.grad_splits; s += 4) {  // 4 independent vector loads in flight
      const float4 a = load_slab4(d, base + (long)(s + 0) * d.grad_split_stride, slab_unscale(d, up, s + 0));
      const float4 b = load_slab4(d, base + (long)(s + 1) * d.grad_split_stride, slab_unscale(d, up, s + 1));
      const float4 e = load_slab4(d, base + (long)(s + 2) * d.grad_split_stride, slab_unscale(d, up, s + 2));
      const float4 f = load_slab4(d, base + (long)(s + 3) * d.grad_split_stride, slab_unscale(d, up, s + 3));
      g.x += (a.x + b.x) + (e.x + f.x); g.y += (a.y + b.y) + (e.y + f.y);
      g.z += (a.z + b.z) + (e.z + f.z); g.w += (a.w + b.w) + (e.w + f.w);
    }
    for (; s < d.grad_splits; ++s) {
      const float4 a = load_slab4(d, base + (long)s * d.grad_split_stride, slab_unscale(d, up, s));
      g.x += a.x; g.y += a.y; g.z += a.z; g.w += a.w;
    }
  } else {
    float t[4] = {0.f, 0.f, 0.f, 0.f};
    for (; s + 4 <= d.grad_splits; s += 4) {
      const float u0 = slab_unscale(d, up, s), u1 = slab_unscale(d, up, s + 1), u2 = slab_unscale(d, up, s + 2),
                  u3 = slab_unscale(d, up, s + 3);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j < nvalid) {
          const long q = base + j + (long)s * d.grad_split_stride;
          t[j] += (load_slab1(d, q, u0) + load_slab1(d, q + d.grad_split_stride, u1)) +
                  (load_slab1(d, q + 2 * d.grad_split_stride, u2) + load_slab1(d, q + 3 * d.grad_split_stride, u3));
        }
    }
    for (; s < d.grad_splits; ++s) {
      const float u0 = slab_unscale(d, up, s);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j < nvalid) t[j] += load_slab1(d, base + j + (long)s * d.grad_split_stride, u0);
    }
    g = make_float4(t[0], t[1], t[2], t[3]);
  }
  return g;
}

// The Adam update of one element (torch.optim.Adam, single-tensor form; train.py:163,193).  Every kernel that
// updates parameters goes through this one function with floating-point contraction OFF, so that the stand-alone
// kernel, the optimizer blocks of the weight-gradient launch and the sharded flat kernel (vector and scalar paths)
// round identically whatever the surrounding code looks like -- replicas and schedules stay bit-equal.
// `inv_bc2s` = 1 / sqrt(1 - beta2^t).  The square root and the reciprocal are the hardware's 1-ulp instructions
// (v_sqrt_f32, v_rcp_f32), not the correctly rounded library sequences (~10 instructions each): the update differs
// from torch's by ~3e-7 relative, far inside the 1e-5 the Adam tests allow, and an optimizer block that shares a CU
// with nothing else (rv_linear_wgrad_adam) is bound by the length of this dependency chain.
__device__ __forceinline__ void adam_update(float& m, float& v, float& w, const float g, const float step_size,
                                            const float inv_bc2s) {
#pragma clang fp contract(off)
  m = 0.9f * m + 0.1f * g;
  v = 0.999f * v + (0.001f * g) * g;
  const float denom = __builtin_amdgcn_sqrtf(v) * inv_bc2s + 1e-8f;
  w = w - step_size * (m * __builtin_amdgcn_rcpf(denom));
}

// Bias corrections of step t = *step_counter: (lr / (1 - beta1^t), 1 / sqrt(1 - beta2^t)).  1 - beta^t = -expm1(t ln beta):
// the direct form 1 - exp2(t log2 beta) cancels catastrophically for small t (1 - 0.999 carries 6e-8 / 1e-3 = 6e-5 of
// relative error in fp32, which went straight into the first updates: torch computes these in double precision).
__device__ __forceinline__ void adam_step_consts(const long long* step_counter, float lr, float* step_size, float* inv_bc2s) {
  const float tt = (float)(*step_counter);
  const float bc1 = -expm1f(tt * -0.10536051565782628f);               // ln(0.9)
  const float bc2 = -expm1f(tt * -0.0010005003335835344f);             // ln(0.999)
  *step_size = lr / bc1;
  *inv_bc2s = 1.0f / sqrtf(bc2);
}

// fp8(w * scale) for up to 4 consecutive elements of one row of the padded fp8 shadow.
__device__ __forceinline__ void store_fp8x4(const rv_param_desc& d, long r, long c, const float (&wv)[4], int nvalid) {
  const float sc = *d.fp8_scale;
  float q[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) q[j] = fminf(fmaxf(wv[j] * sc, -448.f), 448.f);
  unsigned w = 0;
  w = __builtin_amdgcn_cvt_pk_fp8_f32(q[0], q[1], w, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(q[2], q[3], w, true);
  unsigned char* sp = reinterpret_cast<unsigned char*>(d.shadow_fp8) + r * d.shadow_ld + c;
  if (nvalid == 4 && (d.shadow_ld & 3) == 0) *reinterpret_cast<unsigned*>(sp) = w;
  else
    for (int j = 0; j < nvalid; ++j) sp[j] = (unsigned char)(w >> (8 * j));
}

__host__ __device__ inline bool adam_coop(const rv_param_desc& d) { return d.rows == 1 && d.grad_splits >= 16; }

// fp16 slabs: a thread takes EIGHT consecutive elements, so that its slab loads are 16 bytes like everything else it
// touches (4 elements of fp16 are 8 bytes, and 8-byte accesses run at 0.54-0.70 of the 16-byte rate:
// MI355X_MICROARCH.md, "Workgroup dispatch ..." table).  Needs rows of whole 8-element groups, 16-byte aligned.
__host__ __device__ inline bool adam_wide(const rv_param_desc& d) {
  return d.grad_half && d.rows > 1 && !d.shadow_f32 && d.grad_splits <= 8 &&
         ((d.cols | d.grad_ld | d.grad_split_stride | d.offset | d.shadow_ld) & 7) == 0 &&
         ((reinterpret_cast<uintptr_t>(d.grad_slabs) & 15) == 0);
}
__host__ __device__ inline long adam_groups(const rv_param_desc& d) {   // thread-sized groups of one tensor
  return adam_wide(d) ? d.rows * (d.cols / 8) : d.rows * ((d.cols + 3) / 4);
}

// The 8-elements-per-thread form of adam_block for fp16-slab tensors (adam_wide): same arithmetic per element (slab
// sum in slab_sum4's order, adam_update), 16-byte accesses throughout.
template <bool UPDATE>
__device__ __forceinline__ void adam_block_wide(const rv_param_desc& d, const long blk, const int tid,
                                                float* __restrict__ param, float* __restrict__ m_arena,
                                                float* __restrict__ v_arena, float* __restrict__ grad_out, float lr,
                                                float grad_scale, const long long* __restrict__ step_counter,
                                                bf16_t* __restrict__ grad_out_bf16) {
  typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
  const long gpr = d.cols / 8;
  const long grp = blk * 256 + tid;
  if (grp >= gpr * d.rows) return;
  const unsigned r32 = (unsigned)grp / (unsigned)gpr;
  const long r = r32, c = (long)((unsigned)grp - r32 * (unsigned)gpr) * 8;
  const long o = d.offset + r * d.cols + c;
  float4 m4[2], v4[2], w4[2];
  if constexpr (UPDATE) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      m4[h] = *reinterpret_cast<const float4*>(m_arena + o + 4 * h);
      v4[h] = *reinterpret_cast<const float4*>(v_arena + o + 4 * h);
      w4[h] = *reinterpret_cast<const float4*>(param + o + 4 * h);
    }
  }
  const long base = r * d.grad_ld + c;
  const float* up = slab_unscale_ptr(d, r, c);   // 8 consecutive elements share a 32-column granule
  const _Float16* sl = reinterpret_cast<const _Float16*>(d.grad_slabs);
  float gv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int s = 0;
  for (; s + 4 <= d.grad_splits; s += 4) {   // (a + b) + (e + f), as slab_sum4
    f16x8 q[4];
    float us[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      q[u] = *reinterpret_cast<const f16x8*>(sl + base + (long)(s + u) * d.grad_split_stride);
      us[u] = slab_unscale(d, up, s + u);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e)
      gv[e] += ((float)q[0][e] * us[0] + (float)q[1][e] * us[1]) + ((float)q[2][e] * us[2] + (float)q[3][e] * us[3]);
  }
  for (; s < d.grad_splits; ++s) {
    const f16x8 q = *reinterpret_cast<const f16x8*>(sl + base + (long)s * d.grad_split_stride);
    const float us = slab_unscale(d, up, s);
#pragma unroll
    for (int e = 0; e < 8; ++e) gv[e] += (float)q[e] * us;
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) gv[e] *= grad_scale;
  if (grad_out) {
    *reinterpret_cast<float4*>(grad_out + o) = make_float4(gv[0], gv[1], gv[2], gv[3]);
    *reinterpret_cast<float4*>(grad_out + o + 4) = make_float4(gv[4], gv[5], gv[6], gv[7]);
  }
  if (grad_out_bf16) {
    const bf16x8 b8 = {(bf16_t)gv[0], (bf16_t)gv[1], (bf16_t)gv[2], (bf16_t)gv[3], (bf16_t)gv[4], (bf16_t)gv[5], (bf16_t)gv[6], (bf16_t)gv[7]};
    *reinterpret_cast<bf16x8*>(grad_out_bf16 + o) = b8;
  }
  if constexpr (UPDATE) {
    float step_size, inv_bc2s;
    adam_step_consts(step_counter, lr, &step_size, &inv_bc2s);
    float mv[8] = {m4[0].x, m4[0].y, m4[0].z, m4[0].w, m4[1].x, m4[1].y, m4[1].z, m4[1].w};
    float vv[8] = {v4[0].x, v4[0].y, v4[0].z, v4[0].w, v4[1].x, v4[1].y, v4[1].z, v4[1].w};
    float wv[8] = {w4[0].x, w4[0].y, w4[0].z, w4[0].w, w4[1].x, w4[1].y, w4[1].z, w4[1].w};
#pragma unroll
    for (int e = 0; e < 8; ++e) adam_update(mv[e], vv[e], wv[e], gv[e], step_size, inv_bc2s);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      *reinterpret_cast<float4*>(m_arena + o + 4 * h) = make_float4(mv[4 * h], mv[4 * h + 1], mv[4 * h + 2], mv[4 * h + 3]);
      *reinterpret_cast<float4*>(v_arena + o + 4 * h) = make_float4(vv[4 * h], vv[4 * h + 1], vv[4 * h + 2], vv[4 * h + 3]);
      *reinterpret_cast<float4*>(param + o + 4 * h) = make_float4(wv[4 * h], wv[4 * h + 1], wv[4 * h + 2], wv[4 * h + 3]);
    }
    if (d.shadow_bf16) {
      const bf16x8 b8 = {(bf16_t)wv[0], (bf16_t)wv[1], (bf16_t)wv[2], (bf16_t)wv[3], (bf16_t)wv[4], (bf16_t)wv[5], (bf16_t)wv[6], (bf16_t)wv[7]};
      *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(d.shadow_bf16) + r * d.shadow_ld + c) = b8;
    }
    if (d.shadow_fp8) {
      const float lo[4] = {wv[0], wv[1], wv[2], wv[3]}, hi[4] = {wv[4], wv[5], wv[6], wv[7]};
      store_fp8x4(d, r, c, lo, 4);
      store_fp8x4(d, r, c + 4, hi, 4);
    }
  }
}

// One virtual block of 256 threads (`vblock` of tab.blk_start[tab.n], thread `tid` of it).  Each thread owns
// 4 consecutive elements of one row (rows are processed in 4-element groups, so a group never straddles a
// row).  The caller maps real blocks to virtual ones: 1:1 in k_adam, a strided loop in the GEMM launch.
template <bool UPDATE>
__device__ __forceinline__ void adam_block(const DescTable& tab, const long vblock, const int tid,
                                           float* __restrict__ param, float* __restrict__ m_arena,
                                           float* __restrict__ v_arena, float* __restrict__ grad_out, float lr,
                                           float grad_scale, const long long* __restrict__ step_counter,
                                           bf16_t* __restrict__ grad_out_bf16,
                                           const bf16_t* __restrict__ grad_in_bf16) {
  int t = 0;
  while (t + 1 < tab.n && vblock >= tab.blk_start[t + 1]) ++t;
  t = __builtin_amdgcn_readfirstlane(t);   // vblock is wave-uniform in every caller
  const rv_param_desc d = tab.d[t];
  if (adam_wide(d) && !grad_in_bf16) {   // wave-uniform (the descriptor is); a bf16 flat gradient comes with fp32-slab descriptors
    adam_block_wide<UPDATE>(d, vblock - tab.blk_start[t], tid, param, m_arena, v_arena, grad_out, lr, grad_scale,
                            step_counter, grad_out_bf16);
    return;
  }
  const long gpr = (d.cols + 3) / 4;  // 4-element groups per row
  const bool coop = adam_coop(d);     // bias rows with many partials: one WAVE per group
  const long blk = vblock - tab.blk_start[t];
  const long grp = coop ? blk * 4 + (tid >> 6) : blk * 256 + tid;
  if (grp >= gpr * d.rows) return;
  // 32-bit division (adam_build_table checks that a tensor has fewer than 2^31 groups): the 64-bit one is ~4x the code
  const unsigned r32 = (unsigned)grp / (unsigned)gpr;
  const long r = r32, c = (long)((unsigned)grp - r32 * (unsigned)gpr) * 4;
  const int nvalid = (int)(d.cols - c < 4 ? d.cols - c : 4);
  const long o = d.offset + r * d.cols + c;
  const bool vec = nvalid == 4 && ((d.cols | d.grad_ld | d.grad_split_stride | d.offset) & 3) == 0 &&
                   ((reinterpret_cast<uintptr_t>(d.grad_slabs) & 15) == 0);   // (fp16 slabs need 8: implied)
  // issue the optimizer-state loads first so they are in flight under the slab sums
  float mv[4] = {0.f, 0.f, 0.f, 0.f}, vv[4] = {0.f, 0.f, 0.f, 0.f}, wv[4] = {0.f, 0.f, 0.f, 0.f};
  if constexpr (UPDATE) {
    if (!coop || (tid & 63) == 0) {
      if (vec) {
        const float4 m4 = *reinterpret_cast<const float4*>(m_arena + o);
        const float4 v4 = *reinterpret_cast<const float4*>(v_arena + o);
        const float4 w4 = *reinterpret_cast<const float4*>(param + o);
        mv[0] = m4.x; mv[1] = m4.y; mv[2] = m4.z; mv[3] = m4.w;
        vv[0] = v4.x; vv[1] = v4.y; vv[2] = v4.z; vv[3] = v4.w;
        wv[0] = w4.x; wv[1] = w4.y; wv[2] = w4.z; wv[3] = w4.w;
      } else {
        for (int j = 0; j < nvalid; ++j) {
          mv[j] = m_arena[o + j];
          vv[j] = v_arena[o + j];
          wv[j] = param[o + j];
        }
      }
    }
  }
  float4 g;
  if (grad_in_bf16) {
    // gradient = flat bf16 arena (the data-parallel payload after its all-reduce), same element offsets
    if (coop && (tid & 63) != 0) return;
    float t4[4] = {0.f, 0.f, 0.f, 0.f};
    if (vec && (reinterpret_cast<uintptr_t>(grad_in_bf16) & 7) == 0) {
      const bf16x4 b4 = *reinterpret_cast<const bf16x4*>(grad_in_bf16 + o);
      t4[0] = (float)b4[0]; t4[1] = (float)b4[1]; t4[2] = (float)b4[2]; t4[3] = (float)b4[3];
    } else {
      for (int j = 0; j < nvalid; ++j) t4[j] = (float)grad_in_bf16[o + j];
    }
    g = make_float4(t4[0], t4[1], t4[2], t4[3]);
  } else if (coop) {
    // lanes stride over the partial slabs, then a fixed-order butterfly: deterministic
    const int lane = tid & 63;
    float tsum[4] = {0.f, 0.f, 0.f, 0.f};
    const float* up = slab_unscale_ptr(d, r, c);
    for (int s = lane; s < d.grad_splits; s += 64) {
      const long q = (long)s * d.grad_split_stride + r * d.grad_ld + c;
      const float us = slab_unscale(d, up, s);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j < nvalid) tsum[j] += load_slab1(d, q + j, us);
    }
    g = make_float4(wave_sum(tsum[0]), wave_sum(tsum[1]), wave_sum(tsum[2]), wave_sum(tsum[3]));
    if (lane != 0) return;
  } else {
    g = vec ? slab_sum4<true>(d, r, c, 4) : slab_sum4<false>(d, r, c, nvalid);
  }
  g.x *= grad_scale; g.y *= grad_scale; g.z *= grad_scale; g.w *= grad_scale;
  float gv[4] = {g.x, g.y, g.z, g.w};
  if (grad_out) {
    if (vec) *reinterpret_cast<float4*>(grad_out + o) = g;
    else
      for (int j = 0; j < nvalid; ++j) grad_out[o + j] = gv[j];
  }
  if (grad_out_bf16) {
    if (vec && (reinterpret_cast<uintptr_t>(grad_out_bf16) & 7) == 0) {
      const bf16x4 b4 = {(bf16_t)gv[0], (bf16_t)gv[1], (bf16_t)gv[2], (bf16_t)gv[3]};
      *reinterpret_cast<bf16x4*>(grad_out_bf16 + o) = b4;
    } else {
      for (int j = 0; j < nvalid; ++j) grad_out_bf16[o + j] = (bf16_t)gv[j];
    }
  }
  if constexpr (UPDATE) {
    float step_size, inv_bc2s;
    adam_step_consts(step_counter, lr, &step_size, &inv_bc2s);
#pragma unroll
    for (int j = 0; j < 4; ++j) adam_update(mv[j], vv[j], wv[j], gv[j], step_size, inv_bc2s);
    if (vec) {
      *reinterpret_cast<float4*>(m_arena + o) = make_float4(mv[0], mv[1], mv[2], mv[3]);
      *reinterpret_cast<float4*>(v_arena + o) = make_float4(vv[0], vv[1], vv[2], vv[3]);
      *reinterpret_cast<float4*>(param + o) = make_float4(wv[0], wv[1], wv[2], wv[3]);
    } else {
      for (int j = 0; j < nvalid; ++j) {
        m_arena[o + j] = mv[j];
        v_arena[o + j] = vv[j];
        param[o + j] = wv[j];
      }
    }
    if (d.shadow_bf16) {
      bf16_t* sp = reinterpret_cast<bf16_t*>(d.shadow_bf16) + r * d.shadow_ld + c;
      if (nvalid == 4 && (d.shadow_ld & 3) == 0) {
        bf16x4 b4 = {(bf16_t)wv[0], (bf16_t)wv[1], (bf16_t)wv[2], (bf16_t)wv[3]};
        *reinterpret_cast<bf16x4*>(sp) = b4;
      } else {
        for (int j = 0; j < nvalid; ++j) sp[j] = (bf16_t)wv[j];
      }
    }
    if (d.shadow_f32)
      for (int j = 0; j < nvalid; ++j) d.shadow_f32[r * d.shadow_ld + c + j] = wv[j];
    if (d.shadow_fp8) store_fp8x4(d, r, c, wv, nvalid);
  }
}

// U virtual blocks per thread with all loads of all of them issued before the first dependent instruction (twice
// the bytes in flight per lane): for callers that walk the table with few resident threads (the optimizer blocks
// of rv_linear_wgrad_adam get one 512-thread block per CU).  Same arithmetic as adam_block; anything off the
// aligned 4-wide path (ragged row ends, bias rows summed by a wave) falls back to it.
struct AdamItem {
  int state;  // 0: nothing to do, 1: aligned 4-wide group, 2: general path
  int t;
  long r, c, o;
};

// `vblock` must be wave-uniform (it is in both callers: a virtual block is 256 consecutive threads); the
// readfirstlane makes that provable, so the descriptor is read with scalar loads instead of a per-lane loop.
__device__ __forceinline__ AdamItem adam_locate(const DescTable& tab, const long vblock, const int tid, const long limit = -1) {
  AdamItem it{0, 0, 0, 0, 0};
  if (vblock >= (limit >= 0 ? limit : tab.blk_start[tab.n])) return it;   // (limit: the caller walks only a prefix of the table)
  int t = 0;
  while (t + 1 < tab.n && vblock >= tab.blk_start[t + 1]) ++t;
  t = __builtin_amdgcn_readfirstlane(t);
  const rv_param_desc d = tab.d[t];
  it.t = t;
  if (adam_coop(d) || adam_wide(d)) { it.state = 2; return it; }   // adam_block's own paths
  const long gpr = (d.cols + 3) / 4;
  const long grp = (vblock - tab.blk_start[t]) * 256 + tid;
  if (grp >= gpr * d.rows) return it;
  const unsigned r32 = (unsigned)grp / (unsigned)gpr;
  it.r = r32;
  it.c = (long)((unsigned)grp - r32 * (unsigned)gpr) * 4;
  it.o = d.offset + it.r * d.cols + it.c;
  const bool vec = d.cols - it.c >= 4 && ((d.cols | d.grad_ld | d.grad_split_stride | d.offset) & 3) == 0 &&
                   ((reinterpret_cast<uintptr_t>(d.grad_slabs) & 15) == 0) && d.grad_splits <= 4 &&
                   (!d.shadow_bf16 || (d.shadow_ld & 3) == 0) && !d.shadow_f32;
  it.state = vec ? 1 : 2;
  return it;
}

template <int U>
__device__ __forceinline__ void adam_group(const DescTable& tab, const long vb0, const long vb_stride, const int tid,
                                           float* __restrict__ param, float* __restrict__ m_arena,
                                           float* __restrict__ v_arena, const float lr, const float grad_scale,
                                           const long long* __restrict__ step_counter, const long limit = -1) {
  AdamItem it[U];
  bool all_vec = true;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    it[u] = adam_locate(tab, vb0 + u * vb_stride, tid, limit);
    all_vec = all_vec && it[u].state != 2;   // state 0 (past the end of the table) is skipped item by item
  }
  if (all_vec) {
    float4 m4[U], v4[U], w4[U], sl[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (it[u].state != 1) continue;
      const rv_param_desc d = tab.d[__builtin_amdgcn_readfirstlane(it[u].t)];
      m4[u] = *reinterpret_cast<const float4*>(m_arena + it[u].o);
      v4[u] = *reinterpret_cast<const float4*>(v_arena + it[u].o);
      w4[u] = *reinterpret_cast<const float4*>(param + it[u].o);
      const long base = it[u].r * d.grad_ld + it[u].c;
      const float* up = slab_unscale_ptr(d, it[u].r, it[u].c);
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_)
        sl[u][s_] = s_ < d.grad_splits ? load_slab4(d, base + (long)s_ * d.grad_split_stride, slab_unscale(d, up, s_))
                                       : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float step_size, inv_bc2s;
    adam_step_consts(step_counter, lr, &step_size, &inv_bc2s);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (it[u].state != 1) continue;
      const rv_param_desc d = tab.d[__builtin_amdgcn_readfirstlane(it[u].t)];
      float gv[4];
      if (d.grad_splits == 4) {  // the summation order of slab_sum4
        gv[0] = (sl[u][0].x + sl[u][1].x) + (sl[u][2].x + sl[u][3].x);
        gv[1] = (sl[u][0].y + sl[u][1].y) + (sl[u][2].y + sl[u][3].y);
        gv[2] = (sl[u][0].z + sl[u][1].z) + (sl[u][2].z + sl[u][3].z);
        gv[3] = (sl[u][0].w + sl[u][1].w) + (sl[u][2].w + sl[u][3].w);
      } else {
        gv[0] = gv[1] = gv[2] = gv[3] = 0.f;
#pragma unroll
        for (int s_ = 0; s_ < 3; ++s_)
          if (s_ < d.grad_splits) { gv[0] += sl[u][s_].x; gv[1] += sl[u][s_].y; gv[2] += sl[u][s_].z; gv[3] += sl[u][s_].w; }
      }
      float mv[4] = {m4[u].x, m4[u].y, m4[u].z, m4[u].w}, vv[4] = {v4[u].x, v4[u].y, v4[u].z, v4[u].w};
      float wv[4] = {w4[u].x, w4[u].y, w4[u].z, w4[u].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        gv[j] *= grad_scale;
        adam_update(mv[j], vv[j], wv[j], gv[j], step_size, inv_bc2s);
      }
      *reinterpret_cast<float4*>(m_arena + it[u].o) = make_float4(mv[0], mv[1], mv[2], mv[3]);
      *reinterpret_cast<float4*>(v_arena + it[u].o) = make_float4(vv[0], vv[1], vv[2], vv[3]);
      *reinterpret_cast<float4*>(param + it[u].o) = make_float4(wv[0], wv[1], wv[2], wv[3]);
      if (d.shadow_bf16) {
        const bf16x4 b4 = {(bf16_t)wv[0], (bf16_t)wv[1], (bf16_t)wv[2], (bf16_t)wv[3]};
        *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(d.shadow_bf16) + it[u].r * d.shadow_ld + it[u].c) = b4;
      }
      if (d.shadow_fp8) store_fp8x4(d, it[u].r, it[u].c, wv, 4);
    }
    return;
  }
#pragma unroll
  for (int u = 0; u < U; ++u)
    if (it[u].state)
      adam_block<true>(tab, vb0 + u * vb_stride, tid, param, m_arena, v_arena, nullptr, lr, grad_scale, step_counter,
                       nullptr, nullptr);
}

// ------------------------------------------------------------------ streamed update (LDS-DMA ring per wave)
// The same update for callers that own a whole CU with few waves (the optimizer blocks of rv_linear_wgrad_adam).
// A loop of plain loads and stores streams ~20-25 GB/s per CU: loads and stores share vmcnt and complete out of
// order with respect to each other, so the compiler drains the queue before every trip.  Here each wave moves its
// operands with global_load_lds (no VGPR destination, so nothing for the compiler to wait on) into a private
// two-slot LDS ring, one slot = 64 four-element groups x 7 streams (4 gradient slabs, exp_avg, exp_avg_sq, param)
// x 16 B = 7 KB, and retires a slot with a COUNTED wait: memory reads return in order among themselves, so once at
// most 7 vector-memory operations are outstanding none of them can be a load older than the 7 loads of the next
// slot -- whatever the stores in between are doing.  One "wave chunk" = a quarter of a 256-thread virtual block;
// every chunk issues exactly 7 loads (lanes or chunks that do not take the aligned path read a valid dummy
// address and go through adam_block), which keeps the count static.  Arithmetic: adam_update, slab order of slab_sum4.
constexpr int AS_SLOT = 7 * 1024;

__device__ __forceinline__ void adam_stream_issue(const DescTable& tab, const AdamItem& it, lds_char* slot,
                                                  const float* param, const float* m_arena, const float* v_arena) {
  const rv_param_desc d = tab.d[__builtin_amdgcn_readfirstlane(it.t)];
  const bool on = it.state == 1 && !d.grad_half;
  const long o = on ? it.o : d.offset;
  const long base = on ? it.r * d.grad_ld + it.c : 0;
#pragma unroll
  for (int s_ = 0; s_ < 4; ++s_) {
    const float* g = d.grad_slabs + base + (long)(s_ < d.grad_splits ? s_ : 0) * d.grad_split_stride;
    __builtin_amdgcn_global_load_lds((glb_cptr)g, (__attribute__((address_space(3))) void*)(slot + s_ * 1024), 16, 0, 0);
  }
  __builtin_amdgcn_global_load_lds((glb_cptr)(m_arena + o), (__attribute__((address_space(3))) void*)(slot + 4096), 16, 0, 0);
  __builtin_amdgcn_global_load_lds((glb_cptr)(v_arena + o), (__attribute__((address_space(3))) void*)(slot + 5120), 16, 0, 0);
  __builtin_amdgcn_global_load_lds((glb_cptr)(param + o), (__attribute__((address_space(3))) void*)(slot + 6144), 16, 0, 0);
}

// `lds`: 2 * AS_SLOT bytes private to the calling wave.  Chunks wc_first, wc_first + wc_stride, ... (wave-uniform).
__device__ __forceinline__ void adam_stream(const DescTable& tab, const long wc_first, const long wc_stride, lds_char* lds,
                                            const int lane, float* param, float* m_arena, float* v_arena, const float lr,
                                            const float grad_scale, const long long* __restrict__ step_counter) {
  const long total = tab.blk_start[tab.n] * 4;
  if (wc_first >= total) return;
  auto locate = [&](const long wc) {
    return wc < total ? adam_locate(tab, wc >> 2, (int)(wc & 3) * 64 + lane) : AdamItem{0, 0, 0, 0, 0};
  };
  AdamItem a = locate(wc_first), b = locate(wc_first + wc_stride);
  adam_stream_issue(tab, a, lds, param, m_arena, v_arena);
  adam_stream_issue(tab, b, lds + AS_SLOT, param, m_arena, v_arena);
  float step_size, inv_bc2s;
  adam_step_consts(step_counter, lr, &step_size, &inv_bc2s);
  int slot = 0;
  for (long wc = wc_first; wc < total; wc += wc_stride) {
    asm volatile("s_waitcnt vmcnt(7)" ::: "memory");   // slot `slot` has landed; the next slot's 7 loads may fly
    const lds_char* sl = lds + slot * AS_SLOT + lane * 16;
    typedef const __attribute__((address_space(3))) f32x4* lds_f4;
    f32x4 sv[4];
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) sv[s_] = *reinterpret_cast<lds_f4>(sl + s_ * 1024);
    const f32x4 m4 = *reinterpret_cast<lds_f4>(sl + 4096);
    const f32x4 v4 = *reinterpret_cast<lds_f4>(sl + 5120);
    const f32x4 w4 = *reinterpret_cast<lds_f4>(sl + 6144);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the reads are done before the DMA below refills the slot
    const AdamItem c = locate(wc + 2 * wc_stride);
    adam_stream_issue(tab, c, lds + slot * AS_SLOT, param, m_arena, v_arena);
    const rv_param_desc d = tab.d[__builtin_amdgcn_readfirstlane(a.t)];
    if (a.state == 1 && !d.grad_half) {
      float gv[4];
      if (d.grad_splits == 4) {  // the summation order of slab_sum4
#pragma unroll
        for (int j = 0; j < 4; ++j) gv[j] = (sv[0][j] + sv[1][j]) + (sv[2][j] + sv[3][j]);
      } else {
        gv[0] = gv[1] = gv[2] = gv[3] = 0.f;
#pragma unroll
        for (int s_ = 0; s_ < 3; ++s_)
          if (s_ < d.grad_splits) {
#pragma unroll
            for (int j = 0; j < 4; ++j) gv[j] += sv[s_][j];
          }
      }
      float mv[4] = {m4[0], m4[1], m4[2], m4[3]}, vv[4] = {v4[0], v4[1], v4[2], v4[3]}, wv[4] = {w4[0], w4[1], w4[2], w4[3]};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        gv[j] *= grad_scale;
        adam_update(mv[j], vv[j], wv[j], gv[j], step_size, inv_bc2s);
      }
      *reinterpret_cast<float4*>(m_arena + a.o) = make_float4(mv[0], mv[1], mv[2], mv[3]);
      *reinterpret_cast<float4*>(v_arena + a.o) = make_float4(vv[0], vv[1], vv[2], vv[3]);
      *reinterpret_cast<float4*>(param + a.o) = make_float4(wv[0], wv[1], wv[2], wv[3]);
      if (d.shadow_bf16) {
        const bf16x4 b4 = {(bf16_t)wv[0], (bf16_t)wv[1], (bf16_t)wv[2], (bf16_t)wv[3]};
        *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(d.shadow_bf16) + a.r * d.shadow_ld + a.c) = b4;
      }
      if (d.shadow_fp8) store_fp8x4(d, a.r, a.c, wv, 4);
    } else if (a.state != 0) {
      adam_block<true>(tab, wc >> 2, (int)(wc & 3) * 64 + lane, param, m_arena, v_arena, nullptr, lr, grad_scale,
                       step_counter, nullptr, nullptr);
    }
    a = b;
    b = c;
    slot ^= 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the last two (dummy) refills land before the LDS is given back
}

// Parameters (unless `param` is null) and every operand shadow of the table's tensors from a flat fp32 source (the
// all-gather's output, or the parameter arena itself): element at flat arena offset o is flat[o - flat_base].
// One virtual block = 256 threads, 4 elements each.  No __restrict__: `flat` may be the parameter arena.
__device__ __forceinline__ void refresh_block(const DescTable& tab, const long vblock, const int tid,
                                              const float* flat, const long flat_base, float* param) {
  int t = 0;
  while (t + 1 < tab.n && vblock >= tab.blk_start[t + 1]) ++t;
  t = __builtin_amdgcn_readfirstlane(t);
  const rv_param_desc d = tab.d[t];
  const bool wide = adam_wide(d);   // the table gives such tensors 8 elements per thread: two 4-element halves here
  const long gpr = wide ? d.cols / 8 : (d.cols + 3) / 4;
  const bool coop = adam_coop(d);   // the table's block layout gives such rows one WAVE per group
  const long blk = vblock - tab.blk_start[t];
  const long grp = coop ? blk * 4 + (tid >> 6) : blk * 256 + tid;
  if (grp >= gpr * d.rows || (coop && (tid & 63) != 0)) return;
  const unsigned r32 = (unsigned)grp / (unsigned)gpr;
  const long r = r32, c0 = (long)((unsigned)grp - r32 * (unsigned)gpr) * (wide ? 8 : 4);
  for (int half = 0; half < (wide ? 2 : 1); ++half) {
  const long c = c0 + 4 * half;
  const int nvalid = (int)(d.cols - c < 4 ? d.cols - c : 4);
  const long o = d.offset + r * d.cols + c;
  float wv[4] = {0.f, 0.f, 0.f, 0.f};
  const bool vec = nvalid == 4 && (((o - flat_base) | o) & 3) == 0 && ((reinterpret_cast<uintptr_t>(flat) & 15) == 0);
  if (vec) {
    const float4 w4 = *reinterpret_cast<const float4*>(flat + (o - flat_base));
    wv[0] = w4.x; wv[1] = w4.y; wv[2] = w4.z; wv[3] = w4.w;
    if (param) *reinterpret_cast<float4*>(param + o) = w4;
  } else {
    for (int j = 0; j < nvalid; ++j) {
      wv[j] = flat[o - flat_base + j];
      if (param) param[o + j] = wv[j];
    }
  }
  if (d.shadow_bf16) {
    bf16_t* sp = reinterpret_cast<bf16_t*>(d.shadow_bf16) + r * d.shadow_ld + c;
    if (nvalid == 4 && (d.shadow_ld & 3) == 0) {
      const bf16x4 b4 = {(bf16_t)wv[0], (bf16_t)wv[1], (bf16_t)wv[2], (bf16_t)wv[3]};
      *reinterpret_cast<bf16x4*>(sp) = b4;
    } else {
      for (int j = 0; j < nvalid; ++j) sp[j] = (bf16_t)wv[j];
    }
  }
  if (d.shadow_f32)
    for (int j = 0; j < nvalid; ++j) d.shadow_f32[r * d.shadow_ld + c + j] = wv[j];
  if (d.shadow_fp8) store_fp8x4(d, r, c, wv, nvalid);
  }
}

template <bool UPDATE>
__global__ void __launch_bounds__(256)
k_adam(const DescTable tab, float* __restrict__ param, float* __restrict__ m_arena,
       float* __restrict__ v_arena, float* __restrict__ grad_out, float lr, float grad_scale,
       const long long* __restrict__ step_counter, bf16_t* __restrict__ grad_out_bf16,
       const bf16_t* __restrict__ grad_in_bf16, const int* __restrict__ poison,
       const float* __restrict__ grad_scale_dev) {
  // `poison` (the data-parallel step: its count of cross-stream flag waits that ran out, plan.hip): a wait in front of
  // this launch gave up, so the gradient it guards is incomplete -- the update is NOT applied, on any tensor, from then
  // on (the count is never cleared on the device; the host raises when it reads it).  One scalar load per block.
  if (poison && *poison != 0) return;
  if (grad_scale_dev) grad_scale *= *grad_scale_dev;   // a scale that lives on the device (an upstream gradient: plan.hip)
  adam_block<UPDATE>(tab, (long)blockIdx.x, (int)threadIdx.x, param, m_arena, v_arena, grad_out, lr, grad_scale,
                     step_counter, grad_out_bf16, grad_in_bf16);
}

// Host side: descriptor table with the first virtual block of every tensor.
inline int adam_build_table(const rv_param_desc* descs, int n, DescTable* tab) {
  RV_REQUIRE(descs && n > 0 && n <= MAX_DESC, RV_ERR_SHAPE, "param desc count %d out of range", n);
  tab->n = n;
  long blk = 0;
  for (int i = 0; i < n; ++i) {
    tab->d[i] = descs[i];
    RV_REQUIRE(descs[i].rows > 0 && descs[i].cols > 0 && descs[i].grad_slabs && descs[i].grad_splits >= 1,
               RV_ERR_SHAPE, "param desc %d invalid", i);
    RV_REQUIRE(descs[i].rows * ((descs[i].cols + 3) / 4) < 0x7fffffffL, RV_ERR_SHAPE, "param desc %d: tensor too large", i);
    tab->blk_start[i] = blk;
    {
      const long groups = adam_groups(descs[i]);
      blk += adam_coop(descs[i]) ? (groups + 3) / 4 : (groups + 255) / 256;
    }
  }
  tab->blk_start[n] = blk;
  return RV_OK;
}

}  // namespace rv
