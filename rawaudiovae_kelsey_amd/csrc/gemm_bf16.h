// bf16 MFMA GEMM family for the VAE Linear stacks (gfx950).
//
//   C[M,N] = sum_k A(m,k) * B(k,n)      fp32 accumulate, v_mfma_f32_16x16x32_bf16
//
// One kernel template serves the three contractions of a Linear layer
// (reference: nn.Linear at rawvae/model.py:13-17, autograd at train.py:191):
//   forward  Y  = X  W^T      A = X  [M,K] K-major      B = W  [N,K] K-major
//   dgrad    dX = dY W        A = dY [M,K] K-major      B = W  [K,N] MN-major
//   wgrad    dW = dY^T X      A = dY [K,M] MN-major     B = X  [K,N] MN-major
// so no transposed copy of any weight or activation is ever materialised.
//
// Data movement (per 64-deep K tile, double buffered):
//   HBM/L2 -> LDS by 16-byte global_load_lds (LDS image is lane-linear, the
//   bank-conflict swizzle is applied to the per-lane SOURCE address and undone
//   on the read address);
//   K-major operand : [rows][64] image, 128-B rows, fragments by ds_read_b128,
//                     16-B chunk c of row r stored at chunk c ^ ((r>>1)&7);
//   MN-major operand: [64 k][rows] image, fragments by two ds_read_b64_tr_b16
//                     (hardware transpose), 32-B chunk c of k-row k stored at
//                     c ^ swz(k).
// All extents are multiples of the tile (the host pads), so the
// main loop carries no bounds checks.  Epilogues fuse bias/ReLU/tanh, the MSE
// partial sums + d(pre-tanh) emission, ReLU-mask application and the bias-grad
// column sums.
#pragma once
#include "common.h"

namespace rv {

enum : int {
  EPI_BIAS_ACT_BF16 = 0,  // out_bf16 = act(acc + bias)            (fc1, fc3)
  EPI_F32 = 1,            // out_f32[split] = acc (+bias)          (heads, dz, wgrads)
  EPI_TANH_LOSS = 2,      // recon = tanh(acc+bias); mse; dP4      (fc4)
  EPI_MASK_BF16 = 3,      // out_bf16 = mask>0 ? acc : 0; colsum   (dgrad + ReLU')
};

struct GemmArgs {
  const bf16_t* A;
  const bf16_t* B;
  long lda, ldb;
  int k_tiles;           // 64-deep K tiles per split (grid.z = splits)
  int M_valid, N_valid;  // unpadded extents (row/col masks in epilogues)
  int relu;              // EPI_BIAS_ACT_BF16: apply ReLU
  float* out_f32;
  long ld_f32, split_stride_f32;
  bf16_t* out_bf16;
  long ld_bf16;
  const float* bias;  // [N] (padded), may be null
  const bf16_t* mask;
  long ld_mask;
  const float* x;  // EPI_TANH_LOSS target frames, exact [M_valid, N_valid]
  long ld_x;
  float* recon;  // optional exact-shape fp32 reconstruction
  long ld_recon;
  float* colsum;    // [grid.y][grid.x*BN] per-row-tile column sums of the bf16 output
  float* blocksum;  // [grid.y*grid.x] per-block sum of (recon-x)^2
  float scale;      // 2/(B*S)
};

template <int ROWS>
__device__ __forceinline__ int swz_mn(int k) {
  if constexpr (ROWS == 128)
    return (k & 3) | (((k >> 3) & 1) << 2);  // 8 x 32-B chunks per 256-B k-row
  else
    return ((k >> 1) & 1) | (((k >> 3) & 1) << 1);  // 4 x 32-B chunks per 128-B k-row
}

// Stage one ROWS x 64 operand tile into LDS.  `g` is the tile origin.
template <int ROWS, bool KMAJ>
__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ g, long ld, lds_char* lds,
                                           int wave, int lane) {
  constexpr int NINSTR = ROWS * 128 / 1024;  // 1-KiB wave-instructions per tile
#pragma unroll
  for (int i = 0; i < NINSTR / 4; ++i) {
    const int t = wave + 4 * i;
    const bf16_t* src;
    if constexpr (KMAJ) {
      const int r = 8 * t + (lane >> 3);
      const int c = (lane & 7) ^ ((r >> 1) & 7);
      src = g + (long)r * ld + c * 8;
    } else {
      constexpr int LPR = ROWS * 2 / 16;  // lanes per k-row
      const int kr = t * (64 / LPR) + lane / LPR;
      const int p16 = lane % LPR;
      const int c32 = (p16 >> 1) ^ swz_mn<ROWS>(kr);
      src = g + (long)kr * ld + (c32 * 2 + (p16 & 1)) * 8;
    }
    __builtin_amdgcn_global_load_lds((glb_cptr)src,
                                     (__attribute__((address_space(3))) void*)(lds + t * 1024), 16,
                                     0, 0);
  }
}

// One 16(rows) x 32(k) MFMA operand fragment: lane l holds rows row0+(l&15),
// k = 32*kk + 8*(l>>4) + j, j = 0..7.
template <int ROWS, bool KMAJ>
__device__ __forceinline__ bf16x8 load_frag(const lds_char* lds, int row0, int kk, int lane) {
  if constexpr (KMAJ) {
    const int r = row0 + (lane & 15);
    const int c = (kk * 4 + (lane >> 4)) ^ ((r >> 1) & 7);
    return *(const __attribute__((address_space(3))) bf16x8*)(lds + r * 128 + c * 16);
  } else {
    constexpr int RB = ROWS * 2;
    const int k = kk * 32 + 8 * (lane >> 4) + ((lane & 15) >> 2);
    const int c32 = (row0 >> 4) ^ swz_mn<ROWS>(k);
    const lds_char* a = lds + k * RB + c32 * 32 + (lane & 3) * 8;
    const s16x4 lo =
        __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(a + 4 * RB));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  }
}

template <int BM, int BN, bool A_KMAJ, bool B_KMAJ, int EPI>
__global__ void __launch_bounds__(256) gemm_bf16_kernel(const GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem_generic[];
  lds_char* smem = (lds_char*)smem_generic;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int WTM = BM / 2, WTN = BN / 2, MI = WTM / 16, NI = WTN / 16;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int tile_n = blockIdx.x, tile_m = blockIdx.y, split = blockIdx.z;
  const long m0 = (long)tile_m * BM, n0 = (long)tile_n * BN;
  const long k0 = (long)split * p.k_tiles * 64;

  const bf16_t* Ag = A_KMAJ ? p.A + m0 * p.lda + k0 : p.A + k0 * p.lda + m0;
  const bf16_t* Bg = B_KMAJ ? p.B + n0 * p.ldb + k0 : p.B + k0 * p.ldb + n0;
  const long a_step = A_KMAJ ? 64 : 64 * p.lda;
  const long b_step = B_KMAJ ? 64 : 64 * p.ldb;

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

  stage_tile<BM, A_KMAJ>(Ag, p.lda, smem, wave, lane);
  stage_tile<BN, B_KMAJ>(Bg, p.ldb, smem + A_BYTES, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int nk = p.k_tiles;
  for (int kt = 0; kt < nk; ++kt) {
    lds_char* cur = smem + (kt & 1) * STAGE;
    if (kt + 1 < nk) {
      lds_char* nxt = smem + ((kt + 1) & 1) * STAGE;
      stage_tile<BM, A_KMAJ>(Ag + (kt + 1) * a_step, p.lda, nxt, wave, lane);
      stage_tile<BN, B_KMAJ>(Bg + (kt + 1) * b_step, p.ldb, nxt + A_BYTES, wave, lane);
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 af[MI], bfr[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
        af[mi] = load_frag<BM, A_KMAJ>(cur, wm * WTM + mi * 16, kk, lane);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        bfr[ni] = load_frag<BN, B_KMAJ>(cur + A_BYTES, wn * WTN + ni * 16, kk, lane);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][ni] =
              __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mi], bfr[ni], acc[mi][ni], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ------------------------------ epilogue ------------------------------
  // acc[mi][ni][j] = C[row][col], row = m0 + wm*WTM + mi*16 + (lane>>4)*4 + j,
  //                              col = n0 + wn*WTN + ni*16 + (lane&15).
  const long row_base = m0 + wm * WTM + (lane >> 4) * 4;
  const long col_base = n0 + wn * WTN + (lane & 15);
  float cs[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) cs[ni] = 0.f;
  float sq = 0.f;

#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const long col = col_base + ni * 16;
      float b = 0.f;
      if constexpr (EPI == EPI_F32) {
        if (p.bias && split == 0) b = p.bias[col];  // slab 0 carries the bias
      } else if constexpr (EPI != EPI_MASK_BF16) {
        if (p.bias) b = p.bias[col];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const long row = row_base + mi * 16 + j;
        float v = acc[mi][ni][j] + b;
        if constexpr (EPI == EPI_BIAS_ACT_BF16) {
          if (p.relu) v = fmaxf(v, 0.f);
          p.out_bf16[row * p.ld_bf16 + col] = (bf16_t)v;
        } else if constexpr (EPI == EPI_F32) {
          p.out_f32[split * p.split_stride_f32 + row * p.ld_f32 + col] = v;
        } else if constexpr (EPI == EPI_TANH_LOSS) {
          const float r = fast_tanh(v);
          const bool valid = row < p.M_valid && col < p.N_valid;
          if (p.recon && valid) p.recon[row * p.ld_recon + col] = r;
          if (p.x) {
            float d = 0.f;
            if (valid) d = r - p.x[row * p.ld_x + col];
            sq += d * d;
            const float g = p.scale * d * (1.f - r * r);
            cs[ni] += g;
            p.out_bf16[row * p.ld_bf16 + col] = (bf16_t)g;
          }
        } else {  // EPI_MASK_BF16
          const float mk = (float)p.mask[row * p.ld_mask + col];
          v = mk > 0.f ? v : 0.f;
          cs[ni] += v;
          p.out_bf16[row * p.ld_bf16 + col] = (bf16_t)v;
        }
      }
    }
  }

  if constexpr (EPI == EPI_TANH_LOSS || EPI == EPI_MASK_BF16) {
    float* red = (float*)smem_generic;  // LDS is free: the loop ended on a barrier
    if (p.colsum) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        float s = cs[ni];
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        if (lane < 16) red[wm * BN + wn * WTN + ni * 16 + lane] = s;
      }
      __syncthreads();
      if (tid < BN)
        p.colsum[(long)tile_m * gridDim.x * BN + n0 + tid] = red[tid] + red[BN + tid];
      __syncthreads();
    }
    if constexpr (EPI == EPI_TANH_LOSS) {
      if (p.blocksum) {
        const float s = block_sum_256(sq, red);
        if (tid == 0) p.blocksum[tile_m * gridDim.x + tile_n] = s;
      }
    }
  }
}

}  // namespace rv
