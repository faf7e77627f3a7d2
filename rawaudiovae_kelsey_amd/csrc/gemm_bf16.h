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
#include <type_traits>

#include "common.h"
#include "philox.h"

namespace rv {

enum : int {
  EPI_BIAS_ACT_BF16 = 0,  // out_bf16 = act(acc + bias)            (fc1, fc3)
  EPI_F32 = 1,            // out_f32[split] = acc (+bias)          (heads, dz, wgrads)
  EPI_TANH_LOSS = 2,      // recon = tanh(acc+bias); mse; dP4      (fc4)
  EPI_MASK_BF16 = 3,      // out_bf16 = mask>0 ? acc : 0; colsum   (dgrad + ReLU')
  // latent widths above 64 (the reference's own latent_dim = 256, default.ini:18): the latent-sized contractions are
  // GEMMs in their own right there, and the elementwise steps between them are epilogues
  EPI_REPARAM = 4,        // heads GEMM: acc = mu | logvar of the same latents; z = mu + eps exp(logvar / 2), KL partial
                          // (model.py:21-26,45)
  EPI_REPARAM_BWD = 5,    // dz GEMM: dmu, dlogvar from dz (autograd of model.py:23-26 + the KL term), their column sums
};

struct GemmArgs {
  const bf16_t* A;
  const bf16_t* B;
  long lda, ldb;
  int k_tiles;           // 64-deep K tiles per split
  int tiles_m, tiles_n;  // output tiles
  int splits;            // K splits (grid.x = tiles_m * tiles_n * splits)
  int M_valid, N_valid;  // unpadded extents (row/col masks in epilogues)
  int relu;              // EPI_BIAS_ACT_BF16: apply ReLU
  float* out_f32;
  long ld_f32, split_stride_f32;
  // EPI_F32 with fp16 split-K slabs (block floating point): same element strides, values stored as
  // fp16(value * 2^e) with ONE exponent per wave tile of one slab, chosen from that tile's own largest magnitude
  // (so that it lands in [2^14, 2^15): any gradient magnitude keeps fp16's 11 significant bits relative to its
  // tile).  The factor that undoes it, 2^-e, goes to f16_unscale[split * us_split_stride + (row / 32) * us_ld +
  // col / 32] for every 32 x 32 granule the wave tile covers -- a fixed granule, so the reader (Adam) does not need
  // to know the GEMM's tile configuration.  Halves the slab bytes that a split-K weight gradient writes and that
  // Adam reads back; the sum over slabs stays fp32.
  void* out_f16;
  float* f16_unscale;
  long us_ld, us_split_stride;
  bf16_t* out_bf16;
  long ld_bf16;
  const float* bias;  // [N] (padded), may be null
  const bf16_t* mask;
  long ld_mask;
  // fp8 (e4m3) operand path, 256 x 256 ping-pong dgrad only: non-zero = `mask` points at the fp8 image of the activation
  // (one byte per element, ld_mask in BYTES): the ReLU mask is read from the image the forward already wrote for the next
  // GEMM, and the bf16 copy of that activation need not exist (a positive e4m3 byte is a positive int8)
  int mask_fp8;
  const float* x;  // EPI_TANH_LOSS target frames, exact [M_valid, N_valid]
  long ld_x;
  // ... or, when x_hop != 0, hop-strided frames of a waveform: `x` is the waveform (x_nsamples samples, zero past
  // them) and row r is x[f*x_hop : f*x_hop + N_valid] with f = x_idx ? x_idx[r] : x_first + r (AudioDataset,
  // rawvae/dataset.py:108-118) -- the target is read where the audio lives, no framed copy of the batch exists
  const long long* x_idx;
  long x_first, x_hop, x_nsamples;
  float* recon;  // optional exact-shape fp32 reconstruction
  long ld_recon;
  float* colsum;    // [grid.y][grid.x*BN] per-row-tile column sums of the bf16 output
  float* blocksum;  // [grid.y*grid.x] per-block sum of (recon-x)^2
  float scale;      // 2/(B*S)
  // fp8 (e4m3) operand path: A and B point at fp8 bytes viewed as bf16 pairs (lda/ldb and the K extent count
  // PAIRS), so one staged "64-deep" tile is 128 fp8 values per row and all tile geometry is unchanged
  const float* dq;       // device scalar: 1 / (scale_A * scale_B), applied to the accumulator; null = 1
  unsigned char* out_fp8;  // EPI_BIAS_ACT_BF16: also store the output as fp8(out * *q_scale) (next layer's operand)
  long ld_fp8;
  const float* q_scale;  // device scalar
  float* amax_part;      // [number of blocks]: max|out| of each block (delayed scaling: reduced by the next step's first kernel), or null
  // A operand gathered from a resident waveform (fc1 forward on the real-data path, rv_plan_step_frames): when
  // a_hop != 0, `A` is the waveform as bf16 (cast once when it was uploaded) and row r of the operand is the frame
  // A[f * a_hop : f * a_hop + K], f = a_idx ? a_idx[r] : a_first + r (AudioDataset.__getitem__, rawvae/dataset.py:
  // 108-118); rows >= a_rows repeat row a_rows - 1 (their outputs are padding).  a_hop is a multiple of 8 and A is
  // 16-byte aligned (16-byte LDS-DMA pieces), the buffer extends K elements past the last frame's start.  No cast /
  // gather kernel runs and no fp32 frame is read; the framed bf16 matrix the fc1 weight gradient needs later is a
  // by-product: the block with tile_n == kt % tiles_n copies K tile kt of its rows from LDS to a_copy (every block
  // writes 32 KB instead of one kernel writing 8.4 MB).
  const long long* a_idx;
  long a_first, a_hop;
  int a_rows;
  bf16_t* a_copy;
  long ld_copy;
  long long* step_inc;   // EPI_BIAS_ACT_BF16: block 0 bumps the device step counter (the step's first kernel does)
  // EPI_REPARAM / EPI_REPARAM_BWD (latent-sized GEMMs with the reparameterisation in the epilogue).  EPI_REPARAM: B is the
  // stacked head weight [2 lat_lp, K] (fc21 rows, then fc22 rows); output tile column block tn covers latents
  // [64 tn, 64 tn + 64) of BOTH heads -- the B tile's 128 rows are gathered as 16 mu rows, 16 logvar rows, 16 mu rows ...
  // so that a wave's two column fragments hold mu and logvar of the same 16 latents and every lane owns both values of
  // its (row, latent) pairs.  `bias` = the stacked head bias [2 lat_lp].
  long lat_lp, lat_l;            // padded / exact latent width
  const float* eps_in;           // [M_valid, lat_l] fp32, or NULL: Philox draws (seed, step_counter), written to eps_out
  float* eps_out;
  unsigned long long seed;
  const long long* step_counter;
  float* mulv;                   // EPI_REPARAM: out [Mp, 2 lat_lp] fp32 (mu | logvar).  EPI_REPARAM_BWD: the same tensor, read
  bf16_t* z;                     // EPI_REPARAM: out [Mp, lat_lp] bf16
  float* kl_partial;             // EPI_REPARAM: 4 slots per block (sum, 0, 0, 0): one per 1024 elements of the [Mp, lat_lp] grid
  const float* eps;              // EPI_REPARAM_BWD: the eps the forward used [M_valid, lat_l]
  float kl_beta, inv_nk;         // EPI_REPARAM_BWD: kl_beta, 1 / (B L)
  const float* dmu_ext;          // EPI_REPARAM_BWD: gradients arriving from outside [M_valid, lat_l], or NULL
  const float* dlv_ext;
  bf16_t* dmulv;                 // EPI_REPARAM_BWD: out [Mp, 2 lat_lp] bf16 (dmu | dlogvar)
  float* dbh_partial;            // EPI_REPARAM_BWD: [Mp / 16][2 lat_lp] column sums: row 4 tile_m holds the tile's, rows + 1..3 zeros
  int wt;                // non-zero: the epilogue's outputs are written through (common.h store_wt16; the launchers copy rv_store_wt)
};

template <int ROWS>
__device__ __forceinline__ int swz_mn(int k) {
  if constexpr (ROWS >= 128)
    return (k & 3) | (((k >> 3) & 1) << 2);  // >= 8 x 32-B chunks per k-row (256/512-B rows)
  else
    return ((k >> 1) & 1) | (((k >> 3) & 1) << 1);  // 4 x 32-B chunks per 128-B k-row
}

// MN-major fp8 (e4m3) operand of the 256 x 256 ping-pong kernels (the fp8 fc4 backward): a half tile is 128 k-rows of
// 128 bytes (128 rows of the operand, one byte each); 16-byte chunk c of k-row k is stored at chunk c ^ swz_mn8(k).
// ds_read_b64_tr_b8 hands each group of 16 lanes an 8 (k) x 16 (byte) block -- lane i addresses row i >> 1, byte
// 8 (i & 1) of it; lane j receives column j, eight consecutive k -- so one 32-lane half touches 16 k-rows of one chunk:
// rows alternate between the two halves of the 64 banks by themselves (128-byte pitch), the swizzle spreads the four
// rows of equal parity of each 16-lane group and the two groups (k apart by 16) over the eight chunks: no conflicts.
__device__ __forceinline__ int swz_mn8(int k) { return ((k >> 1) & 3) | (((k >> 4) & 1) << 2); }

// Per-lane element offsets (from the tile origin) of the 16-byte pieces this lane stages for one
// ROWS x 64 operand tile: computed once, so the K loop only adds a wave-uniform tile base
// (scalar) to a 32-bit per-lane offset -- global_load_lds then uses its saddr+voffset form and
// the loop carries one VGPR per piece instead of a 64-bit pointer.
template <int ROWS, bool KMAJ, int NW, bool F8MN = false>
struct StageOffsets {
  static constexpr int NINSTR = ROWS * 128 / 1024;  // 1-KiB wave-instructions per tile
  static constexpr int PER_WAVE = NINSTR / NW;
  static_assert(NINSTR % NW == 0, "tile must split evenly over the waves");
  static_assert(!F8MN || (!KMAJ && ROWS == 128), "MN-major fp8: half tiles of 128 k-rows x 128 bytes");
  unsigned off[PER_WAVE];

  __device__ __forceinline__ void init(long ld, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
      const int t = wave + NW * i;
      if constexpr (F8MN) {
        // k-row pitch `ld` in 2-byte units (the operand pointers are typed bf16); 8 k-rows of 8 chunks per instruction
        const int k = 8 * t + (lane >> 3);
        const int c = (lane & 7) ^ swz_mn8(k);
        off[i] = (unsigned)(k * (int)ld + c * 8);
      } else if constexpr (KMAJ) {
        const int r = 8 * t + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        off[i] = (unsigned)(r * (int)ld + c * 8);
      } else {
        constexpr int LPR = ROWS * 2 / 16;  // lanes per k-row
        const int kr = LPR >= 64 ? t / (LPR / 64) : t * (64 / LPR) + lane / LPR;
        const int p16 = LPR >= 64 ? (t % (LPR / 64)) * 64 + lane : lane % LPR;
        const int c32 = (p16 >> 1) ^ swz_mn<ROWS>(kr);
        off[i] = (unsigned)(kr * (int)ld + (c32 * 2 + (p16 & 1)) * 8);
      }
    }
  }

  // K-major tile whose row r (tile-relative) starts at element offset rowoff(r) instead of r * ld
  template <typename F>
  __device__ __forceinline__ void init_rows(F rowoff, int wave, int lane) {
    static_assert(KMAJ, "row gather: K-major operands");
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
      const int t = wave + NW * i;
      const int r = 8 * t + (lane >> 3);
      const int c = (lane & 7) ^ ((r >> 1) & 7);
      off[i] = (unsigned)(rowoff(r) + c * 8);
    }
  }

  // Stage the tile whose origin is the wave-uniform pointer `g` into the LDS slot `lds`.
  __device__ __forceinline__ void stage(const bf16_t* __restrict__ g, lds_char* lds, int wave) const {
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
      const int t = wave + NW * i;
      __builtin_amdgcn_global_load_lds((glb_cptr)(g + (size_t)off[i]),
                                       (__attribute__((address_space(3))) void*)(lds + t * 1024), 16, 0,
                                       0);
    }
  }
};

// One 16(rows) x 32(k) MFMA operand fragment: lane l holds rows row0+(l&15),
// k = 32*kk + 8*(l>>4) + j, j = 0..7.
typedef int v2i32_ __attribute__((ext_vector_type(2)));
typedef int v4i32_ __attribute__((ext_vector_type(4)));
template <int ROWS, bool KMAJ, bool F8 = false>
__device__ __forceinline__ bf16x8 load_frag(const lds_char* lds, int row0, int kk, int lane) {
  if constexpr (F8 && !KMAJ) {
    // 16 consecutive k (bytes) of operand row row0 + (lane & 15), k = 16 (4 kk + (lane >> 4)) ..: the positions the
    // K-major fragment of the other operand holds (chunk 4 kk + (lane >> 4) of a 128-byte row)
    static_assert(ROWS == 128, "MN-major fp8: 128-byte k-rows");
    const int i = lane & 15;
    const int k0 = 16 * (kk * 4 + (lane >> 4)) + (i >> 1), k1 = k0 + 8;
    const int c = row0 >> 4;
    const lds_char* a0 = lds + k0 * 128 + ((c ^ swz_mn8(k0)) << 4) + 8 * (i & 1);
    const lds_char* a1 = lds + k1 * 128 + ((c ^ swz_mn8(k1)) << 4) + 8 * (i & 1);
    const v2i32_ lo = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) v2i32_*)a0);
    const v2i32_ hi = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) v2i32_*)a1);
    const v4i32_ v = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(bf16x8, v);
  } else if constexpr (KMAJ) {
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

// v_permlane16_swap_b32: lanes 16-31 / 48-63 of `a` trade places with lanes 0-15 / 32-47 of `b`.  Inline asm
// on scalars: hipcc (ROCm 7.2) miscompiles __builtin_amdgcn_permlane16_swap when its results are inserted into
// vector elements (tools/probe_vector_elements.hip: elements 1..3 come back as copies of other lanes' element 0).  The
// s_nop covers the VALU-write -> permlane-read hazard (2 wait states), which nothing pads inside an asm.
__device__ __forceinline__ void swap_rows16(float& a, float& b) {
  asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}

// One accumulate step on a pair of 16-byte fragments.  bf16: 16x16x32.  fp8: the two fragments of a staged tile
// (kk = 0, 1: 2 x 16 fp8 values per lane) feed ONE v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales
// (E8M0 127): twice the MFMA rate of bf16 per k.  A and B fragments take their bytes from the same k positions,
// so the instruction's internal k order does not matter.
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma_fp8_k128(const bf16x8 f0, const bf16x8 f1, const bf16x8 s0, const bf16x8 s1,
                                               const f32x4 c) {
  const i32x4 a0 = __builtin_bit_cast(i32x4, f0), a1 = __builtin_bit_cast(i32x4, f1);
  const i32x4 b0 = __builtin_bit_cast(i32x4, s0), b1 = __builtin_bit_cast(i32x4, s1);
  const i32x8 a = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
  const i32x8 b = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
  return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
}

// The same on operands that were LOADED as 8-dword tuples (the ping-pong loop: with a dozen fragments alive, assembling
// the tuples at the MFMA from separate 4-dword halves doubled the fragment registers and spilled).
__device__ __forceinline__ f32x4 mfma_fp8_k128(const i32x8 a, const i32x8 b, const f32x4 c) {
  return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
}
// ... accumulating IN PLACE, as inline asm: through the builtin hipcc (ROCm 7.2) gave a third of the ping-pong loop's
// MFMAs a destination other than their accumulator input (the three-address form), which doubles the live accumulators
// of a kernel that has none to spare: 317 dwords spilled.  "+v" ties destination and accumulator.  The results are first
// read by ordinary instructions in the epilogue, behind two workgroup barriers (the wait states an MFMA result needs
// before a VALU read have long passed; nothing pads them inside an asm).
__device__ __forceinline__ void mfma_fp8_k128_acc(const i32x8 a, const i32x8 b, f32x4& c, const int unit_scale) {
  asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]"
               : "+v"(c) : "v"(a), "v"(b), "v"(unit_scale));
}
// Both k halves (kk = 0, 1) of one fp8 fragment as the instruction's 8-dword operand.
template <int ROWS, bool KMAJ>
__device__ __forceinline__ i32x8 load_frag8(const lds_char* lds, int row0, int lane) {
  if constexpr (KMAJ) {
    const i32x4 f0 = __builtin_bit_cast(i32x4, load_frag<ROWS, true, true>(lds, row0, 0, lane));
    const i32x4 f1 = __builtin_bit_cast(i32x4, load_frag<ROWS, true, true>(lds, row0, 1, lane));
    return i32x8{f0[0], f0[1], f0[2], f0[3], f1[0], f1[1], f1[2], f1[3]};
  } else {
    // MN-major: the swizzle of k-rows k, k + 8 and k + 64 is the same (swz_mn8 reads bits 1, 2 and 4 of k; the lane's
    // first row is 16 (lane >> 4) + ((lane & 15) >> 1)), so ONE per-lane address serves the four reads of a fragment
    // through immediate offsets: + 8 k-rows, + 64 k-rows (kk = 1), + 72
    static_assert(ROWS == 128, "MN-major fp8: 128-byte k-rows");
    const int i = lane & 15, k = 16 * (lane >> 4) + (i >> 1);
    const lds_char* a = lds + k * 128 + (((row0 >> 4) ^ swz_mn8(k)) << 4) + 8 * (i & 1);
    typedef __attribute__((address_space(3))) v2i32_* lp;
    const v2i32_ r0 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lp)a);
    const v2i32_ r1 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lp)(a + 8 * 128));
    const v2i32_ r2 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lp)(a + 64 * 128));
    const v2i32_ r3 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lp)(a + 72 * 128));
    return i32x8{r0[0], r0[1], r1[0], r1[1], r2[0], r2[1], r3[0], r3[1]};
  }
}

// fp32 x 8 -> 8 fp8 (e4m3, OCP) bytes, saturating at +-448 (v_cvt_pk_fp8_f32 rounds to nearest even).
__device__ __forceinline__ unsigned long long pack_fp8x8(const float (&v)[8]) {
  float c[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) c[e] = fminf(fmaxf(v[e], -448.f), 448.f);
  unsigned lo = 0, hi = 0;
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(c[0], c[1], lo, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(c[2], c[3], lo, true);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(c[4], c[5], hi, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(c[6], c[7], hi, true);
  return ((unsigned long long)hi << 32) | lo;
}

// Wait until at most `tiles` staged tiles (GL LDS-DMA instructions each) are still in flight.
template <int GL>
__device__ __forceinline__ void wait_tiles_in_flight(int tiles) {
  if (tiles >= 6) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 * GL < 63 ? 6 * GL : 63) : "memory");
  else if (tiles == 5) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * GL < 63 ? 5 * GL : 63) : "memory");
  else if (tiles == 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * GL < 63 ? 4 * GL : 63) : "memory");
  else if (tiles == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * GL) : "memory");
  else if (tiles == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * GL) : "memory");
  else if (tiles == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GL) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// Sum over the block's NW waves; result valid in thread 0.  `red` = NW floats of LDS.
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) red[w] = v;
  __syncthreads();
  float r = 0.f;
  if (threadIdx.x == 0)
    for (int i = 0; i < NW; ++i) r += red[i];
  __syncthreads();
  return r;
}

// ---------------------------------------------------------------------------------------------
// "Ping-pong" main loop for the 256x256 tile (8 waves as 2 x 4 of 128 x 64; NSTAGE == 8 selects it).
//
// The two waves that share a SIMD (wave rows 0 and 1) run the same program ONE BARRIER APART: while one
// is between the two barriers that bracket a cluster of 16 MFMAs, the other issues the LDS fragment
// reads and the LDS-DMA staging of its next phase, so the matrix pipe of every SIMD is fed by one
// wave while the other waits on memory.  Four phases per 64-deep K tile, one quadrant (64 x 32) of the
// wave's 128 x 64 output per phase:
//      phase 0: read B(n0) + A(m0)   MFMA (m0,n0)      stage B-half 1 of tile kt+1
//      phase 1: read B(n1)           MFMA (m0,n1)      stage A-half 0 of tile kt+1
//      phase 2: read A(m1)           MFMA (m1,n1)      stage A-half 1 of tile kt+1
//      phase 3: (no reads)           MFMA (m1,n0)      stage B-half 0 of tile kt+2, wait: tile kt+1 landed
// LDS: 2 buffers x {A half 0, A half 1, B half 0, B half 1} x 16 KiB (128 rows x 64 k).  A half is
// restaged two or more phases after its last fragment read (the partner wave group runs a barrier
// behind), and read two barriers after the counted vmcnt that retires its LDS-DMA.
// `tail_hook` runs once per wave in phase 0 of the LAST K tile, where the loop has nothing left to stage: a place to
// put LDS-DMA of epilogue operands in flight (they are retired by the loop's final vmcnt(0)).
// FP8: e4m3 operands (pointers typed bf16, leading dims in 2-byte units): a K tile is 128 deep, the LDS images hold the
// same 16 KiB per half (K-major: 128 rows x 128 k-bytes; MN-major: 128 k-rows x 128 row-bytes, StageOffsets F8MN),
// and a phase's 16 MFMAs of 16x16x32 become 8 of 16x16x128 -- the same matrix-pipe time for twice the contraction.
// `b_rows` (anything but NoGather: K-major B only): element offset of row r (0..127) of B's first half tile from `Bg`,
// `b_half_gather` the offset of the second half's row r from the first's -- the head-interleaved gather of EPI_REPARAM.
struct NoGather {};
template <bool A_KMAJ, bool B_KMAJ, bool FP8, typename Hook, typename BRows = NoGather>
__device__ __forceinline__ void mainloop_pingpong(const bf16_t* __restrict__ Ag, const bf16_t* __restrict__ Bg,
                                                  const long lda, const long ldb, const int nk, lds_char* smem,
                                                  const int wave, const int lane, f32x4 (&acc)[8][4], Hook tail_hook,
                                                  BRows b_rows = BRows{}, const long b_half_gather = 0,
                                                  const bf16_t* __restrict__ Ag_next = nullptr,
                                                  const bf16_t* __restrict__ Bg_next = nullptr, const bool resumed = false) {
  // Tile lists (gemm_pp_persist_kernel): `Ag_next` / `Bg_next` are the operand origins of the workgroup's NEXT output tile.
  // The loop then keeps staging across the tile boundary -- K tile t >= nk is K tile t - nk of the next tile (nk is even, so
  // the buffer parity carries over) -- and ends in exactly the state its own prologue produces: K tile 0 landed and
  // published, B-half 0 of K tile 1 in flight.  `resumed`: that state was left by the previous tile's loop, skip the prologue.
  constexpr bool GATHER_B = !std::is_same<BRows, NoGather>::value;
  const bool has_next = Ag_next != nullptr;
  constexpr int HALF = 128 * 128;        // bytes of one 128-row x 64-k half tile
  constexpr int BUF = 4 * HALF;          // A0 A1 B0 B1
  const int wr = wave >> 2, wc = wave & 3;
  StageOffsets<128, A_KMAJ, 8, FP8 && !A_KMAJ> sa;
  StageOffsets<128, B_KMAJ, 8, FP8 && !B_KMAJ> sb;
  sa.init(lda, wave, lane);
  if constexpr (GATHER_B) sb.init_rows(b_rows, wave, lane);
  else sb.init(ldb, wave, lane);
  const int unit_scale = 0x7F7F7F7F;     // fp8: E8M0 block scales of 1.0 for both operands (held in one VGPR)
  constexpr int KROWS = FP8 ? 128 : 64;  // k-rows of an MN-major tile
  constexpr int MNH = FP8 ? 64 : 128;    // 128 operand rows of an MN-major image, in 2-byte units
  const long a_step = A_KMAJ ? 64 : KROWS * lda, b_step = B_KMAJ ? 64 : KROWS * ldb;
  const long a_half = A_KMAJ ? 128 * lda : MNH, b_half = GATHER_B ? b_half_gather : (B_KMAJ ? 128 * ldb : MNH);
  auto stage_a = [&](int h, int t) {
    if (t < nk) sa.stage(Ag + h * a_half + (long)t * a_step, smem + (t & 1) * BUF + h * HALF, wave);
    else if (has_next) sa.stage(Ag_next + h * a_half + (long)(t - nk) * a_step, smem + (t & 1) * BUF + h * HALF, wave);
  };
  auto stage_b = [&](int h, int t) {
    if (t < nk) sb.stage(Bg + h * b_half + (long)t * b_step, smem + (t & 1) * BUF + (2 + h) * HALF, wave);
    else if (has_next) sb.stage(Bg_next + h * b_half + (long)(t - nk) * b_step, smem + (t & 1) * BUF + (2 + h) * HALF, wave);
  };
  // fragment registers: bf16 -- [.][kk] 4-dword fragments; fp8 -- the 8 dwords of a fragment's two k halves live in ONE
  // tuple (elements [.][0] and [.][1] of these arrays are then the low and high half of that tuple's storage)
  struct Frag { bf16x8 h[2]; };
  union FragU { Frag f; i32x8 t; __device__ FragU() {} };
  FragU a[4], b0[2], b1[2], b2[2];
  const int brow = (wc & 1) * 64;
  auto rd_a = [&](const lds_char* buf, int mq) {
    const lds_char* base = buf + wr * HALF;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if constexpr (FP8) a[i].t = load_frag8<128, A_KMAJ>(base, mq * 64 + i * 16, lane);
      else {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) a[i].f.h[kk] = load_frag<128, A_KMAJ>(base, mq * 64 + i * 16, kk, lane);
      }
    }
  };
  auto rd_b = [&](FragU (&b)[2], const lds_char* buf, int nq) {
    const lds_char* base = buf + (2 + (wc >> 1)) * HALF;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if constexpr (FP8) b[j].t = load_frag8<128, B_KMAJ>(base, brow + nq * 32 + j * 16, lane);
      else {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) b[j].f.h[kk] = load_frag<128, B_KMAJ>(base, brow + nq * 32 + j * 16, kk, lane);
      }
    }
  };
  // close the memory half of a phase, run its MFMA cluster between two barriers
  auto mma = [&](auto mq_c, auto nq_c, FragU (&b)[2]) {
    constexpr int MQ = decltype(mq_c)::value, NQ = decltype(nq_c)::value;
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    if constexpr (FP8) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          mfma_fp8_k128_acc(b[j].t, a[i].t, acc[MQ * 4 + i][NQ * 2 + j], unit_scale);
    } else {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[MQ * 4 + i][NQ * 2 + j] =
                __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j].f.h[kk], a[i].f.h[kk], acc[MQ * 4 + i][NQ * 2 + j], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  // One K tile = four phases.  `cur`/`nxt` are the (compile-time distinct) LDS buffers of tile kt and
  // kt+1.  EARLY_B (both operands K-major: few address registers, so a third B fragment set fits):
  // bq holds B(n0) of tile kt on entry and bn receives B(n0) of tile kt+1 in phase 3, which spreads the
  // fragment reads 8 / 4 / 8 / 4 over the phases; otherwise B(n0) is read in phase 0 (12 / 4 / 8 / 0).
  constexpr bool EARLY_B = A_KMAJ && B_KMAJ;
  auto k_tile = [&](int kt, const lds_char* cur, const lds_char* nxt, FragU (&bq)[2], FragU (&bn)[2]) {
    // phase 0: (B(n0),) A(m0);  stage B-half 1 of tile kt+1
    if constexpr (!EARLY_B) rd_b(bq, cur, 0);
    rd_a(cur, 0);
    stage_b(1, kt + 1);
    if (kt + 1 >= nk) tail_hook();
    mma(I0{}, I0{}, bq);
    // phase 1: B(n1);  stage A-half 0 of tile kt+1
    rd_b(b1, cur, 1);
    stage_a(0, kt + 1);
    mma(I0{}, I1{}, b1);
    // phase 2: A(m1);  stage A-half 1 of tile kt+1.  EARLY_B: both B halves of tile kt+1 must have landed
    // for the phase-3 reads (the two A halves just issued may stay in flight)
    rd_a(cur, 1);
    stage_a(1, kt + 1);
    if constexpr (EARLY_B) {
      if (kt + 1 < nk || has_next) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    mma(I1{}, I1{}, b1);
    // phase 3: (B(n0) of tile kt+1;)  stage B-half 0 of tile kt+2; all of tile kt+1 must have landed
    if constexpr (EARLY_B) {
      if (kt + 1 < nk) rd_b(bn, nxt, 0);
    }
    stage_b(0, kt + 2);
    if (kt + 2 < nk || has_next) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    mma(I1{}, I0{}, bq);
  };

  // prologue: tile 0 complete, B-half 0 of tile 1 in flight
  if (!resumed) {
    stage_a(0, 0); stage_a(1, 0); stage_b(0, 0); stage_b(1, 0);
    stage_b(0, 1);
    if (nk > 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  // (resumed: the previous tile's last phase waited for this K tile and ran two barriers behind that wait; its B(n0) fragments
  // are read here rather than in that phase -- the registers do not outlive the call)
  if constexpr (EARLY_B) rd_b(b0, smem, 0);
  if (wr == 1) __builtin_amdgcn_s_barrier();   // wave row 1 runs one barrier behind wave row 0
  __builtin_amdgcn_sched_barrier(0);

  for (int kt = 0; kt < nk; kt += 2) {   // nk is even (the launcher falls back to the ring loop otherwise)
    if constexpr (EARLY_B) {
      k_tile(kt, smem, smem + BUF, b0, b2);
      k_tile(kt + 1, smem + BUF, smem, b2, b0);
    } else {
      k_tile(kt, smem, smem + BUF, b0, b0);
      k_tile(kt + 1, smem + BUF, smem, b0, b0);
    }
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();   // re-align the two wave rows
  if constexpr (FP8) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the asm MFMAs' results, before any VALU reads them
  __builtin_amdgcn_sched_barrier(0);
}

// Block tile BM x BN computed by a WGM x WGN grid of waves (wave tile BM/WGM x BN/WGN).
//
// NSTAGE-deep LDS ring, software-pipelined so it also runs at one wave per SIMD:
//   * all NSTAGE slots are staged up front; tiles are retired with a COUNTED vmcnt
//     (never 0 in steady state) and ONE raw s_barrier per K tile;
//   * the barrier sits in the MIDDLE of a tile's MFMAs: after the second-half fragments of
//     tile kt are in registers, wait(tile kt+1 landed) -> barrier -> refill the slot of
//     tile kt with tile kt+NSTAGE -> read the first-half fragments of tile kt+1 -> second
//     half of tile kt's MFMAs.  Every ds_read is therefore issued one MFMA half-phase
//     before its consumer, and the DMA refill has NSTAGE-1 tiles of MFMA time to land.
// The barrier publishes every wave's share of tile kt+1 and orders the refill after all
// reads of the vacated slot (each wave's fragment reads have returned before it arrives).
template <int BM, int BN, int WGM, int WGN, bool A_KMAJ, bool B_KMAJ, int EPI, int NSTAGE, bool FP8 = false>
__device__ __forceinline__ void gemm_body(const GemmArgs& p, const int bid, char* smem_generic, const int next_bid = -1,
                                          const bool resumed = false) {
  static_assert(!FP8 || NSTAGE == 8 || (A_KMAJ && B_KMAJ), "fp8 operands: K-major (forward) GEMMs on the ring loop, any layout on the ping-pong loop");
  lds_char* smem = (lds_char*)smem_generic;
  constexpr int NW = WGM * WGN;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int B_OFF = A_BYTES;             // the B image inside a stage
  constexpr int WTM = BM / WGM, WTN = BN / WGN, MI = WTM / 16, NI = WTN / 16;
  constexpr int GL = STAGE / 1024 / NW;  // LDS-DMA instructions per wave per tile
  constexpr bool PINGPONG = NSTAGE == 8;    // 256x256 ping-pong main loop (2 LDS buffers)
  // (deep rings -- 6 or 7 slots of a small tile -- exist for the latent-sized GEMMs: what a CU pulls through its L2 -> LDS port
  // is bytes in flight / latency, and three 24 KB tiles in flight gave 43 GB/s where the port does 60-70)
  static_assert(PINGPONG || (NSTAGE >= 2 && NSTAGE <= 7 && (NSTAGE - 1) * GL < 64), "vmcnt is a 6-bit counter");
  static_assert(!PINGPONG || (BM == 256 && BN == 256 && WGM == 2 && WGN == 4), "ping-pong loop: 256x256, 2x4 waves");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  // XCD-aware work order over the 1-D grid of (split, tile_m, tile_n) items: workgroups are
  // dealt round-robin over the 8 XCDs (blocks b and b+8 share an L2), so give each XCD a
  // contiguous run of items -- one K split and a few tile rows -- whose A and B panels it then
  // shares through its own L2 instead of every XCD streaming every panel.  Bijective for any
  // grid size; a different placement only changes speed.
  const int tiles_n = p.tiles_n, nt = p.tiles_n * p.tiles_m, nwg = nt * p.splits;
  // within a split, tiles are enumerated in 4x4 super-tiles where the grid allows it, so that an XCD's
  // contiguous run of items is a near-square patch: fewer distinct A row panels + B column panels stream
  // through its private L2 than with row-major order (a 2 x 16 strip needs 1 + 4 MB at C2, a 4 x 8 patch 2 + 2)
  auto locate = [&](const int b_, int& split_, int& tile_m_, int& tile_n_) {
    const int xcd = b_ & 7, q = nwg >> 3, rr = nwg & 7;
    const int item = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (b_ >> 3);
    split_ = item / nt;
    const int tid_lin = item - split_ * nt;
    if (((tiles_n | p.tiles_m) & 3) == 0) {
      const int sn = tiles_n >> 2, s4 = tid_lin >> 4, w4 = tid_lin & 15;
      tile_m_ = (s4 / sn) * 4 + (w4 >> 2);
      tile_n_ = (s4 % sn) * 4 + (w4 & 3);
    } else {
      tile_m_ = tid_lin / tiles_n;
      tile_n_ = tid_lin - tile_m_ * tiles_n;
    }
  };
  int split, tile_m, tile_n;
  locate(bid, split, tile_m, tile_n);
  const long m0 = (long)tile_m * BM, n0 = (long)tile_n * BN;
  const long k0 = (long)split * p.k_tiles * 64;
  // (fp8: one byte per element and pointers typed bf16 -- a K-major row advances by k0 two-byte units for 2 k0 bytes, an
  // MN-major image starts 2 k0 k-rows down and m0 / 2 two-byte units in)
  constexpr int F8S = FP8 ? 2 : 1;
  auto a_origin = [&](const long m_, const long k_) { return A_KMAJ ? p.A + m_ * p.lda + k_ : p.A + (F8S * k_) * p.lda + m_ / F8S; };
  auto b_origin = [&](const long n_, const long k_) { return B_KMAJ ? p.B + n_ * p.ldb + k_ : p.B + (F8S * k_) * p.ldb + n_ / F8S; };
  const bf16_t* Ag = a_origin(m0, k0);
  const bf16_t* Bg = b_origin(n0, k0);
  // the workgroup's next output tile, when it walks a list of them (gemm_pp_persist_kernel)
  const bf16_t *Ag_next = nullptr, *Bg_next = nullptr;
  if (next_bid >= 0) {
    int s2, tm2, tn2;
    locate(next_bid, s2, tm2, tn2);
    Ag_next = a_origin((long)tm2 * BM, (long)s2 * p.k_tiles * 64);
    Bg_next = b_origin((long)tn2 * BN, (long)s2 * p.k_tiles * 64);
  }
  const long a_step = A_KMAJ ? 64 : 64 * p.lda;
  const long b_step = B_KMAJ ? 64 : 64 * p.ldb;

  // Epilogue geometry (see the epilogue below): after the column-block pairing lane (eq = l>>4, ej = l&15) owns
  // the (row, 8-column) items  row = roww + 16 * mi,  col = colw + 32 * t.
  static_assert(NI % 2 == 0, "epilogue pairs column fragments");
  constexpr int NP = NI / 2;                       // column-block pairs per wave tile
  // (the reparameterisation backward on the ping-pong tile keeps 128 accumulators and 32 column sums live and reads 24 operand
  // registers per item: one fragment row per chunk there, or the epilogue spills)
  constexpr int CM = (EPI == EPI_REPARAM_BWD && NSTAGE == 8) ? 1 : NP >= 2 ? 2 : (MI >= 4 ? 4 : MI);  // fragment rows per chunk (bounds live registers)
  static_assert(MI % CM == 0, "chunking must divide the wave tile");
  constexpr int CH = CM * NP;                      // (row, 8-column) items per chunk
  const int eq = lane >> 4, ej = lane & 15;
  const long colw = n0 + wn * WTN + (eq & 1) * 16 + (eq >> 1) * 8;   // + 32 * t
  const long roww = m0 + wm * WTM + ej;                              // + 16 * mi
  typedef int i32x4_ __attribute__((ext_vector_type(4)));

  // Epilogue operands that do not depend on the accumulators are fetched BEFORE the main loop, for the first
  // chunk of the epilogue (the only one on the tiles the step uses for these GEMMs): the loss target of
  // EPI_TANH_LOSS (fp32 frames) and the ReLU mask of EPI_MASK_BF16.  Issued ahead of every LDS-DMA of the ring
  // they are the oldest entries of the in-order vmcnt queue, so the counted waits of the loop are unchanged, and
  // the round trip to memory is over long before the epilogue, which used to start with it (3.7 us of the fc4
  // forward, profiles/r02_gemm_decomp.txt).  Costs CH x 8 (target) / CH x 4 (mask) VGPRs across the loop.
  // (The 256 x 256 ping-pong kernels have no registers to spare: their mask goes through LDS, below.)
  constexpr bool PF_ROOM = !PINGPONG && MI * NI <= 16;   // accumulators take at most 64 VGPRs
  constexpr bool PF_X = EPI == EPI_TANH_LOSS && PF_ROOM;
  constexpr bool PF_MASK = EPI == EPI_MASK_BF16 && PF_ROOM;
  float xpf[CH][8];
  i32x4_ mpf[CH];
  bool pf_on = false;   // wave-uniform
  if constexpr (PF_X) {
    const bool x_al0 = p.x && !p.x_hop && (p.ld_x & 3) == 0 && ((reinterpret_cast<uintptr_t>(p.x) & 15) == 0);
    if (x_al0 && m0 + BM <= p.M_valid && n0 + BN <= p.N_valid) {
      pf_on = true;
#pragma unroll
      for (int it = 0; it < CH; ++it) {
        const float* xs = p.x + (roww + 16 * (it / NP)) * p.ld_x + colw + 32 * (it % NP);
        const f32x4 lo = *(const f32x4*)xs, hi = *(const f32x4*)(xs + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { xpf[it][e] = lo[e]; xpf[it][4 + e] = hi[e]; }
      }
    }
  }
  if constexpr (PF_MASK) {
    if (p.mask) {
      pf_on = true;
#pragma unroll
      for (int it = 0; it < CH; ++it)
        mpf[it] = *(const i32x4_*)(p.mask + (roww + 16 * (it / NP)) * p.ld_mask + colw + 32 * (it % NP));
    }
  }
  // ... and the reparameterisation epilogues' operands: mu | logvar and eps of the lane's (row, latent) items (EPI_REPARAM_BWD:
  // 6 x 16 bytes per item), eps of its rows (EPI_REPARAM); eps as 16-byte loads where the exact latent width is a multiple
  // of 4 and the item lies inside the valid extent -- everything else is read in the epilogue, element by element.
  constexpr bool PF_RB = EPI == EPI_REPARAM_BWD, PF_RF = EPI == EPI_REPARAM;
  constexpr bool PF_RB_ON = PF_RB && MI == CM;   // one chunk (the 64-row tiles): its operands are fetched ahead
  constexpr int RF_N = MI * (NI / 2);   // EPI_REPARAM: (row fragment, mu | logvar fragment pair) items of a wave
  f32x4 rb_m[PF_RB ? CH : 1][2], rb_l[PF_RB ? CH : 1][2], rb_e[PF_RB ? CH : (PF_RF ? RF_N : 1)][2];
  bool rb_vec[PF_RB ? CH : (PF_RF ? RF_N : 1)];
  if constexpr (PF_RB_ON) {
    const long L2p_ = 2 * p.lat_lp, L_ = p.lat_l;
    const bool al = (L_ & 3) == 0 && (reinterpret_cast<uintptr_t>(p.eps) & 15) == 0;
#pragma unroll
    for (int it = 0; it < CH; ++it) {
      const long r_ = roww + 16 * (it / NP), c_ = colw + 32 * (it % NP);
      const float* mp = p.mulv + r_ * L2p_ + c_;
      rb_m[it][0] = *(const f32x4*)mp; rb_m[it][1] = *(const f32x4*)(mp + 4);
      rb_l[it][0] = *(const f32x4*)(mp + p.lat_lp); rb_l[it][1] = *(const f32x4*)(mp + p.lat_lp + 4);
      rb_vec[it] = al && r_ < p.M_valid && c_ + 8 <= L_;
      rb_e[it][0] = rb_e[it][1] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (rb_vec[it]) {
        rb_e[it][0] = *(const f32x4*)(p.eps + r_ * L_ + c_);
        rb_e[it][1] = *(const f32x4*)(p.eps + r_ * L_ + c_ + 4);
      }
    }
  }
  if constexpr (PF_RF) {
    // (with more than four items -- the 256-row tiles of large batches -- eps is read in the epilogue: the loop has no
    // registers to spare for it, and one round trip in 16 tiles' worth of work is not what bounds such a block)
    const long L_ = p.lat_l;
    const bool al = RF_N <= 4 && p.eps_in && (L_ & 3) == 0 && (reinterpret_cast<uintptr_t>(p.eps_in) & 15) == 0;
#pragma unroll
    for (int it = 0; it < RF_N; ++it) {
      const long b_ = m0 + wm * WTM + (it / (NI / 2)) * 16 + (lane & 15);
      const long l_ = (long)tile_n * (BN / 2) + wn * (WTN / 2) + (it % (NI / 2)) * 16 + 4 * (lane >> 4);
      rb_vec[it] = al && b_ < p.M_valid && l_ + 4 <= L_;
      rb_e[it][0] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (rb_vec[it]) rb_e[it][0] = *(const f32x4*)(p.eps_in + b_ * L_ + l_);
    }
  }
  if constexpr (PF_X || PF_MASK || PF_RB || PF_RF) {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);   // keep these loads ahead of the ring's first LDS-DMA
  }

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

  // The 256 x 256 ping-pong kernels take the ReLU mask of their wave tile (128 rows x 128 B) through LDS: each wave
  // issues its mask pieces as LDS-DMA (1 KiB = 8 rows each; 16-byte piece c of row r lands at position c ^ (r & 7),
  // swizzled on the source address as in the main loop) and retires them chunk by chunk with a counted vmcnt -- one
  // round trip to memory for the whole epilogue instead of one per chunk (loads into registers, stores, next chunk's
  // loads ...: 4 dependent round trips per wave in a tail where all 256 blocks issue theirs together).  The first
  // chunk's pieces are put in flight during the LAST K tile, into the 32 KiB of LDS behind the two ring buffers, and
  // have landed when the loop ends; the other three follow into the ring once it is free (epilogue).
  constexpr bool MASK_LDS = EPI == EPI_MASK_BF16 && PINGPONG;
  static_assert(!MASK_LDS || (WTN == 64 && CH == 2 * CM), "mask through LDS: 128-byte mask rows, as many stores as DMA pieces per chunk");
  constexpr int MK_C0 = 2 * CM * 1024;                       // bytes of one chunk's pieces per wave (4 KiB)
  lds_char* mk_lds = smem + wave * (WTM * 128);              // chunks 1.. : the wave's 16 KiB of the ring (first 4 KiB unused)
  lds_char* mk_lds0 = smem + 2 * (BM + BN) * 128 + wave * MK_C0;   // chunk 0: behind the ring
  bool mask_lds = false;
  const char* mk_g = nullptr;       // this lane's piece of row (lane >> 3) of the wave tile's mask, as a byte address
  long mk_pitch8 = 0;               // bytes between DMA instructions (8 mask rows)
  const bool mask8 = FP8 && MASK_LDS && p.mask_fp8;   // wave-uniform
  if constexpr (MASK_LDS) {
    if (p.mask) {
      mask_lds = true;
      const int mrow = lane >> 3;
      if (mask8) {
        // fp8 image: a mask row of the wave tile is 64 bytes = four 16-byte pieces; LDS position s (0..3) of row r holds
        // piece s ^ (r & 3), positions 4..7 of the 128-byte LDS row repeat them (never read: the instruction count, and
        // with it the counted vmcnt waits of the epilogue, stay those of the bf16 mask)
        const int mpc = ((lane & 3) ^ (mrow & 3));
        mk_g = (const char*)p.mask + (m0 + wm * WTM + mrow) * p.ld_mask + n0 + wn * WTN + mpc * 16;
        mk_pitch8 = 8 * p.ld_mask;
      } else {
        const int mpc = (lane & 7) ^ (lane >> 3);
        mk_g = (const char*)(p.mask + (m0 + wm * WTM + mrow) * p.ld_mask + n0 + wn * WTN + mpc * 8);
        mk_pitch8 = 16 * p.ld_mask;
      }
    }
  }
  auto mask_dma = [&](int i0, int i1, lds_char* base) {     // pieces [i0, i1) of the wave tile (8 rows each)
#pragma unroll
    for (int i = i0; i < i1; ++i)
      __builtin_amdgcn_global_load_lds((glb_cptr)(mk_g + (long)i * mk_pitch8),
                                       (__attribute__((address_space(3))) void*)(base + (i - i0) * 1024), 16, 0, 0);
  };
  if constexpr (PINGPONG && EPI == EPI_REPARAM) {
    // B's rows gathered head-interleaved, as on the ring loop below: row r of the 256-row tile is head (r >> 4) & 1, latent
    // tile_n * 128 + (r >> 5) * 16 + (r & 15) -- the second half tile (rows 128..255) starts 64 latents after the first
    static_assert(B_KMAJ && !FP8, "reparameterisation epilogue: bf16 K-major weights");
    mainloop_pingpong<A_KMAJ, B_KMAJ, FP8>(Ag, p.B + k0, p.lda, p.ldb, p.k_tiles, smem, wave, lane, acc, [&]() {},
        [&](int r) { return (((r >> 4) & 1) * p.lat_lp + (long)tile_n * (BN / 2) + (r >> 5) * 16 + (r & 15)) * p.ldb; }, 64 * p.ldb);
  } else if constexpr (PINGPONG) {
    mainloop_pingpong<A_KMAJ, B_KMAJ, FP8>(Ag, Bg, p.lda, p.ldb, p.k_tiles, smem, wave, lane, acc, [&]() {
      if constexpr (MASK_LDS) {
        if (mask_lds) mask_dma(0, 2 * CM, mk_lds0);
      }
    }, NoGather{}, 0, Ag_next, Bg_next, resumed);
  } else {
  StageOffsets<BM, A_KMAJ, NW> sa;
  StageOffsets<BN, B_KMAJ, NW> sb;
  const bf16_t* Agr = Ag;
  bool gathered = false;
  if constexpr (A_KMAJ && EPI == EPI_BIAS_ACT_BF16 && !FP8) {
    if (p.a_hop) {   // rows are frames of the resident bf16 waveform
      gathered = true;
      Agr = p.A + k0;
      sa.init_rows([&](int r) {
        long row = m0 + r;
        row = row < p.a_rows ? row : p.a_rows - 1;
        return (p.a_idx ? (long)p.a_idx[row] : p.a_first + row) * p.a_hop;
      }, wave, lane);
    }
    if (p.step_inc && bid == 0 && tid == 0) *p.step_inc += 1;
  }
  if (!gathered) sa.init(p.lda, wave, lane);
  const bf16_t* Bgr = Bg;
  if constexpr (EPI == EPI_REPARAM) {
    static_assert(B_KMAJ && WTN % 32 == 0, "reparameterisation epilogue: tiles of BN / 2 latents x 2 heads, wave tiles of whole (mu, logvar) fragment pairs");
    Bgr = p.B + k0;
    sb.init_rows([&](int r) { return (((r >> 4) & 1) * p.lat_lp + (long)tile_n * (BN / 2) + (r >> 5) * 16 + (r & 15)) * p.ldb; }, wave, lane);
  } else {
    sb.init(p.ldb, wave, lane);
  }
  // by-product of the gathered operand: K tile kt of this block's rows, from its LDS image to HBM
  auto copy_out = [&](int kt, const lds_char* slot_) {
    if constexpr (A_KMAJ && EPI == EPI_BIAS_ACT_BF16 && !FP8) {
      if (gathered && p.a_copy && kt % tiles_n == tile_n) {
#pragma unroll
        for (int i = 0; i < StageOffsets<BM, A_KMAJ, NW>::PER_WAVE; ++i) {
          const int t = wave + NW * i;
          const int r = 8 * t + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
          const bf16x8 v_ = *(const __attribute__((address_space(3))) bf16x8*)(slot_ + t * 1024 + lane * 16);
          store_out16((bf16x8*)(p.a_copy + (m0 + r) * p.ld_copy + (long)kt * 64 + c * 8), v_, p.wt);
        }
      }
    }
  };

  const int nk = p.k_tiles;
#pragma unroll
  for (int s = 0; s < NSTAGE; ++s)
    if (s < nk) {
      sa.stage(Agr + s * a_step, smem + s * STAGE, wave);
      sb.stage(Bgr + s * b_step, smem + s * STAGE + B_OFF, wave);
    }
  // tile 0 landed (tiles 1..NSTAGE-1 may still be in flight)
  wait_tiles_in_flight<GL>(nk - 1 < NSTAGE - 1 ? nk - 1 : NSTAGE - 1);
  __builtin_amdgcn_s_barrier();

  bf16x8 a0[MI], b0[NI], a1[MI], b1[NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) a0[mi] = load_frag<BM, A_KMAJ>(smem, wm * WTM + mi * 16, 0, lane);
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) b0[ni] = load_frag<BN, B_KMAJ>(smem + B_OFF, wn * WTN + ni * 16, 0, lane);

  // ds_read instructions per half-tile of fragments, MFMAs per half-tile
  constexpr int NRD = MI * (A_KMAJ ? 1 : 2) + NI * (B_KMAJ ? 1 : 2);
  constexpr int NMF = MI * NI;

  // First half of tile kt: MFMAs on (a0, b0) with the reads of (a1, b1) slotted between them
  // (one MFMA first: its operands were loaded across the loop back-edge, and the lgkmcnt(0) the
  // compiler puts in front of it must not also drain the reads issued in this half).
  auto first_half = [&](const lds_char* cur) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) a1[mi] = load_frag<BM, A_KMAJ>(cur, wm * WTM + mi * 16, 1, lane);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) b1[ni] = load_frag<BN, B_KMAJ>(cur + B_OFF, wn * WTN + ni * 16, 1, lane);
    if constexpr (!FP8) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[ni], a0[mi], acc[mi][ni], 0, 0, 0);
    }
    // issue order: MFMA, then {k ds_reads, MFMA} ...  (fp8: no MFMA in this half -- a K = 128 instruction
    // takes both fragments of the tile, in the second half)
    if constexpr (!FP8) {
      constexpr int K1 = (NRD + NMF - 2) / (NMF - 1);  // reads per MFMA gap
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#pragma unroll
      for (int i = 0; i < NRD; i += K1) {
        __builtin_amdgcn_sched_group_barrier(0x100, K1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  // Second half of tile kt, entered right after the mid-tile barrier: the LDS-DMA refill of the
  // vacated slot and the first-half fragment reads of tile kt+1 are slotted between the MFMAs on
  // (a1, b1), so the matrix pipe restarts immediately after the barrier.
  auto second_half = [&](auto refill_c, auto next_c, int kt, int slot, int nslot) {
    constexpr bool REFILL = decltype(refill_c)::value, NEXT = decltype(next_c)::value;
    if constexpr (REFILL) {
      lds_char* rf = smem + slot * STAGE;
      sa.stage(Agr + (long)(kt + NSTAGE) * a_step, rf, wave);
      sb.stage(Bgr + (long)(kt + NSTAGE) * b_step, rf + B_OFF, wave);
    }
    // fp8: this tile's first-half fragments are still needed by the MFMAs below, so the next tile's go to a
    // second register set and are moved over afterwards
    bf16x8 a0n[FP8 ? MI : 1], b0n[FP8 ? NI : 1];
    if constexpr (NEXT) {
      const lds_char* nxt = smem + nslot * STAGE;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const bf16x8 f = load_frag<BM, A_KMAJ>(nxt, wm * WTM + mi * 16, 0, lane);
        if constexpr (FP8) a0n[mi] = f; else a0[mi] = f;
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const bf16x8 f = load_frag<BN, B_KMAJ>(nxt + B_OFF, wn * WTN + ni * 16, 0, lane);
        if constexpr (FP8) b0n[ni] = f; else b0[ni] = f;
      }
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        if constexpr (FP8) acc[mi][ni] = mfma_fp8_k128(b0[ni], b1[ni], a0[mi], a1[mi], acc[mi][ni]);
        else acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[ni], a1[mi], acc[mi][ni], 0, 0, 0);
      }
    if constexpr (FP8 && NEXT) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) a0[mi] = a0n[mi];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) b0[ni] = b0n[ni];
    }
    if constexpr (NEXT && !FP8) {
      constexpr int NV = REFILL ? GL : 0;
      constexpr int K2 = (NV + NRD + NMF - 2) / (NMF - 1);  // memory instructions per MFMA gap
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#pragma unroll
      for (int i = 0; i < NV; i += K2) {
        __builtin_amdgcn_sched_group_barrier(0x020, K2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
#pragma unroll
      for (int i = 0; i < NRD; i += K2) {
        __builtin_amdgcn_sched_group_barrier(0x100, K2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  // Mid-tile hand-off: this wave is done reading slot `slot` (register USE of the second-half
  // fragments makes the compiler place the lgkmcnt wait here), tile kt+1 has landed for this wave's
  // share (counted vmcnt), then the barrier publishes both facts to the block.
  auto hand_off = [&](int newer) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) asm volatile("" ::"v"(a1[mi]));
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) asm volatile("" ::"v"(b1[ni]));
    wait_tiles_in_flight<GL>(newer);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  using T_ = std::integral_constant<bool, true>;
  using F_ = std::integral_constant<bool, false>;
  int slot = 0;  // ring slot of tile kt
  int kt = 0;
  // steady state: a refill is issued every tile
  for (; kt + NSTAGE < nk; ++kt) {
    const int nslot = slot + 1 == NSTAGE ? 0 : slot + 1;
    copy_out(kt, smem + slot * STAGE);
    first_half(smem + slot * STAGE);
    hand_off(NSTAGE - 2);
    second_half(T_{}, T_{}, kt, slot, nslot);
    slot = nslot;
  }
  // drain: tiles already staged, nothing left to prefetch
  for (; kt + 1 < nk; ++kt) {
    const int nslot = slot + 1 == NSTAGE ? 0 : slot + 1;
    copy_out(kt, smem + slot * STAGE);
    first_half(smem + slot * STAGE);
    hand_off(nk - 2 - kt < NSTAGE - 2 ? nk - 2 - kt : NSTAGE - 2);
    second_half(F_{}, T_{}, kt, slot, nslot);
    slot = nslot;
  }
  // last tile
  copy_out(kt, smem + slot * STAGE);
  first_half(smem + slot * STAGE);
  second_half(F_{}, F_{}, kt, slot, slot);
  }
  // every wave is done with the ring before the epilogue's reductions reuse LDS -- not in a tile list's inner tiles: the ring
  // already holds the next tile's first K tile (LDS-DMA still in flight: a __syncthreads() would drain it), and the
  // epilogues that run there keep out of LDS (gemm_pp_persist_kernel)
  if (next_bid < 0) __syncthreads();
  // ------------------------------ epilogue ------------------------------
  // The MFMAs are issued with the operands swapped (first operand = the B fragment), so the accumulator
  // of fragment (mi, ni) holds C^T: lane l owns row mi*16 + (l&15) of the wave tile and the FOUR CONSECUTIVE
  // columns ni*16 + 4*(l>>4) + r, r = 0..3 -- row-contiguous straight out of the registers, no LDS staging.
  // One v_permlane16_swap per register between the fragments of a column-block pair (2t, 2t+1) widens
  // that to EIGHT consecutive columns per lane: afterwards lane (q = l>>4, j = l&15) holds columns
  // [c, c+8), c = (2t + (q&1))*16 + (q>>1)*8, of row mi*16 + j (first four in the pair's first fragment
  // register set, last four in the second), i.e. 16-byte bf16 / 2 x 16-byte fp32 accesses, and one store
  // instruction covers 16 rows x 64 B (bf16).  Within a chunk all global LOADS are issued before the first
  // store (loads and stores share the in-order vmcnt counter).
  if constexpr (MASK_LDS) {
    if (mask_lds) mask_dma(2 * CM, WTM / 8, mk_lds + MK_C0);   // chunks 1..: the ring is free now
  }

  if constexpr (EPI == EPI_REPARAM) {
    // The B tile's rows were gathered head-interleaved (above), so fragment (mi, 0) of this wave holds mu and fragment
    // (mi, 1) logvar of the SAME (row, latent) pairs: lane (q, j) owns row mi * 16 + j of the wave tile and the four
    // consecutive latents l .. l + 3.  No pairing swap, no hand-over between lanes: bias, eps, exp, z and the KL term
    // straight on the accumulators (k_reparam_fwd's arithmetic and eps draws -- counter = the (row, 4-latent group) index
    // over the padded [Mp, lat_lp / 4] grid --, elementwise.hip), 16-byte fp32 / 8-byte bf16 stores, 64 bytes per row
    // and store instruction.
    static_assert(NI % 2 == 0, "reparameterisation epilogue: (mu, logvar) fragment pairs");
    const int q = lane >> 4, j = lane & 15;
    const long Lp_ = p.lat_lp, L2p_ = 2 * Lp_, L_ = p.lat_l;
    const uint64_t step_ = p.step_counter ? (uint64_t)*p.step_counter : 0;
    const bool eps_al = (L_ & 3) == 0 && (reinterpret_cast<uintptr_t>(p.eps_in ? (const void*)p.eps_in : (const void*)p.eps_out) & 15) == 0;
    float kl = 0.f;
#pragma unroll
    for (int pr = 0; pr < NI / 2; ++pr) {
      const long l = (long)tile_n * (BN / 2) + wn * (WTN / 2) + pr * 16 + 4 * q;
      const f32x4 bm = *(const f32x4*)(p.bias + l), bv = *(const f32x4*)(p.bias + Lp_ + l);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int it = mi * (NI / 2) + pr;
        const long b = m0 + wm * WTM + mi * 16 + j;
        float mua[4] = {0.f, 0.f, 0.f, 0.f}, lva[4] = {0.f, 0.f, 0.f, 0.f}, zz[4] = {0.f, 0.f, 0.f, 0.f};
        if (b < p.M_valid && l < L_) {
          float ev[4];
          // (eps as one 16-byte access per item where the exact latent width allows it; the 64-row tiles fetched theirs before
          // the main loop)
          const bool v4 = eps_al && l + 4 <= L_;
          bool have = false;
          if (!p.eps_in) {
            normal4_fast(p.seed, (uint64_t)(b * (Lp_ >> 2) + (l >> 2)), step_, ev);
            if (v4) { *(f32x4*)(p.eps_out + b * L_ + l) = f32x4{ev[0], ev[1], ev[2], ev[3]}; have = true; }
          } else if (rb_vec[it]) {
#pragma unroll
            for (int e_ = 0; e_ < 4; ++e_) ev[e_] = rb_e[it][0][e_];
            have = true;
          } else if (v4) {
            const f32x4 t_ = *(const f32x4*)(p.eps_in + b * L_ + l);
#pragma unroll
            for (int e_ = 0; e_ < 4; ++e_) ev[e_] = t_[e_];
            have = true;
          }
#pragma unroll
          for (int e_ = 0; e_ < 4; ++e_) {
            if (l + e_ < L_) {
              float e;
              if (p.eps_in) {
                e = have ? ev[e_] : p.eps_in[b * L_ + l + e_];
              } else {
                e = ev[e_];
                if (!have) p.eps_out[b * L_ + l + e_] = e;
              }
              mua[e_] = acc[mi][2 * pr][e_] + bm[e_];
              lva[e_] = acc[mi][2 * pr + 1][e_] + bv[e_];
              const float sd = __expf(0.5f * lva[e_]);
              zz[e_] = mua[e_] + e * sd;
              kl += 1.f + lva[e_] - mua[e_] * mua[e_] - sd * sd;
            }
          }
        }
        store_out16((f32x4*)(p.mulv + b * L2p_ + l), f32x4{mua[0], mua[1], mua[2], mua[3]}, p.wt);
        store_out16((f32x4*)(p.mulv + b * L2p_ + Lp_ + l), f32x4{lva[0], lva[1], lva[2], lva[3]}, p.wt);
        const bf16x4 zb = {(bf16_t)zz[0], (bf16_t)zz[1], (bf16_t)zz[2], (bf16_t)zz[3]};
        *(bf16x4*)(p.z + b * Lp_ + l) = zb;
      }
    }
    float* red = (float*)smem_generic;
    const float s_ = block_sum<NW>(kl, red);
    // one KL partial per 1024 elements of the padded [Mp, lat_lp] grid is what the loss reduction sums: this block's
    // BM x BN / 2 elements own BM * BN / 2048 slots -- the sum goes into the first, zeros into the others
    constexpr int KL_SLOTS = BM * BN / 2048;
    if (tid < KL_SLOTS) p.kl_partial[KL_SLOTS * (tile_m * tiles_n + tile_n) + tid] = tid == 0 ? s_ : 0.f;
    return;
  }

  float cs[NP][8];
#pragma unroll
  for (int t = 0; t < NP; ++t)
#pragma unroll
    for (int e = 0; e < 8; ++e) cs[t][e] = 0.f;
  float sq = 0.f;
  float cs2[EPI == EPI_REPARAM_BWD ? NP : 1][8];   // EPI_REPARAM_BWD: column sums of dlogvar (cs: of dmu)
  if constexpr (EPI == EPI_REPARAM_BWD) {
#pragma unroll
    for (int t = 0; t < NP; ++t)
#pragma unroll
      for (int e = 0; e < 8; ++e) cs2[t][e] = 0.f;
  }

  float bias[NP][8];
#pragma unroll
  for (int t = 0; t < NP; ++t)
#pragma unroll
    for (int e = 0; e < 8; ++e) bias[t][e] = 0.f;
  if constexpr (EPI != EPI_MASK_BF16) {
    if (p.bias && (EPI != EPI_F32 || split == 0)) {  // split-K: slab 0 carries the bias
#pragma unroll
      for (int t = 0; t < NP; ++t) {
        const f32x4 lo = *(const f32x4*)(p.bias + colw + 32 * t), hi = *(const f32x4*)(p.bias + colw + 32 * t + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { bias[t][e] = lo[e]; bias[t][4 + e] = hi[e]; }
      }
    }
  }
  const float dq = (FP8 && p.dq) ? *p.dq : 1.f;   // fp8 operands: undo the operand scales on the accumulator
  const float qs = ((EPI == EPI_BIAS_ACT_BF16 || EPI == EPI_TANH_LOSS) && p.out_fp8) ? *p.q_scale : 0.f;
  float amax = 0.f;
  const bool x_al = EPI == EPI_TANH_LOSS && p.x && (p.ld_x & 3) == 0 && ((reinterpret_cast<uintptr_t>(p.x) & 15) == 0);
  // fp16 slabs: the wave tile's exponent (see GemmArgs::out_f16).  max|acc| over the wave (order-independent, so
  // steps stay bit-reproducible) -> biased exponent e of the maximum -> scale 2^(14 - (e - 127)), clamped so that
  // scale and its reciprocal are both normal floats; an infinite / NaN maximum keeps a finite scale and the
  // non-finite values go through the fp16 conversion as they are.
  float f16s = 1.f;
  if constexpr (EPI == EPI_F32) {
    static_assert(WTM % 32 == 0 && WTN % 32 == 0, "fp16 slab exponents are kept per 32 x 32 granule");
    if (p.out_f16) {
      float mx = 0.f;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int r = 0; r < 4; ++r) mx = fmaxf(mx, fabsf(acc[mi][ni][r]));
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
      if constexpr (FP8) mx *= dq;   // the stored values are the accumulators times the (positive) dequantisation factor
      const int e = (int)((__float_as_uint(mx) >> 23) & 0xffu);
      int sb = 268 - e;                       // biased exponent of the scale
      sb = sb < 1 ? 1 : (sb > 253 ? 253 : sb);
      f16s = __uint_as_float((unsigned)sb << 23);
      constexpr int GC = WTN / 32, NG = (WTM / 32) * GC;
      if (lane < NG) {
        const long gr = (m0 + wm * WTM) / 32 + lane / GC, gc = (n0 + wn * WTN) / 32 + lane % GC;
        p.f16_unscale[split * p.us_split_stride + gr * p.us_ld + gc] = __uint_as_float((unsigned)(254 - sb) << 23);
      }
    }
  }

#pragma unroll
  for (int c0 = 0; c0 < MI; c0 += CM) {
    float v[CH][8];
    long rowi[CH], coli[CH];
#pragma unroll
    for (int cm = 0; cm < CM; ++cm)
#pragma unroll
      for (int t = 0; t < NP; ++t) {
        const int it = cm * NP + t;
        f32x4 lo = acc[c0 + cm][2 * t], hi = acc[c0 + cm][2 * t + 1];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          // rows 1 and 3 (16-lane groups) of `lo` trade places with rows 0 and 2 of `hi`
          float a_ = lo[r], b_ = hi[r];
          swap_rows16(a_, b_);
          lo[r] = a_;
          hi[r] = b_;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[it][e] = lo[e]; v[it][4 + e] = hi[e]; }
        if constexpr (FP8) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[it][e] *= dq;
        }
        rowi[it] = roww + 16 * (c0 + cm);
        coli[it] = colw + 32 * t;
      }

    if constexpr (EPI == EPI_BIAS_ACT_BF16) {
      // ReLU as one max against a wave-uniform floor (a per-element `relu ? max : id` costs a compare-select more);
      // fmax(x, NaN) = x, so a NaN floor is the identity, NaN inputs included
      const float floor_ = p.relu ? 0.f : __builtin_nanf("");
#pragma unroll
      for (int it = 0; it < CH; ++it) {
        bf16x8 o;
        float tt[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          tt[e] = __builtin_fmaxf(v[it][e] + bias[it % NP][e], floor_);
          o[e] = (bf16_t)tt[e];
        }
        store_out16((bf16x8*)(p.out_bf16 + rowi[it] * p.ld_bf16 + coli[it]), o, p.wt);
        if (p.out_fp8 || p.amax_part) {   // fp8 forward only
          float q8[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            amax = fmaxf(amax, fabsf(tt[e]));
            q8[e] = p.relu ? fp8_keep_positive(tt[e], tt[e] * qs) : tt[e] * qs;
          }
          if (p.out_fp8) *(unsigned long long*)(p.out_fp8 + rowi[it] * p.ld_fp8 + coli[it]) = pack_fp8x8(q8);
        }
      }
    } else if constexpr (EPI == EPI_F32) {
      float* out = p.out_f32 + split * p.split_stride_f32;
#pragma unroll
      for (int it = 0; it < CH; ++it) {
        f32x4 lo, hi;
#pragma unroll
        for (int e = 0; e < 4; ++e) { lo[e] = v[it][e] + bias[it % NP][e]; hi[e] = v[it][4 + e] + bias[it % NP][4 + e]; }
        if (p.out_f16) {
          typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
          f16x8 h;
#pragma unroll
          for (int e = 0; e < 4; ++e) { h[e] = (_Float16)(lo[e] * f16s); h[4 + e] = (_Float16)(hi[e] * f16s); }
          store_out16((f16x8*)((_Float16*)p.out_f16 + split * p.split_stride_f32 + rowi[it] * p.ld_f32 + coli[it]), h, p.wt);
        } else {
          store_out16((f32x4*)(out + rowi[it] * p.ld_f32 + coli[it]), lo, p.wt);
          store_out16((f32x4*)(out + rowi[it] * p.ld_f32 + coli[it] + 4), hi, p.wt);
        }
      }
    } else if constexpr (EPI == EPI_TANH_LOSS) {
      // target frames: exact [M_valid, N_valid] fp32.  Rows/columns past the valid extent are
      // read from a clamped address and masked by select (no per-element branches).
      float xin[CH][8];
#pragma unroll
      for (int it = 0; it < CH; ++it)
#pragma unroll
        for (int e = 0; e < 8; ++e) xin[it][e] = 0.f;
      if (PF_X && pf_on && c0 == 0) {
#pragma unroll
        for (int it = 0; it < CH; ++it)
#pragma unroll
          for (int e = 0; e < 8; ++e) xin[it][e] = xpf[it][e];
      } else if (p.x) {
#pragma unroll
        for (int it = 0; it < CH; ++it) {
          const long r = rowi[it], col = coli[it];
          const long rc = r < p.M_valid ? r : p.M_valid - 1;
          if (p.x_hop) {   // frame rc of the resident waveform
            const long start = (p.x_idx ? (long)p.x_idx[rc] : p.x_first + rc) * p.x_hop + col;
            if ((p.x_hop & 3) == 0 && ((reinterpret_cast<uintptr_t>(p.x) & 15) == 0) && col + 8 <= p.N_valid && start >= 0 &&
                start + 8 <= p.x_nsamples) {
              const f32x4 lo = *(const f32x4*)(p.x + start);
              const f32x4 hi = *(const f32x4*)(p.x + start + 4);
#pragma unroll
              for (int e = 0; e < 4; ++e) { xin[it][e] = lo[e]; xin[it][4 + e] = hi[e]; }
            } else {
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const long a_ = start + e;
                xin[it][e] = (col + e < p.N_valid && a_ >= 0 && a_ < p.x_nsamples) ? p.x[a_] : 0.f;
              }
            }
          } else if (x_al && col + 8 <= p.N_valid) {
            const f32x4 lo = *(const f32x4*)(p.x + rc * p.ld_x + col);
            const f32x4 hi = *(const f32x4*)(p.x + rc * p.ld_x + col + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { xin[it][e] = lo[e]; xin[it][4 + e] = hi[e]; }
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const long cc = col + e < p.N_valid ? col + e : p.N_valid - 1;
              xin[it][e] = p.x[rc * p.ld_x + cc];
            }
          }
        }
      }
      // a tile that lies wholly inside the valid extent (every tile at C2) skips the per-element validity selects
      const bool interior = m0 + BM <= p.M_valid && n0 + BN <= p.N_valid;
#pragma unroll
      for (int it = 0; it < CH; ++it) {
        const long r = rowi[it], col = coli[it];
        const bool rv_ = r < p.M_valid;
        bf16x8 o;
        float rec[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) rec[e] = fast_tanh(v[it][e] + bias[it % NP][e]);
        float gq[8];   // the same gradient for the fp8 image of dP4 (the fp8 fc4 backward's operand)
        if (p.x && interior) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float d = rec[e] - xin[it][e];
            sq += d * d;
            const float g = p.scale * d * (1.f - rec[e] * rec[e]);
            cs[it % NP][e] += g;
            o[e] = (bf16_t)g;
            gq[e] = g * qs;
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const bool valid = rv_ && col + e < p.N_valid;
            float g = 0.f;
            if (p.x) {
              const float d = valid ? rec[e] - xin[it][e] : 0.f;
              sq += d * d;
              g = p.scale * d * (1.f - rec[e] * rec[e]);
              cs[it % NP][e] += g;
            }
            o[e] = (bf16_t)g;
            gq[e] = g * qs;
          }
        }
        if (p.x && p.out_bf16) store_out16((bf16x8*)(p.out_bf16 + r * p.ld_bf16 + col), o, p.wt);
        if (p.x && p.out_fp8) *(unsigned long long*)(p.out_fp8 + r * p.ld_fp8 + col) = pack_fp8x8(gq);
        if (p.recon && rv_) {
          if ((p.ld_recon & 3) == 0 && col + 8 <= p.N_valid) {
            store_out16((f32x4*)(p.recon + r * p.ld_recon + col), f32x4{rec[0], rec[1], rec[2], rec[3]}, p.wt);
            store_out16((f32x4*)(p.recon + r * p.ld_recon + col + 4), f32x4{rec[4], rec[5], rec[6], rec[7]}, p.wt);
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e)
              if (col + e < p.N_valid) p.recon[r * p.ld_recon + col + e] = rec[e];
          }
        }
      }
    } else if constexpr (EPI == EPI_REPARAM_BWD) {
      // v = dz of (row, 8 consecutive latents): k_reparam_bwd's arithmetic (elementwise.hip) on the accumulators
      const long Lp_ = p.lat_lp, L2p_ = 2 * Lp_, L_ = p.lat_l;
      // (64-row tiles: mu | logvar and, where 16-byte loads were possible, eps of these items were fetched before the main
      // loop; the 256-row tiles of large batches fetch them here, chunk by chunk)
      if constexpr (!PF_RB_ON) {
        const bool al = (L_ & 3) == 0 && (reinterpret_cast<uintptr_t>(p.eps) & 15) == 0;
#pragma unroll
        for (int it = 0; it < CH; ++it) {
          const float* mp = p.mulv + rowi[it] * L2p_ + coli[it];
          rb_m[it][0] = *(const f32x4*)mp; rb_m[it][1] = *(const f32x4*)(mp + 4);
          rb_l[it][0] = *(const f32x4*)(mp + Lp_); rb_l[it][1] = *(const f32x4*)(mp + Lp_ + 4);
          rb_vec[it] = al && rowi[it] < p.M_valid && coli[it] + 8 <= L_;
          rb_e[it][0] = rb_e[it][1] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (rb_vec[it]) {
            rb_e[it][0] = *(const f32x4*)(p.eps + rowi[it] * L_ + coli[it]);
            rb_e[it][1] = *(const f32x4*)(p.eps + rowi[it] * L_ + coli[it] + 4);
          }
        }
      }
#pragma unroll
      for (int it = 0; it < CH; ++it) {
        const long b = rowi[it], l = coli[it];
        bf16x8 om, ov;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float dmu = 0.f, dlv = 0.f;
          if (b < p.M_valid && l + e < L_) {
            const float ee = rb_vec[it] ? rb_e[it][e >> 2][e & 3] : p.eps[b * L_ + l + e];
            const float mu_ = rb_m[it][e >> 2][e & 3], lv_ = rb_l[it][e >> 2][e & 3];
            const float sd = __expf(0.5f * lv_);
            dmu = v[it][e] + p.kl_beta * mu_ * p.inv_nk;
            dlv = v[it][e] * ee * 0.5f * sd + p.kl_beta * 0.5f * (sd * sd - 1.f) * p.inv_nk;
            if (p.dmu_ext) dmu += p.dmu_ext[b * L_ + l + e];
            if (p.dlv_ext) dlv += p.dlv_ext[b * L_ + l + e];
          }
          cs[it % NP][e] += dmu;
          cs2[it % NP][e] += dlv;
          om[e] = (bf16_t)dmu;
          ov[e] = (bf16_t)dlv;
        }
        store_out16((bf16x8*)(p.dmulv + b * L2p_ + l), om, p.wt);
        store_out16((bf16x8*)(p.dmulv + b * L2p_ + Lp_ + l), ov, p.wt);
      }
    } else {  // EPI_MASK_BF16
      {
        // activation > 0 tested on the bf16 bit patterns, two per 32-bit word: the high half is positive iff the word,
        // as a signed integer, exceeds 0xFFFF; the low half iff it is a positive int16 (NaNs never occur in a ReLU output)
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        i32x4 mk[CH];
        if (MASK_LDS && mask_lds) {
          // this chunk's 2 * CM pieces have landed once at most (pieces of later chunks) + (stores of earlier
          // chunks) = WTM / 8 - 2 * CM vector-memory operations of this wave are outstanding (in-order counter)
          // chunk 0 landed inside the main loop.  A later chunk's 2 * CM pieces have landed once at most (pieces of the
          // chunks behind it) + (stores of the chunks before it, 2 * CM each) = WTM / 8 - 2 * CM vector-memory
          // operations of this wave are outstanding (in-order counter)
          if (c0 != 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MASK_LDS ? WTM / 8 - 2 * CM : 0) : "memory");
          const lds_char* mb = c0 == 0 ? mk_lds0 : mk_lds;
#pragma unroll
          for (int it = 0; it < CH; ++it) {
            const int rl = (c0 + it / NP) * 16 + ej, pc = (it % NP) * 4 + (eq & 1) * 2 + (eq >> 1);
            if (mask8) {
              // fp8 image: the item's 8 columns are 8 bytes, half (pc & 1) of 16-byte piece pc >> 1; turned into the bf16
              // test's words: byte b is a positive e4m3 iff it is a positive int8 -- spread to 16 bits it keeps its sign
              // and its "non-zero", which is all the test below reads
              typedef int i32x2_ __attribute__((ext_vector_type(2)));
              const i32x2_ b8 = *(const __attribute__((address_space(3))) i32x2_*)(mb + rl * 128 + (((pc >> 1) ^ (rl & 3)) * 16) + (pc & 1) * 8);
#pragma unroll
              for (int w = 0; w < 4; ++w) {
                const int two = (b8[w >> 1] >> (16 * (w & 1))) & 0xFFFF;                 // bytes 2w, 2w + 1
                const int lo = (int)(signed char)(two & 0xFF), hi = (int)(signed char)(two >> 8);
                mk[it][w] = (lo & 0xFFFF) | (hi << 16);                                  // sign-extended to 16 bits each
              }
            } else {
              mk[it] = *(const __attribute__((address_space(3))) i32x4*)(mb + rl * 128 + ((pc ^ (rl & 7)) * 16));
            }
          }
        } else if (PF_MASK && pf_on && c0 == 0) {
#pragma unroll
          for (int it = 0; it < CH; ++it) mk[it] = mpf[it];
        } else {
#pragma unroll
          for (int it = 0; it < CH; ++it) {
            mk[it] = *(const i32x4*)(p.mask + rowi[it] * p.ld_mask + coli[it]);
          }
        }
#pragma unroll
        for (int it = 0; it < CH; ++it) {
          bf16x8 o;
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            const int word = mk[it][w];
            const float t0 = (short)(word & 0xFFFF) > 0 ? v[it][2 * w] : 0.f;
            const float t1 = word > 0xFFFF ? v[it][2 * w + 1] : 0.f;
            cs[it % NP][2 * w] += t0;
            cs[it % NP][2 * w + 1] += t1;
            o[2 * w] = (bf16_t)t0;
            o[2 * w + 1] = (bf16_t)t1;
          }
          store_out16((bf16x8*)(p.out_bf16 + rowi[it] * p.ld_bf16 + coli[it]), o, p.wt);
        }
      }
    }
  }

  if constexpr (EPI == EPI_BIAS_ACT_BF16) {
    if (p.amax_part) {   // one plain store per block (2048 atomics on one word cost 25 us)
      float m_ = amax;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) m_ = fmaxf(m_, __shfl_xor(m_, o, 64));
      float* red = (float*)smem_generic;
      if (lane == 0) red[wave] = m_;
      __syncthreads();
      if (tid == 0) {
        float b_ = red[0];
        for (int w = 1; w < NW; ++w) b_ = fmaxf(b_, red[w]);
        p.amax_part[bid] = b_;
      }
    }
  }
  if constexpr (EPI == EPI_REPARAM_BWD) {
    // the head biases' gradients: column sums of dmu and dlogvar over the tile's rows -- over the 16 row lanes by a
    // butterfly, over the wave rows through LDS (the ring is free: barrier after the main loop) -- into row 4 tile_m of
    // the [Mp / 16][2 lat_lp] partial-row table (k_reparam_bwd writes one row per 16 batch rows; the three rows this
    // tile does not use are zeroed, so the optimizer's descriptors do not depend on which kernel ran)
    static_assert(BM % 16 == 0, "reparameterisation backward epilogue: BM / 16 rows of the bias partial table per tile");
    float* red = (float*)smem_generic;
#pragma unroll
    for (int t = 0; t < NP; ++t)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float a_ = cs[t][e], c_ = cs2[t][e];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { a_ += __shfl_xor(a_, o, 64); c_ += __shfl_xor(c_, o, 64); }
        cs[t][e] = a_; cs2[t][e] = c_;
      }
    if (ej == 0) {
#pragma unroll
      for (int t = 0; t < NP; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int c_ = wn * WTN + (eq & 1) * 16 + (eq >> 1) * 8 + 32 * t + e;
          red[wm * BN + c_] = cs[t][e];
          red[(WGM + wm) * BN + c_] = cs2[t][e];
        }
    }
    __syncthreads();
    if (p.dbh_partial) {
      const long L2p_ = 2 * p.lat_lp;
      for (int i = tid; i < (BM / 16) * 2 * BN; i += 64 * NW) {   // BM / 16 table rows x {dmu, dlogvar} x BN columns
        const int row = i / (2 * BN), hd = (i / BN) & 1, c_ = i % BN;
        float s_ = 0.f;
        if (row == 0) {
#pragma unroll
          for (int w = 0; w < WGM; ++w) s_ += red[(hd * WGM + w) * BN + c_];
        }
        p.dbh_partial[((long)tile_m * (BM / 16) + row) * L2p_ + hd * p.lat_lp + n0 + c_] = s_;
      }
    }
  }
  if constexpr (EPI == EPI_TANH_LOSS || EPI == EPI_MASK_BF16) {
    if (p.colsum) {
      // lanes with equal (lane >> 4) own the same columns: butterfly over the 16 row lanes, then across
      // the block's wave rows through LDS (the ring is no longer read: barrier after the main loop)
      float* red = (float*)smem_generic;
      if constexpr (MASK_LDS) __syncthreads();   // `red` overlays wave 0's mask rows: every wave is done reading its own
#pragma unroll
      for (int t = 0; t < NP; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float s_ = cs[t][e];
#pragma unroll
          for (int o = 1; o < 16; o <<= 1) s_ += __shfl_xor(s_, o, 64);
          cs[t][e] = s_;
        }
      if (ej == 0) {
#pragma unroll
        for (int t = 0; t < NP; ++t)
#pragma unroll
          for (int e = 0; e < 8; ++e)
            red[wm * BN + wn * WTN + (eq & 1) * 16 + (eq >> 1) * 8 + 32 * t + e] = cs[t][e];
      }
      __syncthreads();
      if (tid < BN) {
        float s_ = 0.f;
#pragma unroll
        for (int w = 0; w < WGM; ++w) s_ += red[w * BN + tid];
        p.colsum[(long)tile_m * tiles_n * BN + n0 + tid] = s_;
      }
    }
    if constexpr (EPI == EPI_TANH_LOSS) {
      if (p.blocksum) {
        __syncthreads();
        float* red = (float*)smem_generic;
        const float s_ = block_sum<NW>(sq, red);
        if (tid == 0) p.blocksum[tile_m * tiles_n + tile_n] = s_;
      }
    }
  }
}

template <int BM, int BN, int WGM, int WGN, bool A_KMAJ, bool B_KMAJ, int EPI, int NSTAGE, bool FP8 = false>
__global__ void __launch_bounds__(64 * WGM * WGN) gemm_bf16_kernel(const GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem_dyn[];
  gemm_body<BM, BN, WGM, WGN, A_KMAJ, B_KMAJ, EPI, NSTAGE, FP8>(p, blockIdx.x, smem_dyn);
}

// A workgroup per CU walking a LIST of 256 x 256 output tiles (large batches: 16-32 tiles per CU).  Between two tiles of a
// list the ping-pong loop never stops staging: the last K tile of tile i already pulls the first K tile (and a quarter) of
// tile i + 1 into the ring, so tile i's epilogue runs while that LDS-DMA lands and tile i + 1 starts without a launch, a
// cold first load or a ring fill (mainloop_pingpong, `Ag_next`).  The epilogues that run here keep out of LDS.  Virtual block
// ids b, b + G, b + 2 G .. keep the XCD of the real block (G is a multiple of 8), so the XCD-aware item order holds.
template <bool A_KMAJ, bool B_KMAJ, int EPI>
__global__ void __launch_bounds__(512) gemm_pp_persist_kernel(const GemmArgs p) {
  static_assert(EPI == EPI_BIAS_ACT_BF16, "tile lists: epilogues that do not use LDS");
  extern __shared__ __attribute__((aligned(16))) char smem_dyn[];
  const int nwg = p.tiles_m * p.tiles_n * p.splits, G = (int)gridDim.x;
  bool resumed = false;
  for (int vb = (int)blockIdx.x; vb < nwg; vb += G) {
    gemm_body<256, 256, 2, 4, A_KMAJ, B_KMAJ, EPI, 8>(p, vb, smem_dyn, vb + G < nwg ? vb + G : -1, resumed);
    resumed = true;
  }
}

// Two independent GEMMs in ONE launch (blocks [0, n_first) run the first): neither of the
// paired problems has enough 256x256 output tiles to fill 256 CUs on its own, together they do.
// Used for the backward of a Linear layer: dX = relu'(dY W) (NN) and dW = dY^T X (TN).
template <int BM, int BN, int WGM, int WGN, int NSTAGE, bool FP8 = false>
__global__ void __launch_bounds__(64 * WGM * WGN)
gemm_dgrad_wgrad_kernel(const GemmArgs dgrad, const GemmArgs wgrad, const int n_first) {
  extern __shared__ __attribute__((aligned(16))) char smem_dyn[];
  if ((int)blockIdx.x < n_first)
    gemm_body<BM, BN, WGM, WGN, true, false, EPI_MASK_BF16, NSTAGE, FP8>(dgrad, blockIdx.x, smem_dyn);
  else
    gemm_body<BM, BN, WGM, WGN, false, false, EPI_F32, NSTAGE, FP8>(wgrad, blockIdx.x - n_first, smem_dyn);
}

// Two independent GEMMs of the same block-tile configuration in ONE launch (blocks [0, n_first) run
// the first): the latent-sized backward GEMMs are a few microseconds each, mostly launch, first-tile
// latency and store tail, and come in pairs that read the same activation (dz and dW3 read dP3; the
// heads' dgrad and wgrad read dmulv and h1) -- one grid runs both pairs' blocks side by side.
template <int BM, int BN, int WGM, int WGN, int NSTAGE, bool A1, bool B1, int E1, bool A2, bool B2, int E2>
__global__ void __launch_bounds__(64 * WGM * WGN)
gemm_dual_kernel(const GemmArgs first, const GemmArgs second, const int n_first) {
  extern __shared__ __attribute__((aligned(16))) char smem_dyn[];
  if ((int)blockIdx.x < n_first)
    gemm_body<BM, BN, WGM, WGN, A1, B1, E1, NSTAGE>(first, blockIdx.x, smem_dyn);
  else
    gemm_body<BM, BN, WGM, WGN, A2, B2, E2, NSTAGE>(second, blockIdx.x - n_first, smem_dyn);
}

}  // namespace rv
