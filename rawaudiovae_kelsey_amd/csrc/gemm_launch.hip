// Host-side launchers for the bf16 MFMA GEMM family (C ABI: include/rawvae_hip.h).
#include <hip/hip_ext.h>
#include "gemm_bf16.h"
#include "adam.h"
#include "../../include/rawvae_hip.h"
#include "internal.h"

using namespace rv;

namespace rv {
thread_local int rv_store_wt = 0;   // see common.h; set by plan.hip around the training step's launches
}

namespace {

// Tile configurations (block tile, wave grid, LDS ring depth):
//   0:  64x64  2x2 waves of 32x32, 4 stages ( 64 KiB)  latent-sized extents
//   1: 128x128 2x2 waves of 64x64, 4 stages (128 KiB)
//   2: 256x128 4x2 waves of 64x64, 3 stages (144 KiB)  two waves per SIMD
//   3: 256x128 2x2 waves of 128x64, 3 stages (144 KiB) one wave per SIMD, half the LDS reads
//   4: 128x128 2x4 waves of 64x32, 4 stages (128 KiB)  two waves per SIMD; default 128x128
//      (measured 4-10 % faster than config 1 on every 128-tile GEMM of the step)
//   5: 256x256 2x4 waves of 128x64, 2 stages (128 KiB ring); only
//      picked by the paired dgrad+wgrad launch (no single GEMM of the step has 256 such tiles)
int g_force_tile = -1;   // test hook (include/rawvae_hip_diag.h): pin one tile configuration, -1 = the picker's choice
int g_tile_lists = 1;    // test hook: 256 x 256 forward GEMMs with many tiles as tile lists (launch_tile, case 7)

template <int BM, int BN, int WGM, int WGN, int NSTAGE, bool AK, bool BK, int EPI, bool FP8 = false>
int launch(const GemmArgs& a, long Mp, long Np, int splits, hipStream_t st) {
  constexpr int RING = NSTAGE == 8 ? 2 : NSTAGE;  // NSTAGE 8 = the ping-pong main loop on 2 buffers
  // the ping-pong dgrad keeps its first ReLU-mask chunk behind the ring (gemm_bf16.h MASK_LDS): 4 KiB per wave
  constexpr int mask_extra = (NSTAGE == 8 && EPI == EPI_MASK_BF16) ? 8 * 4096 : 0;
  constexpr int smem_max = RING * (BM + BN) * 128 + mask_extra;
  // short K loops never refill the ring: allocate only the slots they stage (but at least the
  // epilogue's staging area) so several blocks fit on a CU
  constexpr int stage_bytes = (BM + BN) * 128;
  constexpr int epi_bytes = WGM * BN * 4 + 256;  // column-sum / block-sum reductions of the epilogue
  const int used = (NSTAGE >= 8 ? RING : (a.k_tiles < RING ? a.k_tiles : RING)) * stage_bytes + mask_extra;
  const int smem = used > epi_bytes ? used : epi_bytes;
  auto kern = gemm_bf16_kernel<BM, BN, WGM, WGN, AK, BK, EPI, NSTAGE, FP8>;
  static std::atomic<unsigned long long> attr_done{0};
  lds_opt_in((const void*)kern, smem_max, attr_done);
  GemmArgs g = a;
  g.wt = rv_store_wt;
  g.tiles_m = (int)(Mp / BM);
  g.tiles_n = (int)(Np / BN);
  g.splits = splits;
  dim3 grid((unsigned)(g.tiles_m * g.tiles_n * splits), 1, 1);
  hipLaunchKernelGGL(kern, grid, dim3(64 * WGM * WGN), smem, st, g);
  RV_CHECK_LAUNCH();
  return RV_OK;
}

// The 256 x 256 ping-pong forward GEMM as tile lists (gemm_bf16.h gemm_pp_persist_kernel): one workgroup per CU
template <bool AK, bool BK, int EPI>
int launch_persist(const GemmArgs& a, long Mp, long Np, hipStream_t st) {
  constexpr int smem = 2 * (256 + 256) * 128;
  auto kern = gemm_pp_persist_kernel<AK, BK, EPI>;
  static std::atomic<unsigned long long> attr_done{0};
  lds_opt_in((const void*)kern, smem, attr_done);
  GemmArgs g = a;
  g.wt = rv_store_wt;
  g.tiles_m = (int)(Mp / 256);
  g.tiles_n = (int)(Np / 256);
  g.splits = 1;
  hipLaunchKernelGGL(kern, dim3(256), dim3(512), smem, st, g);
  RV_CHECK_LAUNCH();
  return RV_OK;
}

// Smallest power-of-two split count that yields `target` blocks (each split costs a partial slab
// that a later kernel re-reads, so no more than needed).
int splits_for(long tiles, long k_tiles, int max_splits, long target = 256) {
  int s = 1;
  while (tiles * s < target && 2 * s <= max_splits && k_tiles % (2 * s) == 0 && k_tiles / (2 * s) >= 2) s *= 2;
  return s;
}

bool tile_fits(int tile, long Mp, long Np) {
  switch (tile) {
    case 0: return Mp % 64 == 0 && Np % 64 == 0;
    case 1: return Mp % 128 == 0 && Np % 128 == 0;
    case 2: case 3: return Mp % 256 == 0 && Np % 128 == 0;
    case 4: return Mp % 128 == 0 && Np % 128 == 0;
    case 5: case 7: return Mp % 256 == 0 && Np % 256 == 0;
    default: return false;
  }
}

void tile_dims(int tile, int* bm, int* bn) {
  *bm = tile == 0 ? 64 : (tile == 1 || tile == 4) ? 128 : 256;
  *bn = tile == 0 ? 64 : (tile == 5 || tile == 7) ? 256 : 128;
}

template <bool AK, bool BK, int EPI>
int launch_tile(int tile, const GemmArgs& a, long Mp, long Np, long Kp, int splits, hipStream_t st) {
  RV_REQUIRE(Mp > 0 && Np > 0 && Kp > 0 && Kp % 64 == 0, RV_ERR_SHAPE, "gemm: bad extents %ld %ld %ld", Mp, Np, Kp);
  RV_REQUIRE(tile_fits(tile, Mp, Np), RV_ERR_SHAPE, "gemm: tile %d does not divide %ld x %ld", tile, Mp, Np);
  RV_REQUIRE(splits >= 1 && (Kp / 64) % splits == 0, RV_ERR_SHAPE,
             "gemm: K tiles %ld not divisible by splits %d", Kp / 64, splits);
  RV_REQUIRE(a.lda % 8 == 0 && a.ldb % 8 == 0, RV_ERR_SHAPE, "gemm: leading dims must be multiples of 8");
  RV_REQUIRE((((uintptr_t)a.A | (uintptr_t)a.B) & 15) == 0, RV_ERR_SHAPE, "gemm: operands must be 16-byte aligned");
  switch (tile) {
    case 0: return launch<64, 64, 2, 2, 4, AK, BK, EPI>(a, Mp, Np, splits, st);
    case 1: return launch<128, 128, 2, 2, 4, AK, BK, EPI>(a, Mp, Np, splits, st);
    case 2: return launch<256, 128, 4, 2, 3, AK, BK, EPI>(a, Mp, Np, splits, st);
    case 4: return launch<128, 128, 2, 4, 4, AK, BK, EPI>(a, Mp, Np, splits, st);
    case 5: return launch<256, 256, 2, 4, 2, AK, BK, EPI>(a, Mp, Np, splits, st);
    case 7:
      if ((Kp / 64 / splits) % 2) return launch<256, 256, 2, 4, 2, AK, BK, EPI>(a, Mp, Np, splits, st);
      if constexpr (EPI == EPI_BIAS_ACT_BF16) {
        // more than two tiles per CU: tile lists (the fp8 forward's block maxima go through LDS: not there)
        // (default.ini's shape, one box: 4592 -> 4546 us per step for fc1's and fc3's forward together, 741 -> 696 us)
        if (g_tile_lists && splits == 1 && !a.amax_part && !a.a_hop && (Mp / 256) * (Np / 256) >= 512) return launch_persist<AK, BK, EPI>(a, Mp, Np, st);
      }
      return launch<256, 256, 2, 4, 8, AK, BK, EPI>(a, Mp, Np, splits, st);
    default: return launch<256, 128, 2, 2, 3, AK, BK, EPI>(a, Mp, Np, splits, st);
  }
}

// A weight-gradient GEMM whose launch also carries optimizer work: blocks [0, n_gemm) run the 256x256 TN GEMM
// (dW = dY^T X as split-K slabs); the blocks behind them walk the fused Adam update of OTHER tensors, whose
// gradients earlier launches have completed, in a strided loop over the 256-thread "virtual blocks" of adam.h.
// Why one launch: dW1 is the last GEMM of the backward and runs alone on the chip (32 output tiles x 4 K splits
// = 128 blocks of this tile), while Adam is pure HBM streaming that the GEMM's MFMA loop leaves idle.  Every
// block of a launch gets the same dynamic LDS, so an optimizer block cannot share a CU with a GEMM block: the
// optimizer blocks take the CUs the GEMM does not fill.  (Two streams instead cost a 6 us bubble per event
// record on this runtime and slowed the co-running GEMMs by more than was hidden: profiles/r02_sched2_*.)
template <int NSTAGE, bool FP8 = false>
__global__ void __launch_bounds__(512)
gemm_wgrad_adam_kernel(const GemmArgs wgrad, const int n_gemm, const DescTable tab, float* __restrict__ param,
                       float* __restrict__ m_arena, float* __restrict__ v_arena, const float lr,
                       const float grad_scale, const long long* __restrict__ step_counter, const int stream_mode,
                       float* __restrict__ fin_f32, bf16_t* __restrict__ fin_bf16, const int tail_vb) {
  extern __shared__ __attribute__((aligned(16))) char smem_dyn[];
  if ((int)blockIdx.x < n_gemm) {
    gemm_body<256, 256, 2, 4, false, false, EPI_F32, NSTAGE, FP8>(wgrad, blockIdx.x, smem_dyn);
    if (tail_vb > 0) {
      // The LAST `tail_vb` virtual blocks of the riders' table are left to the GEMM blocks: when its tile is done, every
      // wave takes quarters of them (a virtual block is 256 threads that do not talk to each other).  A static share, sized
      // by the host so that both kinds of block finish together -- where the GEMM is the shorter half of the launch (fp8
      // operands) its CUs would otherwise idle while the riders stream on at their per-CU rate.
      const long total = tab.blk_start[tab.n];
      const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = (int)(threadIdx.x & 63);
      for (long u = (long)blockIdx.x * 8 + wave; u < 4L * tail_vb; u += 8L * n_gemm)
        adam_block<true>(tab, total - tail_vb + (u >> 2), (int)(u & 3) * 64 + lane, param, m_arena, v_arena, nullptr, lr,
                         grad_scale, step_counter, nullptr, nullptr);
    }
  } else {
    if (fin_f32 || fin_bf16) {
      // riders that only SUM the slabs of the table's tensors into a flat gradient payload (rv_grad_finalize's work;
      // rv_linear_wgrad_finalize): the data-parallel step's second bucket minus the gradient this GEMM produces
      const long total = tab.blk_start[tab.n];
      const int half = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
      for (long vb = 2L * ((long)blockIdx.x - n_gemm) + half; vb < total; vb += 2L * ((long)gridDim.x - n_gemm))
        adam_block<false>(tab, vb, (int)(threadIdx.x & 255), nullptr, nullptr, nullptr, fin_f32, 0.f, 1.f, nullptr, fin_bf16, nullptr);
      return;
    }
    if (stream_mode) {
      // each wave streams its own chunks through a private two-slot LDS ring (adam_stream)
      const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
      const long n_waves = 8L * ((long)gridDim.x - n_gemm);
      adam_stream(tab, 8L * ((long)blockIdx.x - n_gemm) + wave, n_waves, (lds_char*)smem_dyn + wave * 2 * AS_SLOT,
                  (int)(threadIdx.x & 63), param, m_arena, v_arena, lr, grad_scale, step_counter);
      return;
    }
    const long total = tab.blk_start[tab.n] - tail_vb;     // (the rest is the GEMM blocks' share, above)
    const long stride = 2L * ((long)gridDim.x - n_gemm);   // virtual blocks taken per sweep of the optimizer blocks
    const int half = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));   // wave-uniform
    constexpr int U = 2;   // virtual blocks in flight per thread (4 measured 0.6 us slower: more registers, same bytes/s)
    for (long vb = 2L * ((long)blockIdx.x - n_gemm) + half; vb < total; vb += U * stride)
      adam_group<U>(tab, vb, stride, (int)(threadIdx.x & 255), param, m_arena, v_arena, lr, grad_scale, step_counter, total);
  }
}

// fp8 (e4m3) forward GEMMs: operands are fp8 bytes viewed as bf16 pairs, `Kp2` = K / 2 in such pairs (so a staged
// 64-pair tile holds 128 fp8 values per row and every tile / swizzle / ring rule of the bf16 kernels carries over).
template <int EPI>
int launch_tile_fp8(int tile, const GemmArgs& a, long Mp, long Np, long Kp2, hipStream_t st) {
  RV_REQUIRE(Mp > 0 && Np > 0 && Kp2 > 0 && Kp2 % 64 == 0, RV_ERR_SHAPE, "fp8 gemm: K must be a multiple of 128 (got %ld)", 2 * Kp2);
  RV_REQUIRE(tile_fits(tile, Mp, Np), RV_ERR_SHAPE, "fp8 gemm: tile %d does not divide %ld x %ld", tile, Mp, Np);
  RV_REQUIRE(a.lda % 8 == 0 && a.ldb % 8 == 0, RV_ERR_SHAPE, "fp8 gemm: leading dims must be multiples of 16 bytes");
  RV_REQUIRE((((uintptr_t)a.A | (uintptr_t)a.B) & 15) == 0, RV_ERR_SHAPE, "fp8 gemm: operands must be 16-byte aligned");
  switch (tile) {
    case 0: return launch<64, 64, 2, 2, 4, true, true, EPI, true>(a, Mp, Np, 1, st);
    case 2: return launch<256, 128, 4, 2, 3, true, true, EPI, true>(a, Mp, Np, 1, st);
    case 5: case 7: return launch<256, 256, 2, 4, 2, true, true, EPI, true>(a, Mp, Np, 1, st);   // (large batches: choose_tile)
    default: return launch<128, 128, 2, 4, 4, true, true, EPI, true>(a, Mp, Np, 1, st);
  }
}

// Split-K slab destination of a weight-gradient GEMM [Mp, Np]: fp32 (dtype 0), or fp16 with one power-of-two scale
// per wave tile and slab (dtype 1, GemmArgs::out_f16): `unscale` then receives the factors that undo the scales,
// [splits][Mp / 32][Np / 32] floats.  Same element strides either way.
int set_slabs(GemmArgs& g, void* dw, long lddw, long split_stride, int dtype, float* unscale, long Mp, long Np) {
  RV_REQUIRE(dtype == RV_SLAB_F32 || dtype == RV_SLAB_F16, RV_ERR_UNSUPPORTED, "weight gradient: slab dtype %d", dtype);
  RV_REQUIRE(dtype == RV_SLAB_F32 || unscale, RV_ERR_NULL, "weight gradient: fp16 slabs need the table of per-tile scales");
  g.ld_f32 = lddw; g.split_stride_f32 = split_stride;
  g.out_f32 = (float*)dw; g.out_f16 = nullptr; g.f16_unscale = nullptr;
  if (dtype == RV_SLAB_F16) {
    g.out_f16 = dw; g.f16_unscale = unscale;
    g.us_ld = Np / 32; g.us_split_stride = (Mp / 32) * (Np / 32);
  }
  return RV_OK;
}

int g_pair_loop = 8;  // main loop of the paired 256x256 kernel: 8 = ping-pong (default), 2 = two-slot ring

// One-shot: the next paired launch signals this event on completion (rv_pair_stop_event; the data-parallel step's first
// fork).  Thread-local: a plan is driven from one host thread at a time.
static thread_local hipEvent_t g_pair_stop_event = nullptr;
template <int NSTAGE, bool FP8 = false>
int launch_pair(const GemmArgs& d_in, const GemmArgs& g_in, hipStream_t st) {
  constexpr int BM = 256, BN = 256, WGM = 2, WGN = 4;
  GemmArgs d = d_in, g = g_in;
  d.wt = g.wt = rv_store_wt;
  const int n_d = d.tiles_m * d.tiles_n, n_w = g.tiles_m * g.tiles_n * g.splits;
  constexpr int smem = 2 * (BM + BN) * 128 + 8 * 4096;  // the ring (the epilogue's reductions reuse its first bytes) + the
                                                         // first ReLU-mask chunk of the dgrad blocks (gemm_bf16.h MASK_LDS)
  auto kern = gemm_dgrad_wgrad_kernel<BM, BN, WGM, WGN, NSTAGE, FP8>;
  static std::atomic<unsigned long long> attr_done{0};
  lds_opt_in((const void*)kern, smem, attr_done);
  if (g_pair_stop_event) {
    // the launch's own completion signal as a HIP event: 3.7 us of bubble behind this kernel on its stream instead of the
    // 5.7 a hipEventRecord behind it costs, and 7 us instead of 11 to the dependent kernel on the other stream
    // (tools/probe_extlaunch.hip)
    hipExtLaunchKernelGGL(kern, dim3((unsigned)(n_d + n_w)), dim3(64 * WGM * WGN), smem, st, nullptr, g_pair_stop_event, 0, d, g, n_d);
    g_pair_stop_event = nullptr;
  } else {
    hipLaunchKernelGGL(kern, dim3((unsigned)(n_d + n_w)), dim3(64 * WGM * WGN), smem, st, d, g, n_d);
  }
  RV_CHECK_LAUNCH();
  return RV_OK;
}

// Two GEMMs of one tile configuration in one launch (gemm_dual_kernel).  `a`/`b` carry operands,
// k_tiles and outputs; tiles/splits are filled in here.
template <int BM, int BN, int WGM, int WGN, int NSTAGE, bool A1, bool B1, int E1, bool A2, bool B2, int E2>
int launch_dual(const GemmArgs& a, long Mp1, long Np1, int splits1, const GemmArgs& b, long Mp2, long Np2, int splits2,
                hipStream_t st) {
  constexpr int stage_bytes = (BM + BN) * 128, smem_max = NSTAGE * stage_bytes;
  constexpr int epi_bytes = WGM * BN * 4 + 256;
  const int kt = a.k_tiles > b.k_tiles ? a.k_tiles : b.k_tiles;
  const int used = (kt < NSTAGE ? kt : NSTAGE) * stage_bytes;
  const int smem = used > epi_bytes ? used : epi_bytes;
  auto kern = gemm_dual_kernel<BM, BN, WGM, WGN, NSTAGE, A1, B1, E1, A2, B2, E2>;
  static std::atomic<unsigned long long> attr_done{0};
  lds_opt_in((const void*)kern, smem_max, attr_done);
  GemmArgs g1 = a, g2 = b;
  g1.wt = g2.wt = rv_store_wt;
  g1.tiles_m = (int)(Mp1 / BM); g1.tiles_n = (int)(Np1 / BN); g1.splits = splits1;
  g2.tiles_m = (int)(Mp2 / BM); g2.tiles_n = (int)(Np2 / BN); g2.splits = splits2;
  const int n1 = g1.tiles_m * g1.tiles_n * splits1, n2 = g2.tiles_m * g2.tiles_n * splits2;
  hipLaunchKernelGGL(kern, dim3((unsigned)(n1 + n2)), dim3(64 * WGM * WGN), smem, st, g1, g2, n1);
  RV_CHECK_LAUNCH();
  return RV_OK;
}

// Dual launch on tile 0 (64x64) or tile 4 (128x128, 8 waves); false if `tile` is neither.
template <bool A1, bool B1, int E1, bool A2, bool B2, int E2>
bool try_dual(int tile, const GemmArgs& a, long Mp1, long Np1, int s1, const GemmArgs& b, long Mp2, long Np2, int s2,
              hipStream_t st, int* rc) {
  // dual launches run on two ring slots: more co-resident blocks hide the latency of these short GEMMs
  // (measured ~1 us/step better than four slots for the 64x64 pair; the opposite holds for a lone 64x64 GEMM)
  if (tile == 0) { *rc = launch_dual<64, 64, 2, 2, 2, A1, B1, E1, A2, B2, E2>(a, Mp1, Np1, s1, b, Mp2, Np2, s2, st); return true; }
  // two ring slots (64 KiB) so that two blocks share a CU: these GEMMs are short and latency-bound
  if (tile == 4) { *rc = launch_dual<128, 128, 2, 4, 2, A1, B1, E1, A2, B2, E2>(a, Mp1, Np1, s1, b, Mp2, Np2, s2, st); return true; }
  return false;
}

}  // namespace

// Tile used for a GEMM launched with a given split count (deterministic: callers size their
// partial-sum buffers from it).
// Large batches (the reference's default.ini trains at batch_size = 131072, default.ini:27): once a GEMM has at least two
// full rounds of 256 x 256 tiles -- or, for a split-K weight gradient, 2048-deep K slices -- the 256 x 256 ping-pong loop
// is the faster form (measured at B = 131072, profiles/r06_big_batch_gemms.txt: fc1 forward 0.39 of the MFMA peak against
// 0.35 on 256 x 128, fc4's dgrad 0.35 against 0.25, the weight gradients 0.49-0.52 against 0.30).  `Kp` = 0: unknown (the
// plan's queries concern unsplit forward GEMMs, whose choice does not depend on it).
static bool big_tiles(long Mp, long Np, int splits, long Kp) {
  if (!tile_fits(7, Mp, Np)) return false;
  const long t = (Mp / 256) * (Np / 256);
  if (splits == 1) return t >= 512;
  return Kp > 0 && Kp / splits >= 2048 && t * splits >= 192;
}

static int choose_tile(long Mp, long Np, int splits, long Kp = 0, bool allow_big = true) {
  if (g_force_tile >= 0 && tile_fits(g_force_tile, Mp, Np)) return g_force_tile;
  if (allow_big && big_tiles(Mp, Np, splits, Kp)) return 7;
  // skinny outputs (the heads: N = 2 Lp = 128): 64x64 tiles, 64 KiB of LDS, two blocks per CU
  // (measured 7.4 us vs 8.4-9.0 on 128x128 for 4096x128x2048)
  if (Np <= 128) return 0;
  // 256x128 only when it fills the chip without slicing K finely (small-N GEMMs do better on 128x128)
  if (tile_fits(2, Mp, Np) && splits <= 4 && (Mp / 256) * (Np / 128) * splits >= 192) return 2;
  return tile_fits(4, Mp, Np) ? 4 : 0;
}

// (the fused loss forward on fp8 operands keeps the tiles below 256 x 256: its epilogue does not fit the registers there)
extern "C" int rv_gemm_tile_fp8_loss(long Mp, long Np, int* bm, int* bn) {
  int m, n;
  tile_dims(choose_tile(Mp, Np, 1, 0, false), &m, &n);
  if (bm) *bm = m;
  if (bn) *bn = n;
  return RV_OK;
}

extern "C" int rv_gemm_tile(long Mp, long Np, int splits, int* bm, int* bn) {
  RV_REQUIRE(Mp > 0 && Np > 0 && Mp % 64 == 0 && Np % 64 == 0 && splits >= 1, RV_ERR_SHAPE,
             "rv_gemm_tile: extents must be positive multiples of 64 (got %ld %ld)", Mp, Np);
  int m, n;
  tile_dims(choose_tile(Mp, Np, splits), &m, &n);
  if (bm) *bm = m;
  if (bn) *bn = n;
  return RV_OK;
}

extern "C" int rv_gemm_pick(long Mp, long Np, long Kp, int max_splits, int* bm, int* bn, int* splits) {
  RV_REQUIRE(Mp > 0 && Np > 0 && Kp > 0 && Mp % 64 == 0 && Np % 64 == 0 && Kp % 64 == 0, RV_ERR_SHAPE,
             "rv_gemm_pick: extents must be positive multiples of 64 (got %ld %ld %ld)", Mp, Np, Kp);
  if (max_splits < 1) max_splits = 1;
  const long kt = Kp / 64;
  int t = -1;
  if (g_force_tile >= 0 && tile_fits(g_force_tile, Mp, Np)) {
    t = g_force_tile;
  } else if (tile_fits(7, Mp, Np) && Kp >= 16384 && (Mp / 256) * (Np / 256) <= 256) {
    // a weight gradient over a large batch: 256 x 256 tiles, as many K splits as fill the chip once (see big_tiles)
    const int s7 = splits_for((Mp / 256) * (Np / 256), kt, max_splits);
    if (big_tiles(Mp, Np, s7, Kp)) t = 7;
  }
  if (t >= 0) {
  } else if (Np <= 128) {
    t = 0;
  } else if (tile_fits(2, Mp, Np)) {
    const long tl = (Mp / 256) * (Np / 128);
    const int sp = splits_for(tl, kt, max_splits);
    if (sp <= 4 && tl * sp >= 192) t = 2;
  }
  if (t < 0) t = tile_fits(4, Mp, Np) ? 4 : 0;
  int m, n;
  tile_dims(t, &m, &n);
  // 128x128 8-wave blocks saturate at ~128 blocks for the short K loops they get here (measured:
  // head GEMM 8.4 us at 4 splits vs 8.9 at 8; head wgrad 9.4 us at 8 and at 16 splits)
  const int s = splits_for((Mp / m) * (Np / n), kt, max_splits, t == 4 ? 128 : 256);
  if (bm) *bm = m;
  if (bn) *bn = n;
  if (splits) *splits = s;
  return RV_OK;
}

// Test hook (include/rawvae_hip_diag.h, not part of the product ABI): pin the block tile every later launch uses
// (-1: the picker's choice), or -- 102 / 108 -- the main loop of the paired 256x256 launch (two-slot ring / ping-pong).
extern "C" int rv_gemm_force_tile(int tile) {
  if (tile == 102 || tile == 108) { g_pair_loop = tile - 100; return RV_OK; }
  if (tile == 110 || tile == 111) { g_tile_lists = tile - 110; return RV_OK; }
  g_force_tile = tile;
  return RV_OK;
}

template <bool AK, bool BK, int EPI>
static int launch_auto(const GemmArgs& a, long Mp, long Np, long Kp, int splits, hipStream_t st) {
  RV_REQUIRE(Mp > 0 && Np > 0 && Mp % 64 == 0 && Np % 64 == 0 && splits >= 1, RV_ERR_SHAPE,
             "gemm: extents must be positive multiples of 64 (got %ld %ld)", Mp, Np);
  int tile = choose_tile(Mp, Np, splits, Kp);
  // (the fused loss epilogue holds 16 more values per item than the others: on 256 x 256 tiles the two-slot ring is its
  // faster loop -- 0.375 against 0.345 of the MFMA peak at B = 131072)
  if (EPI == EPI_TANH_LOSS && tile == 7 && g_force_tile < 0) tile = 5;
  return launch_tile<AK, BK, EPI>(tile, a, Mp, Np, Kp, splits, st);
}

extern "C" {

int rv_linear_fwd(const void* x, long ldx, const void* w, long ldw, const float* bias, long Mp,
                  long Np, long Kp, int act, void* y, long ldy, void* stream) {
  RV_REQUIRE(x && w && y, RV_ERR_NULL, "rv_linear_fwd: null operand");
  RV_REQUIRE(act == RV_ACT_NONE || act == RV_ACT_RELU, RV_ERR_UNSUPPORTED, "rv_linear_fwd: act %d", act);
  GemmArgs a{};
  a.A = (const bf16_t*)x; a.lda = ldx; a.B = (const bf16_t*)w; a.ldb = ldw;
  a.k_tiles = (int)(Kp / 64); a.M_valid = (int)Mp; a.N_valid = (int)Np;
  a.relu = act == RV_ACT_RELU; a.bias = bias; a.out_bf16 = (bf16_t*)y; a.ld_bf16 = ldy;
  return launch_auto<true, true, EPI_BIAS_ACT_BF16>(a, Mp, Np, Kp, 1, (hipStream_t)stream);
}

int rv_linear_fwd_ex(const void* x, long ldx, const void* w, long ldw, const float* bias, long Mp, long Np, long Kp,
                     int act, void* y, long ldy, void* y_fp8, long ldy_fp8, const float* q_scale, float* amax_part,
                     void* stream) {
  RV_REQUIRE(x && w && y, RV_ERR_NULL, "rv_linear_fwd_ex: null operand");
  RV_REQUIRE(act == RV_ACT_NONE || act == RV_ACT_RELU, RV_ERR_UNSUPPORTED, "rv_linear_fwd_ex: act %d", act);
  RV_REQUIRE(!y_fp8 || (q_scale && ldy_fp8 % 8 == 0 && ((uintptr_t)y_fp8 & 7) == 0), RV_ERR_SHAPE,
             "rv_linear_fwd_ex: fp8 output needs a scale and 8-byte aligned rows");
  GemmArgs a{};
  a.A = (const bf16_t*)x; a.lda = ldx; a.B = (const bf16_t*)w; a.ldb = ldw;
  a.k_tiles = (int)(Kp / 64); a.M_valid = (int)Mp; a.N_valid = (int)Np;
  a.relu = act == RV_ACT_RELU; a.bias = bias; a.out_bf16 = (bf16_t*)y; a.ld_bf16 = ldy;
  a.out_fp8 = (unsigned char*)y_fp8; a.ld_fp8 = ldy_fp8; a.q_scale = q_scale; a.amax_part = amax_part;
  return launch_auto<true, true, EPI_BIAS_ACT_BF16>(a, Mp, Np, Kp, 1, (hipStream_t)stream);
}

// fc1 forward on the real-data path: the A operand's rows are hop-strided frames of the resident bf16 waveform
// (GemmArgs::a_hop), the framed bf16 matrix is written as a by-product, block 0 bumps the step counter.
int rv_linear_fwd_frames(const void* audio_bf16, const long long* frame_index, long first_frame, long hop, long B,
                         const void* w, long ldw, const float* bias, long Mp, long Np, long Kp, int act, void* y, long ldy,
                         void* frames_bf16, long ld_frames, long long* step_counter, void* stream) {
  RV_REQUIRE(audio_bf16 && w && y, RV_ERR_NULL, "rv_linear_fwd_frames: null operand");
  RV_REQUIRE(act == RV_ACT_NONE || act == RV_ACT_RELU, RV_ERR_UNSUPPORTED, "rv_linear_fwd_frames: act %d", act);
  RV_REQUIRE(hop > 0 && hop % 8 == 0 && ((uintptr_t)audio_bf16 & 15) == 0, RV_ERR_SHAPE,
             "rv_linear_fwd_frames: hop %ld must be a multiple of 8 and the waveform 16-byte aligned (16-byte LDS-DMA pieces)", hop);
  RV_REQUIRE(B >= 1 && B <= Mp && first_frame >= 0, RV_ERR_SHAPE, "rv_linear_fwd_frames: %ld frames for %ld rows", B, Mp);
  RV_REQUIRE(!frames_bf16 || (ld_frames >= Kp && ld_frames % 8 == 0 && ((uintptr_t)frames_bf16 & 15) == 0), RV_ERR_SHAPE,
             "rv_linear_fwd_frames: the framed copy needs 16-byte aligned rows of at least Kp elements");
  GemmArgs a{};
  a.A = (const bf16_t*)audio_bf16; a.lda = 8; a.B = (const bf16_t*)w; a.ldb = ldw;
  a.k_tiles = (int)(Kp / 64); a.M_valid = (int)Mp; a.N_valid = (int)Np;
  a.relu = act == RV_ACT_RELU; a.bias = bias; a.out_bf16 = (bf16_t*)y; a.ld_bf16 = ldy;
  a.a_idx = frame_index; a.a_first = first_frame; a.a_hop = hop; a.a_rows = (int)B;
  a.a_copy = (bf16_t*)frames_bf16; a.ld_copy = ld_frames; a.step_inc = step_counter;
  const int tile = choose_tile(Mp, Np, 1);
  RV_REQUIRE(tile != 5 && tile != 7, RV_ERR_UNSUPPORTED,
             "rv_linear_fwd_frames: the gathered operand is implemented in the ring main loop only (tile %d)", tile);
  return launch_tile<true, true, EPI_BIAS_ACT_BF16>(tile, a, Mp, Np, Kp, 1, (hipStream_t)stream);
}

int rv_linear_fwd_fp8(const void* x_fp8, long ldx, const void* w_fp8, long ldw, const float* bias, const float* dq,
                      long Mp, long Np, long Kp, int act, void* y, long ldy, void* stream) {
  RV_REQUIRE(x_fp8 && w_fp8 && y && dq, RV_ERR_NULL, "rv_linear_fwd_fp8: null operand");
  RV_REQUIRE(act == RV_ACT_NONE || act == RV_ACT_RELU, RV_ERR_UNSUPPORTED, "rv_linear_fwd_fp8: act %d", act);
  RV_REQUIRE(Kp % 128 == 0 && ldx % 16 == 0 && ldw % 16 == 0, RV_ERR_SHAPE, "rv_linear_fwd_fp8: K and leading dims must be multiples of 128 / 16 fp8 elements");
  GemmArgs a{};
  a.A = (const bf16_t*)x_fp8; a.lda = ldx / 2; a.B = (const bf16_t*)w_fp8; a.ldb = ldw / 2;
  a.k_tiles = (int)(Kp / 128); a.M_valid = (int)Mp; a.N_valid = (int)Np;
  a.relu = act == RV_ACT_RELU; a.bias = bias; a.out_bf16 = (bf16_t*)y; a.ld_bf16 = ldy; a.dq = dq;
  return launch_tile_fp8<EPI_BIAS_ACT_BF16>(choose_tile(Mp, Np, 1), a, Mp, Np, Kp / 2, (hipStream_t)stream);
}

int rv_decode_out_loss_fwd_fp8(const void* h3_fp8, long ldh, const void* w4_fp8, long ldw, const float* b4, const float* dq,
                               long Bp, long Sp, long Hp, long B, long S, const float* x, long ldx, float* recon,
                               long ld_recon, void* dP4, long ld_dp4, void* dP4_fp8, long ld_dp4q, const float* dp4_scale,
                               float* mse_partial, float* db4_partial, void* stream) {
  RV_REQUIRE(h3_fp8 && w4_fp8 && dq, RV_ERR_NULL, "rv_decode_out_loss_fwd_fp8: null operand");
  RV_REQUIRE(B <= Bp && S <= Sp, RV_ERR_SHAPE, "rv_decode_out_loss_fwd_fp8: B,S exceed padded extents");
  RV_REQUIRE(!x || dP4 || dP4_fp8, RV_ERR_NULL, "rv_decode_out_loss_fwd_fp8: x given without dP4 output");
  RV_REQUIRE(!dP4_fp8 || (dp4_scale && ld_dp4q % 8 == 0 && ((uintptr_t)dP4_fp8 & 7) == 0), RV_ERR_SHAPE,
             "rv_decode_out_loss_fwd_fp8: the fp8 image of dP4 needs its scale and 8-byte aligned rows");
  RV_REQUIRE(Hp % 128 == 0 && ldh % 16 == 0 && ldw % 16 == 0, RV_ERR_SHAPE, "rv_decode_out_loss_fwd_fp8: K and leading dims must be multiples of 128 / 16 fp8 elements");
  GemmArgs a{};
  a.A = (const bf16_t*)h3_fp8; a.lda = ldh / 2; a.B = (const bf16_t*)w4_fp8; a.ldb = ldw / 2;
  a.k_tiles = (int)(Hp / 128); a.M_valid = (int)B; a.N_valid = (int)S;
  a.bias = b4; a.x = x; a.ld_x = ldx; a.recon = recon; a.ld_recon = ld_recon;
  a.out_bf16 = (bf16_t*)dP4; a.ld_bf16 = ld_dp4; a.blocksum = mse_partial; a.colsum = db4_partial;
  a.out_fp8 = (unsigned char*)dP4_fp8; a.ld_fp8 = ld_dp4q; a.q_scale = dp4_scale;
  a.scale = 2.0f / ((float)B * (float)S); a.dq = dq;
  return launch_tile_fp8<EPI_TANH_LOSS>(choose_tile(Bp, Sp, 1, 0, false), a, Bp, Sp, Hp / 2, (hipStream_t)stream);
}

int rv_linear_fwd_f32(const void* x, long ldx, const void* w, long ldw, const float* bias,
                      long Mp, long Np, long Kp, int splits, float* y, long ldy, void* stream) {
  RV_REQUIRE(x && w && y, RV_ERR_NULL, "rv_linear_fwd_f32: null operand");
  GemmArgs a{};
  a.A = (const bf16_t*)x; a.lda = ldx; a.B = (const bf16_t*)w; a.ldb = ldw;
  a.k_tiles = (int)(Kp / 64 / (splits > 0 ? splits : 1)); a.M_valid = (int)Mp; a.N_valid = (int)Np;
  a.bias = bias; a.out_f32 = y; a.ld_f32 = ldy; a.split_stride_f32 = Mp * ldy;
  return launch_auto<true, true, EPI_F32>(a, Mp, Np, Kp, splits, (hipStream_t)stream);
}

int rv_decode_out_loss_fwd(const void* h3, long ldh, const void* w4, long ldw, const float* b4,
                           long Bp, long Sp, long Hp, long B, long S, const float* x, long ldx,
                           float* recon, long ld_recon, void* dP4, long ld_dp4,
                           float* mse_partial, float* db4_partial, void* stream) {
  RV_REQUIRE(h3 && w4, RV_ERR_NULL, "rv_decode_out_loss_fwd: null operand");
  RV_REQUIRE(B <= Bp && S <= Sp, RV_ERR_SHAPE, "rv_decode_out_loss_fwd: B,S exceed padded extents");
  RV_REQUIRE(!x || dP4, RV_ERR_NULL, "rv_decode_out_loss_fwd: x given without dP4 output");
  GemmArgs a{};
  a.A = (const bf16_t*)h3; a.lda = ldh; a.B = (const bf16_t*)w4; a.ldb = ldw;
  a.k_tiles = (int)(Hp / 64); a.M_valid = (int)B; a.N_valid = (int)S;
  a.bias = b4; a.x = x; a.ld_x = ldx; a.recon = recon; a.ld_recon = ld_recon;
  a.out_bf16 = (bf16_t*)dP4; a.ld_bf16 = ld_dp4; a.blocksum = mse_partial; a.colsum = db4_partial;
  a.scale = 2.0f / ((float)B * (float)S);
  return launch_auto<true, true, EPI_TANH_LOSS>(a, Bp, Sp, Hp, 1, (hipStream_t)stream);
}

// rv_decode_out_loss_fwd (bf16 or fp8 operands: h3/w4 fp8 when `dq` is given) with the target frames read from the
// resident waveform: frame r = audio[f*hop : f*hop + S], f = frame_index ? frame_index[r] : first_frame + r.
int rv_decode_out_loss_fwd_frames(const void* h3, long ldh, const void* w4, long ldw, const float* b4, const float* dq,
                                  long Bp, long Sp, long Hp, long B, long S, const float* audio, long n_samples,
                                  const long long* frame_index, long first_frame, long hop, float* recon, long ld_recon,
                                  void* dP4, long ld_dp4, void* dP4_fp8, long ld_dp4q, const float* dp4_scale,
                                  float* mse_partial, float* db4_partial, void* stream) {
  RV_REQUIRE(h3 && w4 && audio && (dP4 || dP4_fp8), RV_ERR_NULL, "rv_decode_out_loss_fwd_frames: null operand");
  RV_REQUIRE(!dP4_fp8 || (dq && dp4_scale && ld_dp4q % 8 == 0 && ((uintptr_t)dP4_fp8 & 7) == 0), RV_ERR_SHAPE,
             "rv_decode_out_loss_fwd_frames: the fp8 image of dP4 belongs to the fp8 forward and needs its scale");
  RV_REQUIRE(B <= Bp && S <= Sp && hop > 0 && n_samples > 0, RV_ERR_SHAPE, "rv_decode_out_loss_fwd_frames: bad extents");
  GemmArgs a{};
  a.M_valid = (int)B; a.N_valid = (int)S;
  a.bias = b4; a.x = audio; a.ld_x = 0; a.x_idx = frame_index; a.x_first = first_frame; a.x_hop = hop; a.x_nsamples = n_samples;
  a.recon = recon; a.ld_recon = ld_recon;
  a.out_bf16 = (bf16_t*)dP4; a.ld_bf16 = ld_dp4; a.blocksum = mse_partial; a.colsum = db4_partial;
  a.out_fp8 = (unsigned char*)dP4_fp8; a.ld_fp8 = ld_dp4q; a.q_scale = dp4_scale;
  a.scale = 2.0f / ((float)B * (float)S);
  if (dq) {
    RV_REQUIRE(Hp % 128 == 0 && ldh % 16 == 0 && ldw % 16 == 0, RV_ERR_SHAPE, "rv_decode_out_loss_fwd_frames: fp8 K and leading dims must be multiples of 128 / 16");
    a.A = (const bf16_t*)h3; a.lda = ldh / 2; a.B = (const bf16_t*)w4; a.ldb = ldw / 2; a.k_tiles = (int)(Hp / 128); a.dq = dq;
    return launch_tile_fp8<EPI_TANH_LOSS>(choose_tile(Bp, Sp, 1, 0, false), a, Bp, Sp, Hp / 2, (hipStream_t)stream);
  }
  a.A = (const bf16_t*)h3; a.lda = ldh; a.B = (const bf16_t*)w4; a.ldb = ldw; a.k_tiles = (int)(Hp / 64);
  return launch_auto<true, true, EPI_TANH_LOSS>(a, Bp, Sp, Hp, 1, (hipStream_t)stream);
}

int rv_linear_dgrad(const void* dy, long lddy, const void* w, long ldw, long Mp, long Np, long Kp,
                    const void* mask, long ldmask, void* dx, long lddx, float* colsum,
                    float* dx32, long lddx32, int splits, void* stream) {
  RV_REQUIRE(dy && w, RV_ERR_NULL, "rv_linear_dgrad: null operand");
  GemmArgs a{};
  a.A = (const bf16_t*)dy; a.lda = lddy; a.B = (const bf16_t*)w; a.ldb = ldw;
  a.M_valid = (int)Mp; a.N_valid = (int)Np;
  if (mask) {
    RV_REQUIRE(dx, RV_ERR_NULL, "rv_linear_dgrad: mask given without bf16 output");
    a.k_tiles = (int)(Kp / 64); a.mask = (const bf16_t*)mask; a.ld_mask = ldmask;
    a.out_bf16 = (bf16_t*)dx; a.ld_bf16 = lddx; a.colsum = colsum;
    return launch_auto<true, false, EPI_MASK_BF16>(a, Mp, Np, Kp, 1, (hipStream_t)stream);
  }
  RV_REQUIRE(dx32, RV_ERR_NULL, "rv_linear_dgrad: no output given");
  a.k_tiles = (int)(Kp / 64 / (splits > 0 ? splits : 1));
  a.out_f32 = dx32; a.ld_f32 = lddx32; a.split_stride_f32 = Mp * lddx32;
  return launch_auto<true, false, EPI_F32>(a, Mp, Np, Kp, splits, (hipStream_t)stream);
}

// Weight gradient with every option: a named block tile (RV_TILE_AUTO = the picker's choice) and the slab element type.
int rv_linear_wgrad(const void* dy, long lddy, const void* x, long ldx, long Mp, long Np, long Kp, int splits,
                       int tile, void* dw, long lddw, int slab_dtype, float* slab_unscale, void* stream) {
  RV_REQUIRE(dy && x && dw, RV_ERR_NULL, "rv_linear_wgrad: null operand");
  RV_REQUIRE(tile == RV_TILE_AUTO || tile == RV_TILE_256x256 || tile == RV_TILE_256x128 || tile == RV_TILE_128x128 ||
                 tile == RV_TILE_64x64, RV_ERR_UNSUPPORTED, "rv_linear_wgrad: unknown tile %d", tile);
  RV_REQUIRE(splits >= 1, RV_ERR_SHAPE, "rv_linear_wgrad: splits %d", splits);
  GemmArgs a{};
  a.A = (const bf16_t*)dy; a.lda = lddy; a.B = (const bf16_t*)x; a.ldb = ldx;
  a.k_tiles = (int)(Kp / 64 / splits); a.M_valid = (int)Mp; a.N_valid = (int)Np;
  int rc = set_slabs(a, dw, lddw, Mp * lddw, slab_dtype, slab_unscale, Mp, Np);
  if (rc) return rc;
  if (tile == RV_TILE_AUTO) return launch_auto<false, false, EPI_F32>(a, Mp, Np, Kp, splits, (hipStream_t)stream);
  return launch_tile<false, false, EPI_F32>(tile, a, Mp, Np, Kp, splits, (hipStream_t)stream);
}

int rv_wgrad_adam_fits(long Mp, long Np, long Kp, int splits) {   // (plan.hip's schedule; not in the public header)
  return Mp > 0 && Np > 0 && Kp > 0 && Mp % 256 == 0 && Np % 256 == 0 && Kp % 64 == 0 && splits >= 1 &&
         (Kp / 64) % splits == 0 && g_force_tile < 0;
}

}  // extern "C" (the rider launchers below carry their own linkage)

constexpr int TAIL_PCT_BF16 = 0, TAIL_PCT_FP8 = 15;   // C2 sweep: bf16 176 us per step at 0, 178-182 above; fp8 163 at 0, 158 from 12 to 22
static int wgrad_riders(const char* who, const void* dy, long lddy, const void* x, long ldx, long Mp, long Np, long Kp, int splits,
                        void* dw, long lddw, int slab_dtype, float* slab_unscale, const rv_param_desc* descs, int n_desc,
                        float* param, float* exp_avg, float* exp_avg_sq, float lr, float grad_scale,
                        const long long* step_counter, float* fin_f32, bf16_t* fin_bf16, int n_rider_blocks, void* stream,
                        const float* fp8_dq = nullptr) {
  const bool fp8 = fp8_dq != nullptr;   // operands are e4m3 bytes (leading dims in bytes), K tiles 128 deep, ping-pong loop only
  const long kt = fp8 ? 128 : 64;
  RV_REQUIRE(Mp > 0 && Np > 0 && Kp > 0 && Mp % 256 == 0 && Np % 256 == 0 && Kp % kt == 0 && splits >= 1 &&
                 (Kp / kt) % splits == 0, RV_ERR_SHAPE,
             "%s: %ld x %ld x %ld / %d splits does not tile by 256x256x%ld", who, Mp, Np, Kp, splits, kt);
  RV_REQUIRE(!fp8 || ((Kp / kt / splits) % 2 == 0 && lddy % 16 == 0 && ldx % 16 == 0), RV_ERR_SHAPE,
             "%s: fp8 operands need an even number of 128-deep K tiles per split and leading dims that are multiples of 16 bytes", who);
  RV_REQUIRE(lddy % 8 == 0 && ldx % 8 == 0 && (((uintptr_t)dy | (uintptr_t)x) & 15) == 0, RV_ERR_SHAPE,
             "%s: operands must be 16-byte aligned with leading dims multiples of 8", who);
  RV_REQUIRE(n_rider_blocks >= 1 && n_rider_blocks <= 4096, RV_ERR_SHAPE, "%s: %d rider blocks", who, n_rider_blocks);
  DescTable tab;
  int rc = adam_build_table(descs, n_desc, &tab);
  if (rc) return rc;
  GemmArgs g{};
  g.A = (const bf16_t*)dy; g.lda = fp8 ? lddy / 2 : lddy; g.B = (const bf16_t*)x; g.ldb = fp8 ? ldx / 2 : ldx;
  g.k_tiles = (int)(Kp / kt / splits); g.M_valid = (int)Mp; g.N_valid = (int)Np; g.dq = fp8_dq;
  rc = set_slabs(g, dw, lddw, Mp * lddw, slab_dtype, slab_unscale, Mp, Np);
  if (rc) return rc;
  g.tiles_m = (int)(Mp / 256); g.tiles_n = (int)(Np / 256); g.splits = splits; g.wt = rv_store_wt;
  const int n_gemm = g.tiles_m * g.tiles_n * splits;
  constexpr int smem = 2 * (256 + 256) * 128;
  static_assert(8 * 2 * AS_SLOT <= smem, "the optimizer waves' LDS rings live in the launch's dynamic LDS");
  // fp16 slabs are 8 B per group, below the 16-byte LDS-DMA piece: such tables take the plain-load walk (adam_group)
  int stream_mode = 1;
  for (int i = 0; i < n_desc; ++i)
    if (descs[i].grad_half) stream_mode = 0;
  // Share of the riders' table the GEMM blocks take over behind their tiles (percent of its virtual blocks; update mode
  // with plain-load riders only).  The riders stream at a per-CU rate whatever the GEMM does, so the share is what evens
  // the two halves out: measured per operand type at C2 (the sweep: profiles/r04_tail_sweep.txt).
  int tail_vb = 0;
  if (!fin_f32 && !fin_bf16 && !stream_mode) tail_vb = (int)(tab.blk_start[n_desc] * (fp8 ? TAIL_PCT_FP8 : TAIL_PCT_BF16) / 100);
  const bool pp = g.k_tiles % 2 == 0;
  const int which = fp8 ? 2 : (pp ? 1 : 0);
  auto kern = fp8 ? gemm_wgrad_adam_kernel<8, true> : (pp ? gemm_wgrad_adam_kernel<8, false> : gemm_wgrad_adam_kernel<2, false>);
  static std::atomic<unsigned long long> attr_done[3];
  lds_opt_in((const void*)kern, smem, attr_done[which]);
  hipLaunchKernelGGL(kern, dim3((unsigned)(n_gemm + n_rider_blocks)), dim3(512), smem, (hipStream_t)stream, g, n_gemm, tab,
                     param, exp_avg, exp_avg_sq, lr, grad_scale, step_counter, stream_mode, fin_f32, fin_bf16, tail_vb);
  RV_CHECK_LAUNCH();
  return RV_OK;
}

// ... and on fp8 operands (as rv_linear_wgrad_adam_fp8 below: bytes, both MN-major, dq = 1 / (scale_dy * scale_x)).
extern "C" int rv_linear_wgrad_finalize_fp8(const void* dy_fp8, long lddy, const void* x_fp8, long ldx, const float* dq, long Mp,
                                 long Np, long Kp, int splits, void* dw, long lddw, int slab_dtype, float* slab_unscale,
                                 const rv_param_desc* descs, int n_desc, void* grad_out, int out_bf16, int n_rider_blocks,
                                 void* stream) {
  RV_REQUIRE(dy_fp8 && x_fp8 && dq && dw && grad_out, RV_ERR_NULL, "rv_linear_wgrad_finalize_fp8: null pointer");
  return wgrad_riders("rv_linear_wgrad_finalize_fp8", dy_fp8, lddy, x_fp8, ldx, Mp, Np, Kp, splits, dw, lddw, slab_dtype, slab_unscale,
                      descs, n_desc, nullptr, nullptr, nullptr, 0.f, 1.f, nullptr, out_bf16 ? nullptr : (float*)grad_out,
                      out_bf16 ? (bf16_t*)grad_out : nullptr, n_rider_blocks, stream, dq);
}

// rv_linear_wgrad_adam on fp8 (e4m3) operands (RV_OPT_FP8 = 1; plan.hip): dy_fp8 [Kp(batch), Mp] and x_fp8 [Kp, Np], one
// byte per element, both read MN-major through ds_read_b64_tr_b8 (the contraction index is the row of both matrices);
// dq: device scalar 1 / (scale_dy * scale_x).  Same slabs, riders and launch shape.  Not in the public header.
extern "C" int rv_linear_wgrad_adam_fp8(const void* dy_fp8, long lddy, const void* x_fp8, long ldx, const float* dq, long Mp, long Np,
                             long Kp, int splits, void* dw, long lddw, int slab_dtype, float* slab_unscale,
                             const rv_param_desc* descs, int n_desc, float* param, float* exp_avg, float* exp_avg_sq, float lr,
                             float grad_scale, const long long* step_counter, int n_adam_blocks, void* stream) {
  RV_REQUIRE(dy_fp8 && x_fp8 && dq && dw && param && exp_avg && exp_avg_sq && step_counter, RV_ERR_NULL, "rv_linear_wgrad_adam_fp8: null pointer");
  return wgrad_riders("rv_linear_wgrad_adam_fp8", dy_fp8, lddy, x_fp8, ldx, Mp, Np, Kp, splits, dw, lddw, slab_dtype, slab_unscale, descs,
                      n_desc, param, exp_avg, exp_avg_sq, lr, grad_scale, step_counter, nullptr, nullptr, n_adam_blocks, stream, dq);
}

extern "C" int rv_pair_stop_event(void* ev) {   // returns whether an armed event was still pending (no paired launch took it)
  const int pending = g_pair_stop_event != nullptr;
  g_pair_stop_event = (hipEvent_t)ev;
  return pending;
}

extern "C" int rv_linear_wgrad_adam(const void* dy, long lddy, const void* x, long ldx, long Mp, long Np, long Kp, int splits,
                         void* dw, long lddw, int slab_dtype, float* slab_unscale, const rv_param_desc* descs, int n_desc,
                         float* param, float* exp_avg,
                         float* exp_avg_sq, float lr, float grad_scale, const long long* step_counter,
                         int n_adam_blocks, void* stream) {
  RV_REQUIRE(dy && x && dw && param && exp_avg && exp_avg_sq && step_counter, RV_ERR_NULL, "rv_linear_wgrad_adam: null pointer");
  return wgrad_riders("rv_linear_wgrad_adam", dy, lddy, x, ldx, Mp, Np, Kp, splits, dw, lddw, slab_dtype, slab_unscale, descs,
                      n_desc, param, exp_avg, exp_avg_sq, lr, grad_scale, step_counter, nullptr, nullptr, n_adam_blocks, stream);
}

// The same launch whose rider blocks only sum OTHER tensors' gradient slabs into a flat payload arena (fp32, or bf16
// when out_bf16): rv_grad_finalize's work on the CUs the GEMM leaves idle (plan.hip: the data-parallel step's second
// bucket, except the gradient this GEMM is producing).  Not in the public header.
extern "C" int rv_linear_wgrad_finalize(const void* dy, long lddy, const void* x, long ldx, long Mp, long Np, long Kp, int splits,
                             void* dw, long lddw, int slab_dtype, float* slab_unscale, const rv_param_desc* descs, int n_desc,
                             void* grad_out, int out_bf16, int n_rider_blocks, void* stream) {
  RV_REQUIRE(dy && x && dw && grad_out, RV_ERR_NULL, "rv_linear_wgrad_finalize: null pointer");
  return wgrad_riders("rv_linear_wgrad_finalize", dy, lddy, x, ldx, Mp, Np, Kp, splits, dw, lddw, slab_dtype, slab_unscale, descs,
                      n_desc, nullptr, nullptr, nullptr, 0.f, 1.f, nullptr, out_bf16 ? nullptr : (float*)grad_out,
                      out_bf16 ? (bf16_t*)grad_out : nullptr, n_rider_blocks, stream);
}


extern "C" {

// ---- paired backward of one Linear layer: dX = relu'(dY W) and dW = dY^T X in one launch ----
// dy [Mp(batch), Kp(out features)], w [Kp, Np] ([out,in]), x [Mp, Np] is BOTH the ReLU output that
// masks dX and the right operand of dW.  dW[Kp, Np] leaves as `splits` slabs over the batch.
// Unpaired backward of a Linear layer: when the wgrad runs on a dual-capable tile (64x64 or 128x128 with
// 8 waves) that also divides the dgrad's output, the dgrad takes the same tile and both go out in one launch.
static int dgrad_tile_unpaired(long Mp, long Np, long Kp, int wgrad_splits) {
  const int tw = choose_tile(Kp, Np, wgrad_splits, Mp);
  if (g_force_tile < 0 && (tw == 0 || tw == 4) && tile_fits(tw, Mp, Np)) return tw;
  return choose_tile(Mp, Np, 1);
}

int rv_dgrad_wgrad_pick(long Mp, long Np, long Kp, int* paired, int* bm_dgrad, int* splits) {
  RV_REQUIRE(Mp > 0 && Np > 0 && Kp > 0 && Mp % 64 == 0 && Np % 64 == 0 && Kp % 64 == 0, RV_ERR_SHAPE,
             "rv_dgrad_wgrad_pick: extents must be positive multiples of 64");
  const bool fits = Mp % 256 == 0 && Np % 256 == 0 && Kp % 256 == 0;
  int pr = 0, bm = 0, sp = 1;
  if (fits && (g_force_tile == 5 || g_force_tile < 0)) {
    const long t_d = (Mp / 256) * (Np / 256), t_w = (Kp / 256) * (Np / 256);
    // wgrad splits: even out the K work per block (dgrad blocks loop over Kp, wgrad blocks over Mp/sp)
    while (2 * sp <= 16 && (Mp / 64) % (2 * sp) == 0 && Mp / (2 * sp) >= Kp && Mp / (2 * sp) >= 128) sp *= 2;
    // one launch while its blocks fill one round of the chip's 256 CUs, or two to four rounds to at least three quarters (B = 8192:
    // fc4's backward 104 -> 68 us, the heads' 65 -> 38 at L = 256; B = 16384: the heads' 127 -> 81; beyond four rounds the big-tile
    // launches of choose_tile are as fast or faster -- profiles/r06_batch_sweep.jsonl)
    const long tot = t_d + t_w * sp, rounds = (tot + 255) / 256;
    const bool whole = rounds >= 2 && rounds <= 4 && 4 * tot >= 3 * 256 * rounds;
    if (g_force_tile == 5 || whole || (tot >= 192 && tot <= 320)) { pr = 1; bm = 256; }
  }
  if (!pr) {
    int bn;
    rv_gemm_pick(Kp, Np, Mp, 16, nullptr, nullptr, &sp);
    tile_dims(dgrad_tile_unpaired(Mp, Np, Kp, sp), &bm, &bn);
  }
  if (paired) *paired = pr;
  if (bm_dgrad) *bm_dgrad = bm;
  if (splits) *splits = sp;
  return RV_OK;
}

int rv_linear_dgrad_wgrad(const void* dy, long lddy, const void* w, long ldw, const void* x, long ldx,
                             long Mp, long Np, long Kp, void* dx_bf16, long lddx,
                             float* colsum_partial, void* dw_slabs, long lddw, int splits, int slab_dtype, float* slab_unscale,
                             void* stream) {
  RV_REQUIRE(dy && w && x && dx_bf16 && dw_slabs, RV_ERR_NULL, "rv_linear_dgrad_wgrad: null operand");
  int paired, bm, sp;
  int rc = rv_dgrad_wgrad_pick(Mp, Np, Kp, &paired, &bm, &sp);
  if (rc) return rc;
  RV_REQUIRE(sp == splits, RV_ERR_STATE, "rv_linear_dgrad_wgrad: caller passed %d splits, rv_dgrad_wgrad_pick says %d",
             splits, sp);
  if (!paired) {
    const int td = dgrad_tile_unpaired(Mp, Np, Kp, splits), tw = choose_tile(Kp, Np, splits, Mp);
    GemmArgs d{}, g{};
    d.A = (const bf16_t*)dy; d.lda = lddy; d.B = (const bf16_t*)w; d.ldb = ldw;
    d.k_tiles = (int)(Kp / 64); d.M_valid = (int)Mp; d.N_valid = (int)Np;
    d.mask = (const bf16_t*)x; d.ld_mask = ldx; d.out_bf16 = (bf16_t*)dx_bf16; d.ld_bf16 = lddx; d.colsum = colsum_partial;
    g.A = (const bf16_t*)dy; g.lda = lddy; g.B = (const bf16_t*)x; g.ldb = ldx;
    g.k_tiles = (int)(Mp / 64 / splits); g.M_valid = (int)Kp; g.N_valid = (int)Np;
    rc = set_slabs(g, dw_slabs, lddw, Kp * lddw, slab_dtype, slab_unscale, Kp, Np);
    if (rc) return rc;
    if (td == tw && (Mp / 64) % splits == 0 &&
        try_dual<false, false, EPI_F32, true, false, EPI_MASK_BF16>(td, g, Kp, Np, splits, d, Mp, Np, 1, (hipStream_t)stream, &rc))  // long (wgrad) blocks first
      return rc;
    rc = launch_tile<true, false, EPI_MASK_BF16>(td, d, Mp, Np, Kp, 1, (hipStream_t)stream);
    if (rc) return rc;
    return rv_linear_wgrad(dy, lddy, x, ldx, Kp, Np, Mp, splits, RV_TILE_AUTO, dw_slabs, lddw, slab_dtype, slab_unscale, stream);
  }
  RV_REQUIRE(lddy % 8 == 0 && ldw % 8 == 0 && ldx % 8 == 0, RV_ERR_SHAPE, "rv_linear_dgrad_wgrad: leading dims must be multiples of 8");
  RV_REQUIRE((((uintptr_t)dy | (uintptr_t)w | (uintptr_t)x) & 15) == 0, RV_ERR_SHAPE, "rv_linear_dgrad_wgrad: operands must be 16-byte aligned");
  constexpr int BM = 256, BN = 256;
  GemmArgs d{}, g{};
  d.A = (const bf16_t*)dy; d.lda = lddy; d.B = (const bf16_t*)w; d.ldb = ldw;
  d.k_tiles = (int)(Kp / 64); d.M_valid = (int)Mp; d.N_valid = (int)Np;
  d.mask = (const bf16_t*)x; d.ld_mask = ldx; d.out_bf16 = (bf16_t*)dx_bf16; d.ld_bf16 = lddx; d.colsum = colsum_partial;
  d.tiles_m = (int)(Mp / BM); d.tiles_n = (int)(Np / BN); d.splits = 1;
  g.A = (const bf16_t*)dy; g.lda = lddy; g.B = (const bf16_t*)x; g.ldb = ldx;
  g.k_tiles = (int)(Mp / 64 / splits); g.M_valid = (int)Kp; g.N_valid = (int)Np;
  rc = set_slabs(g, dw_slabs, lddw, Kp * lddw, slab_dtype, slab_unscale, Kp, Np);
  if (rc) return rc;
  g.tiles_m = (int)(Kp / BM); g.tiles_n = (int)(Np / BN); g.splits = splits;
  if (g_pair_loop == 8 && d.k_tiles % 2 == 0 && g.k_tiles % 2 == 0) return launch_pair<8>(d, g, (hipStream_t)stream);
  return launch_pair<2>(d, g, (hipStream_t)stream);
}


// Backward of a Linear layer whose input had no activation (fc3: its input is z): dX as fp32 split-K
// slabs and dW slabs, in one launch when both GEMMs run on the same dual-capable tile.
int rv_linear_dgrad_wgrad_f32(const void* dy, long lddy, const void* w, long ldw, const void* x, long ldx, long Mp,
                              long Np, long Kp, float* dx_slabs, long lddx, int dgrad_splits, float* dw_slabs,
                              long lddw, int wgrad_splits, void* stream) {
  RV_REQUIRE(dy && w && x && dx_slabs && dw_slabs, RV_ERR_NULL, "rv_linear_dgrad_wgrad_f32: null operand");
  RV_REQUIRE(dgrad_splits >= 1 && wgrad_splits >= 1 && (Kp / 64) % dgrad_splits == 0 && (Mp / 64) % wgrad_splits == 0,
             RV_ERR_SHAPE, "rv_linear_dgrad_wgrad_f32: splits %d / %d do not divide the K tiles", dgrad_splits, wgrad_splits);
  const int td = choose_tile(Mp, Np, dgrad_splits, Kp), tw = choose_tile(Kp, Np, wgrad_splits, Mp);
  int rc;
  if (td == tw) {
    GemmArgs d{}, g{};
    d.A = (const bf16_t*)dy; d.lda = lddy; d.B = (const bf16_t*)w; d.ldb = ldw;
    d.k_tiles = (int)(Kp / 64 / dgrad_splits); d.M_valid = (int)Mp; d.N_valid = (int)Np;
    d.out_f32 = dx_slabs; d.ld_f32 = lddx; d.split_stride_f32 = Mp * lddx;
    g.A = (const bf16_t*)dy; g.lda = lddy; g.B = (const bf16_t*)x; g.ldb = ldx;
    g.k_tiles = (int)(Mp / 64 / wgrad_splits); g.M_valid = (int)Kp; g.N_valid = (int)Np;
    g.out_f32 = dw_slabs; g.ld_f32 = lddw; g.split_stride_f32 = Kp * lddw;
    RV_REQUIRE(lddy % 8 == 0 && ldw % 8 == 0 && ldx % 8 == 0, RV_ERR_SHAPE, "rv_linear_dgrad_wgrad_f32: leading dims must be multiples of 8");
    RV_REQUIRE((((uintptr_t)dy | (uintptr_t)w | (uintptr_t)x) & 15) == 0, RV_ERR_SHAPE, "rv_linear_dgrad_wgrad_f32: operands must be 16-byte aligned");
    if (try_dual<true, false, EPI_F32, false, false, EPI_F32>(td, d, Mp, Np, dgrad_splits, g, Kp, Np, wgrad_splits,
                                                              (hipStream_t)stream, &rc))
      return rc;
  }
  rc = rv_linear_dgrad(dy, lddy, w, ldw, Mp, Np, Kp, nullptr, 0, nullptr, 0, nullptr, dx_slabs, lddx, dgrad_splits, stream);
  if (rc) return rc;
  return rv_linear_wgrad(dy, lddy, x, ldx, Kp, Np, Mp, wgrad_splits, RV_TILE_AUTO, dw_slabs, lddw, RV_SLAB_F32, nullptr, stream);
}

// The paired backward of fc4 on fp8 (e4m3) operands (RV_OPT_FP8; plan.hip): dX = relu'(dYq Wq) and dW = dYq^T Xq in
// one 256 x 256 ping-pong launch, every operand one byte per element -- dy_fp8 [Mp(batch), Kp(out)] (the fp8 image of
// dP4 that the fc4 forward's epilogue wrote, K-major for the dgrad and MN-major for the wgrad), w_fp8 [Kp, Np] (the
// fp8 weight shadow, MN-major through ds_read_b64_tr_b8), x_fp8 [Mp, Np] (the fp8 image of h3, the wgrad's MN-major
// right operand); mask is the bf16 h3 (ReLU'), or -- mask_is_fp8 -- its fp8 image again (ldmask in bytes): the bf16 copy of
// h3 then need not exist.  dq_dgrad / dq_wgrad: device scalars 1 / (scale_dy * scale_w) and
// 1 / (scale_dy * scale_x).  Same outputs, splits and slab formats as rv_linear_dgrad_wgrad.  Not in the public header.
int rv_linear_dgrad_wgrad_fp8(const void* dy_fp8, long lddy, const void* w_fp8, long ldw, const void* x_fp8, long ldx,
                              const void* mask, long ldmask, int mask_is_fp8, const float* dq_dgrad, const float* dq_wgrad,
                              long Mp, long Np, long Kp, void* dx_bf16, long lddx, float* colsum_partial, void* dw_slabs,
                              long lddw, int splits, int slab_dtype, float* slab_unscale, void* stream) {
  const void* mask_bf16 = mask;
  RV_REQUIRE(dy_fp8 && w_fp8 && x_fp8 && mask_bf16 && dx_bf16 && dw_slabs && dq_dgrad && dq_wgrad, RV_ERR_NULL,
             "rv_linear_dgrad_wgrad_fp8: null operand");
  RV_REQUIRE(rv_dgrad_wgrad_fp8_fits(Mp, Np, Kp, splits), RV_ERR_SHAPE,
             "rv_linear_dgrad_wgrad_fp8: %ld x %ld x %ld / %d splits: extents must tile by 256 x 256 with an even number of 128-deep K tiles", Mp, Np, Kp, splits);
  RV_REQUIRE(lddy % 16 == 0 && ldw % 16 == 0 && ldx % 16 == 0 && ldmask % (mask_is_fp8 ? 16 : 8) == 0, RV_ERR_SHAPE,
             "rv_linear_dgrad_wgrad_fp8: leading dims must be multiples of 16 bytes");
  RV_REQUIRE((((uintptr_t)dy_fp8 | (uintptr_t)w_fp8 | (uintptr_t)x_fp8 | (uintptr_t)mask_bf16) & 15) == 0, RV_ERR_SHAPE,
             "rv_linear_dgrad_wgrad_fp8: operands must be 16-byte aligned");
  constexpr int BM = 256, BN = 256;
  GemmArgs d{}, g{};
  d.A = (const bf16_t*)dy_fp8; d.lda = lddy / 2; d.B = (const bf16_t*)w_fp8; d.ldb = ldw / 2;
  d.k_tiles = (int)(Kp / 128); d.M_valid = (int)Mp; d.N_valid = (int)Np;
  d.mask = (const bf16_t*)mask_bf16; d.ld_mask = ldmask; d.mask_fp8 = mask_is_fp8 ? 1 : 0;
  d.out_bf16 = (bf16_t*)dx_bf16; d.ld_bf16 = lddx; d.colsum = colsum_partial;
  d.tiles_m = (int)(Mp / BM); d.tiles_n = (int)(Np / BN); d.splits = 1; d.dq = dq_dgrad;
  g.A = (const bf16_t*)dy_fp8; g.lda = lddy / 2; g.B = (const bf16_t*)x_fp8; g.ldb = ldx / 2;
  g.k_tiles = (int)(Mp / 128 / splits); g.M_valid = (int)Kp; g.N_valid = (int)Np; g.dq = dq_wgrad;
  const int rc = set_slabs(g, dw_slabs, lddw, Kp * lddw, slab_dtype, slab_unscale, Kp, Np);
  if (rc) return rc;
  g.tiles_m = (int)(Kp / BM); g.tiles_n = (int)(Np / BN); g.splits = splits;
  return launch_pair<8, true>(d, g, (hipStream_t)stream);
}

int rv_dgrad_wgrad_fp8_fits(long Mp, long Np, long Kp, int splits) {
  if (!(Mp > 0 && Np > 0 && Kp > 0 && Mp % 256 == 0 && Np % 256 == 0 && Kp % 256 == 0 && splits >= 1)) return 0;
  if (Kp % 128 || (Kp / 128) % 2 || (Kp / 128) < 2) return 0;                       // dgrad: even count of 128-deep K tiles
  if (Mp % (128L * splits) || (Mp / 128 / splits) % 2 || (Mp / 128 / splits) < 2) return 0;   // wgrad, per split
  int paired = 0, bm = 0, sp = 0;
  if (rv_dgrad_wgrad_pick(Mp, Np, Kp, &paired, &bm, &sp) || !paired || sp != splits) return 0;
  return g_force_tile < 0 && g_pair_loop == 8;
}

int rv_gemm_plan(int what, long Mp, long Np, long Kp, int splits_in, int* bm, int* bn, int* splits, int* paired) {
  switch (what) {
    case RV_PLAN_GEMM: return rv_gemm_pick(Mp, Np, Kp, splits_in, bm, bn, splits);
    case RV_PLAN_TILE: {
      if (splits) *splits = splits_in;
      return rv_gemm_tile(Mp, Np, splits_in, bm, bn);
    }
    case RV_PLAN_PAIR: return rv_dgrad_wgrad_pick(Mp, Np, Kp, paired, bm, splits);
  }
  RV_REQUIRE(false, RV_ERR_UNSUPPORTED, "rv_gemm_plan: unknown query %d", what);
  return RV_OK;
}

}  // extern "C"
