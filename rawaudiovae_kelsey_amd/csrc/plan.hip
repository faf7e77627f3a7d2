// Whole-step plan: one host call enqueues the 14 kernels of a training step
// (train.py:184-193) on a stream; hipGraph capture/replay and HIP-event timing
// helpers.  C ABI: include/rawvae_hip.h.
#include "common.h"
#include "../../include/rawvae_hip.h"
#include "internal.h"

#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <string>
#include <vector>


namespace {

struct Buf {
  const char* name;
  long bytes;
  long off;
};

long align256(long b) { return (b + 255) / 256 * 256; }

int splits_of(long Mp, long Np, long Kp) {
  int s = 1;
  rv_gemm_pick(Mp, Np, Kp, 16, nullptr, nullptr, &s);
  return s;
}

// Block tile of the fc1 weight gradient when it runs as a launch of its own: the 256x256 tile of the launch that also
// carries optimizer blocks (rv_linear_wgrad_adam) wherever that one applies, so that every schedule of the step --
// full, phase by phase, data-parallel -- writes bit-identical slabs (the per-wave-tile exponents of fp16 slabs
// follow the tile).
int w1_tile(const rv_plan* p);

int row_tiles(long Mp, long Np) {
  int bm = 128;
  rv_gemm_tile(Mp, Np, 1, &bm, nullptr);
  return (int)(Mp / bm);
}

}  // namespace

// roctx ranges around the phases of a step (RV_OPT_ROCTX): the marker library is looked up at run time, once; without it
// (or with the option off) a range is two untaken branches.  Ranges bracket the HOST calls that enqueue a phase: in a
// rocprofv3 --marker-trace --kernel-trace timeline every kernel dispatch falls inside the range that launched it.
namespace {
typedef int (*roctx_push_fn)(const char*);
typedef int (*roctx_pop_fn)();
roctx_push_fn g_roctx_push = nullptr;
roctx_pop_fn g_roctx_pop = nullptr;
bool roctx_load() {
  static int state = 0;   // 0 not tried, 1 loaded, -1 unavailable
  if (state == 0) {
    state = -1;
    for (const char* name : {"librocprofiler-sdk-roctx.so", "libroctx64.so"}) {
      void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (!h) continue;
      g_roctx_push = (roctx_push_fn)dlsym(h, "roctxRangePushA");
      g_roctx_pop = (roctx_pop_fn)dlsym(h, "roctxRangePop");
      if (g_roctx_push && g_roctx_pop) { state = 1; break; }
    }
  }
  return state == 1;
}
struct Range {   // scope guard
  bool on;
  Range(bool enabled, const char* name) : on(enabled && g_roctx_push) { if (on) g_roctx_push(name); }
  ~Range() { if (on) g_roctx_pop(); }
};
}  // namespace

struct rv_plan {
  long B, S, H, L, Bp, Sp, Hp, Lp, L2p;
  int s_heads, s_dz, s_w4, s_w3, s_wh, s_w1;  // split-K factors
  int n_mse, n_kl, n_mt4, n_mt3, n_mt1;       // partial counts (row tiles of the producing GEMMs)
  // the heads' backward has two forms with different partial counts: the generic dual launch (s_wh_gen slabs of dWh,
  // n_mt1_gen partial rows of fc1's bias gradient) and the streaming kernel rv_heads_bwd (hb_groups of both; 0 = the
  // shape does not allow it); s_wh / n_mt1 and the descriptors follow the form in use (heads_mode_apply)
  int s_wh_gen = 1, n_mt1_gen = 1, hb_groups = 0;
  // the fc4 forward's partial counts per operand type: [0] bf16 operands, [1] fp8 operands (whose fused-loss kernel keeps
  // the smaller tiles at large batches); n_mt4 / n_mse and fc4.bias's descriptor follow the type in use (fwd4_mode_apply)
  int n_mt4_of[2] = {1, 1}, n_mse_of[2] = {1, 1};
  int heads_pair_gen = 0;   // the generic form is the paired 256 x 256 launch (rv_dgrad_wgrad_pick): its dWh slabs may be fp16
  long off[10];                               // element offsets of the 10 params in the flat arenas
  long n_params;
  std::vector<Buf> bufs;
  long ws_bytes;
  rv_plan_buffers b;
  bool bound;
  rv_param_desc d_slab[10], d_flat[10];
  // gradients supplied by the caller for the next backward phases (rv_plan_set_external_grads); all null = the
  // fused loss of the forward phase
  const float* ext_d_recon = nullptr; const float* ext_recon = nullptr;
  const float* ext_dmu = nullptr; const float* ext_dlv = nullptr;
  float* ext_grad_out = nullptr;
  const float* loss_grad_dev = nullptr;   // rv_plan_set_loss_grad: the FINALIZE launches multiply by this device scalar
  int latent_fused = 1;          // RV_OPT_LATENT_FUSED: heads + reparam + fc3 as one launch (rv_latent_fwd) where it applies
  bool shadows_padded = false;   // rv_plan_refresh_shadows has zeroed the shadows' padding once
  // data-parallel step: the collective library's all-reduce (RCCL's ncclAllReduce), its communicator,
  // a dedicated stream for it, and "bucket ready" / "bucket reduced" events per gradient bucket
  rv_allreduce_fn allreduce = nullptr;
  void* comm = nullptr;
  int world = 1;
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev_ready[3] = {nullptr, nullptr, nullptr}, ev_done[3] = {nullptr, nullptr, nullptr};
  int rank = 0;
  hipEvent_t ev_flush = nullptr;   // rv_plan_step behind a deferred data-parallel update on another stream
  int fp8 = 0;                 // fc1 / fc4 forward on fp8 operands (RV_OPT_FP8)
  int n_amax_cap = 4096;       // entries of the workspace buffer "h3_amax" for h3's maxima
  int n_amax_dp1 = 0;          // ... and for dP1's behind them (fp8 weight gradient of fc1)
  int n_amax_h3 = 0;           // how many of h3's the forward in use writes (set by the forward phase)
  bool heads_half = false;     // the streaming heads' backward writes fp16 dWh slabs (heads_mode_apply)
  bool fwd_for_fp8_w1 = false; // set by rv_plan_step_ddp around its forward call (fp8_w1)
  bool last_fwd_f8_w1 = false; // what the most recent forward phase decided (fp8_w1): the backward follows IT, not its own phase mask
  bool last_fwd_no_h3 = false; // ... and whether it left the bf16 h3 unwritten (fc4's dgrad then masks with the fp8 image)
  int ddp_seq = 0;             // data-parallel steps enqueued with device-side flags (their sequence number)
  int ddp_signal = 1;          // RV_OPT_DDP_SIGNAL: 1 device-side flags between the two streams (default), 0 HIP events
  long ddp_wait_ms = 30000;    // RV_OPT_DDP_WAIT_MS: bound of a flag wait whose setter sits behind a collective (peers)
  // RV_OPT_DDP_DEFER_TAIL: the all-reduce schedule leaves its last wait (second exchange done) and the update behind it
  // to the NEXT call, which enqueues its own cast launch first -- that launch needs no parameter, and the compute stream
  // has nothing else to do while the exchange is on the links.  tail_*: what the deferred half needs from its step.
  int ddp_defer = 0, tail_pending = 0, tail_seq = 0;
  float tail_lr = 0.f, tail_scale = 1.f;
  void* tail_stream = nullptr;   // the stream the deferring step was enqueued on: its other half goes there and nowhere else
  int cast_done = 0;           // the forward's cast launch went out ahead of its phase (ddp_finish_tail's caller)
  int s_w1_ddp = 1;            // split-K of fc1's weight gradient in the data-parallel step (RV_OPT_DDP_W1_WIDE)
  int ddp_w1_wide = 0;
  int roctx = 0;               // RV_OPT_ROCTX: roctx ranges around the step's phases
  unsigned skip = 0;           // rv_plan_diag_skip (include/rawvae_hip_diag.h): launches of the full step left out
  int slab_dtype = RV_SLAB_F16;   // element type of the dW1 / dW4 split-K slabs (RV_OPT_SLAB_DTYPE)
  float* us_w1 = nullptr;         // per-granule scale tables of the fp16 slabs (workspace "dW1_us" / "dW4_us")
  float* us_w4 = nullptr;
  // frame source of the step in flight (rv_plan_step_frames): `x` is then the resident waveform
  const long long* fr_idx = nullptr;
  long fr_first = 0, fr_hop = 0, fr_nsamples = 0;
  const void* fr_bf16 = nullptr;   // the waveform as bf16 (fc1's operand is gathered from it), or null
  int payload_bf16 = 0;
  void* grad_bf16 = nullptr;   // caller's flat bf16 payload arena (rv_comm_desc.grad_bf16)

  char* ws(const char* name, long* nbytes = nullptr) const {
    for (const Buf& x : bufs)
      if (!strcmp(x.name, name)) {
        if (nbytes) *nbytes = x.bytes;
        return (char*)b.workspace + x.off;
      }
    return nullptr;
  }
  void add(const char* name, long bytes) {
    bufs.push_back({name, bytes, ws_bytes});
    ws_bytes += align256(bytes);
  }
};

namespace {
int w1_tile(const rv_plan* p) {
  return rv_wgrad_adam_fits(p->Hp, p->Sp, p->Bp, p->s_w1) && (p->Hp / 256) * (p->Sp / 256) * p->s_w1 <= 192 ? RV_TILE_256x256
                                                                                                          : RV_TILE_AUTO;
}
}  // namespace

// Which form of the heads' backward runs (see rv_plan: s_wh_gen / hb_groups) and the partial counts that follow from it.
static bool heads_streaming(const rv_plan* p) {
  // (large batches: the streaming kernel re-reads its Wh slice per 512 rows at 0.12 of the MFMA peak; the paired / tiled
  // GEMM forms take over with the other latent-sized launches)
  return p->latent_fused && p->hb_groups > 0 && rv_latent_rowlocal(p->Bp, p->Hp, p->Lp);
}
static bool latent_bwd_fused(const rv_plan* p);
static void heads_mode_apply(rv_plan* p) {
  const bool st = heads_streaming(p);
  {
    // the head biases' partial rows [Bp / 16][2 Lp]: rv_reparam_bwd and the row-local backward fill every row, the GEMM forms of
    // rv_latent_bwd one row per dz tile (64 or 256 batch rows) and zeros between -- the descriptors step over those (at
    // default.ini's batch 8192 rows of which 512 count: the optimizer's cooperative reduction walked them all, 120 us)
    const long rows = latent_bwd_fused(p) ? rv_latent_bwd_tile_rows(p->Bp, p->Hp, p->Lp) : 16;
    for (int i : {3, 5}) {
      p->d_slab[i].grad_splits = (int)(p->Bp / rows);
      p->d_slab[i].grad_split_stride = (rows / 16) * p->L2p;
    }
  }
  p->s_wh = st ? p->hb_groups : p->s_wh_gen;
  p->n_mt1 = st ? p->hb_groups : p->n_mt1_gen;
  p->d_slab[1].grad_splits = p->n_mt1;
  p->d_slab[2].grad_splits = p->s_wh;
  p->d_slab[4].grad_splits = p->s_wh;
  // The streaming heads' backward writes its dWh slabs in the element type of the large weight gradients' (RV_OPT_SLAB_DTYPE):
  // block-floating-point fp16 by default (rv_heads_bwd_ex), and so does the generic form where it is the paired 256 x 256
  // launch (the reference's own latent_dim = 256: eight 512 x 2048 slabs, 32 MB as fp32 -- written by that launch and read
  // back by the optimizer); fp32 in the generic form's smaller tilings and in the strict mode
  const bool half = p->slab_dtype == RV_SLAB_F16 && p->Hp % 32 == 0 && (st ? p->Lp == 64 : p->heads_pair_gen != 0);
  p->heads_half = half;
  float* dWh = (float*)p->ws("dWh");
  float* us = (float*)p->ws("dWh_us");
  const long us_ld = p->Hp / 32;
  for (int i : {2, 4}) {   // fc21.weight = rows [0, Lp) of the stacked slab, fc22.weight = rows [Lp, 2 Lp)
    rv_param_desc* d = p->d_slab + i;
    d->grad_half = half;
    d->grad_unscale = half ? us + (i == 4 ? (p->Lp / 32) * us_ld : 0) : nullptr;
    d->us_ld = us_ld;
    d->us_split_stride = (2 * p->Lp / 32) * us_ld;
    // (grad_slabs is typed float*; with fp16 elements row Lp of the slab sits Lp * Hp HALF-elements behind its start)
    d->grad_slabs = i == 2 ? dWh : (half ? dWh + p->Lp * p->Hp / 2 : dWh + p->Lp * p->Hp);
  }
  if (!p->b.grad) {   // without a flat gradient arena d_flat mirrors the slab descriptors
    p->d_flat[1].grad_splits = p->n_mt1;
    for (int i : {3, 5}) { p->d_flat[i].grad_splits = p->d_slab[i].grad_splits; p->d_flat[i].grad_split_stride = p->d_slab[i].grad_split_stride; }
    for (int i : {2, 4}) {
      p->d_flat[i].grad_splits = p->s_wh;
      p->d_flat[i].grad_half = p->d_slab[i].grad_half; p->d_flat[i].grad_unscale = p->d_slab[i].grad_unscale;
      p->d_flat[i].us_ld = p->d_slab[i].us_ld; p->d_flat[i].us_split_stride = p->d_slab[i].us_split_stride;
      p->d_flat[i].grad_slabs = p->d_slab[i].grad_slabs;
    }
  }
}

extern "C" {

int rv_plan_create(rv_plan** out, long B, long S, long H, long L) {
  RV_REQUIRE(out, RV_ERR_NULL, "rv_plan_create: null out");
  rv_plan* p = new (std::nothrow) rv_plan();
  RV_REQUIRE(p, RV_ERR_STATE, "rv_plan_create: out of host memory");
  p->B = B; p->S = S; p->H = H; p->L = L;
  int rc = rv_pad_dims(B, S, H, L, &p->Bp, &p->Sp, &p->Hp, &p->Lp);
  if (rc) { delete p; return rc; }
  p->L2p = 2 * p->Lp;
  const long Bp = p->Bp, Sp = p->Sp, Hp = p->Hp, Lp = p->Lp, L2p = p->L2p;
  p->s_heads = splits_of(Bp, L2p, Hp);
  p->s_dz = splits_of(Bp, Lp, Hp);
  {
    // fc4 backward runs as one paired launch (dgrad + wgrad) when 256x256 tiles apply
    int paired = 0, bm = 128, sp = 1;
    rv_dgrad_wgrad_pick(Bp, Hp, Sp, &paired, &bm, &sp);
    p->s_w4 = sp;
    p->n_mt3 = (int)(Bp / bm);
  }
  p->s_w3 = splits_of(Hp, Lp, Bp);
  if (!rv_latent_rowlocal(Bp, Hp, Lp) && Hp % 128 == 0) {
    // GEMM form of the latent backward (csrc/latent.hip): dW3 runs beside the dz tiles on 128 x 128 (128 x 64 at a padded
    // latent width of 64) tiles; enough K splits that its blocks fill the chip once, at least 8 K tiles each
    const long tiles = (Hp / 128) * (Lp >= 128 ? Lp / 128 : 1), kt = Bp / 64;
    int s = 1;
    while (tiles * s < 256 && 2 * s <= 16 && kt % (2 * s) == 0 && kt / (2 * s) >= 8) s *= 2;
    if (rv_latent_bwd_pp(Bp, Hp, Lp)) {   // 256 x 256 ping-pong tiles: Hp / 256 of them per split, an even number of K tiles each
      s = 1;
      // (half a round of them: 16 splits measured 394 us for the launch at default.ini's shape, 8 and 32: 464, 435 -- and every
      // split is another 2 MB slab for Adam to read)
      while ((Hp / 256) * s < 128 && 2 * s <= 64 && kt % (4 * s) == 0 && kt / (2 * s) >= 16) s *= 2;
    }
    p->s_w3 = s;
  }
  {
    // heads backward (dgrad with the ReLU mask of h1 + wgrad) also goes through rv_linear_dgrad_wgrad
    int paired = 0, bm = 128, sp = 1;
    rv_dgrad_wgrad_pick(Bp, Hp, L2p, &paired, &bm, &sp);
    p->s_wh = p->s_wh_gen = sp;
    p->heads_pair_gen = paired;
    p->n_mt1 = p->n_mt1_gen = (int)(Bp / bm);
    p->hb_groups = (Lp == 64 && Bp % 512 == 0 && Hp % 64 == 0) ? (int)(Bp / 512) : 0;
  }
  p->s_w1 = splits_of(Hp, Sp, Bp);
  // per-row-tile partial counts follow the tile each producing GEMM will use
  for (int f8 = 0; f8 < 2; ++f8) {   // fc4 fwd: dP4 column sums (db4) and MSE partials
    int bm = 128, bn = 128;
    if (f8) rv_gemm_tile_fp8_loss(Bp, Sp, &bm, &bn);
    else rv_gemm_tile(Bp, Sp, 1, &bm, &bn);
    p->n_mt4_of[f8] = (int)(Bp / bm);
    p->n_mse_of[f8] = (int)((Bp / bm) * (Sp / bn));
  }
  p->n_mt4 = p->n_mt4_of[0];
  p->n_mse = p->n_mse_of[0];
  p->n_kl = (int)(Bp * Lp / 1024);
  const long sizes[10] = {H * S, H, L * H, L, L * H, L, H * L, H, S * H, S};
  long o = 0;
  for (int i = 0; i < 10; ++i) { p->off[i] = o; o += sizes[i]; }
  p->n_params = o;
  p->ws_bytes = 0;
  p->add("xb", Bp * Sp * 2);
  p->add("W1b", Hp * Sp * 2);
  p->add("Whb", L2p * Hp * 2);
  p->add("W3b", Hp * Lp * 2);
  p->add("W4b", Sp * Hp * 2);
  p->add("b1p", Hp * 4);
  p->add("bhp", L2p * 4);
  p->add("b3p", Hp * 4);
  p->add("b4p", Sp * 4);
  p->add("h1", Bp * Hp * 2);
  p->add("mulv_slabs", (long)p->s_heads * Bp * L2p * 4);
  p->add("mulv", Bp * L2p * 4);
  p->add("eps", Bp * Lp * 4);
  p->add("z", Bp * Lp * 2);
  p->add("h3", Bp * Hp * 2);
  p->add("dP4", Bp * Sp * 2);
  p->add("dP3", Bp * Hp * 2);
  p->add("dz_slabs", (long)p->s_dz * Bp * Lp * 4);
  p->add("dmulv", Bp * L2p * 2);
  p->add("dP1", Bp * Hp * 2);
  // data-parallel step: fc1's weight gradient has no optimizer riders beside it (the reduced gradients they would need
  // have not arrived), so it runs on ALL CUs with twice the K splits where the extents allow (256 blocks of 256 x 256
  // with an even number of >= 2 K tiles each)
  p->s_w1_ddp = p->s_w1;
  if (Hp % 256 == 0 && Sp % 256 == 0 && (Hp / 256) * (Sp / 256) * 2 * p->s_w1 <= 256 && (Bp / 64) % (2 * p->s_w1) == 0 &&
      (Bp / 64) / (2 * p->s_w1) >= 2 && ((Bp / 64) / (2 * p->s_w1)) % 2 == 0 && 2 * p->s_w1 <= 8)
    p->s_w1_ddp = 2 * p->s_w1;
  p->add("dW1", (long)p->s_w1_ddp * Hp * Sp * 4);
  p->add("dWh", (long)(p->s_wh_gen > p->hb_groups ? p->s_wh_gen : p->hb_groups) * L2p * Hp * 4);
  // fp16 dWh slabs: 2^-e per slab and 32 x 32 granule ([slabs][2 Lp / 32][Hp / 32])
  p->add("dWh_us", (long)(p->hb_groups > p->s_wh_gen ? p->hb_groups : p->s_wh_gen) * (L2p / 32) * (Hp / 32 + 1) * 4);
  p->add("dW3", (long)p->s_w3 * Hp * Lp * 4);
  p->add("dW4", (long)p->s_w4 * Sp * Hp * 4);
  p->add("dW1_us", (long)p->s_w1_ddp * (Hp / 32) * (Sp / 32) * 4);   // fp16 slabs: 2^-e per 32 x 32 granule and slab
  p->add("dW4_us", (long)p->s_w4 * (Sp / 32) * (Hp / 32) * 4);
  p->add("db1p", (long)(p->n_mt1_gen > p->hb_groups ? p->n_mt1_gen : p->hb_groups) * Hp * 4);
  p->add("dbhp", (Bp / 16) * L2p * 4);
  p->add("db3p", (long)p->n_mt3 * Hp * 4);
  p->add("db4p", (long)(p->n_mt4_of[0] > p->n_mt4_of[1] ? p->n_mt4_of[0] : p->n_mt4_of[1]) * Sp * 4);
  p->add("xq", Bp * Sp);          // fp8 operands of the fp8 forward path (RV_OPT_FP8)
  p->add("W1q", Hp * Sp);
  p->add("W4q", Sp * Hp);
  p->add("h3q", Bp * Hp);
  p->add("dP4q", Bp * Sp);        // fp8 image of dP4: the fp8 fc4 backward's operand (RV_OPT_FP8 = 1)
  p->add("dP1q", Bp * Hp);        // fp8 image of dP1: left operand of fc1's fp8 weight gradient (RV_OPT_FP8 = 1)
  p->add("fp8_state", (32 + 2 * 1024) * 4);   // 16 state floats (+16 pad), then 2 x 1024 max|W| slots
  // per-wave (fused latent forward: 8 per 16 batch rows) or per-tile max|h3| of the fc3 forward (zero until it has run)
  p->n_amax_cap = (int)(Bp / 2 > 4096 ? Bp / 2 : 4096);
  p->n_amax_dp1 = Bp % 512 == 0 ? (int)(8 * (Bp / 512) * (Hp / 64)) : 0;   // one maximum per wave of rv_heads_bwd_ex
  p->add("h3_amax", ((long)p->n_amax_cap + p->n_amax_dp1) * 4);            // h3's maxima, then dP1's right behind them
  p->add("ddp_flags", 64 * 4);   // data-parallel step: cross-stream sequence flags [0..3], timeout counter [8]
  p->add("mse_part", (long)(p->n_mse_of[0] > p->n_mse_of[1] ? p->n_mse_of[0] : p->n_mse_of[1]) * 4);
  p->add("kl_part", (long)p->n_kl * 4);
  p->bound = false;
  *out = p;
  return RV_OK;
}

void rv_plan_destroy(rv_plan* p) {
  if (!p) return;
  for (hipEvent_t e : p->ev_ready)
    if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : p->ev_done)
    if (e) (void)hipEventDestroy(e);
  if (p->ev_flush) (void)hipEventDestroy(p->ev_flush);
  // p->comm_stream is the process-wide collective stream (helper_stream below): not destroyed here
  delete p;
}

// The collective stream (highest priority) exists ONCE per process and is created on first need.  One per plan was
// measured harmful: every extra HIP stream may land on another of the runtime's few hardware queues
// (GPU_MAX_HW_QUEUES, default 4), and with an unlucky mapping every kernel of a data-parallel step started ~50 us
// late (profiles/r02_hw_queue_hazard.txt).  One process drives one GPU (header, Conventions).
static hipStream_t g_comm_stream = nullptr;
static int helper_stream(bool, hipStream_t* out) {
  if (!g_comm_stream) {
    int lo = 0, hi = 0;  // collectives ahead of compute when both are runnable
    RV_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
    RV_HIP(hipStreamCreateWithPriority(&g_comm_stream, hipStreamNonBlocking, hi));
  }
  *out = g_comm_stream;
  return RV_OK;
}

int rv_heads_reparam_fwd(const void* h_bf16, long ldh, const void* wh_bf16, long ldw, const float* bias_heads,
                         long Bp, long Lp, long Kp, long B, long L, int splits, float* mulv_slabs,
                         const float* eps_in, float* eps_out, unsigned long long seed, const long long* step_counter,
                         float* mulv, void* z_bf16, float* kl_partial, void* stream) {
  RV_REQUIRE(mulv_slabs, RV_ERR_NULL, "rv_heads_reparam_fwd: null slab workspace");
  const int rc = rv_linear_fwd_f32(h_bf16, ldh, wh_bf16, ldw, bias_heads, Bp, 2 * Lp, Kp, splits, mulv_slabs, 2 * Lp, stream);
  if (rc) return rc;
  return rv_reparam_fwd(mulv_slabs, splits, Bp, Lp, B, L, eps_in, eps_out, seed, step_counter, mulv, z_bf16, kl_partial,
                        stream);
}

int rv_plan_set_external_grads(rv_plan* p, const float* d_recon, const float* recon, const float* dmu,
                               const float* dlogvar, float* grad_out) {
  RV_REQUIRE(p, RV_ERR_NULL, "rv_plan_set_external_grads: null plan");
  RV_REQUIRE(!d_recon || recon, RV_ERR_NULL, "rv_plan_set_external_grads: d_recon needs recon (tanh')");
  p->ext_d_recon = d_recon; p->ext_recon = recon; p->ext_dmu = dmu; p->ext_dlv = dlogvar; p->ext_grad_out = grad_out;
  return RV_OK;
}

// The plan's own loss as an autograd node (the drop-in loop: loss_function on the untouched outputs of the fused forward).
// rv_plan_loss: (total, mse, kld) of the forward phase that ran last, into out3 -- one small launch, the summation order of
// the value the backward later writes to the loss ring.  rv_plan_set_loss_grad: the following backward + FINALIZE phases
// run on the forward's own fused loss gradient (no gradients from outside) and leave d_loss * gradient in `grad_out`
// (exact-shape fp32 [n_params]; NULL: the plan's grad arena), d_loss read from the device at finalize time.  Both NULL
// switches it off again.
int rv_plan_loss(rv_plan* p, float kl_beta, float* out3, void* stream) {
  RV_REQUIRE(p && p->bound && out3, RV_ERR_STATE, "rv_plan_loss: plan not bound / null output");
  return rv_loss_from_partials((const float*)p->ws("mse_part"), p->n_mse, (const float*)p->ws("kl_part"), p->n_kl, p->B, p->S,
                               p->L, kl_beta, out3, stream);
}

int rv_plan_set_loss_grad(rv_plan* p, const float* d_loss_dev, float* grad_out) {
  RV_REQUIRE(p, RV_ERR_NULL, "rv_plan_set_loss_grad: null plan");
  RV_REQUIRE(!(d_loss_dev && (p->ext_d_recon || p->ext_dmu || p->ext_dlv)), RV_ERR_STATE,
             "rv_plan_set_loss_grad: gradients from outside are set (rv_plan_set_external_grads)");
  p->loss_grad_dev = d_loss_dev;
  p->ext_grad_out = grad_out;
  return RV_OK;
}

static void fwd4_mode_apply(rv_plan* p) {
  const int f8 = p->fp8 ? 1 : 0;
  p->n_mt4 = p->n_mt4_of[f8];
  p->n_mse = p->n_mse_of[f8];
  p->d_slab[9].grad_splits = p->n_mt4;
  if (!p->b.grad) p->d_flat[9].grad_splits = p->n_mt4;
}

static int plan_set_fp8(rv_plan* p, int enable) {
  RV_REQUIRE(enable >= 0 && enable <= 2, RV_ERR_UNSUPPORTED, "rv_plan_set_option: RV_OPT_FP8 takes 0, 1 or 2 (got %d)", enable);
  p->fp8 = enable;   // 1: forward of fc1 / fc4 and backward of fc4; 2: forward only
  fwd4_mode_apply(p);
  float* st = (float*)p->ws("fp8_state");
  // Adam keeps the fp8 shadows of fc1.weight / fc4.weight current (descriptor 0 and 8)
  for (rv_param_desc* d : {p->d_slab, p->d_flat}) {
    d[0].shadow_fp8 = enable ? p->ws("W1q") : nullptr; d[0].fp8_scale = st + 1;
    d[8].shadow_fp8 = enable ? p->ws("W4q") : nullptr; d[8].fp8_scale = st + 2;
  }
  return RV_OK;
}

static int plan_set_slab_dtype(rv_plan* p, int slab_dtype) {
  RV_REQUIRE(slab_dtype == RV_SLAB_F32 || slab_dtype == RV_SLAB_F16, RV_ERR_UNSUPPORTED, "rv_plan_set_option: slab element type %d", slab_dtype);
  p->slab_dtype = slab_dtype;
  const long Hp = p->Hp, Sp = p->Sp;
  const bool half = slab_dtype == RV_SLAB_F16;
  rv_param_desc* d1 = p->d_slab + 0;   // fc1.weight [H, S] and fc4.weight [S, H]: the two 8 MB gradients
  d1->grad_half = half; d1->grad_unscale = half ? p->us_w1 : nullptr; d1->us_ld = Sp / 32; d1->us_split_stride = (Hp / 32) * (Sp / 32);
  rv_param_desc* d4 = p->d_slab + 8;
  d4->grad_half = half; d4->grad_unscale = half ? p->us_w4 : nullptr; d4->us_ld = Hp / 32; d4->us_split_stride = (Sp / 32) * (Hp / 32);
  if (!p->b.grad)   // without a flat gradient arena d_flat describes the same slabs (rv_plan_bind): keep their element type
    for (int i : {0, 8}) {
      p->d_flat[i].grad_half = p->d_slab[i].grad_half; p->d_flat[i].grad_unscale = p->d_slab[i].grad_unscale;
      p->d_flat[i].us_ld = p->d_slab[i].us_ld; p->d_flat[i].us_split_stride = p->d_slab[i].us_split_stride;
    }
  heads_mode_apply(p);   // the heads' slabs follow
  return RV_OK;
}

int rv_plan_set_option(rv_plan* p, int option, int value) {
  RV_REQUIRE(p && p->bound, RV_ERR_STATE, "rv_plan_set_option: plan not bound");
  switch (option) {
    case RV_OPT_LATENT_FUSED: p->latent_fused = value ? 1 : 0; heads_mode_apply(p); return RV_OK;
    case RV_OPT_DDP_DEFER_TAIL:
      RV_REQUIRE(value || !p->tail_pending, RV_ERR_STATE, "rv_plan_set_option: a deferred update is pending (rv_plan_ddp_flush first)");
      p->ddp_defer = value ? 1 : 0;
      return RV_OK;
    case RV_OPT_FP8: return plan_set_fp8(p, value);
    case RV_OPT_SLAB_DTYPE: return plan_set_slab_dtype(p, value);
    case RV_OPT_DDP_SIGNAL: p->ddp_signal = value ? 1 : 0; return RV_OK;
    case RV_OPT_DDP_W1_WIDE: p->ddp_w1_wide = value ? 1 : 0; return RV_OK;
    case RV_OPT_DDP_WAIT_MS:
      RV_REQUIRE(value >= 1, RV_ERR_SHAPE, "rv_plan_set_option: RV_OPT_DDP_WAIT_MS must be at least 1 (got %d)", value);
      p->ddp_wait_ms = value;
      return RV_OK;
    case RV_OPT_ROCTX:
      RV_REQUIRE(!value || roctx_load(), RV_ERR_UNSUPPORTED, "rv_plan_set_option: no roctx library (librocprofiler-sdk-roctx.so / libroctx64.so) could be loaded");
      p->roctx = value ? 1 : 0;
      return RV_OK;
  }
  RV_REQUIRE(false, RV_ERR_UNSUPPORTED, "rv_plan_set_option: unknown option %d", option);
  return RV_OK;
}

// Test / measurement hook (include/rawvae_hip_diag.h): leave launches out of the following full steps.
int rv_plan_diag_skip(rv_plan* p, unsigned mask) {
  RV_REQUIRE(p, RV_ERR_NULL, "rv_plan_diag_skip: null plan");
  p->skip = mask;
  return RV_OK;
}

long rv_plan_workspace_bytes(const rv_plan* p) { return p ? p->ws_bytes : 0; }

void* rv_plan_buffer(rv_plan* p, const char* name, long* n_bytes) {
  if (!p || !p->bound || !name) return nullptr;
  return p->ws(name, n_bytes);
}

int rv_plan_bind(rv_plan* p, const rv_plan_buffers* b) {
  RV_REQUIRE(p && b, RV_ERR_NULL, "rv_plan_bind: null");
  RV_REQUIRE(b->param && b->exp_avg && b->exp_avg_sq && b->workspace && b->step_counter && b->loss_ring &&
                 b->ring > 0,
             RV_ERR_NULL, "rv_plan_bind: missing buffer");
  RV_REQUIRE(((uintptr_t)b->workspace & 255) == 0, RV_ERR_SHAPE, "rv_plan_bind: workspace must be 256-byte aligned");
  p->b = *b;
  p->bound = true;
  const long H = p->H, S = p->S, L = p->L, Hp = p->Hp, Sp = p->Sp, Lp = p->Lp, L2p = p->L2p, Bp = p->Bp;
  float* dW1 = (float*)p->ws("dW1"); float* dWh = (float*)p->ws("dWh");
  float* dW3 = (float*)p->ws("dW3"); float* dW4 = (float*)p->ws("dW4");
  float* db1 = (float*)p->ws("db1p"); float* dbh = (float*)p->ws("dbhp");
  float* db3 = (float*)p->ws("db3p"); float* db4 = (float*)p->ws("db4p");
  char* W1b = p->ws("W1b"); char* Whb = p->ws("Whb"); char* W3b = p->ws("W3b"); char* W4b = p->ws("W4b");
  float* b1p = (float*)p->ws("b1p"); float* bhp = (float*)p->ws("bhp");
  float* b3p = (float*)p->ws("b3p"); float* b4p = (float*)p->ws("b4p");
  const int n_mt3 = p->n_mt3, n_mt1 = p->n_mt1, n_mt4 = p->n_mt4, n_b64 = (int)(Bp / 16);  // reparam_bwd: one partial row per 16 batch rows
  //                 offset     rows cols slabs        ld   split_stride  splits     bf16 shadow          f32 shadow  ld
  rv_param_desc d[10] = {
      {p->off[0], H, S, dW1, Sp, Hp * Sp, p->s_w1, W1b, nullptr, Sp},
      {p->off[1], 1, H, db1, Hp, Hp, n_mt1, nullptr, b1p, Hp},
      {p->off[2], L, H, dWh, Hp, L2p * Hp, p->s_wh, Whb, nullptr, Hp},
      {p->off[3], 1, L, dbh, L2p, L2p, n_b64, nullptr, bhp, L2p},
      {p->off[4], L, H, dWh + Lp * Hp, Hp, L2p * Hp, p->s_wh, Whb + Lp * Hp * 2, nullptr, Hp},
      {p->off[5], 1, L, dbh + Lp, L2p, L2p, n_b64, nullptr, bhp + Lp, L2p},
      {p->off[6], H, L, dW3, Lp, Hp * Lp, p->s_w3, W3b, nullptr, Lp},
      {p->off[7], 1, H, db3, Hp, Hp, n_mt3, nullptr, b3p, Hp},
      {p->off[8], S, H, dW4, Hp, Sp * Hp, p->s_w4, W4b, nullptr, Hp},
      {p->off[9], 1, S, db4, Sp, Sp, n_mt4, nullptr, b4p, Sp},
  };
  p->us_w1 = (float*)p->ws("dW1_us");
  p->us_w4 = (float*)p->ws("dW4_us");
  for (int i = 0; i < 10; ++i) {
    p->d_slab[i] = d[i];
    p->d_flat[i] = d[i];
    if (b->grad) {
      p->d_flat[i].grad_slabs = b->grad + d[i].offset;
      p->d_flat[i].grad_ld = d[i].cols;
      p->d_flat[i].grad_split_stride = 0;
      p->d_flat[i].grad_splits = 1;
    }
  }
  heads_mode_apply(p);
  fwd4_mode_apply(p);
  return plan_set_slab_dtype(p, p->slab_dtype);
}

// Which tensors' optimizer updates ride beside fc1's weight gradient in the full local step (launch 7: the GEMM fills half
// the chip for ~30 us, its rider blocks stream ~2.8 TB/s from the other half): tensors [first, last) of the table, the
// rest goes into the step's last launch, which runs on all CUs at ~4.8 TB/s.  [2, 10) (everything but fc1, ~91 MB) at C2's
// latent width of 64.  At the reference's own latent_dim = 256 that set is 136-152 MB and made launch 7 twice as long as
// its GEMM (60 us; profiles/r06_first_look.txt): there the heads and fc3 ride, [2, 8) = 65 MB, and the last launch takes
// fc1 and fc4 -- the two tensors whose four fp16 slabs stream at the optimizer's best rate: 229-231 us per step against
// 237 with fc3 + fc4 riding and 240-242 with everything but fc1 (one box, interleaved: profiles/r06_riders_ab.txt).
static long desc_bytes(const rv_param_desc& d) {
  const long el = d.rows * d.cols;
  return el * (d.shadow_bf16 ? 26 : 28) + el * d.grad_splits * (d.grad_half ? 2 : 4);
}
static void rider_range(const rv_plan* p, int* first, int* last) {
  const long budget = 100L * 1000 * 1000;
  const int cand[4][2] = {{2, 10}, {2, 8}, {6, 10}, {8, 10}};
  for (const auto& c : cand) {
    long n = 0;
    for (int i = c[0]; i < c[1]; ++i) n += desc_bytes(p->d_slab[i]);
    if (n <= budget) { *first = c[0]; *last = c[1]; return; }
  }
  *first = 8; *last = 10;
}

int rv_plan_riders(const rv_plan* p, int* first, int* last) {
  RV_REQUIRE(p && p->bound && first && last, RV_ERR_STATE, "rv_plan_riders: plan not bound");
  rider_range(p, first, last);
  return RV_OK;
}

int rv_plan_descs(const rv_plan* p, rv_param_desc* out10, int from_flat) {
  RV_REQUIRE(p && p->bound && out10, RV_ERR_STATE, "rv_plan_descs: plan not bound");
  for (int i = 0; i < 10; ++i) out10[i] = from_flat ? p->d_flat[i] : p->d_slab[i];
  return RV_OK;
}

int rv_plan_refresh_shadows(rv_plan* p, void* stream) {
  RV_REQUIRE(p && p->bound, RV_ERR_STATE, "rv_plan_refresh_shadows: plan not bound");
  hipStream_t st = (hipStream_t)stream;
  // after the first refresh the padding of every shadow is zero and stays zero (nothing writes it): one launch
  // rewrites the valid elements of all ten shadows from the parameter arena
  if (p->shadows_padded) return rv_params_from_flat(p->d_slab, 10, p->b.param, 0, nullptr, stream);
  for (int i = 0; i < 10; ++i) {
    const rv_param_desc& d = p->d_slab[i];
    const float* src = p->b.param + d.offset;
    if (d.shadow_bf16) {
      // weights: every [rows,cols] block is cast into its padded home; padding is zeroed
      long rows_p, cols_p = d.shadow_ld;
      if (i == 0 || i == 6) rows_p = p->Hp;
      else if (i == 8) rows_p = p->Sp;
      else rows_p = p->Lp;
      int rc = rv_cast_pad_bf16(src, d.rows, d.cols, d.cols, d.shadow_bf16, rows_p, cols_p, cols_p, nullptr, stream);
      if (rc) return rc;
      if (d.shadow_fp8) {
        rc = rv_cast_pad_fp8(src, d.rows, d.cols, d.cols, d.shadow_fp8, rows_p, cols_p, cols_p, d.fp8_scale, stream);
        if (rc) return rc;
      }
    } else {
      long pad = (i == 3 || i == 5) ? p->Lp : d.shadow_ld;
      RV_HIP(hipMemsetAsync(d.shadow_f32, 0, pad * sizeof(float), st));
      RV_HIP(hipMemcpyAsync(d.shadow_f32, src, d.cols * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
  }
  p->shadows_padded = true;
  return RV_OK;
}

// Launches issued inside a plan call write their epilogue outputs through (common.h store_wt16) while the step's activations
// are small enough for the next launch to find them in the L2s / MALL anyway: up to a padded batch of 8192.  Beyond that the
// consumer reads from HBM either way, and write-back lets the L2 merge the epilogues' 64-byte row segments into whole lines
// before they leave (same boxes, write-through against write-back per step: 178.8 / 185.8 us at B = 4096, 344 / 353 at 8192,
// 588 / 580 at 16384, 1397 / 1340 at 32768, 4757 / 4663 at 131072; L = 256 alike).  Scope guard:
struct WtScope {
  int prev;
  explicit WtScope(const rv_plan* p) : prev(rv::rv_store_wt) { rv::rv_store_wt = p->Bp <= 8192; }
  ~WtScope() { rv::rv_store_wt = prev; }
};

// fp8 forward: behind every optimizer update the maxima of the two fp8 weight shadows are measured for the next step's
// weight scales (RV_OPT_FP8, state block [8] / [9]).
static int fp8_after_update(rv_plan* p, void* stream) {
  if (!p->fp8) return RV_OK;
  return rv_fp8_wmax(p->ws("W1q"), p->Hp * p->Sp, p->ws("W4q"), p->Sp * p->Hp, (float*)p->ws("fp8_state"), stream);
}

// fc4's backward on fp8 operands (RV_OPT_FP8 = 1): the fp8 images of dP4 (written by the fc4 forward's epilogue), of
// W4 (the weight shadow Adam keeps) and of h3 (written by the fc3 forward) feed ONE 256 x 256 ping-pong launch whose K
// tiles are 128 deep -- half the LDS fill per flop of the bf16 pair, which is what bounds that loop.  Not with gradients
// from outside (rv_plan_set_external_grads: dP4 then comes from rv_tanh_bwd_pack in bf16) and only where the extents
// tile (256 x 256 tiles, an even count of 128-deep K tiles per block).
static bool fp8_bwd(const rv_plan* p) {
  return p->fp8 == 1 && !p->ext_d_recon && rv_dgrad_wgrad_fp8_fits(p->Bp, p->Hp, p->Sp, p->s_w4);
}
static bool fp8_bwd_possible(const rv_plan* p) {   // (at forward time: which images of dP4 to write)
  return p->fp8 == 1 && rv_dgrad_wgrad_fp8_fits(p->Bp, p->Hp, p->Sp, p->s_w4);
}
// fc1's weight gradient on fp8 operands (RV_OPT_FP8 = 1), in the schedules whose dW1 launch has an fp8 form -- the one
// with rider blocks: the full local step (riders = optimizer) and the data-parallel all-reduce step (riders = slab sums;
// it announces itself to its forward call through `fwd_for_fp8_w1`): the heads' streaming backward writes dP1 as fp8 (its
// scale follows the maximum it measured in the previous step), the frames' fp8 image is the one fc1's forward read, and
// neither bf16 copy is written.
static bool latent_bwd_fused(const rv_plan* p);
static bool fp8_w1(const rv_plan* p, bool full_local) {
  return p->fp8 == 1 && (full_local || p->fwd_for_fp8_w1) && latent_bwd_fused(p) && heads_streaming(p) && p->n_amax_dp1 > 0 &&
         rv_wgrad_adam_fits(p->Hp, p->Sp, p->Bp, p->s_w1) && (p->Hp / 256) * (p->Sp / 256) * p->s_w1 <= 192 &&
         p->Bp % (128L * p->s_w1) == 0 && (p->Bp / 128 / p->s_w1) % 2 == 0 && p->Hp % 16 == 0 && p->Sp % 16 == 0;
}
static int fc4_backward(rv_plan* p, void* stream) {
  const long Bp = p->Bp, Sp = p->Sp, Hp = p->Hp;
  if (fp8_bwd(p)) {
    float* f8 = (float*)p->ws("fp8_state");
    // the ReLU mask: the bf16 h3, or -- where the forward of this step did not write it (last_fwd_no_h3) -- the fp8 image
    // of h3 that is this launch's weight-gradient operand anyway
    const bool m8 = p->last_fwd_no_h3;
    return rv_linear_dgrad_wgrad_fp8(p->ws("dP4q"), Sp, p->ws("W4q"), Hp, p->ws("h3q"), Hp, m8 ? p->ws("h3q") : p->ws("h3"), Hp,
                                     m8 ? 1 : 0, f8 + 10, f8 + 11, Bp, Hp, Sp, p->ws("dP3"), Hp, (float*)p->ws("db3p"), p->ws("dW4"),
                                     Hp, p->s_w4, p->slab_dtype, p->us_w4, stream);
  }
  return rv_linear_dgrad_wgrad(p->ws("dP4"), Sp, p->ws("W4b"), Hp, p->ws("h3"), Hp, Bp, Hp, Sp, p->ws("dP3"), Hp,
                               (float*)p->ws("db3p"), p->ws("dW4"), Hp, p->s_w4, p->slab_dtype, p->us_w4, stream);
}

// The latent-sized backward between the fc4 pair and fc1's weight gradient: dz, the reparameterisation backward (which
// also finishes the loss), fc3's weight gradient, and the heads' dgrad + wgrad.  Row-local form (RV_OPT_LATENT_FUSED,
// padded latent width 64; its GEMM form above that): rv_latent_bwd (dz + reparam backward with dW3 on extra workgroups of
// the same launch) and the heads' backward -- two launches.  Otherwise three: dz + dW3 as split-K slabs, rv_reparam_bwd,
// the heads' backward.
// (padded latent width 64 at batches up to 8192: the row-local kernels, hidden width a multiple of 512 up to 2048; 128 / 256
// -- the reference's own latent_dim = 256 --, large batches and every other hidden width: the GEMM forms with the
// reparameterisation in their epilogues; rv_latent_rowlocal, csrc/latent.hip)
static bool latent_bwd_fused(const rv_plan* p) { return p->latent_fused && p->Hp % 128 == 0; }

static int latent_heads_bwd(rv_plan* p, const float* eps_used, float kl_beta, const float* dmu_ext, const float* dlv_ext,
                            void* stream, bool f8_w1 = false) {
  const long Bp = p->Bp, Hp = p->Hp, Lp = p->Lp, L2p = 2 * p->Lp, B = p->B, L = p->L, S = p->S;
  void* dP3 = p->ws("dP3"); void* z = p->ws("z"); void* h1 = p->ws("h1"); void* dP1 = p->ws("dP1"); void* dmulv = p->ws("dmulv");
  float* mulv = (float*)p->ws("mulv"); float* dz_slabs = (float*)p->ws("dz_slabs");
  float* mse_part = (float*)p->ws("mse_part"); float* kl_part = (float*)p->ws("kl_part");
  const bool do_latent = !(p->skip >> 5 & 1), do_heads = !(p->skip >> 6 & 1);   // rv_plan_diag_skip
  int rc;
  if (latent_bwd_fused(p)) {
    if (do_latent) {
      rc = rv_latent_bwd(dP3, Hp, p->ws("W3b"), Lp, Bp, Hp, Lp, B, L, S, mulv, eps_used, kl_beta, dmu_ext, dlv_ext, dmulv,
                         (float*)p->ws("dbhp"), mse_part, p->n_mse, kl_part, p->n_kl, p->b.loss_ring, p->b.step_counter,
                         p->b.ring, z, Lp, (float*)p->ws("dW3"), Lp, p->s_w3, stream);
      if (rc) return rc;
    }
    if (!do_heads) return RV_OK;
    if (f8_w1) {   // dP1 as fp8 only, its maxima behind h3's (fp8_w1)
      float* f8 = (float*)p->ws("fp8_state");
      return rv_heads_bwd_ex(dmulv, p->ws("Whb"), Hp, h1, Hp, Bp, Hp, Lp, nullptr, 0, (float*)p->ws("db1p"), (float*)p->ws("dWh"), Hp,
                             p->ws("dP1q"), Hp, f8 + 13, (float*)p->ws("h3_amax") + p->n_amax_h3,
                             p->heads_half ? (float*)p->ws("dWh_us") : nullptr, stream);
    }
    if (heads_streaming(p))
      return rv_heads_bwd_ex(dmulv, p->ws("Whb"), Hp, h1, Hp, Bp, Hp, Lp, dP1, Hp, (float*)p->ws("db1p"), (float*)p->ws("dWh"), Hp,
                             nullptr, 0, nullptr, nullptr, p->heads_half ? (float*)p->ws("dWh_us") : nullptr, stream);
    return rv_linear_dgrad_wgrad(dmulv, L2p, p->ws("Whb"), Hp, h1, Hp, Bp, Hp, L2p, dP1, Hp, (float*)p->ws("db1p"),
                                 p->ws("dWh"), Hp, p->s_wh, p->heads_half ? RV_SLAB_F16 : RV_SLAB_F32,
                                 p->heads_half ? (float*)p->ws("dWh_us") : nullptr, stream);
  }
  if (do_latent) {
    rc = rv_linear_dgrad_wgrad_f32(dP3, Hp, p->ws("W3b"), Lp, z, Lp, Bp, Lp, Hp, dz_slabs, Lp, p->s_dz, (float*)p->ws("dW3"), Lp,
                                   p->s_w3, stream);
    if (rc) return rc;
    rc = rv_reparam_bwd(dz_slabs, p->s_dz, Bp, Lp, B, L, S, mulv, eps_used, kl_beta, dmu_ext, dlv_ext, dmulv, (float*)p->ws("dbhp"),
                        mse_part, p->n_mse, kl_part, p->n_kl, p->b.loss_ring, p->b.step_counter, p->b.ring, stream);
    if (rc) return rc;
  }
  if (!do_heads) return RV_OK;
  return rv_linear_dgrad_wgrad(dmulv, L2p, p->ws("Whb"), Hp, h1, Hp, Bp, Hp, L2p, dP1, Hp, (float*)p->ws("db1p"), p->ws("dWh"), Hp,
                               p->s_wh, p->heads_half ? RV_SLAB_F16 : RV_SLAB_F32, p->heads_half ? (float*)p->ws("dWh_us") : nullptr,
                               stream);
}

int rv_plan_step(rv_plan* p, int phases, const float* x, const float* eps, float* recon_out,
                 float kl_beta, float lr, float grad_scale, int adam_from_flat,
                 unsigned long long seed, void* stream) {
  RV_REQUIRE(p && p->bound, RV_ERR_STATE, "rv_plan_step: plan not bound");
  if (p->tail_pending) {   // a data-parallel step left its last update to "the next call": this is it
    void* ts = p->tail_stream;
    const int frc = rv_plan_ddp_flush(p, nullptr);   // (on the stream that step was enqueued on)
    if (frc) return frc;
    if (ts != stream) {
      // ... and this step's kernels read the parameters and shadows that update writes: an edge from that stream to this
      // one (round-5 advisor: without it the forward could start while the deferred Adam was still running)
      if (!p->ev_flush) RV_HIP(hipEventCreateWithFlags(&p->ev_flush, hipEventDisableTiming));
      RV_HIP(hipEventRecord(p->ev_flush, (hipStream_t)ts));
      RV_HIP(hipStreamWaitEvent((hipStream_t)stream, p->ev_flush, 0));
    }
  }
  WtScope wt_scope(p);
  const long B = p->B, S = p->S, L = p->L, Bp = p->Bp, Sp = p->Sp, Hp = p->Hp, Lp = p->Lp, L2p = p->L2p;
  void* xb = p->ws("xb"); void* h1 = p->ws("h1"); void* z = p->ws("z"); void* h3 = p->ws("h3");
  void* dP4 = p->ws("dP4"); void* dP3 = p->ws("dP3"); void* dmulv = p->ws("dmulv"); void* dP1 = p->ws("dP1");
  float* mulv_slabs = (float*)p->ws("mulv_slabs"); float* mulv = (float*)p->ws("mulv");
  float* eps_buf = (float*)p->ws("eps"); float* dz_slabs = (float*)p->ws("dz_slabs");
  float* mse_part = (float*)p->ws("mse_part"); float* kl_part = (float*)p->ws("kl_part");
  const float* eps_used = eps ? eps : eps_buf;
  int rc;
#define RV_TRY(call) do { rc = (call); if (rc) return rc; } while (0)
#define RV_K(k, call) do { if (!(p->skip >> (k) & 1)) RV_TRY(call); } while (0)   /* launch k of the step (rv_plan_diag_skip) */
  const bool full_local = (phases & (RV_PHASE_BWD_A | RV_PHASE_BWD_B | RV_PHASE_ADAM)) ==
                              (RV_PHASE_BWD_A | RV_PHASE_BWD_B | RV_PHASE_ADAM) &&
                          !(phases & (RV_PHASE_FINALIZE_A | RV_PHASE_FINALIZE_B)) && !adam_from_flat;
  if (phases & RV_PHASE_FWD) {
    RV_REQUIRE(x, RV_ERR_NULL, "rv_plan_step: x is null");
    Range range_fwd(p->roctx, "rv:fwd");
    float* f8 = (float*)p->ws("fp8_state");
    // fp8 backward of fc4: the forward writes dP4 as fp8 (dP4q) INSTEAD of bf16 (a caller that then supplies its own
    // gradients gets its bf16 dP4 from rv_tanh_bwd_pack)
    const bool f8_bwd = fp8_bwd_possible(p);
    const bool f8_w1 = fp8_w1(p, full_local);     // then nothing reads the frames' bf16 copy: it is not written
    p->last_fwd_f8_w1 = f8_w1;
    // ... and nothing but fc4's dgrad reads the bf16 h3 then (its ReLU mask): the fp8 image serves (round 6: 16 MB less
    // written by the latent forward, 8 MB less read by the pair)
    const bool no_h3 = f8_w1 && f8_bwd && rv_latent_rowlocal(Bp, Hp, Lp);
    p->last_fwd_no_h3 = no_h3;
    void* xb_out = f8_w1 ? nullptr : xb;
    const int n_amax2 = f8_w1 ? p->n_amax_dp1 : 0;
    int n_amax = 0;
    // heads -> reparam -> fc3: one launch where the fused kernel exists (padded latent width 64), else three
    const bool latent_fused = latent_bwd_fused(p);   // (same shapes both ways)
    if (p->fp8 && latent_fused && rv_latent_rowlocal(Bp, Hp, Lp)) {
      n_amax = (int)(Bp / 16) * 8;   // one maximum per wave of rv_latent_fwd_ex
      RV_REQUIRE(n_amax <= p->n_amax_cap, RV_ERR_STATE, "rv_plan_step: h3_amax holds %d entries, the fused latent forward writes %d", p->n_amax_cap, n_amax);
    } else if (p->fp8) {
      int bm3 = 128, bn3 = 128;
      rv_gemm_tile(Bp, Hp, 1, &bm3, &bn3);
      n_amax = (int)((Bp / bm3) * (Hp / bn3));
      RV_REQUIRE(n_amax <= p->n_amax_cap, RV_ERR_SHAPE, "rv_plan_step: fp8 path supports up to %d fc3 output tiles (got %d)", p->n_amax_cap, n_amax);
    }
    p->n_amax_h3 = n_amax;
    // in place only when the padded frame length IS the frame length: with S < Sp the loader's columns S..Sp would be
    // the samples that follow the frame instead of zeros (harmless to fc1, whose weight columns there are zero, but
    // they would reach the framed copy and with it fc1's weight gradient and the exponents of its fp16 slabs)
    if (p->fr_hop && p->fr_bf16 && !p->fp8 && p->fr_hop % 8 == 0 && ((uintptr_t)p->fr_bf16 & 15) == 0 && S == Sp) {
      // N1 as SURVEY 8f words it: fc1's A-tile loader reads frame i at i * hop of the resident bf16 waveform; the
      // framed bf16 matrix dW1 needs later is a by-product of that launch; no cast / gather kernel
      RV_K(1, rv_linear_fwd_frames(p->fr_bf16, p->fr_idx, p->fr_first, p->fr_hop, B, p->ws("W1b"), Sp, (float*)p->ws("b1p"),
                                  Bp, Hp, Sp, RV_ACT_RELU, h1, Hp, xb, Sp, p->b.step_counter, stream));
    } else if (p->fr_hop) {
      // frames come straight from the resident waveform: waveform -> bf16 (and fp8) operand in one kernel
      RV_K(0, rv_gather_cast_frames(x, p->fr_nsamples, p->fr_idx, p->fr_first, B, S, p->fr_hop, xb_out, Bp, Sp, Sp,
                                   p->fp8 ? p->ws("xq") : nullptr, Sp, p->fp8 ? f8 : nullptr, (float*)p->ws("h3_amax"), n_amax,
                                   n_amax2, p->b.step_counter, stream));
      if (p->fp8)
        RV_K(1, rv_linear_fwd_fp8(p->ws("xq"), Sp, p->ws("W1q"), Sp, (float*)p->ws("b1p"), f8 + 5, Bp, Hp, Sp, RV_ACT_RELU,
                                 h1, Hp, stream));
      else
        RV_K(1, rv_linear_fwd_ex(xb, Sp, p->ws("W1b"), Sp, (float*)p->ws("b1p"), Bp, Hp, Sp, RV_ACT_RELU, h1, Hp, nullptr, 0,
                                nullptr, nullptr, stream));
    } else if (p->fp8) {
      RV_K(0, rv_cast_pad_bf16_q8(x, B, S, S, xb_out, Bp, Sp, Sp, p->ws("xq"), Sp, f8, (float*)p->ws("h3_amax"), n_amax,
                                 n_amax2, p->b.step_counter, stream));
      RV_K(1, rv_linear_fwd_fp8(p->ws("xq"), Sp, p->ws("W1q"), Sp, (float*)p->ws("b1p"), f8 + 5, Bp, Hp, Sp, RV_ACT_RELU,
                               h1, Hp, stream));
    } else {
      if (p->cast_done) p->cast_done = 0;   // went out ahead of the previous step's deferred update (rv_plan_step_ddp)
      else RV_K(0, rv_cast_pad_bf16(x, B, S, S, xb, Bp, Sp, Sp, p->b.step_counter, stream));
      RV_K(1, rv_linear_fwd_ex(xb, Sp, p->ws("W1b"), Sp, (float*)p->ws("b1p"), Bp, Hp, Sp, RV_ACT_RELU, h1, Hp, nullptr, 0,
                                nullptr, nullptr, stream));
    }
    {
      if (latent_fused)
        RV_K(2, rv_latent_fwd_ex(h1, Hp, p->ws("Whb"), Hp, (float*)p->ws("bhp"), p->ws("W3b"), Lp, (float*)p->ws("b3p"), Bp, Hp, Lp,
                                B, L, eps, eps_buf, seed, p->b.step_counter, mulv, z, kl_part, no_h3 ? nullptr : h3, Hp,
                                p->fp8 ? p->ws("h3q") : nullptr, Hp, p->fp8 ? f8 + 3 : nullptr,
                                p->fp8 ? (float*)p->ws("h3_amax") : nullptr, stream));
      else
        RV_K(2, rv_heads_reparam_fwd(h1, Hp, p->ws("Whb"), Hp, (float*)p->ws("bhp"), Bp, Lp, Hp, B, L, p->s_heads, mulv_slabs,
                                    eps, eps_buf, seed, p->b.step_counter, mulv, z, kl_part, stream));
      if (p->fr_hop) {
        if (p->fp8 && !latent_fused)
          RV_K(2, rv_linear_fwd_ex(z, Lp, p->ws("W3b"), Lp, (float*)p->ws("b3p"), Bp, Hp, Lp, RV_ACT_RELU, h3, Hp,
                                  p->ws("h3q"), Hp, f8 + 3, (float*)p->ws("h3_amax"), stream));
        else if (!latent_fused)
          RV_K(2, rv_linear_fwd_ex(z, Lp, p->ws("W3b"), Lp, (float*)p->ws("b3p"), Bp, Hp, Lp, RV_ACT_RELU, h3, Hp, nullptr, 0,
                                  nullptr, nullptr, stream));
        RV_K(3, rv_decode_out_loss_fwd_frames(p->fp8 ? p->ws("h3q") : h3, Hp, p->fp8 ? p->ws("W4q") : p->ws("W4b"), Hp,
                                             (float*)p->ws("b4p"), p->fp8 ? f8 + 6 : nullptr, Bp, Sp, Hp, B, S, x,
                                             p->fr_nsamples, p->fr_idx, p->fr_first, p->fr_hop, recon_out, S,
                                             f8_bwd ? nullptr : dP4, Sp, f8_bwd ? p->ws("dP4q") : nullptr, Sp, f8 + 12,
                                             mse_part, (float*)p->ws("db4p"), stream));
      } else if (p->fp8) {
        if (!latent_fused)
          RV_K(2, rv_linear_fwd_ex(z, Lp, p->ws("W3b"), Lp, (float*)p->ws("b3p"), Bp, Hp, Lp, RV_ACT_RELU, h3, Hp,
                                  p->ws("h3q"), Hp, f8 + 3, (float*)p->ws("h3_amax"), stream));
        RV_K(3, rv_decode_out_loss_fwd_fp8(p->ws("h3q"), Hp, p->ws("W4q"), Hp, (float*)p->ws("b4p"), f8 + 6, Bp, Sp, Hp, B, S,
                                          x, S, recon_out, S, f8_bwd ? nullptr : dP4, Sp, f8_bwd ? p->ws("dP4q") : nullptr, Sp,
                                          f8 + 12, mse_part, (float*)p->ws("db4p"), stream));
      } else {
        if (!latent_fused)
          RV_K(2, rv_linear_fwd_ex(z, Lp, p->ws("W3b"), Lp, (float*)p->ws("b3p"), Bp, Hp, Lp, RV_ACT_RELU, h3, Hp, nullptr, 0,
                                  nullptr, nullptr, stream));
        RV_K(3, rv_decode_out_loss_fwd(h3, Hp, p->ws("W4b"), Hp, (float*)p->ws("b4p"), Bp, Sp, Hp, B, S, x, S,
                                      recon_out, S, dP4, Sp, mse_part, (float*)p->ws("db4p"), stream));
      }
    }
  }
  // The latent layer's backward (dz + dW3, both read dP3) and the heads' backward (dP1 + dWh, both read
  // dmulv and h1) each go out as ONE launch (rv_linear_dgrad_wgrad_f32 / rv_linear_dgrad_wgrad).
  auto latent_bwd = [&](void* st) {
    return rv_linear_dgrad_wgrad_f32(dP3, Hp, p->ws("W3b"), Lp, z, Lp, Bp, Lp, Hp, dz_slabs, Lp, p->s_dz,
                                     (float*)p->ws("dW3"), Lp, p->s_w3, st);
  };
  auto heads_bwd = [&](void* st) {
    if (heads_streaming(p))
      return rv_heads_bwd_ex(dmulv, p->ws("Whb"), Hp, h1, Hp, Bp, Hp, Lp, dP1, Hp, (float*)p->ws("db1p"), (float*)p->ws("dWh"), Hp,
                             nullptr, 0, nullptr, nullptr, p->heads_half ? (float*)p->ws("dWh_us") : nullptr, st);
    return rv_linear_dgrad_wgrad(dmulv, L2p, p->ws("Whb"), Hp, h1, Hp, Bp, Hp, L2p, dP1, Hp, (float*)p->ws("db1p"),
                                 (float*)p->ws("dWh"), Hp, p->s_wh, p->heads_half ? RV_SLAB_F16 : RV_SLAB_F32,
                                 p->heads_half ? (float*)p->ws("dWh_us") : nullptr, st);
  };
  RV_REQUIRE(!(full_local && (p->ext_d_recon || p->ext_dmu || p->ext_dlv)), RV_ERR_STATE,
             "rv_plan_step: external gradients are set (rv_plan_set_external_grads); run the backward phases without ADAM");
  auto reparam_bwd = [&](void* st) {
    return rv_reparam_bwd(dz_slabs, p->s_dz, Bp, Lp, B, L, S, mulv, eps_used, kl_beta, p->ext_dmu, p->ext_dlv, dmulv,
                              (float*)p->ws("dbhp"), mse_part, p->n_mse, kl_part, p->n_kl, p->b.loss_ring,
                              p->b.step_counter, p->b.ring, st);
  };
  if (full_local && rv_wgrad_adam_fits(Hp, Sp, Bp, p->s_w1) && (Hp / 256) * (Sp / 256) * p->s_w1 <= 192) {
    // Default schedule, one stream.  dW1 is the last GEMM of the backward: 32 tiles x 4 K splits of 256x256 fill
    // half the chip, so its launch also carries the optimizer step of every tensor whose gradient is already
    // complete on the other CUs (fc21, fc22, fc3, fc4); fc1's update is the step's last launch.  An optimizer block
    // streams ~25 GB/s from its CU, so half the chip moves ~3 TB/s -- about what the GEMM blocks take to finish.
    const int n_gemm = (int)((Hp / 256) * (Sp / 256) * p->s_w1);
    int rf = 2, rl = 10;   // tensors [rf, rl) of the table: their updates ride beside fc1's weight gradient (rider_range)
    {
      Range r(p->roctx, "rv:fc4-bwd");
      RV_K(4, fc4_backward(p, stream));
    }
    {
      Range r(p->roctx, "rv:rest-bwd");
      // fc1's weight gradient on fp8 operands only when the FORWARD of this step prepared it (no bf16 copy of the frames,
      // dP1's delayed scale latched from the previous step's maxima): a forward enqueued by a call of its own
      // (RV_PHASE_FWD alone is not a full local step) wrote the bf16 copies, and this backward reads those
      const bool f8_w1 = p->last_fwd_f8_w1 && fp8_w1(p, true);
      RV_TRY(latent_heads_bwd(p, eps_used, kl_beta, nullptr, nullptr, stream, f8_w1));
      // (round 3, with 16-byte slab loads in the optimizer blocks: the heads' tensors ride as well -- 192.0 against
      // 194.8 us per step with only fc3 / fc4 riding, 196.3 with only fc4: profiles/r03_ab_step.txt)
      // (fp8 operands: only fc4's update riding here and the rest in the last launch was tried in round 5 -- 166.2-167.1 us
      // per step against 164.0-164.2 with the whole table riding and the GEMM blocks taking 15 % of it: profiles/r05_fp8_riders.txt)
      rider_range(p, &rf, &rl);
      if (f8_w1)
        RV_K(7, rv_linear_wgrad_adam_fp8(p->ws("dP1q"), Hp, p->ws("xq"), Sp, (float*)p->ws("fp8_state") + 15, Hp, Sp, Bp, p->s_w1,
                                         p->ws("dW1"), Sp, p->slab_dtype, p->us_w1, p->d_slab + rf, rl - rf, p->b.param, p->b.exp_avg,
                                         p->b.exp_avg_sq, lr, grad_scale, p->b.step_counter, 256 - n_gemm, stream));
      else
        RV_K(7, rv_linear_wgrad_adam(dP1, Hp, xb, Sp, Hp, Sp, Bp, p->s_w1, p->ws("dW1"), Sp, p->slab_dtype, p->us_w1, p->d_slab + rf,
                                     rl - rf, p->b.param, p->b.exp_avg, p->b.exp_avg_sq, lr, grad_scale,
                                     p->b.step_counter, 256 - n_gemm, stream));
    }
    Range r(p->roctx, "rv:adam");
    // the last launch: everything that did not ride ([0, rf) and [rl, 10), one table)
    rv_param_desc rest[10];
    int n_rest = 0;
    for (int i = 0; i < 10; ++i)
      if (i < rf || i >= rl) rest[n_rest++] = p->d_slab[i];
    RV_K(8, rv_adam_multi(rest, n_rest, p->b.param, p->b.exp_avg, p->b.exp_avg_sq, nullptr, nullptr, lr, grad_scale,
                          p->b.step_counter, stream));
    return fp8_after_update(p, stream);
  }
  // ---- backward / finalize / Adam as an ordered list of steps, each enabled by the phase mask ----
  const bool old_a = phases & RV_PHASE_BWD_A, old_b = phases & RV_PHASE_BWD_B;
  const bool do_pair = old_a || (phases & RV_PHASE_BWD_FC4);
  const bool do_chain_a = old_a || (phases & RV_PHASE_BWD_CHAIN);   // dz, reparam_bwd
  const bool do_chain_b = old_b || (phases & RV_PHASE_BWD_CHAIN);   // heads dgrad + wgrad (one launch), fc1 wgrad
  const bool do_w3 = old_a || (phases & RV_PHASE_BWD_REST);         // fc3 wgrad
  if (do_pair && p->ext_d_recon) {
    // dP4 = d_recon * (1 - recon^2) and its column sums (fc4.bias) from the caller's gradient instead of the
    // forward's fused MSE gradient; the partial-sum rows this does not write are zero
    RV_HIP(hipMemsetAsync(p->ws("db4p"), 0, (size_t)p->n_mt4 * Sp * sizeof(float), (hipStream_t)stream));
    RV_TRY(rv_tanh_bwd_pack(p->ext_d_recon, p->ext_recon, B, S, dP4, Bp, Sp, stream));
    RV_TRY(rv_colsum_partial(dP4, 1, Bp, Sp, Sp, (float*)p->ws("db4p"), Sp, stream));
  }
  if (do_pair) RV_K(4, fc4_backward(p, stream));
  bool w3_done = false;
  if (do_chain_a && do_chain_b && do_w3) {
    RV_TRY(latent_heads_bwd(p, eps_used, kl_beta, p->ext_dmu, p->ext_dlv, stream));
    w3_done = true;
  } else {
    if (do_chain_a && latent_bwd_fused(p)) {
      RV_TRY(rv_latent_bwd(dP3, Hp, p->ws("W3b"), Lp, Bp, Hp, Lp, B, L, S, mulv, eps_used, kl_beta, p->ext_dmu, p->ext_dlv, dmulv,
                           (float*)p->ws("dbhp"), mse_part, p->n_mse, kl_part, p->n_kl, p->b.loss_ring, p->b.step_counter,
                           p->b.ring, do_w3 ? z : nullptr, Lp, (float*)p->ws("dW3"), Lp, p->s_w3, stream));
      w3_done = do_w3;
    } else if (do_chain_a) {
      if (do_w3) {
        RV_TRY(latent_bwd(stream));
        w3_done = true;
      } else {
        RV_TRY(rv_linear_dgrad(dP3, Hp, p->ws("W3b"), Lp, Bp, Lp, Hp, nullptr, 0, nullptr, 0, nullptr, dz_slabs, Lp,
                               p->s_dz, stream));
      }
      RV_TRY(reparam_bwd(stream));
    }
    if (do_chain_b) RV_TRY(heads_bwd(stream));
  }
  if (do_chain_b) {
    RV_K(7, rv_linear_wgrad(dP1, Hp, xb, Sp, Hp, Sp, Bp, p->s_w1, w1_tile(p), p->ws("dW1"), Sp, p->slab_dtype, p->us_w1, stream));
  }
  if (do_w3 && !w3_done) RV_TRY(rv_linear_wgrad(dP3, Hp, z, Lp, Hp, Lp, Bp, p->s_w3, RV_TILE_AUTO, p->ws("dW3"), Lp, RV_SLAB_F32, nullptr, stream));

  // tensor masks (bit i = parameter i in state_dict order)
  unsigned fin = 0, adam = 0;
  if (phases & RV_PHASE_FINALIZE_A) fin |= 0x3C0;  // fc3, fc4
  if (phases & RV_PHASE_FINALIZE_B) fin |= 0x03F;  // fc1, fc21, fc22
  if (phases & RV_PHASE_FIN_FC4) fin |= 0x300;
  if (phases & RV_PHASE_FIN_FC1) fin |= 0x003;
  if (phases & RV_PHASE_FIN_MID) fin |= 0x0FC;
  if (phases & RV_PHASE_ADAM) adam |= 0x3FF;
  if (phases & RV_PHASE_ADAM_A) adam |= 0x3C0;
  if (phases & RV_PHASE_ADAM_B) adam |= 0x03F;
  if (phases & RV_PHASE_ADAM_FC4) adam |= 0x300;
  if (phases & RV_PHASE_ADAM_FC1) adam |= 0x003;
  if (phases & RV_PHASE_ADAM_MID) adam |= 0x0FC;
  float* fin_out = p->ext_grad_out ? p->ext_grad_out : p->b.grad;
  if (fin) RV_REQUIRE(fin_out, RV_ERR_STATE, "rv_plan_step: FINALIZE needs a grad arena");
  if (adam) RV_REQUIRE(!adam_from_flat || p->b.grad, RV_ERR_STATE, "rv_plan_step: adam_from_flat needs a grad arena");
  const rv_param_desc* ad = adam_from_flat ? p->d_flat : p->d_slab;
  for (int i = 0; i < 10;) {   // contiguous runs of selected tensors -> one launch each
    if (!((fin >> i) & 1)) { ++i; continue; }
    int j = i;
    while (j < 10 && ((fin >> j) & 1)) ++j;
    RV_TRY(rv_grad_finalize_scaled(p->d_slab + i, j - i, fin_out, 0, p->loss_grad_dev, stream));
    i = j;
  }
  for (int i = 0; i < 10;) {
    if (!((adam >> i) & 1)) { ++i; continue; }
    int j = i;
    while (j < 10 && ((adam >> j) & 1)) ++j;
    RV_K(8, rv_adam_multi(ad + i, j - i, p->b.param, p->b.exp_avg, p->b.exp_avg_sq, nullptr, nullptr, lr, grad_scale,
                          p->b.step_counter, stream));
    i = j;
  }
  if (adam & 0x101) RV_TRY(fp8_after_update(p, stream));   // fc1.weight or fc4.weight were updated
#undef RV_K
#undef RV_TRY
  return RV_OK;
}

int rv_plan_step_frames(rv_plan* p, int phases, const float* audio, const void* audio_bf16, long n_samples,
                        const long long* frame_index, long first_frame, long hop, const float* eps, float* recon_out,
                        float kl_beta, float lr, float grad_scale, int adam_from_flat, unsigned long long seed, void* stream) {
  RV_REQUIRE(p && p->bound, RV_ERR_STATE, "rv_plan_step_frames: plan not bound");
  RV_REQUIRE(audio && n_samples > 0 && hop > 0, RV_ERR_SHAPE, "rv_plan_step_frames: bad waveform / hop");
  p->fr_idx = frame_index; p->fr_first = first_frame; p->fr_hop = hop; p->fr_nsamples = n_samples; p->fr_bf16 = audio_bf16;
  const int rc = rv_plan_step(p, phases, audio, eps, recon_out, kl_beta, lr, grad_scale, adam_from_flat, seed, stream);
  p->fr_idx = nullptr; p->fr_first = 0; p->fr_hop = 0; p->fr_nsamples = 0; p->fr_bf16 = nullptr;
  return rc;
}

// ------------------------------------------------------------ data-parallel step
int rv_plan_attach_comm(rv_plan* p, const rv_comm_desc* c) {
  RV_REQUIRE(p && c, RV_ERR_NULL, "rv_plan_attach_comm: null argument");
  if (!c->comm) {   // detach: the stream choice is the one field that may be set ahead of a communicator
    p->allreduce = nullptr; p->comm = nullptr;
    p->comm_stream = (hipStream_t)c->comm_stream;
    return RV_OK;
  }
  RV_REQUIRE(c->world >= 1 && c->rank >= 0 && c->rank < c->world, RV_ERR_SHAPE, "rv_plan_attach_comm: rank %d of %d", c->rank, c->world);
  RV_REQUIRE(c->allreduce, RV_ERR_NULL, "rv_plan_attach_comm: no collective given");
  p->comm_stream = (hipStream_t)c->comm_stream;   // NULL: the library's own
  if (!p->comm_stream) {
    const int src = helper_stream(true, &p->comm_stream);
    if (src) return src;
  }
  if (!p->ev_ready[0]) {
    // (hipEventReleaseToDevice on the fork event was tried: no change in the ~6 us an event costs the compute stream)
    for (hipEvent_t& e : p->ev_ready) RV_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (hipEvent_t& e : p->ev_done) RV_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  p->allreduce = c->allreduce;
  p->comm = c->comm; p->world = c->world; p->rank = c->rank;
  p->grad_bf16 = c->grad_bf16; p->payload_bf16 = c->grad_bf16 ? 1 : 0;
  return RV_OK;
}

// All-reduce schedule (round 4; DESIGN.md section 5 has the timeline and the model it was sized against):
//   * two buckets, fc4 | everything else, BOTH exchanged on the collective stream, in order (one communicator, one
//     stream): fc4's forks off behind the paired fc4 backward and the compute stream never waits for it before the
//     backward is complete -- the exchange has latent backward + heads backward + fc1's weight gradient (~55 us) to
//     hide behind;
//   * behind the backward the compute stream sums the second bucket's slabs, hands it to the collective stream, and runs
//     Adam(fc4) from the reduced payload while that exchange is on the links -- the update of the step's largest tensor
//     costs nothing and disturbs no GEMM; only Adam(fc1, heads, fc3) is left behind the last byte;
//   * the cross-stream edges are device-side flags, not HIP events: an event costs the compute stream a ~5 us bubble
//     per record and ~9 us from record to the dependent kernel on the other stream, a flag ~1.8 us per crossing
//     (measured); two of the four edges are on the critical path;
//   * cutting fc1's gradient into halves buys nothing once each collective pays its own start-up latency, and an early
//     small bucket (heads + fc3) only delays the last one (modelled: tools/ddp_model.py).
// The deferred half of an all-reduce step (RV_OPT_DDP_DEFER_TAIL): wait for the second exchange, update its bucket.
// The step number comes from the copy edge 1's flag kernel latched (ddp_flags[16..17]): the device counter itself may
// already have been bumped by the next step's cast launch.
static int ddp_finish_tail(rv_plan* p) {
  void* stream = p->tail_stream;
  {
    hipStreamCaptureStatus cap0 = hipStreamCaptureStatusNone;
    RV_HIP(hipStreamIsCapturing((hipStream_t)stream, &cap0));
    RV_REQUIRE(cap0 == hipStreamCaptureStatusNone, RV_ERR_STATE,
               "a deferred data-parallel update cannot be enqueued into a stream capture (rv_plan_ddp_flush before capturing)");
  }
  int* fl = (int*)p->ws("ddp_flags");
  int rc = rv_flag_wait(fl + 3, p->tail_seq, fl + 8, p->ddp_wait_ms, stream);
  if (rc) return rc;
  rc = rv_adam_multi_guarded(p->d_flat, 8, p->b.param, p->b.exp_avg, p->b.exp_avg_sq, nullptr,
                             p->payload_bf16 ? p->grad_bf16 : nullptr, p->tail_lr, p->tail_scale,
                             (const long long*)(fl + 16), fl + 8, stream);
  p->tail_pending = 0;
  return rc;
}

int rv_plan_ddp_flush(rv_plan* p, void* stream) {
  RV_REQUIRE(p && p->bound, RV_ERR_STATE, "rv_plan_ddp_flush: plan not bound");
  if (!p->tail_pending) return RV_OK;
  RV_REQUIRE(!stream || stream == p->tail_stream, RV_ERR_STATE,
             "rv_plan_ddp_flush: the deferred half of a step belongs on the stream that step was enqueued on (pass it, or NULL)");
  WtScope wt_scope(p);
  return ddp_finish_tail(p);
}

int rv_plan_step_ddp(rv_plan* p, const float* x, const float* eps, float* recon_out, float kl_beta, float lr,
                     unsigned long long seed, void* stream) {
  RV_REQUIRE(p && p->bound, RV_ERR_STATE, "rv_plan_step_ddp: plan not bound");
  RV_REQUIRE(p->allreduce && p->comm, RV_ERR_STATE, "rv_plan_step_ddp: no communicator attached (rv_plan_attach_comm)");
  RV_REQUIRE(p->b.grad, RV_ERR_STATE, "rv_plan_step_ddp: needs a grad arena (the all-reduce payload)");
  WtScope wt_scope(p);
  RV_REQUIRE(stream, RV_ERR_NULL, "rv_plan_step_ddp: needs a non-default stream");
  if (p->tail_pending) {
    // the previous step's deferred half: this step's cast first (it reads x and writes the bf16 frames, nothing else --
    // fc1's weight gradient, the frames' last reader, is long done), then the wait for the exchange and the update
    RV_REQUIRE(x, RV_ERR_NULL, "rv_plan_step_ddp: x is null");
    RV_REQUIRE(stream == p->tail_stream, RV_ERR_STATE,
               "rv_plan_step_ddp: the previous step deferred its last update on another stream (rv_plan_ddp_flush it first)");
    {
      // a captured graph would re-apply the previous step's update on every replay
      hipStreamCaptureStatus cap0 = hipStreamCaptureStatusNone;
      RV_HIP(hipStreamIsCapturing((hipStream_t)stream, &cap0));
      RV_REQUIRE(cap0 == hipStreamCaptureStatusNone, RV_ERR_STATE,
                 "rv_plan_step_ddp: a deferred update is pending and the stream is capturing (rv_plan_ddp_flush before the capture)");
    }
    if (!p->fp8 && !(p->skip & 1)) {
      const int crc = rv_cast_pad_bf16(x, p->B, p->S, p->S, p->ws("xb"), p->Bp, p->Sp, p->Sp, p->b.step_counter, stream);
      if (crc) return crc;
      p->cast_done = 1;
    }
    // (a failure here leaves cast_done set: the cast HAS run and bumped the device step counter, so a retried call must not
    // run it -- and count the step -- again; round-5 advisor)
    const int frc = ddp_finish_tail(p);
    if (frc) return frc;
  }
  const long Bp = p->Bp, Sp = p->Sp, Hp = p->Hp;
  void* xb = p->ws("xb"); void* dP1 = p->ws("dP1");
  const float* eps_used = eps ? eps : (float*)p->ws("eps");
  hipStream_t s0 = (hipStream_t)stream, sc = p->comm_stream;
  const float scale = 1.0f / (float)p->world;
  int rc;
#define RV_TRY(call) do { rc = (call); if (rc) return rc; } while (0)
  // fc1's weight gradient with its finalize riders has an fp8 form (the wide form and the other tiles have none)
  const int s_w1 = (p->ddp_w1_wide && w1_tile(p) == RV_TILE_256x256) ? p->s_w1_ddp : p->s_w1;
  const int n_gemm = (int)((Hp / 256) * (Sp / 256) * s_w1);
  const bool riders = w1_tile(p) == RV_TILE_256x256 && n_gemm <= 192;
  // Bucket = tensors [t0, t1) of the flat arena: slabs -> flat payload (caller's stream), then the SUM over ranks
  // on stream `on`
  // fc1's weight gradient on all CUs (see rv_plan_create): its descriptor for the payload kernel carries the split count
  rv_param_desc dd[10];
  for (int i = 0; i < 10; ++i) dd[i] = p->d_slab[i];
  dd[0].grad_splits = s_w1;
  auto payload = [&](int t0, int t1, hipStream_t on) -> int {
    if (p->payload_bf16) return rv_grad_finalize(dd + t0, t1 - t0, p->grad_bf16, 1, (void*)on);
    return rv_grad_finalize(dd + t0, t1 - t0, p->b.grad, 0, (void*)on);
  };
  auto reduce = [&](int b, int t0, int t1, hipStream_t on) -> int {
    const long lo = p->off[t0], hi = t1 < 10 ? p->off[t1] : p->n_params;
    int nrc;
    if (p->payload_bf16) {
      char* g = (char*)p->grad_bf16 + 2 * lo;
      nrc = p->allreduce(g, g, (size_t)(hi - lo), /*ncclBfloat16*/ 9, /*ncclSum*/ 0, p->comm, (void*)on);
    } else {
      nrc = p->allreduce(p->b.grad + lo, p->b.grad + lo, (size_t)(hi - lo), /*ncclFloat32*/ 7, /*ncclSum*/ 0,
                         p->comm, (void*)on);
    }
    if (nrc != 0) return rv_fail(RV_ERR_HIP, "all-reduce of gradient bucket %d failed (collective library code %d)", b, nrc);
    return RV_OK;
  };
  // (guarded: a flag wait in front of an update that ran out has left a non-zero count in ddp_flags[8]; from then on no
  // update is applied -- the parameters stay what the last complete exchange made them -- until the host has seen the
  // count and raised: a partial all-reduce never reaches the weights)
  const int* poison = (const int*)p->ws("ddp_flags") + 8;
  auto adam_bucket = [&](int t0, int n) -> int {
    return rv_adam_multi_guarded(p->d_flat + t0, n, p->b.param, p->b.exp_avg, p->b.exp_avg_sq, nullptr,
                                 p->payload_bf16 ? p->grad_bf16 : nullptr, lr, scale, p->b.step_counter, poison, stream);
  };
  // forward + loss and the paired fc4 backward (as rv_plan_step)
  p->fwd_for_fp8_w1 = riders && s_w1 == p->s_w1;
  const bool f8_w1 = fp8_w1(p, false);
  rc = rv_plan_step(p, RV_PHASE_FWD, x, eps, recon_out, kl_beta, lr, 1.f, 0, seed, stream);
  p->fwd_for_fp8_w1 = false;
  if (rc) return rc;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  RV_HIP(hipStreamIsCapturing(s0, &cap));
  bool edge0_armed = false;   // the paired launch itself signals ev_ready[0] (see `signal`, edge 0)
  {
    Range r(p->roctx, "rv:fc4-bwd");
    if (cap == hipStreamCaptureStatusNone) rv_pair_stop_event(p->ev_ready[0]);
    rc = fc4_backward(p, stream);
    edge0_armed = cap == hipStreamCaptureStatusNone && !rv_pair_stop_event(nullptr);   // taken by a paired launch
    if (rc) return rc;
  }
  Range range_rest(p->roctx, "rv:rest-bwd+exchange+adam");
  // Cross-stream edges: device-side flags (elementwise.hip, k_flag_set / k_flag_wait; ~1.8 us per crossing) when
  // launching eagerly, HIP events (~9 us per crossing, ~5 us of bubble per record on the compute stream) under stream
  // capture -- a captured graph needs the event edges to know the collective stream belongs to it -- or when
  // RV_OPT_DDP_SIGNAL is 0.  Edge 0: fc4's slabs complete (s0 -> sc); 1: second payload complete (s0 -> sc);
  // 2: fc4's exchange done (sc -> s0); 3: second exchange done (sc -> s0).  Edges 1 and 3 are on the critical path.
  const bool flags = p->ddp_signal && cap == hipStreamCaptureStatusNone;
  int* fl = (int*)p->ws("ddp_flags");
  const int seq = flags ? ++p->ddp_seq : 0;
  constexpr long LOCAL_WAIT_MS = 5000;
  auto signal = [&](int edge, hipStream_t from, hipStream_t to) -> int {
    // Edge 0 is an event even with flags on: its waiter would sit on the collective stream from the end of the previous
    // step, spinning on one wave slot of one CU all through the forward and the paired fc4 backward -- and that kernel
    // needs EVERY CU whole (256 workgroups, two 256-VGPR waves per SIMD): with one CU short it runs in two rounds
    // (measured: 34 -> 60 us).  The other waiters start spinning late in the backward, beside kernels that leave room.
    if (flags && edge != 0) {
      // (edge 1 also latches the step number for a deferred update: see ddp_finish_tail)
      RV_TRY(rv_flag_set_copy(fl + edge, seq, edge == 1 ? p->b.step_counter : nullptr, (long long*)(fl + 16), (void*)from));
      return rv_flag_wait(fl + edge, seq, fl + 8, LOCAL_WAIT_MS, (void*)to);   // edges 0, 1: set behind this device's own kernels
    }
    hipEvent_t e = edge < 2 ? p->ev_ready[edge] : p->ev_done[edge - 2];
    // (edge 0 outside a capture: the event IS the completion signal of the paired launch -- rv_pair_stop_event -- which
    // leaves 3.7 us of bubble behind that kernel instead of the 5.7 of a record behind it)
    if (!(edge == 0 && edge0_armed)) RV_HIP(hipEventRecord(e, from));
    RV_HIP(hipStreamWaitEvent(to, e, 0));
    return RV_OK;
  };
  // Edges whose consumer is enqueued later than the signal (the compute stream joins after more of its own work):
  // the two halves of `signal`
  auto post = [&](int edge, hipStream_t from) -> int {
    if (flags) return rv_flag_set(fl + edge, seq, (void*)from);
    RV_HIP(hipEventRecord(edge < 2 ? p->ev_ready[edge] : p->ev_done[edge - 2], from));
    return RV_OK;
  };
  auto await = [&](int edge, hipStream_t on) -> int {
    if (flags) return rv_flag_wait(fl + edge, seq, fl + 8, p->ddp_wait_ms, (void*)on);   // edges 2, 3: set behind a collective
    RV_HIP(hipStreamWaitEvent(on, edge < 2 ? p->ev_ready[edge] : p->ev_done[edge - 2], 0));
    return RV_OK;
  };
  // (Tried: the second bucket's payload kernel publishing edge 1's flag from its last workgroup instead of a k_flag_set
  // launch behind it.  Every workgroup then needs an agent-scope release of its own before it counts itself done -- an L2
  // write-back each, 1000 of them: +27 us per step.  The kernel boundary does that once.)
  RV_TRY(signal(0, s0, sc));                               // fork 1: fc4 (8.4 MB of gradient at C2) is summed over its
  RV_TRY(payload(8, 10, sc));                              // slabs and travels behind ALL the rest of the backward
  RV_TRY(reduce(0, 8, 10, sc));
  RV_TRY(post(2, sc));
  RV_TRY(latent_heads_bwd(p, eps_used, kl_beta, nullptr, nullptr, stream, f8_w1));
  // dW1 runs WITHOUT optimizer riders here: at several ranks fc4's sum has not arrived when this launch starts (an 8.4 MB
  // bucket needs 40-65 us on the links; the latent-sized backward in front of this launch lasts 25).  It keeps the local
  // step's 128 workgroups by default: this launch runs beside fc4's exchange, whose workgroups hold CUs, and a GEMM that
  // needs every CU whole then runs in two rounds (RV_OPT_DDP_W1_WIDE; modelled both ways in tools/ddp_model.py).
  if (riders) {
    // the GEMM leaves CUs idle: rider blocks sum the slabs of everything else in the second bucket (fc1.bias, heads, fc3:
    // complete since the heads' backward) into the payload meanwhile, and only fc1.weight's own slabs are left to sum
    void* pay = p->payload_bf16 ? p->grad_bf16 : (void*)p->b.grad;
    if (f8_w1)
      RV_TRY(rv_linear_wgrad_finalize_fp8(p->ws("dP1q"), Hp, p->ws("xq"), Sp, (float*)p->ws("fp8_state") + 15, Hp, Sp, Bp, s_w1,
                                          p->ws("dW1"), Sp, p->slab_dtype, p->us_w1, dd + 1, 7, pay, p->payload_bf16, 256 - n_gemm,
                                          stream));
    else
      RV_TRY(rv_linear_wgrad_finalize(dP1, Hp, xb, Sp, Hp, Sp, Bp, s_w1, p->ws("dW1"), Sp, p->slab_dtype, p->us_w1, dd + 1, 7,
                                      pay, p->payload_bf16, 256 - n_gemm, stream));
    RV_TRY(payload(0, 1, s0));
  } else {
    RV_TRY(rv_linear_wgrad(dP1, Hp, xb, Sp, Hp, Sp, Bp, s_w1, w1_tile(p), p->ws("dW1"), Sp, p->slab_dtype, p->us_w1, stream));
    RV_TRY(payload(0, 8, s0));                             // fc1, fc21, fc22, fc3: contiguous in the arena
  }
  RV_TRY(signal(1, s0, sc));                               // fork 2: the second exchange follows the first on the
  RV_TRY(reduce(1, 0, 8, sc));                             // collective stream (one communicator, one stream, in order)
  RV_TRY(post(3, sc));
  RV_TRY(await(2, s0));                                    // fc4's sum has arrived (long ago, if the links keep up): its
  RV_TRY(adam_bucket(8, 2));                               // update runs while the second exchange is on the links
  if (flags && p->ddp_defer && !p->fp8) {
    // the join and the last update wait for the next call (or rv_plan_ddp_flush): see RV_OPT_DDP_DEFER_TAIL
    p->tail_pending = 1; p->tail_seq = seq; p->tail_lr = lr; p->tail_scale = scale; p->tail_stream = stream;
    return RV_OK;
  }
  RV_TRY(await(3, s0));                                    // the join
  RV_TRY(adam_bucket(0, 8));
  RV_TRY(fp8_after_update(p, stream));
#undef RV_TRY
  return RV_OK;
}

// --------------------------------------------------------------------- hipGraph
struct rv_graph {
  hipGraph_t graph;
  hipGraphExec_t exec;
};

int rv_graph_begin(void* stream) {
  RV_REQUIRE(stream, RV_ERR_NULL, "rv_graph_begin: capture needs a non-default stream");
  RV_HIP(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
  return RV_OK;
}

int rv_graph_end(void* stream, rv_graph** out) {
  RV_REQUIRE(stream && out, RV_ERR_NULL, "rv_graph_end: null");
  hipGraph_t g = nullptr;
  RV_HIP(hipStreamEndCapture((hipStream_t)stream, &g));
  RV_REQUIRE(g, RV_ERR_STATE, "rv_graph_end: capture produced no graph");
  hipGraphExec_t e = nullptr;
  hipError_t err = hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
  if (err != hipSuccess) {
    (void)hipGraphDestroy(g);
    return rv_fail(RV_ERR_HIP, "hipGraphInstantiate -> %s", hipGetErrorString(err));
  }
  rv_graph* r = new (std::nothrow) rv_graph{g, e};
  RV_REQUIRE(r, RV_ERR_STATE, "rv_graph_end: out of host memory");
  *out = r;
  return RV_OK;
}

int rv_graph_launch(rv_graph* g, void* stream) {
  RV_REQUIRE(g, RV_ERR_NULL, "rv_graph_launch: null graph");
  RV_HIP(hipGraphLaunch(g->exec, (hipStream_t)stream));
  return RV_OK;
}

void rv_graph_destroy(rv_graph* g) {
  if (!g) return;
  (void)hipGraphExecDestroy(g->exec);
  (void)hipGraphDestroy(g->graph);
  delete g;
}

}  // extern "C"
