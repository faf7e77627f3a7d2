/* rawvae_hip.h -- C ABI of librawvae_hip.so: the MI355X (gfx950) training path of
 * the raw-audio VAE.
 *
 * The reference (kelseyicotton/rawaudiovae_kelsey) has no FFI of its own: its hot
 * path sits behind a Python module surface (rawvae/model.py, train.py).  Each entry
 * point below names the reference statement(s) whose arithmetic it replaces; the
 * Python classes in rawaudiovae_kelsey_amd/ (re-exported as rawvae.model) keep the
 * reference's signatures and call these through ctypes.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no torch / HIP types in signatures
 *     (`stream` is a hipStream_t passed as void*; NULL = the default stream).
 *   - every pointer is a DEVICE pointer owned by the caller; the library allocates no
 *     device memory at all -- only the opaque rv_plan / rv_graph host objects and a
 *     plan's internal streams and events.
 *   - one process per GPU, one host thread calling in at a time: the per-kernel
 *     "dynamic LDS attribute set" latches (and the test hook of
 *     rawvae_hip_diag.h) are plain process-wide statics (not per device, not thread-safe).
 *   - every function returns 0 on success or a negative RV_ERR_* code;
 *     rv_last_error() gives the message.  Nothing throws, aborts or synchronises
 *     unless its name ends in _sync; all launches are safe under stream capture.
 *   - "bf16 padded" operands: row-major bfloat16 whose extents are multiples of the
 *     GEMM tile (rows of the batch: 128; feature dims: 128; latent: 64) with zero
 *     padding -- see rv_pad_dims().  fp32 tensors at the reference boundary
 *     (frames, recon, mu, logvar, parameters, gradients) keep their exact shapes.
 */
#ifndef RAWVAE_HIP_H
#define RAWVAE_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RV_OK 0
#define RV_ERR_SHAPE (-1)
#define RV_ERR_NULL (-2)
#define RV_ERR_HIP (-3)
#define RV_ERR_UNSUPPORTED (-4)
#define RV_ERR_STATE (-5)

#define RV_ACT_NONE 0
#define RV_ACT_RELU 1

int rv_version(void);
const char* rv_last_error(void);

/* Padded extents used by every bf16 operand: Bp, Sp, Hp multiples of 128, Lp of 64. */
int rv_pad_dims(long B, long S, long H, long L, long* Bp, long* Sp, long* Hp, long* Lp);

/* GEMM tiling decisions, exposed because callers size partial-sum buffers from them (NULL outputs are skipped):
 *   RV_PLAN_GEMM  recommended split-K count *splits (<= splits_in, a power of two) and block tile (*bm x *bn) of a
 *                 padded Mp x Np x Kp GEMM;
 *   RV_PLAN_TILE  the tile the library uses when that GEMM is launched with exactly splits_in splits; per-row-tile
 *                 outputs (column-sum partials, MSE partials) then have Mp / bm row tiles and Np / bn column tiles;
 *   RV_PLAN_PAIR  for rv_linear_dgrad_wgrad(Mp, Np, Kp): *paired (one 256 x 256 launch for both GEMMs), the row tile
 *                 *bm of its column-sum partials and the weight gradient's split count *splits. */
enum { RV_PLAN_GEMM = 0, RV_PLAN_TILE = 1, RV_PLAN_PAIR = 2 };
int rv_gemm_plan(int what, long Mp, long Np, long Kp, int splits_in, int* bm, int* bn, int* splits, int* paired);

/* fp32 [rows, cols] (leading dim ld_src) -> zero-padded bf16 [rows_p, cols_p] (leading dim ld_dst).
 * Replaces the implicit fp32 operand read of F.linear (model.py:20) for frames and
 * is how weight shadows are (re)built after load_state_dict.  If `step_counter` is
 * non-NULL the kernel also increments *step_counter (device int64) once: it is the
 * first kernel of a training step. */
int rv_cast_pad_bf16(const float* src, long rows, long cols, long ld_src, void* dst_bf16,
                     long rows_p, long cols_p, long ld_dst, long long* step_counter, void* stream);

/* y = act(x W^T + b) -> bf16.  nn.Linear + F.relu, model.py:20 (fc1) and :29 (fc3).
 * x [Mp,Kp] bf16, w [Np,Kp] bf16 (nn.Linear [out,in] layout), bias [Np] fp32 or NULL. */
int rv_linear_fwd(const void* x_bf16, long ldx, const void* w_bf16, long ldw, const float* bias,
                  long Mp, long Np, long Kp, int act, void* y_bf16, long ldy, void* stream);

/* Same contraction, fp32 output written as `splits` partial slabs of [Mp,Np]
 * (slab s covers K range s*Kp/splits..); bias (may be NULL) is added by slab 0 only.
 * Used for the fused mu|logvar head GEMM, model.py:21 (fc21, fc22). */
int rv_linear_fwd_f32(const void* x_bf16, long ldx, const void* w_bf16, long ldw,
                      const float* bias, long Mp, long Np, long Kp, int splits, float* y_f32,
                      long ldy, void* stream);

/* Exact-fp32 y = act(x W^T + b) for the inference surface: `model(test_sample)[0]` under
 * torch.no_grad() (train.py:218-232) and the notebooks' encode / decode calls
 * (tutorial.ipynb:461,505-506,922-923).  F.linear + F.relu / F.tanh, model.py:20-21,29-30, in the
 * reference's own precision: f32-input MFMA (a k-ordered fmaf chain), so outputs match the
 * reference to f32 summation order.  Exact shapes (no padding), x [M,K], w [N,K] ([out,in]),
 * bias [N] or NULL, y [M,N]; act: 0 none, 1 relu, 2 tanh. */
int rv_linear_fp32(const float* x, long ldx, const float* w, long ldw, const float* bias, long M,
                   long N, long K, int act, float* y, long ldy, void* stream);

/* recon = tanh(h3 W4^T + b4), model.py:30, fused with the reconstruction half of
 * loss_function (model.py:39) and its derivative:
 *   recon   (optional) exact [B,S] fp32
 *   if x != NULL: mse_partial[block] = sum (recon-x)^2 over the block's valid elements
 *                 dP4 bf16 [Bp,Sp]   = (2/(B*S)) (recon-x)(1-recon^2)   (0 in padding)
 *                 db4_partial [Bp/bm][Sp] column sums of dP4 (optional)
 * with (bm, bn) = rv_gemm_plan(RV_PLAN_TILE, Bp, Sp, Hp, 1, ...): n_mse_partials = (Bp/bm)*(Sp/bn). */
int rv_decode_out_loss_fwd(const void* h3_bf16, long ldh, const void* w4_bf16, long ldw,
                           const float* b4, long Bp, long Sp, long Hp, long B, long S,
                           const float* x, long ldx, float* recon, long ld_recon,
                           void* dP4_bf16, long ld_dp4, float* mse_partial, float* db4_partial,
                           void* stream);

/* dX = dY W (autograd of F.linear, train.py:191).  dy [Mp,Kp] bf16, w [Kp,Np] bf16
 * ([out,in] layout, consumed as-is through transposing LDS reads).
 *   mask != NULL : dx_bf16 = (mask > 0) ? dX : 0   (ReLU', threshold_backward) and
 *                  colsum_partial [Mp/bm][Np] (optional) = column sums = bias grads,
 *                  bm from rv_gemm_plan(RV_PLAN_TILE, Mp, Np, Kp, 1, ...)
 *   mask == NULL : dx_f32 written as `splits` fp32 partial slabs [Mp,Np]. */
int rv_linear_dgrad(const void* dy_bf16, long lddy, const void* w_bf16, long ldw, long Mp,
                    long Np, long Kp, const void* mask_bf16, long ldmask, void* dx_bf16,
                    long lddx, float* colsum_partial, float* dx_f32, long lddx32, int splits,
                    void* stream);

/* Split-K slab element type of a weight gradient: fp32, or block-floating-point fp16 -- fp16(partial * 2^e) with
 * one exponent e per wave tile of one slab, taken from that tile's own largest magnitude, so gradients of ANY
 * magnitude keep fp16's 11 significant bits relative to their tile (same element strides as fp32 slabs).  The GEMM
 * writes the factors that undo the scales, 2^-e, to `slab_unscale`: [splits][Mp / 32][Np / 32] fp32, one per
 * 32 x 32 granule of each slab (required with RV_SLAB_F16, ignored with RV_SLAB_F32).  The sum over slabs stays fp32
 * in rv_adam_multi / rv_grad_finalize, which are told by rv_param_desc.grad_half / grad_unscale.  Halves the bytes
 * the GEMM writes and the optimizer reads back. */
enum { RV_SLAB_F32 = 0, RV_SLAB_F16 = 1 };

/* Both halves of a Linear layer's backward in ONE launch when the extents allow 256x256 tiles
 * (neither GEMM alone has enough such tiles to fill 256 CUs; together they do):
 *   dx_bf16 [Mp,Np] = (x > 0) ? dY W : 0     with column-sum partials [Mp/bm][Np] (bias grads)
 *   dw_slabs [splits][Kp][Np] = dY^T x        (split over the batch)
 * dy [Mp(batch), Kp(out)], w [Kp, Np] ([out,in]), x [Mp, Np] = the layer's ReLU output (mask AND
 * wgrad operand).  `splits` and `bm` must come from rv_gemm_plan(RV_PLAN_PAIR, Mp, Np, Kp, ...); when the
 * 256x256 pairing does not apply (e.g. the heads: Kp = 2 Lp) the two GEMMs still go out in one
 * launch if they share a small tile, otherwise as rv_linear_dgrad + rv_linear_wgrad. */
int rv_linear_dgrad_wgrad(const void* dy_bf16, long lddy, const void* w_bf16, long ldw, const void* x_bf16, long ldx,
                          long Mp, long Np, long Kp, void* dx_bf16, long lddx, float* colsum_partial, void* dw_slabs,
                          long lddw, int splits, int slab_dtype, float* slab_unscale, void* stream);

/* Backward of a Linear layer whose input had no activation (fc3, whose input is z): dX = dY W as
 * `dgrad_splits` fp32 slabs [Mp, Np] and dW = dY^T X as `wgrad_splits` slabs [Kp, Np], in ONE launch
 * when both GEMMs run on the same small tile (they read the same dY; each alone is mostly launch and
 * store-tail time).  dy [Mp(batch), Kp(out)], w [Kp, Np], x [Mp, Np].  Autograd of F.linear, train.py:191. */
int rv_linear_dgrad_wgrad_f32(const void* dy_bf16, long lddy, const void* w_bf16, long ldw,
                              const void* x_bf16, long ldx, long Mp, long Np, long Kp, float* dx_slabs,
                              long lddx, int dgrad_splits, float* dw_slabs, long lddw, int wgrad_splits,
                              void* stream);

/* dW = dY^T X as `splits` partial slabs [Mp(out), Np(in)] (split over the batch) of element type slab_dtype.
 * dy [Kp(batch), Mp] bf16, x [Kp(batch), Np] bf16; both read through transposing LDS
 * reads.  Autograd of F.linear w.r.t. weight, train.py:191.  `tile`: RV_TILE_AUTO (the picker's choice) or a named
 * block tile (extents must be multiples of it; `splits` must divide Kp/64; RV_TILE_256x256 runs the ping-pong main
 * loop when Kp/64/splits is even). */
enum { RV_TILE_AUTO = -1, RV_TILE_64x64 = 0, RV_TILE_128x128 = 4, RV_TILE_256x128 = 2, RV_TILE_256x256 = 7 };
int rv_linear_wgrad(const void* dy_bf16, long lddy, const void* x_bf16, long ldx, long Mp, long Np, long Kp,
                    int splits, int tile, void* dw_slabs, long lddw, int slab_dtype, float* slab_unscale, void* stream);

/* Reparameterisation forward, model.py:23-26, fused with the KL half of
 * loss_function (model.py:45):
 *   mulv_slabs [splits][Bp][2Lp] fp32 partial head outputs (mu at col l, logvar at Lp+l)
 *   -> mulv [Bp][2Lp] fp32 (summed, zero in padding), z bf16 [Bp][Lp],
 *      kl_partial[block] = sum over valid (b,l) of 1 + logvar - mu^2 - exp(logvar).
 * eps: explicit [B,L] fp32 when eps_in != NULL (parity runs), otherwise generated
 * on-device (Philox4x32-10 + Box-Muller, keyed by seed and *step_counter) and
 * written to eps_out [B,L].  n_kl_partials = Bp*Lp/1024. */
int rv_reparam_fwd(const float* mulv_slabs, int splits, long Bp, long Lp, long B, long L,
                   const float* eps_in, float* eps_out, unsigned long long seed,
                   const long long* step_counter, float* mulv, void* z_bf16, float* kl_partial,
                   void* stream);

/* encode's heads + reparameterize in one call, model.py:21-26 (fc21 | fc22 as ONE [2Lp, Kp] weight, then
 * z = mu + eps * exp(logvar / 2)) with the KL partials of model.py:45: rv_linear_fwd_f32 into `mulv_slabs`
 * (workspace, [splits][Bp][2Lp] fp32) followed by rv_reparam_fwd -- two launches back to back on `stream`.
 * A one-launch form was priced and not built: at 2L = 128 outputs the GEMM only fills the chip as 512 split-K
 * blocks, and combining split-K partials inside a launch (release fence + arrival ticket + acquire, 5-13 us per seam
 * on this chip: MI355X_MICROARCH.md, price list row "splitk-seam") costs more than the ~1.5 us kernel boundary it
 * removes; without split-K every block re-streams the whole head weight through its CU's ~60 GB/s L2->LDS port
 * (12-14 us for any row tile from 16 to 64 against 12.9 us for the two launches).  See DESIGN.md section 3. */
int rv_heads_reparam_fwd(const void* h_bf16, long ldh, const void* wh_bf16, long ldw, const float* bias_heads,
                         long Bp, long Lp, long Kp, long B, long L, int splits, float* mulv_slabs,
                         const float* eps_in, float* eps_out, unsigned long long seed, const long long* step_counter,
                         float* mulv, void* z_bf16, float* kl_partial, void* stream);

/* The latent-sized forward in ONE launch for a padded latent width of 64 (L <= 64; BASELINE's C2): heads GEMM
 * (model.py:21) + reparameterisation and KL partials (model.py:23-26,45) + fc3 with bias and ReLU (model.py:29),
 * 16 batch rows per workgroup, all three steps row-local.  Every workgroup streams the whole head weight and W3
 * through its CU (832 KB at C2, the L2 -> LDS port's ~60 GB/s sets the time): the K dimension of the heads GEMM and
 * the columns of fc3 are cut over the 8 waves, each wave running private LDS-DMA rings with counted vmcnt, no
 * workgroup barrier in the streaming loops (csrc/latent.hip).  Replaces rv_heads_reparam_fwd followed by
 * rv_linear_fwd(fc3): same outputs (mulv [Bp][128], z bf16 [Bp][64], kl_partial[Bp / 16], eps_out, h3 bf16
 * [Bp][Hp]); mu / logvar differ from the split-K route by fp32 summation order only; eps draws and the KL partial
 * layout are identical.  RV_ERR_UNSUPPORTED for other latent widths or a padded hidden width that is not a multiple
 * of 512 up to 2048.  18.8 us against 21-22 us for the three launches at C2 (profiles/r03_*): the training plan's
 * default where it applies (rv_plan_set_option, RV_OPT_LATENT_FUSED).
 * w3_bf16 == NULL: heads + reparameterisation only (bias3, h3_bf16 unused; 576 KB per CU instead of 832); fc3 is then
 * the caller's (rv_linear_fwd).
 * Padded latent widths 128 / 256 (the reference's own latent_dim = 256, default.ini:18), batches above 8192 and other
 * hidden widths (multiples of 128): the same call runs its GEMM form -- the heads GEMM on 64 x 128 tiles (256-row tiles at
 * large batches: 256 x 128, from Lp = 128 on 256 x 256 ping-pong) with bias, eps, exp, z and the KL partials in the
 * epilogue, then fc3 as a forward GEMM.  Same outputs and eps draws; kl_partial [Bp Lp / 1024] then holds one non-zero
 * slot per tile and zeros in the others (only the sum is defined). */
int rv_latent_fwd(const void* h_bf16, long ldh, const void* wh_bf16, long ldwh, const float* bias_heads,
                  const void* w3_bf16, long ldw3, const float* bias3, long Bp, long Hp, long Lp, long B, long L,
                  const float* eps_in, float* eps_out, unsigned long long seed, const long long* step_counter,
                  float* mulv, void* z_bf16, float* kl_partial, void* h3_bf16, long ldh3, void* stream);

/* The backward mirror of rv_latent_fwd's first two steps in ONE launch (same shape limits): dz = dP3 W3 (autograd of
 * fc3's input, model.py:29) for 16 batch rows per workgroup over the full contraction, then rv_reparam_bwd's
 * arithmetic on that block's dz while it is still in LDS -- no dz slabs, no second launch.  Arguments as
 * rv_reparam_bwd with dP3 [Bp, Hp] bf16 and W3 [Hp, Lp] bf16 ([out, in]) in place of the dz slabs; dz differs from the
 * split-K route by fp32 summation order only.  With z_bf16 != NULL the launch also computes fc3's weight gradient
 * dW3 [Hp, Lp] = dP3^T z as `dw3_splits` fp32 slabs (rv_linear_wgrad's result, bit for bit) on extra workgroups that
 * share the CUs with the dz workgroups (80 KiB of LDS each): the second read of dP3 fills the bubbles of the first.
 * GEMM form (same shapes as rv_latent_fwd's): dz tiles of 64 rows (256 at large batches; ONE 256 x 256 ping-pong tile per
 * 256 rows at Lp = 256) with the reparameterisation backward in the epilogue, dW3 on the launch's first workgroups.
 * dbh_partial [Bp / 16][2 Lp] keeps its layout: the row-local kernel fills every row, the GEMM forms the first row of each
 * dz tile's rows and ZEROS in the rest (a reader may sum every row, or every (tile rows / 16)-th). */
int rv_latent_bwd(const void* dp3_bf16, long lddp, const void* w3_bf16, long ldw3, long Bp, long Hp, long Lp, long B,
                  long L, long S, const float* mulv, const float* eps, float kl_beta, const float* dmu_ext,
                  const float* dlv_ext, void* dmulv_bf16, float* dbh_partial, const float* mse_partial, int n_mse,
                  const float* kl_partial, int n_kl, float* loss_out, const long long* step_counter, int ring,
                  const void* z_bf16, long ldz, float* dw3_slabs, long lddw3, int dw3_splits, void* stream);

/* The heads' backward (autograd of fc21 | fc22, model.py:21, with fc1's ReLU) for a padded latent width of 64 as ONE
 * streaming launch that reads h1 once: dp1 [Bp, Hp] bf16 = (h1 > 0) ? dmulv Wh : 0, its column sums per 512-row group
 * (db1_partial [Bp / 512][Hp], fc1's bias gradient) and dWh = dmulv^T h1 as one fp32 slab per 512-row group
 * (dwh_slabs [Bp / 512][128][lddw]: rv_linear_dgrad_wgrad's outputs with Bp / 512 splits).  dmulv [Bp, 128] bf16 (mu
 * columns 0..63, logvar 64..127: rv_reparam_bwd's output), wh [128, Hp] bf16 (fc21 | fc22, [out, in]), h1 [Bp, Hp]
 * bf16.  A workgroup keeps its 64-column slice of Wh in LDS and walks 512 rows in tiles of 64; the same staged h1
 * tile is the ReLU mask of the first product and the operand of the second.  RV_ERR_UNSUPPORTED unless Lp == 64,
 * Bp % 512 == 0 and Hp % 64 == 0. */
int rv_heads_bwd(const void* dmulv_bf16, const void* wh_bf16, long ldw, const void* h1_bf16, long ldh, long Bp, long Hp,
                 long Lp, void* dp1_bf16, long ldp, float* db1_partial, float* dwh_slabs, long lddw, void* stream);

/* Backward of reparameterize + KL:
 *   dmu = dz + kl_beta mu/(B L);  dlv = dz eps std/2 + kl_beta (exp(logvar)-1)/(2 B L)
 * dz_slabs [splits][Bp][Lp] fp32 -> dmulv bf16 [Bp][2Lp] and per-block column sums
 * dbh_partial [Bp/16][2Lp] (bias grads of fc21|fc22).  One block also finishes the
 * loss: loss_out[0] = sum(mse_partial)/(B S) + kl_beta*(-0.5*sum(kl_partial)/(B L)),
 * loss_out[1] = mse term, loss_out[2] = KL term (pass NULL partials to skip).  When
 * step_counter != NULL and ring > 0, loss_out is a ring of [ring][4] floats and the
 * slot written is (*step_counter - 1) % ring, so graph replays log every step. */
/* Gradients that arrive from outside are added in (exact [B, L] fp32, either may be NULL):
 * dmu += dmu_ext, dlv += dlv_ext -- what autograd hands the backward of reparameterize when mu / logvar also feed
 * a loss term directly (the KL half of loss_function, model.py:45).  Pass kl_beta = 0 when the KL gradient is
 * already inside dmu_ext / dlv_ext. */
int rv_reparam_bwd(const float* dz_slabs, int splits, long Bp, long Lp, long B, long L, long S,
                       const float* mulv, const float* eps, float kl_beta, const float* dmu_ext,
                       const float* dlv_ext, void* dmulv_bf16, float* dbh_partial, const float* mse_partial,
                       int n_mse, const float* kl_partial, int n_kl, float* loss_out,
                       const long long* step_counter, int ring, void* stream);

/* loss_function(recon_x, x, mu, logvar, kl_beta, segment_length), model.py:38-47, as ONE
 * wave-reduced kernel over exact-shape fp32 tensors; also emits the gradients autograd
 * would produce for (recon, mu, logvar) so loss.backward() needs no second pass.
 * workspace: rv_loss_fused_workspace_bytes() bytes, zero-initialised once by the caller.
 * loss_out[0..2] = total, mse, kld.  Any of the gradient pointers may be NULL. */
long rv_loss_fused_workspace_bytes(void);
int rv_loss_fused(const float* recon, const float* x, const float* mu, const float* logvar,
                  long B, long S, long L, float kl_beta, float* loss_out, float* d_recon,
                  float* d_mu, float* d_logvar, void* workspace, void* stream);

/* z = mu + eps*exp(logvar/2), exact-shape fp32 (VAE.reparameterize called on its own,
 * e.g. tutorial.ipynb:505).  eps_in NULL -> generated from (seed, offset). */
int rv_reparameterize(const float* mu, const float* logvar, long n, const float* eps_in,
                      float* eps_out, unsigned long long seed, unsigned long long offset,
                      float* z, void* stream);

/* Backward of reparameterize for callers that use it on its own: dmu = dz,
 * dlv = dz*eps*exp(logvar/2)/2 (either output may be NULL). */
int rv_reparameterize_bwd(const float* dz, const float* eps, const float* logvar, long n,
                          float* dmu, float* dlv, void* stream);

/* Backward of F.tanh (model.py:30) for VAE.decode used without the fused loss:
 * dP4 = d_recon*(1-recon^2), exact fp32 [B,S] inputs -> zero-padded bf16 [Bp,Sp]. */
int rv_tanh_bwd_pack(const float* d_recon, const float* recon, long B, long S, void* dP4_bf16,
                     long Bp, long Sp, void* stream);

/* fp32 elementwise steps of the strict-fp32 training mode (rawaudiovae_kelsey_amd/strict.py; autograd of
 * model.py:20,29,30): op 0: out = a*(1-b*b) (tanh backward), 1: out = b>0 ? a : 0 (ReLU backward), 2: out = a+b. */
int rv_ew_f32(int op, const float* a, const float* b, long n, float* out, void* stream);

/* Partial column sums (bias gradients): out[rb][c] = sum of rows [256 rb, 256 rb+256) of
 * column c; src is fp32 or bf16 [rows, cols] with leading dim ld.  ceil(rows/256) slabs,
 * summed by rv_grad_finalize / rv_adam_multi. */
int rv_colsum_partial(const void* src, int is_bf16, long rows, long cols, long ld, float* out,
                      long ld_out, void* stream);

/* out[i] = a[i] * scalar[0] (device scalar): applies the upstream gradient of the 0-dim
 * loss tensor to the gradients rv_loss_fused saved. */
int rv_scale_by(const float* a, const float* scalar, long n, float* out, void* stream);
/* The same for up to three tensors in one launch (out_k[i] = a_k[i] * scalar[0]; n_k = 0 skips tensor k): the three
 * gradients loss_function's backward hands on -- the autograd path is bound by host time per launch. */
int rv_scale_by3(const float* a0, float* out0, long n0, const float* a1, float* out1, long n1, const float* a2,
                 float* out2, long n2, const float* scalar, void* stream);

/* Hop-strided framing on the device (AudioDataset, rawvae/dataset.py:99-121): the padded
 * waveform stays in HBM and out[i, :] = audio[f*hop : f*hop + S] with f = frame_index[i]
 * (int64, e.g. one slice of the epoch's shuffle) or first_frame + i when frame_index is NULL
 * (TestDataset, dataset.py:147-157, is hop == S).  Samples past n_samples read as 0. */
int rv_gather_frames(const float* audio, long n_samples, const long long* frame_index,
                     long first_frame, long n_frames, long S, long hop, float* out, void* stream);

/* Standard normal draws (replaces torch.randn_like, model.py:25). */
int rv_randn(float* out, long n, unsigned long long seed, unsigned long long offset,
             void* stream);

/* One parameter tensor as seen by the fused optimizer / gradient finaliser. */
typedef struct rv_param_desc {
  long offset;             /* element offset of the tensor in the flat fp32 arenas   */
  long rows, cols;         /* exact shape ([out,in]; bias: rows=1)                   */
  const float* grad_slabs; /* partial-gradient slabs: element (r,c) of slab s is at  */
  long grad_ld;            /*   grad_slabs[s*grad_split_stride + r*grad_ld + c]      */
  long grad_split_stride;
  int grad_splits;
  void* shadow_bf16;       /* padded bf16 copy refreshed with the new weight (or NULL) */
  float* shadow_f32;       /* padded fp32 copy (biases as read by GEMM epilogues), or NULL */
  long shadow_ld;
  void* shadow_fp8;        /* padded fp8 (e4m3) copy fp8(w * *fp8_scale), leading dim shadow_ld, or NULL */
  const float* fp8_scale;  /* device scalar */
  int grad_half;           /* non-zero: grad_slabs holds fp16 values fp16(partial * 2^e) (same element strides), e per */
  const float* grad_unscale; /* 32 x 32 granule and slab: element (r,c) of slab s is scaled back by                 */
  long us_ld;              /*   grad_unscale[s*us_split_stride + (r/32)*us_ld + c/32]                               */
  long us_split_stride;    /* (the table a weight-gradient GEMM writes with RV_SLAB_F16)                            */
} rv_param_desc;

/* torch.optim.Adam(lr) step (train.py:163,193: betas 0.9/0.999, eps 1e-8, no weight
 * decay, no amsgrad) over the flat arenas in one launch: sums the gradient slabs,
 * updates exp_avg / exp_avg_sq / param, rewrites the bf16 shadow, and (optional)
 * stores the summed gradient to grad_out (flat, exact) for inspection.
 * t = *step_counter (1-based).  grad_scale multiplies the summed gradient first
 * (1/world_size after an all-reduce SUM).  `descs` is HOST memory (copied per call).
 * grad_bf16 != NULL: the gradient is taken from that flat bf16 arena (same element offsets as the fp32 arenas; a bf16
 * all-reduce's result) instead of the descriptors' slabs; grad_out must then be NULL. */
int rv_adam_multi(const rv_param_desc* descs, int n_desc, float* param, float* exp_avg,
                  float* exp_avg_sq, float* grad_out, const void* grad_bf16, float lr, float grad_scale,
                  const long long* step_counter, void* stream);

/* Sum gradient slabs into the flat exact-shape gradient arena only (no update):
 * what loss.backward() leaves in .grad; also the all-reduce payload builder.  out_bf16 != 0: grad_out is a flat
 * bf16 arena (same element offsets) and receives the fp32 sum rounded to bf16 -- half the all-reduce bytes. */
int rv_grad_finalize(const rv_param_desc* descs, int n_desc, void* grad_out, int out_bf16, void* stream);
/* dW = dY^T X on 256x256 tiles (as rv_linear_wgrad with RV_TILE_256x256) in a launch that ALSO runs the
 * fused Adam update (rv_adam_multi) of the `n_desc` tensors in `descs` -- tensors whose gradients earlier
 * launches completed, never the one this GEMM produces -- on `n_adam_blocks` extra 512-thread blocks that take
 * the CUs the GEMM's tiles * splits blocks leave idle (extents must tile by 256 x 256 x 64). */
int rv_linear_wgrad_adam(const void* dy_bf16, long lddy, const void* x_bf16, long ldx, long Mp, long Np, long Kp,
                         int splits, void* dw_slabs, long lddw, int slab_dtype, float* slab_unscale,
                         const rv_param_desc* descs, int n_desc,
                         float* param, float* exp_avg, float* exp_avg_sq, float lr, float grad_scale,
                         const long long* step_counter, int n_adam_blocks, void* stream);

/* Parameters (unless `param` is NULL: shadows only; `flat` may then be the parameter arena itself, which is how
 * rv_plan_refresh_shadows rebuilds all shadows in one launch) AND every operand shadow of the `descs` tensors from a flat
 * fp32 source: arena element o is flat[o - flat_base]. */
int rv_params_from_flat(const rv_param_desc* descs, int n_desc, const float* flat, long flat_base, float* param,
                        void* stream);

/* ---- fp8 (e4m3, OCP) operand path for the two large forward GEMMs (BASELINE configs[4]; a build extension,
 * the reference has no reduced-precision path).  Operands are quantised per tensor:
 * q = fp8(value * scale), the GEMM accumulates in fp32 on v_mfma_scale_f32_16x16x128_f8f6f4 (unit block
 * scales) and multiplies the accumulator by *dq = 1 / (scale_A * scale_B) before bias / activation.
 * K extents and leading dims are in fp8 elements (multiples of 128 / 16). ---- */
/* fp32 [rows, cols] -> zero-padded fp8 [rows_p, cols_p]: fp8(src * *scale), saturating at +-448. */
int rv_cast_pad_fp8(const float* src, long rows, long cols, long ld_src, void* dst_fp8, long rows_p, long cols_p,
                    long ld_dst, const float* scale, void* stream);
/* rv_linear_fwd whose operand rows are read where the audio lives (AudioDataset.__getitem__,
 * rawvae/dataset.py:108-118): row r < B is the frame audio_bf16[f*hop : f*hop + Kp], f = frame_index ? frame_index[r]
 * : first_frame + r, of the waveform kept in HBM as bf16 (cast once when it was uploaded; rounding to bf16 is what
 * rv_cast_pad_bf16 does per batch, so the operand is bit-identical); rows B..Mp repeat row B - 1 (padding).  The A
 * tile loader stages each frame's 16-byte pieces straight from the waveform -- no cast / gather kernel, no fp32
 * read.  hop % 8 == 0, 16-byte aligned waveform, and the buffer must extend Kp elements past the last frame's start.
 * frames_bf16 (or NULL) receives the framed [Mp][ld_frames] bf16 matrix as a by-product (the weight gradient's
 * operand): the block with tile_n == kt % tiles_n copies K tile kt of its rows out of LDS.  step_counter (or NULL)
 * is incremented by block 0 (the step's first kernel does that). */
int rv_linear_fwd_frames(const void* audio_bf16, const long long* frame_index, long first_frame, long hop, long B,
                         const void* w_bf16, long ldw, const float* bias, long Mp, long Np, long Kp, int act,
                         void* y_bf16, long ldy, void* frames_bf16, long ld_frames, long long* step_counter,
                         void* stream);
/* ---- whole-step plan: one call enqueues forward, loss, backward (and Adam) ---- */
typedef struct rv_plan rv_plan;

typedef struct rv_plan_buffers {
  /* flat fp32 arenas, PARAM order fc1.w fc1.b fc21.w fc21.b fc22.w fc22.b fc3.w fc3.b fc4.w fc4.b */
  float* param; float* exp_avg; float* exp_avg_sq; float* grad; /* grad may be NULL */
  void* workspace;             /* rv_plan_workspace_bytes() bytes, zero-initialised   */
  long long* step_counter;     /* device int64, number of steps started               */
  float* loss_ring;            /* [ring][4] fp32: total, mse, kld, unused             */
  int ring;
} rv_plan_buffers;

#define RV_PHASE_FWD 1      /* cast, fc1, heads, reparam, fc3, fc4+loss              */
#define RV_PHASE_BWD_A 2    /* fc4 backward (paired dgrad+wgrad), dz, reparam bwd, fc3 wgrad */
#define RV_PHASE_BWD_B 4    /* heads dgrad/wgrad, fc1 wgrad                           */
#define RV_PHASE_FINALIZE_A 8  /* fc3,fc4 slabs -> flat fp32 grad arena                */
#define RV_PHASE_FINALIZE_B 32 /* fc1,fc21,fc22 slabs -> flat grad arena               */
#define RV_PHASE_ADAM 16       /* optimizer + bf16 shadow refresh (all ten tensors)    */
#define RV_PHASE_ADAM_A 64     /* ... only fc3, fc4                                    */
#define RV_PHASE_ADAM_B 128    /* ... only fc1, fc21, fc22                             */
/* finer split, in gradient-availability order (data-parallel buckets: fc4 | fc1 | the rest):  */
#define RV_PHASE_BWD_FC4 0x0100   /* fc4 backward: fc4.w, fc4.b, fc3.b gradients ready         */
#define RV_PHASE_BWD_CHAIN 0x0200 /* dz, reparam bwd, heads dgrad, fc1 wgrad: fc1.*, head biases */
#define RV_PHASE_BWD_REST 0x0400  /* fc3 wgrad, heads wgrad                                    */
#define RV_PHASE_FIN_FC4 0x0800
#define RV_PHASE_FIN_FC1 0x1000
#define RV_PHASE_FIN_MID 0x2000   /* fc21, fc22, fc3                                           */
#define RV_PHASE_ADAM_FC4 0x4000
#define RV_PHASE_ADAM_FC1 0x8000
#define RV_PHASE_ADAM_MID 0x10000
#define RV_PHASE_ALL_LOCAL (1 | 2 | 4 | 16)

int rv_plan_create(rv_plan** out, long B, long S, long H, long L);
void rv_plan_destroy(rv_plan*);
long rv_plan_workspace_bytes(const rv_plan*);
int rv_plan_bind(rv_plan*, const rv_plan_buffers*);
/* Plan options (the plan must be bound):
 *   RV_OPT_LATENT_FUSED  1 (default): heads GEMM, reparameterisation and fc3 of the forward are ONE launch
 *     (rv_latent_fwd; with the fp8 forward it also emits fc4's fp8 operand) when the padded latent width is 64 and
 *     the padded hidden width a multiple of 512 up to 2048, and dz + the reparameterisation backward likewise (rv_latent_bwd); 0, or any
 *     other shape: three launches (rv_heads_reparam_fwd + fc3) and dz as split-K slabs + rv_reparam_bwd.  The same
 *     switch selects the heads' backward: rv_heads_bwd (one pass over h1; needs a padded batch that is a multiple of
 *     512) or the generic rv_linear_dgrad_wgrad; the partial counts of fc21 / fc22 weights and of fc1's bias in
 *     rv_plan_descs follow the form in use.
 *   RV_OPT_FP8  the fp8 (e4m3) weight path.  2: fp8 forward for fc1 and fc4 (weights AND their input activations in e4m3;
 *     backward, heads, fc3 stay bf16).  1: that forward AND fc4's backward -- dgrad and wgrad in one 256 x 256 launch -- on
 *     fp8 operands: the fc4 forward's epilogue writes dP4 as fp8(dP4 * [12]) instead of bf16 (|dP4| <= 2 * 2 / (B S) by
 *     construction, so the scale is fixed), the dgrad multiplies it with the fp8 weight shadow (read MN-major through
 *     ds_read_b64_tr_b8), the wgrad with the fp8 image of h3 the fc3 forward wrote; K tiles are 128 deep, half the LDS
 *     fill per flop of the bf16 pair.  Where the extents do not tile (256 x 256 tiles, an even number of 128-deep K
 *     tiles per block) and for gradients from outside (rv_plan_set_external_grads) the backward stays bf16.  In the full
 *     local step (all phases in one call) and in rv_plan_step_ddp's all-reduce schedule fc1's weight gradient -- the
 *     launch that also carries rider blocks -- runs on fp8 operands as well, under the same tiling conditions: the heads' backward writes dP1 as fp8(dP1 * [13])
 *     only, the GEMM multiplies it with the fp8 image of the frames fc1's forward read (both MN-major), and neither
 *     bf16 copy is written; [13] follows the maximum of |dP1| measured in the previous step, like h3's scale.
 *     The workspace buffer "fp8_state" holds 16 floats (then 2 x 1024 slots) the caller initialises before rv_plan_refresh_shadows:
 *       [0] scale of x   [1] scale of W1   [2] scale of W4   [3] scale of h3 (this step)
 *       [4] max|h3| of the previous step (reduced from the fc3 forward's per-block maxima, workspace buffer
 *           "h3_amax"; this step's h3 scale is 224 / it: delayed scaling)
 *       [5] 1/([0][1])   [6] 1/([3][2])   (both rewritten at the start of every step)
 *       [7] non-zero: keep [3], [13] and the weight scales fixed (parity runs)
 *       [10] 1/([12][2])   [11] 1/([12][3])   (fc4's fp8 backward: dgrad / wgrad dequantisation, rewritten every step)
 *       [12] scale of dP4's fp8 image: 112 / (2 / (B S)), set by the caller
 *       [13] scale of dP1's fp8 image (this step; the caller's first guess, then 224 / [14])   [14] max|dP1| of the
 *           previous step (reduced from the per-wave maxima rv_heads_bwd left behind h3's in "h3_amax")
 *       [15] 1/([13][0])   (fc1's fp8 weight gradient: dequantisation, rewritten every step)
 *       [8] max|W1|, [9] max|W4| as the last optimizer update left them in the fp8 shadows (0 = none yet; reduced by
 *           the step's first kernel from [32 ..]: 2 x 1024 slots that a small kernel behind the optimizer fills with
 *           max|q| / scale over slices of the two shadows, and that the first kernel resets); [16..31] reserved.  The
 *           caller zero-initialises the whole buffer.
 *     Weight scales start as the caller's (224 / max|W| at refresh).  Adam rewrites the fp8 shadows with the current
 *     scale; the first kernel of the next step, AFTER latching [5] / [6] from the scales the shadows were written
 *     with, moves [1] / [2] to 224 / the measured maximum for the coming update (delayed scaling: a weight that grows
 *     never meets a scale older than one step; one that has saturated reads as 448 / scale, so the scale halves until
 *     it fits).
 *   RV_OPT_SLAB_DTYPE  element type of the split-K slabs of the two large weight gradients (fc1.weight, fc4.weight;
 *     2 x 33.5 MB of fp32 slabs per step at C2): RV_SLAB_F16 (default: block-floating-point fp16, see above), which
 *     halves what the two weight-gradient GEMMs write and Adam reads back, or RV_SLAB_F32.  Each partial is an fp32
 *     sum over a quarter of the batch; rounding it to fp16 adds ~3e-4 relative noise to those two gradients whatever
 *     their magnitude (the sum over slabs stays fp32).  Where the plan runs the streaming heads' backward (rv_heads_bwd's
 *     form: padded latent width 64, padded batch a multiple of 512) the row-group partials of fc21.weight / fc22.weight
 *     follow the same switch (8.4 MB of fp32 slabs per step at C2); rv_plan_descs reports the form in use.
 *   RV_OPT_ROCTX  1: roctx ranges (rocprofv3 --marker-trace) around the host calls that enqueue the step's phases --
 *     "rv:fwd", "rv:fc4-bwd", "rv:rest-bwd", "rv:adam" (local step) / "rv:rest-bwd+exchange+adam" (data-parallel step) --
 *     so that a kernel timeline reads as phases (tracing).  The marker library is loaded at run time
 *     (librocprofiler-sdk-roctx.so, else libroctx64.so); RV_ERR_UNSUPPORTED when neither can be.  Default 0.
 *   RV_OPT_DDP_SIGNAL  how the compute stream and the collective stream of rv_plan_step_ddp (all-reduce schedule) hand
 *     work to each other: 1 (default) device-side sequence flags in the workspace buffer "ddp_flags" -- a one-wave kernel
 *     behind the producer publishes the step's number, a one-wave kernel in front of the consumer waits for it (~1.8 us
 *     per crossing; deadlock-free for any stream -> hardware-queue mapping because every waiter is enqueued after its
 *     setter; every wait is bounded -- 5 s where the setter follows this device's own kernels, RV_OPT_DDP_WAIT_MS where
 *     it follows a collective -- and timeouts are counted in ddp_flags[8], which must stay 0) -- or 0: HIP events (~9 us
 *     per crossing).  Steps enqueued under stream capture always use events (a graph needs the edges).
 *   RV_OPT_DDP_WAIT_MS  bound, in milliseconds, of a flag wait whose setter sits behind a collective, i.e. behind the
 *     slowest peer rank (default 30000 = thirty seconds: a spinning wave cannot be cancelled from the host, so the bound
 *     is what a dead peer costs; ranks reach a step seconds apart around a checkpoint or an evaluation pass as a matter
 *     of course -- a caller whose ranks drift further apart than that raises it, or synchronises the ranks first).  A
 *     wait that runs out is counted in ddp_flags[8] and POISONS the plan on the device: the optimizer launches of
 *     rv_plan_step_ddp read the count and apply no update while it is non-zero, so a partial all-reduce never reaches
 *     the parameters; the host side must read the count (it is never cleared on the device), agree on it across ranks
 *     and stop every rank.
 *   RV_OPT_DDP_DEFER_TAIL  1: rv_plan_step_ddp (all-reduce schedule, device-side flags, bf16 operands) returns with its
 *     LAST wait -- second exchange done -- and the update of that bucket (fc1, heads, fc3) not yet enqueued; the next
 *     rv_plan_step_ddp call enqueues its own cast launch first (it needs no parameter, and the compute stream has nothing
 *     else to do while the exchange is on the links), then that wait and update, then its forward.  Anything else that
 *     follows a step -- reading parameters or optimizer state, a checkpoint, an evaluation pass, the end of training --
 *     needs rv_plan_ddp_flush(plan, stream) first (rv_plan_step and rv_plan_step_frames do it themselves).  The deferred
 *     update takes its step number from a copy latched inside the step, so results are bit-identical to 0 (default).
 *   RV_OPT_DDP_W1_WIDE  1: in rv_plan_step_ddp's all-reduce schedule fc1's weight gradient -- the last GEMM of the
 *     backward, which has no optimizer riders there -- runs with twice the K splits of the local step, i.e. on all 256
 *     CUs instead of 128 (where the extents allow).  0 (default): the local step's split count on 128 CUs, whose rider
 *     blocks sum the slabs of the second bucket's other tensors into the payload meanwhile.  Which is faster depends on
 *     what the collective that runs beside this launch does to the CUs it occupies: with workgroups that leave room
 *     for a 256 x 256 GEMM block beside them the wide form wins (modelled: 241 against 247 us per step at 8 ranks), with
 *     workgroups that take their CUs whole it runs in two rounds and loses (263 against 250) -- the default is the
 *     form whose time does not depend on it.  (With 0 and the fp32 payload a one-rank step reproduces rv_plan_step bit
 *     for bit; the sums over 4 and over 8 partial slabs round differently.) */
enum { RV_OPT_LATENT_FUSED = 0, RV_OPT_FP8 = 1, RV_OPT_SLAB_DTYPE = 2, RV_OPT_ROCTX = 3, RV_OPT_DDP_SIGNAL = 4, RV_OPT_DDP_W1_WIDE = 5,
       RV_OPT_DDP_WAIT_MS = 6, /* 7, 8: retired in round 6 (the paired latent forward and fc3 inside the fc4 forward, both
       measured slower in the step: DESIGN.md section 6, profiles/r05_latent_pair_ab.txt, r05_fc3_in_fc4.txt) */
       RV_OPT_DDP_DEFER_TAIL = 9 };
int rv_plan_set_option(rv_plan*, int option, int value);
/* Enqueue what a data-parallel step left to "the next call" (RV_OPT_DDP_DEFER_TAIL): the wait for the second exchange
 * and the update behind it -- always on the stream that step was enqueued on (`stream`: that stream, or NULL; anything
 * else is RV_ERR_STATE, as is a following rv_plan_step_ddp on another stream).  No-op when nothing is pending. */
int rv_plan_ddp_flush(rv_plan*, void* stream);
/* Gradients from outside for the following BWD / FINALIZE phases (the autograd boundary of rawvae.model.VAE.forward:
 * any loss, not only loss_function).  d_recon [B,S] with recon [B,S] (the forward's output, for tanh'), dmu and
 * dlogvar [B,L], all exact-shape fp32, each NULL = zero; the reparameterisation backward then takes kl_beta from
 * the rv_plan_step call (pass 0 when dmu / dlogvar already hold the KL gradient).  grad_out (or NULL = the bound
 * grad arena) receives the FINALIZE phases' flat fp32 gradients.  All NULL restores the fused loss of the forward
 * phase.  Not honoured by the full-step schedules (phases == RV_PHASE_ALL_LOCAL). */
int rv_plan_set_external_grads(rv_plan*, const float* d_recon, const float* recon, const float* dmu,
                               const float* dlogvar, float* grad_out);
/* The plan's OWN loss across the autograd boundary (loss_function of rawvae/model.py:38-47 called on the untouched
 * outputs of the fused forward: the drop-in loop of train.py:184-193).  rv_plan_loss: (total, mse, kld) of the forward
 * phase that ran last into out3[3] (device), one small launch, the value -- and summation order -- the backward later
 * writes to the loss ring.  rv_plan_set_loss_grad: the following BWD / FINALIZE phases use the forward's fused loss
 * gradient (nothing from outside) and leave (*d_loss_dev) x gradient in grad_out (flat fp32 [n_params]; NULL = the
 * bound grad arena); d_loss_dev is the upstream gradient of the scalar loss, a DEVICE pointer read at finalize time (the
 * host never synchronises on it).  d_loss_dev = NULL and grad_out = NULL switch it off.  RV_ERR_STATE while gradients
 * from outside are set. */
int rv_plan_loss(rv_plan*, float kl_beta, float* out3, void* stream);
int rv_plan_set_loss_grad(rv_plan*, const float* d_loss_dev, float* grad_out);
/* The plan's ten parameter descriptors (PARAM order): gradient slabs of its own workspace (from_flat = 0) or the
 * bound flat gradient arena (1), and the operand shadows Adam must refresh.  For callers that drive
 * rv_adam_multi / rv_params_from_flat themselves. */
int rv_plan_descs(const rv_plan*, rv_param_desc* out10, int from_flat);
/* How the full local step divides the optimizer (train.py:193) between its last two launches: tensors [*first, *last) of
 * the descriptor table are updated by rider blocks beside fc1's weight gradient (rv_linear_wgrad_adam), all others by the
 * step's last launch (rv_adam_multi).  [2, 10) at a latent width of 64 (only fc1 is left for the last launch); [2, 8) --
 * the heads and fc3 -- at the reference's own latent_dim = 256 (default.ini:18), where everything but fc1 would make the
 * riders twice as long as the GEMM beside them.  RV_ERR_STATE for an unbound plan. */
int rv_plan_riders(const rv_plan*, int* first, int* last);
/* Rebuild every bf16 weight shadow from the fp32 param arena (after init / load). */
int rv_plan_refresh_shadows(rv_plan*, void* stream);
/* Enqueue the selected phases of one training step (train.py:184-193) on `stream`.
 * x: exact [B,S] fp32 frames.  eps: explicit [B,L] or NULL (on-device RNG, `seed`).
 * recon_out: optional exact [B,S].  adam_from_flat != 0 makes the Adam phase read the
 * flat grad arena (e.g. after an all-reduce), scaled by grad_scale, instead of the slabs. */
int rv_plan_step(rv_plan*, int phases, const float* x, const float* eps, float* recon_out,
                 float kl_beta, float lr, float grad_scale, int adam_from_flat,
                 unsigned long long seed, void* stream);
/* rv_plan_step whose batch is B hop-strided frames of a waveform resident in HBM (AudioDataset semantics,
 * rawvae/dataset.py:99-121; frame i = audio[f*hop : f*hop + S], f = frame_index ? frame_index[i] : first_frame + i,
 * samples past n_samples read as 0).  With audio_bf16 (the same waveform as bf16, n_samples + Sp + 8 elements, zero
 * past n_samples; hop % 8 == 0) the step launches NO cast or gather kernel: fc1's A-tile loader reads frame f at
 * f*hop of the bf16 waveform (rv_linear_fwd_frames) and fc4's loss epilogue reads its fp32 target at f*hop of `audio`
 * (rv_decode_out_loss_fwd_frames).  audio_bf16 == NULL (or an unaligned hop, a frame length that is not a multiple of
 * 128 -- the padded columns must be zeros, not the samples behind the frame -- or the fp8 forward) falls back to one
 * cast kernel per step (rv_gather_cast_frames). */
int rv_plan_step_frames(rv_plan*, int phases, const float* audio, const void* audio_bf16, long n_samples,
                        const long long* frame_index, long first_frame, long hop, const float* eps, float* recon_out,
                        float kl_beta, float lr, float grad_scale, int adam_from_flat, unsigned long long seed,
                        void* stream);
/* ---- data-parallel step with the collective driven from here (no reference code:
 * the reference is single-process).  `allreduce` is the collective library's in-place-capable
 * all-reduce with RCCL's ncclAllReduce signature -- (sendbuf, recvbuf, count, dtype, op, comm, stream),
 * returning 0 on success -- and `comm` its communicator; the caller creates both (see
 * rawaudiovae_kelsey_amd/ddp.py: RCCL loaded with ctypes, unique id shared over torch.distributed).
 * rv_plan_step_ddp enqueues ONE whole training step: forward, loss, backward with two gradient
 * buckets (fc4 | fc1,fc21,fc22,fc3) summed across ranks on an internal high-priority stream as
 * soon as backward has produced them, and Adam per bucket with the 1/world mean -- one host call,
 * no host synchronisation, capturable in a hipGraph.  Needs a grad arena and a non-default stream. */
typedef int (*rv_allreduce_fn)(const void* sendbuf, void* recvbuf, size_t count, int dtype, int op,
                               void* comm, void* stream);
/* Everything rv_plan_step_ddp needs from the caller, in one descriptor (copied; attach again to change a field,
 * comm == NULL detaches and keeps only comm_stream).  (Rounds 3-5 also carried a sharded-optimizer schedule -- reduce-scatter,
 * Adam on a rank's 1 / world of the arenas, all-gather of parameters or of a 16-bit parameter message; it won no row of the
 * 8-rank model (a second collective's start-up for an update that is 13 us to begin with) and was
 * removed in round 6.) */
typedef struct rv_comm_desc {
  void* comm; int world; int rank;
  rv_allreduce_fn allreduce;
  void* grad_bf16; /* optional: bf16 payload -- each rank's summed gradient rounded to bf16 into this arena */
                   /*   (as many 2-byte elements as the fp32 arenas hold floats) and summed by the collective in bf16;  */
                   /*   NULL: fp32, the exact mean of the ranks' fp32 gradients                                         */
  void* comm_stream; /* the stream the collectives are issued on, or NULL: one high-priority stream per process,     */
                   /*   created by the library.  Why a caller may want to choose: the HIP runtime multiplexes streams   */
                   /*   onto a few hardware queues (GPU_MAX_HW_QUEUES, default 4), and when the collective stream shares */
                   /*   a queue with the compute stream the runtime resolves their cross-stream waits on the host --    */
                   /*   every kernel of the step then starts ~50 us late (880 us instead of 255 us per step at one      */
                   /*   rank).  ddp.pick_comm_stream times a short ping-pong against the compute stream and hands over  */
                   /*   the first candidate that is not affected.  The stream stays the caller's.                       */
} rv_comm_desc;
int rv_plan_attach_comm(rv_plan*, const rv_comm_desc*);
int rv_plan_step_ddp(rv_plan*, const float* x, const float* eps, float* recon_out, float kl_beta,
                     float lr, unsigned long long seed, void* stream);
/* Device pointers into the workspace for tests (name: "mulv","z","h1","h3","dP4",...). */
void* rv_plan_buffer(rv_plan*, const char* name, long* n_bytes);

/* ---- hipGraph capture of any sequence of the calls above on one stream ---- */
typedef struct rv_graph rv_graph;
int rv_graph_begin(void* stream);
int rv_graph_end(void* stream, rv_graph** out);
int rv_graph_launch(rv_graph*, void* stream);
void rv_graph_destroy(rv_graph*);

#ifdef __cplusplus
}
#endif
#endif /* RAWVAE_HIP_H */
