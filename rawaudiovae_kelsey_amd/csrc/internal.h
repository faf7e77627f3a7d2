// Launchers that only the step plan (plan.hip) calls: forms of public entry points with extra operands (fp8 copies,
// per-block maxima, frames read in place).  Not part of the C ABI (include/rawvae_hip.h) and not exported.
#pragma once
#include "../../include/rawvae_hip.h"

#define RV_INTERNAL extern "C" __attribute__((visibility("hidden")))

// The pickers behind rv_gemm_plan (gemm_launch.hip).
RV_INTERNAL int rv_gemm_pick(long Mp, long Np, long Kp, int max_splits, int* bm, int* bn, int* splits);
RV_INTERNAL int rv_gemm_tile(long Mp, long Np, int splits, int* bm, int* bn);
// whether rv_latent_fwd / rv_latent_bwd run their row-local kernels on this shape (1) or the GEMM forms (0): csrc/latent.hip
RV_INTERNAL int rv_latent_rowlocal(long Bp, long Hp, long Lp);
RV_INTERNAL int rv_latent_bwd_pp(long Bp, long Hp, long Lp);
RV_INTERNAL int rv_latent_bwd_tile_rows(long Bp, long Hp, long Lp);
// ... of the fused loss forward on fp8 operands (never 256 x 256)
RV_INTERNAL int rv_gemm_tile_fp8_loss(long Mp, long Np, int* bm, int* bn);
RV_INTERNAL int rv_dgrad_wgrad_pick(long Mp, long Np, long Kp, int* paired, int* bm_dgrad, int* splits);
RV_INTERNAL int rv_wgrad_adam_fits(long Mp, long Np, long Kp, int splits);

// rv_cast_pad_bf16 that also writes the fp8 operand (dst_fp8 may be NULL) and, when `fp8_state` is given, latches the
// delayed activation scale for this step in its first wave from the previous step's per-block maxima
// `amax_part[n_amax]` (state block layout: RV_OPT_FP8 in the public header), and dP1's scale likewise from the
// `n_amax2` maxima that follow them (rv_heads_bwd_ex; 0 = none).  dst_bf16 may be NULL when dst_fp8 is given.
RV_INTERNAL int rv_cast_pad_bf16_q8(const float* src, long rows, long cols, long ld_src, void* dst_bf16, long rows_p,
                                    long cols_p, long ld_dst, void* dst_fp8, long ld_fp8, float* fp8_state,
                                    const float* amax_part, int n_amax, int n_amax2, long long* step_counter, void* stream);
// The same from hop-strided frames of a resident fp32 waveform (rv_gather_frames + the cast in one kernel).
RV_INTERNAL int rv_gather_cast_frames(const float* audio, long n_samples, const long long* frame_index, long first_frame,
                                      long n_frames, long S, long hop, void* dst_bf16, long rows_p, long cols_p,
                                      long ld_dst, void* dst_fp8, long ld_fp8, float* fp8_state, const float* amax_part,
                                      int n_amax, int n_amax2, long long* step_counter, void* stream);
// rv_linear_fwd with every optional output of a bias/ReLU forward GEMM (NULL = not wanted): the output also as
// fp8(y * *q_scale) (the next layer's fp8 operand) and max|y| of every block in amax_part[block], from which the next
// step derives its scale (delayed scaling).
RV_INTERNAL int rv_linear_fwd_ex(const void* x_bf16, long ldx, const void* w_bf16, long ldw, const float* bias, long Mp,
                                 long Np, long Kp, int act, void* y_bf16, long ldy, void* y_fp8, long ldy_fp8,
                                 const float* q_scale, float* amax_part, void* stream);
// rv_linear_fwd / rv_decode_out_loss_fwd on fp8 operands (K extents and leading dims in fp8 elements).
RV_INTERNAL int rv_linear_fwd_fp8(const void* x_fp8, long ldx, const void* w_fp8, long ldw, const float* bias,
                                  const float* dq, long Mp, long Np, long Kp, int act, void* y_bf16, long ldy,
                                  void* stream);
// (dP4_fp8 != NULL: the epilogue also writes fp8(dP4 * *dp4_scale), the fp8 fc4 backward's operand; dP4_bf16 may then be NULL)
RV_INTERNAL int rv_decode_out_loss_fwd_fp8(const void* h3_fp8, long ldh, const void* w4_fp8, long ldw, const float* b4,
                                           const float* dq, long Bp, long Sp, long Hp, long B, long S, const float* x,
                                           long ldx, float* recon, long ld_recon, void* dP4_bf16, long ld_dp4,
                                           void* dP4_fp8, long ld_dp4q, const float* dp4_scale,
                                           float* mse_partial, float* db4_partial, void* stream);
// rv_decode_out_loss_fwd whose fp32 target rows are read in place from the waveform (dq != NULL: fp8 operands).
RV_INTERNAL int rv_decode_out_loss_fwd_frames(const void* h3, long ldh, const void* w4, long ldw, const float* b4,
                                              const float* dq, long Bp, long Sp, long Hp, long B, long S,
                                              const float* audio, long n_samples, const long long* frame_index,
                                              long first_frame, long hop, float* recon, long ld_recon, void* dP4_bf16,
                                              long ld_dp4, void* dP4_fp8, long ld_dp4q, const float* dp4_scale,
                                              float* mse_partial, float* db4_partial, void* stream);
// max|W1|, max|W4| of the fp8 weight shadows (n1 / n4 bytes) into the 2 x 1024 slots behind the fp8 state block
// (RV_OPT_FP8): the plan runs it behind the optimizer, the next step's first kernel turns it into the weight scales.
RV_INTERNAL int rv_fp8_wmax(const void* w1q, long n1, const void* w4q, long n4, float* fp8_state, void* stream);
// rv_latent_fwd with the fp8 forward's extra outputs of fc3 (NULL = not wanted): h3 also as fp8(h3 * *q_scale), and
// max|h3| of every wave's outputs in amax_part[8 * (Bp / 16)].
RV_INTERNAL int rv_latent_fwd_ex(const void* h_bf16, long ldh, const void* wh_bf16, long ldwh, const float* bias_heads,
                                 const void* w3_bf16, long ldw3, const float* bias3, long Bp, long Hp, long Lp, long B, long L,
                                 const float* eps_in, float* eps_out, unsigned long long seed, const long long* step_counter,
                                 float* mulv, void* z_bf16, float* kl_partial, void* h3_bf16, long ldh3, void* h3_fp8, long ldq,
                                 const float* q_scale, float* amax_part, void* stream);
// Device-side cross-stream signalling (elementwise.hip): publish `value` behind the stream's earlier work / hold the
// stream until the flag has reached `value` (bounded; timeouts are counted in *timeouts).
RV_INTERNAL int rv_flag_set(int* flag, int value, void* stream);
// ... and *copy_dst = *copy_src first (same one-wave kernel)
RV_INTERNAL int rv_flag_set_copy(int* flag, int value, const long long* copy_src, long long* copy_dst, void* stream);
RV_INTERNAL int rv_flag_wait(const int* flag, int value, int* timeouts, long max_ms, void* stream);
// rv_adam_multi that withholds the update when `*poison` is non-zero (poison may be NULL): the data-parallel step passes
// its count of flag waits that ran out, so that a step whose exchange did not complete in time changes no parameter.
RV_INTERNAL int rv_adam_multi_guarded(const rv_param_desc* descs, int n_desc, float* param, float* exp_avg, float* exp_avg_sq,
                                      float* grad_out, const void* grad_bf16, float lr, float grad_scale,
                                      const long long* step_counter, const int* poison, void* stream);
// The loss scalar (total, mse, kld) from a forward's partial sums, and rv_grad_finalize times a device-side scalar
// (elementwise.hip): the two pieces of rv_plan_loss / rv_plan_set_loss_grad (plan.hip).
RV_INTERNAL int rv_loss_from_partials(const float* mse_partial, int n_mse, const float* kl_partial, int n_kl, long B, long S,
                                      long L, float kl_beta, float* out3, void* stream);
RV_INTERNAL int rv_grad_finalize_scaled(const rv_param_desc* descs, int n_desc, void* grad_out, int out_bf16,
                                        const float* scale_dev, void* stream);
// rv_linear_wgrad_adam's launch shape (256 x 256 weight-gradient GEMM + rider blocks on the idle CUs) whose riders sum
// the gradient slabs of `descs` into a flat payload arena instead of updating them (gemm_launch.hip).
RV_INTERNAL int rv_linear_wgrad_finalize(const void* dy_bf16, long lddy, const void* x_bf16, long ldx, long Mp, long Np, long Kp,
                                         int splits, void* dw_slabs, long lddw, int slab_dtype, float* slab_unscale,
                                         const rv_param_desc* descs, int n_desc, void* grad_out, int out_bf16,
                                         int n_rider_blocks, void* stream);
// fc1's weight gradient + optimizer riders on fp8 operands (gemm_launch.hip), and the heads' backward that writes its
// fp8 left operand (latent.hip).
RV_INTERNAL int rv_linear_wgrad_adam_fp8(const void* dy_fp8, long lddy, const void* x_fp8, long ldx, const float* dq, long Mp,
                                         long Np, long Kp, int splits, void* dw, long lddw, int slab_dtype, float* slab_unscale,
                                         const rv_param_desc* descs, int n_desc, float* param, float* exp_avg,
                                         float* exp_avg_sq, float lr, float grad_scale, const long long* step_counter,
                                         int n_adam_blocks, void* stream);
RV_INTERNAL int rv_linear_wgrad_finalize_fp8(const void* dy_fp8, long lddy, const void* x_fp8, long ldx, const float* dq, long Mp,
                                             long Np, long Kp, int splits, void* dw, long lddw, int slab_dtype,
                                             float* slab_unscale, const rv_param_desc* descs, int n_desc, void* grad_out,
                                             int out_bf16, int n_rider_blocks, void* stream);
RV_INTERNAL int rv_heads_bwd_ex(const void* dmulv_bf16, const void* wh_bf16, long ldw, const void* h1_bf16, long ldh, long Bp,
                                long Hp, long Lp, void* dp1_bf16, long ldp, float* db1_partial, float* dwh_slabs, long lddw,
                                void* dp1_fp8, long ldq, const float* q_scale, float* amax_part, float* dwh_unscale,
                                void* stream);
// One-shot: the next paired dgrad + wgrad launch (bf16 or fp8) signals `hip_event` when it completes -- the event is the
// launch's own completion signal (hipExtLaunchKernelGGL), cheaper on both streams than a hipEventRecord behind it.
// Not under stream capture.  Returns 1 when an event armed earlier was still pending, i.e. no paired launch took it
// (other tile forms): call with NULL behind the backward to disarm and to learn which.
RV_INTERNAL int rv_pair_stop_event(void* hip_event);
// The paired fc4 backward on fp8 operands (gemm_launch.hip) and whether the extents allow it.
RV_INTERNAL int rv_dgrad_wgrad_fp8_fits(long Mp, long Np, long Kp, int splits);
RV_INTERNAL int rv_linear_dgrad_wgrad_fp8(const void* dy_fp8, long lddy, const void* w_fp8, long ldw, const void* x_fp8, long ldx,
                                          const void* mask, long ldmask, int mask_is_fp8, const float* dq_dgrad,
                                          const float* dq_wgrad, long Mp, long Np, long Kp, void* dx_bf16, long lddx,
                                          float* colsum_partial, void* dw_slabs, long lddw, int splits, int slab_dtype,
                                          float* slab_unscale, void* stream);
