/* Test hook of librawvae_hip.so -- NOT part of the product ABI (nothing in rawaudiovae_kelsey_amd/ calls it; the GPU
 * tests use it to run every block-tile configuration on small shapes).  Process-wide, not thread-safe. */
#ifndef RAWVAE_HIP_DIAG_H
#define RAWVAE_HIP_DIAG_H
#ifdef __cplusplus
extern "C" {
#endif
/* Pin the block tile every later GEMM launch uses wherever it divides the extents (0: 64x64, 1: 128x128 with 4 waves,
 * 2: 256x128 with 8 waves, 3: 256x128 with 4 waves, 4: 128x128 with 8 waves, 5: 256x256 with the two-slot ring loop,
 * 7: 256x256 with the ping-pong loop); -1 = the picker's choice.  108 / 102 select the ping-pong (default) / ring
 * main loop of the paired 256x256 launch (the ring is what odd K-tile counts get); 111 / 110 switch the tile-list form of
 * large forward GEMMs on (default) / off. */
int rv_gemm_force_tile(int tile);
/* Leave launches out of the following FULL steps of a plan (rv_plan_step with RV_PHASE_ALL_LOCAL): bit k of `mask`
 * skips launch k -- 0 cast, 1 fc1 forward, 2 latent forward (heads + reparameterisation + fc3), 3 fc4 forward + loss,
 * 4 paired fc4 backward, 5 latent backward, 6 heads backward, 7 fc1 weight gradient + optimizer riders, 8 Adam(fc1).
 * bench.py times every launch of the step IN the step with it: (a graph of steps) - (the same graph without launch k).
 * Results of a step with launches missing are meaningless.  0 restores the full step. */
struct rv_plan;
int rv_plan_diag_skip(struct rv_plan* plan, unsigned mask);
#ifdef __cplusplus
}
#endif
#endif
