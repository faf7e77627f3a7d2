// Host-side checks of the C ABI under AddressSanitizer + UBSan (tests/test_host_cpu.py builds the four
// translation units with `-fsanitize=address,undefined -fno-gpu-sanitize` and runs this on the CPU):
// tile/split pickers over a sweep of extents, and the argument validation / error paths that return
// before any HIP call.  No GPU needed.
#include <stdio.h>
#include <string.h>
#include "rawvae_hip.h"
int main() {
  long Bp, Sp, Hp, Lp; int bm, bn, sp, paired;
  int fails = 0;
  fails += rv_pad_dims(4096, 1024, 2048, 64, &Bp, &Sp, &Hp, &Lp) != 0;
  fails += !(Bp == 4096 && Sp == 1024 && Hp == 2048 && Lp == 64);
  fails += rv_pad_dims(1, 1, 1, 300, &Bp, &Sp, &Hp, &Lp) == 0;          // latent > 256: rejected
  fails += strstr(rv_last_error(), "latent_dim") == NULL;
  for (long m = 64; m <= 8192; m *= 2) for (long n = 64; n <= 4096; n *= 2) for (long k = 64; k <= 8192; k *= 4) {
    fails += rv_gemm_plan(RV_PLAN_GEMM, m, n, k, 16, &bm, &bn, &sp, NULL) != 0;
    fails += !(m % bm == 0 && n % bn == 0 && (k / 64) % sp == 0);
    fails += rv_gemm_plan(RV_PLAN_PAIR, m, n, k, 0, &bm, NULL, &sp, &paired) != 0;
    fails += !(m % bm == 0);
  }
  fails += rv_gemm_plan(RV_PLAN_GEMM, 100, 64, 64, 16, &bm, &bn, &sp, NULL) == 0;            // not a multiple of 64
  rv_param_desc d[20]; memset(d, 0, sizeof d);
  float x[4];
  fails += rv_adam_multi(d, 20, x, x, x, NULL, NULL, 1e-3f, 1.f, (const long long*)x, NULL) == 0;  // > 16 descriptors
  fails += rv_adam_multi(d, 1, x, x, x, NULL, NULL, 1e-3f, 1.f, (const long long*)x, NULL) == 0;   // invalid descriptor
  fails += rv_linear_fwd(NULL, 0, NULL, 0, NULL, 64, 64, 64, 1, NULL, 0, NULL) == 0;
  fails += rv_linear_fp32(x, 1, x, 1, NULL, 0, 1, 1, 0, x, 1, NULL) == 0;
  rv_plan* pl = NULL;
  fails += rv_plan_create(&pl, 4096, 1024, 2048, 64) != 0;
  { int a_ = 0, b_ = 0; fails += rv_plan_riders(pl, &a_, &b_) == 0; }  // needs a bound plan
  fails += rv_plan_set_option(pl, RV_OPT_FP8, 1) == 0;                 // not bound
  rv_comm_desc c; memset(&c, 0, sizeof c);
  c.comm = x; c.world = 2; c.rank = 2;
  fails += rv_plan_attach_comm(pl, &c) == 0;                           // rank out of range
  c.rank = 0;
  fails += rv_plan_attach_comm(pl, &c) == 0;                           // no collective given
  fails += strstr(rv_last_error(), "no collective") == NULL;
  fails += rv_plan_attach_comm(pl, NULL) == 0;
  memset(&c, 0, sizeof c);
  fails += rv_plan_attach_comm(pl, &c) != 0;                           // comm == NULL: detach, always allowed
  fails += rv_gemm_plan(7, 256, 256, 256, 1, &bm, &bn, &sp, &paired) == 0;   // unknown query
  rv_plan_destroy(pl);
  printf("host checks: %d failures; last error: %s\n", fails, rv_last_error());
  return fails != 0;
}
