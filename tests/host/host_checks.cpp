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
    fails += rv_gemm_pick(m, n, k, 16, &bm, &bn, &sp) != 0;
    fails += !(m % bm == 0 && n % bn == 0 && (k / 64) % sp == 0);
    fails += rv_dgrad_wgrad_pick(m, n, k, &paired, &bm, &sp) != 0;
    fails += !(m % bm == 0);
  }
  fails += rv_gemm_pick(100, 64, 64, 16, &bm, &bn, &sp) == 0;            // not a multiple of 64
  rv_param_desc d[20]; memset(d, 0, sizeof d);
  float x[4];
  fails += rv_adam_multi(d, 20, x, x, x, NULL, 1e-3f, 1.f, (const long long*)x, NULL) == 0;  // > 16 descriptors
  fails += rv_adam_multi(d, 1, x, x, x, NULL, 1e-3f, 1.f, (const long long*)x, NULL) == 0;   // invalid descriptor
  fails += rv_linear_fwd(NULL, 0, NULL, 0, NULL, 64, 64, 64, 1, NULL, 0, NULL) == 0;
  fails += rv_linear_fp32(x, 1, x, 1, NULL, 0, 1, 1, 0, x, 1, NULL) == 0;
  printf("host checks: %d failures; last error: %s\n", fails, rv_last_error());
  return fails != 0;
}
