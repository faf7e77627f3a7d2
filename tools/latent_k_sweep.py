"""K sweep of the latent GEMM forms (csrc/latent.hip) at the reference latent width: fixed cost vs per-K-tile cost.
    python tools/latent_k_sweep.py  ->  profiles/r06_latent_k_sweep.txt"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rawaudiovae_kelsey_amd._lib import lib
Lb = lib(); st = torch.cuda.current_stream().cuda_stream or None
B, Lp, S = 4096, 256, 1024
P = lambda t: None if t is None else t.data_ptr()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def timeit(fn):
    fn(); torch.cuda.synchronize(); v = []
    for r in range(5):
        e0.record()
        for _ in range(20): fn()
        e1.record(); e1.synchronize(); v.append(e0.elapsed_time(e1) / 20 * 1e3)
    return sorted(v)[2]
for H in (128, 256, 512, 1024, 2048, 4096):
    rnd = lambda r, c, s=1.0: (torch.randn(r, c, device="cuda") * s).to(torch.bfloat16)
    dp3, w3, h1, wh = rnd(B, H, 1e-3), rnd(H, Lp, 0.1), rnd(B, H), rnd(2 * Lp, H, 0.05)
    mulv = torch.randn(B, 2 * Lp, device="cuda") * 0.3; eps = torch.randn(B, Lp, device="cuda")
    dmulv = torch.empty(B, 2 * Lp, dtype=torch.bfloat16, device="cuda"); dbh = torch.empty(B // 16, 2 * Lp, device="cuda")
    ctr = torch.ones(1, dtype=torch.int64, device="cuda"); bh = torch.zeros(2 * Lp, device="cuda")
    zo = torch.empty(B, Lp, dtype=torch.bfloat16, device="cuda"); klo = torch.empty(B * Lp // 1024, device="cuda"); mv2 = torch.empty(B, 2 * Lp, device="cuda")
    t_b = timeit(lambda: Lb.rv_latent_bwd(P(dp3), H, P(w3), Lp, B, H, Lp, B, Lp, S, P(mulv), P(eps), 1e-4, None, None, P(dmulv), P(dbh), None, 0, None, 0, None, P(ctr), 4, None, Lp, None, Lp, 0, st))
    t_f = timeit(lambda: Lb.rv_latent_fwd(P(h1), H, P(wh), H, P(bh), None, 0, None, B, H, Lp, B, Lp, P(eps), None, 0, P(ctr), P(mv2), P(zo), P(klo), None, 0, st))
    print("Hp %5d (%3d K tiles): dz + reparam bwd %6.1f us   heads + reparam fwd %6.1f us" % (H, H // 64, t_b, t_f))
