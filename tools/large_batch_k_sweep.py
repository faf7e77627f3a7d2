"""Large batch (default.ini's 131072): launch time against K -- fixed (per-tile prologue + epilogue, HBM streaming of the
epilogue's operands) versus per-K-tile cost -- for the latent GEMM forms (heads + reparameterisation forward with eps given /
generated, dz + reparameterisation backward alone and with dW3 beside it at 8 / 16 / 32 K splits) and for fc4's forward + loss
on the two 256 x 256 loops and on 256 x 128 tiles, with the plain bias + ReLU forward (tile lists) beside it.
    python tools/large_batch_k_sweep.py  ->  profiles/r06_large_batch_k_sweep.txt"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rawaudiovae_kelsey_amd._lib import lib
Lb = lib(); st = torch.cuda.current_stream().cuda_stream or None
B, Lp, S = int(os.environ.get("BB", 131072)), 256, 1024
P = lambda t: None if t is None else t.data_ptr()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize(); v = []
    for r in range(5):
        e0.record()
        for _ in range(n): fn()
        e1.record(); e1.synchronize(); v.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(v)[2]
for H in (128, 512, 1024, 2048):
    rnd = lambda r, c, s=1.0: (torch.randn(r, c, device="cuda") * s).to(torch.bfloat16)
    dp3, w3, h1, wh = rnd(B, H, 1e-3), rnd(H, Lp, 0.1), rnd(B, H), rnd(2 * Lp, H, 0.05)
    mulv = torch.randn(B, 2 * Lp, device="cuda") * 0.3; eps = torch.randn(B, Lp, device="cuda")
    dmulv = torch.empty(B, 2 * Lp, dtype=torch.bfloat16, device="cuda"); dbh = torch.empty(B // 16, 2 * Lp, device="cuda")
    ctr = torch.ones(1, dtype=torch.int64, device="cuda"); bh = torch.zeros(2 * Lp, device="cuda")
    zo = torch.empty(B, Lp, dtype=torch.bfloat16, device="cuda"); klo = torch.empty(B * Lp // 1024, device="cuda"); mv2 = torch.empty(B, 2 * Lp, device="cuda")
    eo = torch.empty(B, Lp, device="cuda")
    zz = rnd(B, Lp); 
    t_b = timeit(lambda: Lb.rv_latent_bwd(P(dp3), H, P(w3), Lp, B, H, Lp, B, Lp, S, P(mulv), P(eps), 1e-4, None, None, P(dmulv), P(dbh), None, 0, None, 0, None, P(ctr), 4, None, Lp, None, Lp, 0, st))
    res = []
    for s3 in (8, 16, 32):
        dw3 = torch.empty(s3, H, Lp, device="cuda")
        res.append(timeit(lambda: Lb.rv_latent_bwd(P(dp3), H, P(w3), Lp, B, H, Lp, B, Lp, S, P(mulv), P(eps), 1e-4, None, None, P(dmulv), P(dbh), None, 0, None, 0, None, P(ctr), 4, P(zz), Lp, P(dw3), Lp, s3, st)))
    t_f = timeit(lambda: Lb.rv_latent_fwd(P(h1), H, P(wh), H, P(bh), None, 0, None, B, H, Lp, B, Lp, P(eps), None, 0, P(ctr), P(mv2), P(zo), P(klo), None, 0, st))
    t_g = timeit(lambda: Lb.rv_latent_fwd(P(h1), H, P(wh), H, P(bh), None, 0, None, B, H, Lp, B, Lp, None, P(eo), 5, P(ctr), P(mv2), P(zo), P(klo), None, 0, st))
    print("B %d Hp %5d (%3d K tiles): dz alone %7.1f  dz + dW3 (8/16/32 splits) %7.1f %7.1f %7.1f   heads fwd eps given %7.1f  generated %7.1f" % (B, H, H // 64, t_b, res[0], res[1], res[2], t_f, t_g), flush=True)

# ---- fc4 forward + loss
S = 1024
xf = torch.rand(B, S, device="cuda") * 2 - 1
outS = torch.empty(B, S, dtype=torch.bfloat16, device="cuda")
bS = torch.randn(S, device="cuda")
for H in (128, 512, 1024, 2048):
    h, W4 = rnd(B, H), rnd(S, H)
    cs = torch.empty(B // 64 * 2048, dtype=torch.float32, device="cuda"); msep = torch.empty(B // 64 * 16, dtype=torch.float32, device="cuda")
    fc4 = lambda: Lb.rv_decode_out_loss_fwd(P(h), H, P(W4), H, P(bS), B, S, H, B, S, P(xf), S, None, S, P(outS), S, P(msep), P(cs), st)
    fwd = lambda: Lb.rv_linear_fwd(P(h), H, P(W4), H, P(bS), B, S, H, 1, P(outS), S, st)
    out = []
    for t in (7, 5, 2):
        Lb.rv_gemm_force_tile(t)
        out.append(timeit(fc4, 3))
        Lb.rv_gemm_force_tile(-1)
    Lb.rv_gemm_force_tile(7); f7 = timeit(fwd, 3); Lb.rv_gemm_force_tile(-1)
    print("B %d K %5d (%2d K tiles): fc4 forward + loss  ping-pong %7.1f  two-slot ring %7.1f  256x128 %7.1f   plain forward (bias + ReLU), tile lists %7.1f" % (B, H, H // 64, out[0], out[1], out[2], f7), flush=True)
