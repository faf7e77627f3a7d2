"""The four large GEMM shapes of the step at a large batch, per tile configuration (standalone, HIP events)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rawaudiovae_kelsey_amd._lib import lib
Lb = lib(); st = torch.cuda.current_stream().cuda_stream or None
B = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
S, H = 1024, 2048
rnd = lambda r, c: torch.randn(r, c, device="cuda").to(torch.bfloat16)
x, h, dp4 = rnd(B, S), rnd(B, H), rnd(B, S)
W1, W4 = rnd(H, S), rnd(S, H)
bH, bS = torch.randn(H, device="cuda"), torch.randn(S, device="cuda")
xf = torch.rand(B, S, device="cuda") * 2 - 1
outH, outS = torch.empty(B, H, dtype=torch.bfloat16, device="cuda"), torch.empty(B, S, dtype=torch.bfloat16, device="cuda")
slabs = torch.empty(64 * 2048 * 1024 // 2, dtype=torch.float32, device="cuda")
us = torch.empty(64 * 64 * 32, dtype=torch.float32, device="cuda")
cs = torch.empty(B // 64 * H, dtype=torch.float32, device="cuda"); msep = torch.empty(B // 64 * 16, dtype=torch.float32, device="cuda")
P = lambda t: t.data_ptr()
def forced(tile, fn):
    def f():
        Lb.rv_gemm_force_tile(tile); fn(); Lb.rv_gemm_force_tile(-1)
    return f
fc1 = lambda: Lb.rv_linear_fwd(P(x), S, P(W1), S, P(bH), B, H, S, 1, P(outH), H, st)
fc4 = lambda: Lb.rv_decode_out_loss_fwd(P(h), H, P(W4), H, P(bS), B, S, H, B, S, P(xf), S, None, S, P(outS), S, P(msep), P(cs), st)
dg4 = lambda: Lb.rv_linear_dgrad(P(dp4), S, P(W4), H, B, H, S, P(h), H, P(outH), H, P(cs), None, 0, 1, st)
def wg(dy, xx, M, N, s, tile): return lambda: Lb.rv_linear_wgrad(P(dy), M, P(xx), N, M, N, B, s, tile, P(slabs), N, 1, P(us), st)
GF = 2.0 * B * S * H
cases = {}
for t, nm in ((-1, "auto"), (2, "256x128"), (7, "256x256pp"), (5, "256x256ring")):
    cases["fc1 fwd  " + nm] = forced(t, fc1) if t >= 0 else fc1
    cases["fc4 fwd+loss " + nm] = forced(t, fc4) if t >= 0 else fc4
    cases["fc4 dgrad " + nm] = forced(t, dg4) if t >= 0 else dg4
for s in (8, 16, 32):
    cases["dW4 wgrad 256x256pp s%d" % s] = wg(dp4, h, S, H, s, 7)
    cases["dW1 wgrad 256x256pp s%d" % s] = wg(h, x, H, S, s, 7)
cases["dW4 wgrad 256x128 s8"] = wg(dp4, h, S, H, 8, 2)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
res = {k: [] for k in cases}
for r in range(4):
    for k, fn in cases.items():
        try:
            fn(); torch.cuda.synchronize()
        except Exception as exc:
            res[k] = str(exc)[:90]; continue
        e0.record()
        for _ in range(3): fn()
        e1.record(); e1.synchronize()
        if r: res[k].append(e0.elapsed_time(e1) / 3 * 1e3)
for k, v in sorted(res.items()):
    if isinstance(v, str): print("%-30s %s" % (k, v)); continue
    v = sorted(v); m = v[len(v) // 2]; print("%-30s %8.1f us  mfma_frac %.3f" % (k, m, GF / (m * 1e-6) / 2.5e15))
