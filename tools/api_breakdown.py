#!/usr/bin/env python3
"""Per-segment host time of the reference-style loop at C2 (perf_counter around each call of the loop and around the two
Python backward functions, which run on autograd's worker thread and are invisible to a profiler of the main thread).
    python tools/api_breakdown.py        -> profiles/r05_api_breakdown.txt"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rawaudiovae_kelsey_amd.synth import make_frames, make_params
from rawaudiovae_kelsey_amd import fused, ops
from rawvae.model import VAE, loss_function
S, H, L, B = 1024, 2048, 64, 4096
m = VAE(S, H, L)
m.load_state_dict({k: torch.from_numpy(v) for k, v in make_params(S, H, L, 0).items()})
m = m.cuda()
opt = torch.optim.Adam(m.parameters(), lr=1e-4)
xs = [torch.from_numpy(make_frames(B, S, i)).cuda() for i in range(4)]
acc = {}
pc = time.perf_counter
def wrap(cls, name):
    f = getattr(cls, name)
    def w(*a, **k):
        t = pc(); r = f(*a, **k); acc[cls.__name__ + "." + name] = acc.get(cls.__name__ + "." + name, 0.0) + pc() - t
        return r
    setattr(cls, name, staticmethod(w))
wrap(fused.VaeFn, "backward"); wrap(ops.LossFn, "backward"); wrap(fused.VaeFn, "forward"); wrap(ops.LossFn, "forward")
wrap(fused.FusedLossFn, "forward"); wrap(fused.FusedLossFn, "backward")
m.fused_loss = os.environ.get("RV_FUSED_LOSS", "1") != "0"     # 0: loss_function through the general autograd route
def step(x, T):
    t0 = pc(); opt.zero_grad(); t1 = pc()
    recon, mu, logvar = m(x); t2 = pc()
    loss = loss_function(recon, x, mu, logvar, 1e-4, S); t3 = pc()
    loss.backward(); t4 = pc()
    opt.step(); t5 = pc()
    for k, v in zip(("zero_grad", "model(x)", "loss_function", "backward", "opt.step"), (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
        T[k] = T.get(k, 0.0) + v
for i in range(20): step(xs[i % 4], {})
torch.cuda.synchronize(); acc.clear()
T = {}; N = 300
t0 = pc()
for i in range(N): step(xs[i % 4], T)
host = pc() - t0
torch.cuda.synchronize()
print("host %.1f us/step" % (host / N * 1e6))
for k, v in T.items(): print("  %-14s %6.1f us" % (k, v / N * 1e6))
for k, v in acc.items(): print("    inside %-18s %6.1f us" % (k, v / N * 1e6))
