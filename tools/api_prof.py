#!/usr/bin/env python3
"""Where the host time of the reference-style loop goes (API path, C2): cProfile over 200 steps, top entries by
cumulative and by own time, plus GPU-only time per step (the same loop, events around 50 steps).
    python tools/api_prof.py"""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rawaudiovae_kelsey_amd.synth import make_frames, make_params  # noqa: E402
from rawvae.model import VAE, loss_function  # noqa: E402

S, H, L, B = 1024, 2048, 64, 4096
m = VAE(S, H, L)
m.load_state_dict({k: torch.from_numpy(v) for k, v in make_params(S, H, L, 0).items()})
m = m.cuda()
kw = {}
if os.environ.get("RV_API_ADAM") == "fused":
    kw["fused"] = True
elif os.environ.get("RV_API_ADAM") == "single":
    kw["foreach"] = False
opt = torch.optim.Adam(m.parameters(), lr=1e-4, **kw)
xs = [torch.from_numpy(make_frames(B, S, i)).cuda() for i in range(4)]


def step(x):
    opt.zero_grad()
    recon, mu, logvar = m(x)
    loss = loss_function(recon, x, mu, logvar, 1e-4, S)
    loss.backward()
    opt.step()
    return loss


for i in range(10):
    step(xs[i % 4])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(100):
    step(xs[i % 4])
host = (time.perf_counter() - t0) / 100
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 100
from rawaudiovae_kelsey_amd import optim_hook  # noqa: E402
print("optimizer hook:", optim_hook.stats)
print("100 steps: host enqueue %.1f us/step, wall %.1f us/step (Adam: %s)" % (host * 1e6, wall * 1e6, os.environ.get("RV_API_ADAM", "default")))
pr = cProfile.Profile()
pr.enable()
for i in range(200):
    step(xs[i % 4])
pr.disable()
torch.cuda.synchronize()
for key in ("cumulative", "tottime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(22)
    print("\n".join(l for l in s.getvalue().splitlines() if l.strip())[:6000])
