#!/usr/bin/env python3
"""Time the reference-style loop (zero_grad / model(x) / loss_function / backward / optimizer.step)
through the drop-in API path at C2, next to the fused engine."""
import os
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
m.fused_training = os.environ.get("RV_API_FUSED", "1") != "0"
opt = torch.optim.Adam(m.parameters(), lr=1e-4)
xs = [torch.from_numpy(make_frames(B, S, i)).cuda() for i in range(4)]


def step(x):
    opt.zero_grad()
    recon, mu, logvar = m(x)
    loss = loss_function(recon, x, mu, logvar, 1e-4, S)
    loss.backward()
    opt.step()
    return loss


for i in range(5):
    step(xs[i % 4])
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 50
for i in range(n):
    loss = step(xs[i % 4])
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print("API path, reference loop unchanged (%s + torch.optim.Adam): %.1f us/step, %.2f M frames/s, loss %.5f"
      % ("one autograd node on a step plan" if getattr(m, "fused_training", True) else "per-layer autograd Functions",
         dt * 1e6, B / dt / 1e6, loss.item()))
