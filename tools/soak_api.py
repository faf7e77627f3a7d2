#!/usr/bin/env python3
"""Long run of the reference's training loop UNCHANGED (train.py:184-193) through the drop-in surface -- `rawvae.model.VAE`,
`loss_function`, `torch.optim.Adam` -- i.e. through the one-node forward, the one-node loss with its direct `backward()`
and the optimizer hook (round 5): loss every N steps, device memory in use, finiteness of parameters and optimizer state
at the end, and the same loop with `fused_loss = False` (round 4's autograd route) from the same weights and batches.
    python tools/soak_api.py [--steps 20000]     -> profiles/r05_soak_api.txt"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from rawaudiovae_kelsey_amd.synth import make_frames, make_params  # noqa: E402
from rawvae.model import VAE, loss_function  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20000)
a = ap.parse_args()
S, H, L, B = 1024, 2048, 64, 4096
xs = [torch.from_numpy(make_frames(B, S, 100 + i)).cuda() for i in range(16)]


def run(fused_loss):
    m = VAE(S, H, L)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in make_params(S, H, L, 0).items()})
    m = m.cuda().manual_seed(5)
    m.fused_loss = fused_loss
    opt = torch.optim.Adam(m.parameters(), lr=1e-4)
    marks, t0 = [], time.perf_counter()
    for i in range(a.steps):
        x = xs[i % 16]
        opt.zero_grad()
        recon, mu, logvar = m(x)
        loss = loss_function(recon, x, mu, logvar, 1e-4, S)
        loss.backward()
        opt.step()
        if (i + 1) % (a.steps // 10) == 0:
            marks.append((i + 1, float(loss.item()), torch.cuda.memory_allocated() / 2 ** 20))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    finite = all(bool(torch.isfinite(p).all()) for p in m.parameters()) and all(
        bool(torch.isfinite(s_["exp_avg"]).all()) and bool(torch.isfinite(s_["exp_avg_sq"]).all()) for s_ in opt.state.values())
    return marks, dt, finite, float(opt.state[m.fc4.weight]["step"])


for name, fl in (("one-node loss, loss.backward() on the calling thread", True), ("loss on the general autograd route (round 4)", False)):
    marks, dt, finite, t = run(fl)
    print("%s: %d steps in %.2f s = %.1f us/step, optimizer at step %d, parameters and moments finite: %s" % (name, a.steps, dt, dt / a.steps * 1e6, int(t), finite))
    print("   step: loss / MiB in use   " + "  ".join("%d: %.5f / %.0f" % mk for mk in marks))
