#!/usr/bin/env python3
"""Every launch group of the training step, timed IN the step (bench.py's method: hipGraph of N steps minus the same
graph without the group), at any shape:  python tools/launches.py [--shape S H L B] [--fp8]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from rawaudiovae_kelsey_amd.engine import TrainEngine  # noqa: E402
from rawaudiovae_kelsey_amd.synth import make_frames, make_params  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shape", type=int, nargs=4, default=[1024, 2048, 64, 4096], metavar=("S", "H", "L", "B"))
ap.add_argument("--fp8", action="store_true")
ap.add_argument("--json", action="store_true")
a = ap.parse_args()
S, H, L, B = a.shape
e = TrainEngine(S, H, L, B, kl_beta=1e-4, lr=1e-4, seed=1, fp8=a.fp8)
e.load_params(make_params(S, H, L, 0))
x = torch.from_numpy(make_frames(B, S, 3)).cuda()
comp = torch.cuda.Stream()
torch.cuda.synchronize()
with torch.cuda.stream(comp):
    for _ in range(5):
        e.step(x, stream=comp)
    comp.synchronize()
    rows, noise = bench.time_launches_in_step(e, x, steps=max(1, 10 * 4096 // max(B, 4096)))
if a.json:
    print(json.dumps(rows))
tot = 0.0
for r in rows:
    tot += r["us"]
    print("%d %8.1f us  mfma %5.3f  hbm %5.3f  %s" % (r["launch"], r["us"], r.get("mfma_frac", 0.0), r["hbm_frac"], r["kernel"][:120]))
F = (10 * S * H + 18 * H * L) * B
print("sum %.1f us  (noise %.2f)  step_mfma_frac of the sum %.4f   shape S=%d H=%d L=%d B=%d" % (tot, noise, F / (tot * 1e-6) / 2.5e15, S, H, L, B))
