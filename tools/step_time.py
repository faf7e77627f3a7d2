#!/usr/bin/env python3
"""us per step of the C2 training step in one engine configuration (for A/B loops over environment knobs on ONE box).
    python tools/step_time.py [--fp8] [--steps 300] [--reps 5] [--tag TEXT]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from rawaudiovae_kelsey_amd.engine import TrainEngine  # noqa: E402
from rawaudiovae_kelsey_amd.synth import make_frames, make_params  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--fp8", action="store_true")
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--tag", default="")
a = ap.parse_args()
S, H, L, B = 1024, 2048, 64, 4096
xs = [torch.from_numpy(make_frames(B, S, 3 + i)).cuda() for i in range(8)]
e = TrainEngine(S, H, L, B, kl_beta=1e-4, lr=1e-4, seed=1, fp8=a.fp8)
e.load_params(make_params(S, H, L, 0))
st = torch.cuda.Stream()
reps = []
with torch.cuda.stream(st):
    for i in range(30):
        e.step(xs[i % 8], stream=st)
    st.synchronize()
    for r in range(a.reps):
        t0 = time.perf_counter()
        for i in range(a.steps):
            e.step(xs[i % 8], stream=st)
        st.synchronize()
        reps.append((time.perf_counter() - t0) / a.steps * 1e6)
reps.sort()
print("%-44s %7.2f us/step (min %.2f max %.2f)  loss %.5f" % (a.tag or ("fp8" if a.fp8 else "bf16"), reps[len(reps) // 2], reps[0], reps[-1],
                                                             e.last_loss()[0]))
