#!/usr/bin/env python3
"""us per step of the C2 training step in one engine configuration (for A/B loops over environment knobs on ONE box).
    python tools/step_time.py [--fp8] [--steps 300] [--reps 5] [--tag TEXT] [--shape S H L B] [--graph]
--shape: any model / batch (default C2); --graph: replay one captured step instead of eager launches."""
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
ap.add_argument("--shape", type=int, nargs=4, default=[1024, 2048, 64, 4096], metavar=("S", "H", "L", "B"))
ap.add_argument("--graph", action="store_true")
a = ap.parse_args()
S, H, L, B = a.shape
a.steps = max(10, a.steps * 4096 // max(B, 4096))
NX = 8 if B <= 16384 else 2
xs = [torch.from_numpy(make_frames(B, S, 3 + i)).cuda() for i in range(NX)]
e = TrainEngine(S, H, L, B, kl_beta=1e-4, lr=1e-4, seed=1, fp8=a.fp8)
e.load_params(make_params(S, H, L, 0))
st = torch.cuda.Stream()
reps = []
with torch.cuda.stream(st):
    for i in range(30):
        e.step(xs[i % NX], stream=st)
    st.synchronize()
    g = None
    if a.graph:
        from rawaudiovae_kelsey_amd.engine import Graph
        g = Graph(st)
        with g:
            e.step(xs[0], stream=st)
        g.launch()
        st.synchronize()
    for r in range(a.reps):
        t0 = time.perf_counter()
        for i in range(a.steps):
            if g is not None:
                g.launch()
            else:
                e.step(xs[i % NX], stream=st)
        st.synchronize()
        reps.append((time.perf_counter() - t0) / a.steps * 1e6)
reps.sort()
F = (10 * S * H + 18 * H * L) * B
med = reps[len(reps) // 2]
print("%-44s S=%d H=%d L=%d B=%d  %8.2f us/step (min %.2f max %.2f)  %.2f Mframes/s  step_mfma_frac %.4f  loss %.5f" % (
    a.tag or ("fp8" if a.fp8 else "bf16"), S, H, L, B, med, reps[0], reps[-1], B / med, F / (med * 1e-6) / 2.5e15, e.last_loss()[0]))
