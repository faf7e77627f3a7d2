#!/usr/bin/env python3
"""Nothing but full local steps of the fp8 weight path at C2 (for a profiler: `tools/prof_fp8.sh`)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from rawaudiovae_kelsey_amd.engine import TrainEngine  # noqa: E402
from rawaudiovae_kelsey_amd.synth import make_frames, make_params  # noqa: E402

S, H, L, B = 1024, 2048, 64, 4096
xs = [torch.from_numpy(make_frames(B, S, 3 + i)).cuda() for i in range(4)]
e = TrainEngine(S, H, L, B, kl_beta=1e-4, lr=1e-4, seed=1, fp8=True)
e.load_params(make_params(S, H, L, 0))
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 510):
        e.step(xs[i % 4], stream=st)
st.synchronize()
print("loss %.5f" % e.last_loss()[0])
