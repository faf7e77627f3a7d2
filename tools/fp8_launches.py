#!/usr/bin/env python3
"""Every launch of the C2 step, timed IN the step with bench.py's method (graph with minus graph without the launch), for the
bf16 step, the fp8 forward and the full fp8 weight path.    python tools/fp8_launches.py   -> profiles/r04_fp8_launches.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from rawaudiovae_kelsey_amd.engine import TrainEngine
from rawaudiovae_kelsey_amd.synth import make_frames, make_params
S, H, L, B = 1024, 2048, 64, 4096
x = torch.from_numpy(make_frames(B, S, 3)).cuda()
st = torch.cuda.Stream()
for mode in (False, "fwd", True):
    e = TrainEngine(S, H, L, B, kl_beta=1e-4, lr=1e-4, seed=1, fp8=mode)
    e.load_params(make_params(S, H, L, 0))
    with torch.cuda.stream(st):
        for _ in range(20): e.step(x, stream=st)
        st.synchronize()
        rows, noise = bench.time_launches_in_step(e, x)
    print("fp8 =", mode, " sum %.1f" % sum(r["us"] for r in rows))
    for r in rows: print("   %d %6.1f  %s" % (r["launch"], r["us"], r["kernel"][:70]))
