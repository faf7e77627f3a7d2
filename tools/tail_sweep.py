#!/usr/bin/env python3
"""Step time at C2, bf16 and full fp8 weight path, for the share of the optimizer riders' table that the weight-gradient GEMM's
blocks take over behind their tiles (RV_WGRAD_TAIL_PCT; one process per value: the library reads it once).
    for p in 0 3 6 9 12 15 18 22; do RV_WGRAD_TAIL_PCT=$p python tools/tail_sweep.py; done   -> profiles/r04_tail_sweep.txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rawaudiovae_kelsey_amd.engine import TrainEngine
from rawaudiovae_kelsey_amd.synth import make_frames, make_params
S, H, L, B = 1024, 2048, 64, 4096
xs = [torch.from_numpy(make_frames(B, S, 3 + i)).cuda() for i in range(4)]
st = torch.cuda.Stream()
out = []
for mode in (False, True):
    e = TrainEngine(S, H, L, B, kl_beta=1e-4, lr=1e-4, seed=1, fp8=mode)
    e.load_params(make_params(S, H, L, 0))
    with torch.cuda.stream(st):
        for i in range(50): e.step(xs[i % 4], stream=st)
        st.synchronize()
        ts = []
        for r in range(5):
            t0 = time.perf_counter()
            for i in range(300): e.step(xs[i % 4], stream=st)
            st.synchronize()
            ts.append((time.perf_counter() - t0) / 300 * 1e6)
    out.append("%s %.1f (loss %.5f)" % ("fp8" if mode else "bf16", sorted(ts)[2], e.last_loss()[0]))
print("tail pct %s: " % os.environ.get("RV_WGRAD_TAIL_PCT", "default") + "   ".join(out))
