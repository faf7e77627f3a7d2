#!/usr/bin/env python3
"""Step time of the C2 model vs per-GPU batch (hipGraph replay): how far the B=4096 tile count,
not the kernels, limits the MFMA fraction.  python tools/batch_sweep.py [B ...]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from rawaudiovae_kelsey_amd.engine import Graph, TrainEngine  # noqa: E402
from rawvae.model import VAE  # noqa: E402

S, H, L = 1024, 2048, 64
F = 10 * S * H + 18 * H * L
for B in [int(v) for v in sys.argv[1:]] or [2048, 4096, 8192, 16384, 32768]:
    torch.manual_seed(0)
    m = VAE(S, H, L).cuda()
    eng = TrainEngine(S, H, L, B, kl_beta=1e-4, lr=1e-4, seed=0)
    eng.adopt(m)
    x = torch.rand(B, S, device="cuda") * 2 - 1
    st = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(st):
        eng.step(x, stream=st)
        g = Graph(st)
        with g:
            eng.step(x, stream=st)
        for _ in range(20):
            g.launch()
        st.synchronize()
        t0 = time.perf_counter()
        n = max(20, 200 * 4096 // B)
        for _ in range(n):
            g.launch()
        st.synchronize()
        dt = (time.perf_counter() - t0) / n
    print(json.dumps({"B": B, "us_per_step": round(dt * 1e6, 1), "frames_per_s": round(B / dt),
                      "step_tflops": round(F * B / dt / 1e12, 1), "step_mfma_frac": round(F * B / dt / 2.5e15, 4)}))
    del eng, m, g
    torch.cuda.empty_cache()
