#!/usr/bin/env python3
"""Step time and every launch group's in-step time against the per-GPU batch (4096 ... 131072 = default.ini's), for the
benchmark model (L = 64) and the reference's own (L = 256): where one tile per CU (B = 4096) stops being the limit.
    python tools/batch_sweep.py [--latent 64 256] [--batches 4096 8192 ...]  ->  one JSON line per point
(profiles/r06_batch_sweep.jsonl).  bench.py's method for `kernels` (hipGraph of steps minus the same graph without the group)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from rawaudiovae_kelsey_amd.engine import Graph, TrainEngine  # noqa: E402
from rawaudiovae_kelsey_amd.synth import flops_per_frame, make_frames, make_params  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--latent", type=int, nargs="+", default=[64, 256])
ap.add_argument("--batches", type=int, nargs="+", default=[4096, 8192, 16384, 32768, 65536, 131072])
a = ap.parse_args()
S, H = 1024, 2048
for L in a.latent:
    for B in a.batches:
        torch.cuda.empty_cache()
        eng = TrainEngine(S, H, L, B, kl_beta=1e-4, lr=1e-4, seed=0)
        eng.load_params(make_params(S, H, L, 0))
        x = torch.from_numpy(make_frames(B, S, 3)).cuda()
        st = torch.cuda.Stream()
        torch.cuda.synchronize()
        with torch.cuda.stream(st):
            eng.step(x, stream=st)
            g = Graph(st)
            with g:
                eng.step(x, stream=st)
            n = max(10, 200 * 4096 // B)
            for _ in range(max(3, n // 10)):
                g.launch()
            st.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                g.launch()
            st.synchronize()
            dt = (time.perf_counter() - t0) / n
            rows, noise = bench.time_launches_in_step(eng, x, steps=max(1, 10 * 4096 // B))
        F = flops_per_frame(S, H, L)
        print(json.dumps({"S": S, "H": H, "L": L, "B": B, "us_per_step": round(dt * 1e6, 1), "frames_per_s": round(B / dt),
                          "step_tflops": round(F * B / dt / 1e12, 1), "step_mfma_frac": round(F * B / dt / 2.5e15, 4),
                          "kernels": [{"launch": r["launch"], "us": round(r["us"], 1), "mfma_frac": round(r.get("mfma_frac", 0.0), 3),
                                       "hbm_frac": round(r["hbm_frac"], 3), "kernel": r["kernel"][:90]} for r in rows]}), flush=True)
        del eng, g, x
