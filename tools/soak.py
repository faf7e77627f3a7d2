#!/usr/bin/env python3
"""Soak run of the real-data training path: `steps` optimizer steps of the C2 model (S=1024, H=2048, L=64, batch 4096)
through `TrainEngine.step_frames` on a synthetic 10-minute waveform resident in HBM (fresh shuffle every epoch, no
cast kernel), printing the loss every `--every` steps and checking at the end that every parameter and both Adam
moments are finite and that the loss went down.
    python tools/soak.py [--steps 200000] [--every 20000]"""
import argparse
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200000)
    ap.add_argument("--every", type=int, default=20000)
    args = ap.parse_args()
    import torch
    from rawaudiovae_kelsey_amd import data as D
    from rawaudiovae_kelsey_amd.engine import TrainEngine
    from rawaudiovae_kelsey_amd.synth import make_params
    S, H, L, B = 1024, 2048, 64, 4096
    sr = 44100
    t = np.arange(600 * sr) / sr
    wave = (0.4 * np.sin(2 * np.pi * 220 * t) + 0.2 * np.sin(2 * np.pi * 3300 * t * (1 + 0.1 * np.sin(2 * np.pi * 0.5 * t)))
            + 0.05 * np.random.default_rng(0).normal(size=t.size)).astype(np.float32)
    ds = D.DeviceAudio(np.clip(wave, -1, 1), S, 128)
    eng = TrainEngine(S, H, L, B, kl_beta=1e-4, lr=1e-4, seed=1)
    eng.load_params(make_params(S, H, L, 0))
    gen = torch.Generator(device="cuda")
    gen.manual_seed(0)
    done, first, t0 = 0, None, time.perf_counter()
    while done < args.steps:
        for idx in ds.index_batches(B, shuffle=True, generator=gen):
            if idx.numel() != B:
                continue
            eng.step_frames(ds, idx)
            done += 1
            if done == 1 or done % args.every == 0 or done == args.steps:
                loss = eng.last_loss()[0]
                first = loss if first is None else first
                dt = time.perf_counter() - t0
                print("step %7d  loss %.6f  (%.1f s, %.2f M frames/s so far)" % (done, loss, dt, done * B / dt / 1e6), flush=True)
                if not np.isfinite(loss):
                    raise SystemExit("loss is not finite")
            if done >= args.steps:
                break
    torch.cuda.synchronize()
    last = eng.last_loss()[0]
    ok = all(bool(torch.isfinite(a).all()) for a in (eng.param, eng.exp_avg, eng.exp_avg_sq))
    print("finite parameters and moments: %s; loss %.6f -> %.6f" % (ok, first, last))
    if not ok or not last < first:
        raise SystemExit("soak failed")


if __name__ == "__main__":
    main()
