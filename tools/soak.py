#!/usr/bin/env python3
"""Soak run of the real-data training path: `steps` optimizer steps of the C2 model (S=1024, H=2048, L=64, batch 4096)
through `TrainEngine.step_frames` on a synthetic 10-minute waveform resident in HBM (fresh shuffle every epoch, no
cast kernel), printing the loss every `--every` steps and checking at the end that every parameter and both Adam
moments are finite and that the loss went down.
    python tools/soak.py [--steps 200000] [--every 20000] [--slab-dtype fp16|fp32]
    python tools/soak.py --ab [--steps 300000]     three arms from the same weights on the same shuffles: fp16 slabs
        (the default), fp32 slabs, and fp32 slabs with another eps seed (the run-to-run spread two arms may differ by)
    python tools/soak.py --fp8-ab [--steps 200000]  the fp8 weight path (all four large GEMM launches on e4m3 operands,
        delayed scaling live) against the bf16 step and against the bf16 step with another eps seed"""
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
    ap.add_argument("--slab-dtype", default="fp16", choices=["fp16", "fp32"])
    ap.add_argument("--ab", action="store_true", help="fp16 slabs vs fp32 slabs vs fp32 slabs with another eps seed")
    ap.add_argument("--fp8-ab", action="store_true", help="fp8 weight path vs bf16 vs bf16 with another eps seed")
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
    p0 = make_params(S, H, L, 0)

    def run(slab_dtype, seed, tag, fp8=False):
        eng = TrainEngine(S, H, L, B, kl_beta=1e-4, lr=1e-4, seed=seed, slab_dtype=slab_dtype, fp8=fp8)
        eng.load_params(p0)
        start = eng.param.clone()
        gen = torch.Generator(device="cuda")
        gen.manual_seed(0)                       # the same shuffles in every arm
        done, curve, t0 = 0, [], time.perf_counter()
        while done < args.steps:
            for idx in ds.index_batches(B, shuffle=True, generator=gen):
                if idx.numel() != B:
                    continue
                eng.step_frames(ds, idx)
                done += 1
                if done == 1 or done % args.every == 0 or done == args.steps:
                    loss = float(np.mean(eng.losses(64))) if done >= 64 else eng.last_loss()[0]   # mean of the last 64 steps
                    curve.append((done, loss))
                    dt = time.perf_counter() - t0
                    print("%s step %7d  loss %.6f  (%.1f s, %.2f M frames/s so far)" % (tag, done, loss, dt, done * B / dt / 1e6), flush=True)
                    if not np.isfinite(loss):
                        raise SystemExit("loss is not finite")
                if done >= args.steps:
                    break
        torch.cuda.synchronize()
        ok = all(bool(torch.isfinite(a).all()) for a in (eng.param, eng.exp_avg, eng.exp_avg_sq))
        print("%s finite parameters and moments: %s; loss %.6f -> %.6f" % (tag, ok, curve[0][1], curve[-1][1]))
        if not ok or not curve[-1][1] < curve[0][1]:
            raise SystemExit("soak failed")
        return curve, eng.param.clone(), start

    if not args.ab and not args.fp8_ab:
        run(args.slab_dtype, 1, "[%s slabs]" % args.slab_dtype)
        return
    if args.fp8_ab:
        ca, pa, start = run("fp16", 1, "[A fp8 weight path, eps seed 1]", fp8=True)
        cb, pb, _ = run("fp16", 1, "[B bf16, eps seed 1]")
        cc, pc, _ = run("fp16", 2, "[C bf16, eps seed 2]")
        what, names = "the fp8 weight path", ("A fp8", "B bf16", "C bf16'")
    else:
        ca, pa, start = run("fp16", 1, "[A fp16 slabs, eps seed 1]")
        cb, pb, _ = run("fp32", 1, "[B fp32 slabs, eps seed 1]")
        cc, pc, _ = run("fp32", 2, "[C fp32 slabs, eps seed 2]")
        what, names = "the slab element type", ("A fp16", "B fp32", "C fp32'")
    print()
    print("loss (mean of the last 64 steps) at the same step counts; |A-B| is the effect of %s, |B-C| what" % what)
    print("two runs differ by when only the eps draws change:")
    print("%9s %10s %10s %10s %11s %11s" % (("step",) + names + ("|A-B|/B", "|B-C|/B")))
    worst_ab = worst_bc = 0.0
    for (n, a), (_, b), (_, c) in zip(ca, cb, cc):
        print("%9d %10.6f %10.6f %10.6f %11.2e %11.2e" % (n, a, b, c, abs(a - b) / b, abs(b - c) / b))
        if n > 1:
            worst_ab, worst_bc = max(worst_ab, abs(a - b) / b), max(worst_bc, abs(b - c) / b)
    trav = float((pb - start).norm())
    print("final parameters: |A - B| / |B - start| = %.4f   |B - C| / |B - start| = %.4f   (|B - start| = %.3f)"
          % (float((pa - pb).norm()) / trav, float((pb - pc).norm()) / trav, trav))
    print("worst relative loss difference after step 1: %s %.2e, eps seed %.2e -> %s"
          % (what, worst_ab, worst_bc, "within run-to-run noise" if worst_ab <= 1.5 * worst_bc else "LARGER than run-to-run noise"))


if __name__ == "__main__":
    main()
