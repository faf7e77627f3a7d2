#!/usr/bin/env python3
"""End-to-end frames/s of `train.py` on real (synthesised) audio: a 10 min, 44.1 kHz wav is written, default.ini's
model (S=1024, H=2048, L=256 -> here L=64 to match C2, hop 128) trains on it for a few epochs through the actual
entry point, and the wall time of the epoch loop is reported next to the device-side step rate.
    python tools/train_bench.py [--epochs 6] [--batch 4096]"""
import argparse
import configparser
import os
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--seconds", type=float, default=600.0)
    args = ap.parse_args()
    import torch
    from rawaudiovae_kelsey_amd import data as D
    import train as T
    work = tempfile.mkdtemp(prefix="rv_trainbench_")
    os.makedirs(os.path.join(work, "audio"))
    os.makedirs(os.path.join(work, "test_audio"))
    sr = 44100
    t = np.arange(int(args.seconds * sr)) / sr
    wave = (0.4 * np.sin(2 * np.pi * 220 * t) + 0.2 * np.sin(2 * np.pi * 3300 * t * (1 + 0.1 * np.sin(2 * np.pi * 0.5 * t)))
            + 0.05 * np.random.default_rng(0).normal(size=t.size)).astype(np.float32)
    D.write_wav(os.path.join(work, "audio", "synth.wav"), np.clip(wave, -1, 1), sr)
    D.write_wav(os.path.join(work, "test_audio", "t.wav"), np.clip(wave[: sr * 2], -1, 1), sr)
    cfg = configparser.ConfigParser(allow_no_value=True)
    cfg.read(os.path.join(REPO, "default.ini"))
    cfg["dataset"]["datapath"] = work
    cfg["dataset"]["test_dataset"] = work
    cfg["dataset"]["generate_test"] = "False"
    cfg["VAE"]["latent_dim"] = "64"
    cfg["training"]["epochs"] = str(args.epochs)
    cfg["training"]["batch_size"] = str(args.batch)
    cfg["training"]["checkpoint_interval"] = str(10 ** 6)
    ini = os.path.join(work, "bench.ini")
    with open(ini, "w") as f:
        cfg.write(f)
    n_frames = D.frame_count(len(wave), 1024, 128)[0]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    T.main(["--config", ini])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("train.py end to end: %d epochs x %d frames (batch %d) in %.2f s = %.2f M frames/s including start-up, wav "
          "decode, model/engine construction and the final checkpoint" % (args.epochs, n_frames, args.batch, dt,
                                                                          args.epochs * n_frames / dt / 1e6))
    # the epoch loop alone: same dataset object, same engine calls, timed around the loop
    from rawaudiovae_kelsey_amd.engine import TrainEngine
    ds = D.DeviceAudio(np.clip(wave, -1, 1), 1024, 128)
    full = args.batch
    eng = TrainEngine(1024, 2048, 64, full)
    tail = TrainEngine(1024, 2048, 64, len(ds) % full, share=eng) if len(ds) % full else None
    gen = torch.Generator(device="cuda").manual_seed(0)
    for warm in range(2):
        for idx in ds.index_batches(full, generator=gen):
            (eng if idx.numel() == full else tail).step_frames(ds, idx)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        for idx in ds.index_batches(full, generator=gen):
            (eng if idx.numel() == full else tail).step_frames(ds, idx)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("epoch loop (shuffle + step_frames on the resident waveform): %.2f M frames/s (%d epochs of %d frames in %.3f s)"
          % (reps * len(ds) / dt / 1e6, reps, len(ds), dt))


if __name__ == "__main__":
    main()
