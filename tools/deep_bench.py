#!/usr/bin/env python3
"""Time the deep-variant training step (BASELINE configs[3]) with hipGraph replay.

  python tools/deep_bench.py [--S 2048 --H 2048 --L 256 --depth 3 --B 4096 --steps 200]
Prints one JSON line (frames/s, us/step, model TFLOP/s).  Not the headline bench (bench.py is C2).
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from rawaudiovae_kelsey_amd.deep import DeepTrainEngine, DeepVAE  # noqa: E402
from rawaudiovae_kelsey_amd.engine import Graph  # noqa: E402


def flops_per_frame(S, H, L, d):
    w = S * H + (d - 1) * H * H + 2 * H * L + L * H + (d - 1) * H * H + H * S
    return 6 * w - 2 * S * H          # fwd + dgrad + wgrad per weight; the first layer has no dgrad


def main():
    ap = argparse.ArgumentParser()
    for k, v in (("S", 2048), ("H", 2048), ("L", 256), ("depth", 3), ("B", 4096), ("steps", 200), ("warmup", 20)):
        ap.add_argument("--" + k, type=int, default=v)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--slab-dtype", default="fp16", choices=["fp16", "fp32"])
    a = ap.parse_args()
    torch.manual_seed(0)
    m = DeepVAE(a.S, a.H, a.L, a.depth).cuda()
    eng = m.engine(a.B, kl_beta=1e-4, lr=1e-4, seed=0, slab_dtype=a.slab_dtype)
    x = torch.rand(a.B, a.S, device="cuda") * 2 - 1
    st = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(st):
        eng.step(x, stream=st)
        if a.no_graph:
            run = lambda: eng.step(x, stream=st)  # noqa: E731
        else:
            g = Graph(st)
            with g:
                eng.step(x, stream=st)
            run = g.launch
        for _ in range(a.warmup):
            run()
        st.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            run()
        st.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
    F = flops_per_frame(a.S, a.H, a.L, a.depth)
    print(json.dumps({"workload": "deep S=%d H=%d L=%d depth=%d B=%d" % (a.S, a.H, a.L, a.depth, a.B),
                      "frames_per_s": a.B / dt, "us_per_step": dt * 1e6, "model_tflops": F * a.B / dt / 1e12,
                      "graph": not a.no_graph, "slab_dtype": a.slab_dtype, "loss_first": eng.losses(a.steps)[0], "loss_last": eng.last_loss()[0]}))


if __name__ == "__main__":
    main()
