#!/usr/bin/env python3
"""Where a GEMM launch's time goes: each large GEMM of the C2 step timed whole, without the global-memory
traffic of its epilogue (dbg 1), without its epilogue (dbg 2), without its main loop (dbg 4) and as an empty
launch (dbg 6); the paired fc4 backward also with only its dgrad / only its wgrad blocks.
Diagnostics only (results are wrong under dbg).  Run on the GPU box: python tools/gemm_decomp.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rawaudiovae_kelsey_amd._lib import dgrad_wgrad_pick, gemm_pick, lib  # noqa: E402

B, S, H, L = 4096, 1024, 2048, 64
Lb = lib()
st = torch.cuda.current_stream().cuda_stream or None
rnd = lambda r, c: torch.randn(r, c, device="cuda").to(torch.bfloat16)  # noqa: E731
x, h, dp4 = rnd(B, S), rnd(B, H), rnd(B, S)
W1, W4 = rnd(H, S), rnd(S, H)
bH, bS = torch.randn(H, device="cuda"), torch.randn(S, device="cuda")
xf = torch.rand(B, S, device="cuda") * 2 - 1
outH, outS = torch.empty(B, H, dtype=torch.bfloat16, device="cuda"), torch.empty(B, S, dtype=torch.bfloat16, device="cuda")
f32buf = torch.empty(16 * 2048 * 2048, dtype=torch.float32, device="cuda")
cs = torch.empty(B // 128 * H, dtype=torch.float32, device="cuda")
msep = torch.empty(1024, dtype=torch.float32, device="cuda")
P = lambda t: t.data_ptr()  # noqa: E731
PAIR_S = dgrad_wgrad_pick(B, H, S)[2]
W1_S = gemm_pick(H, S, B)[2]
cases = {
    "fc1 fwd 256x128 NT": lambda: Lb.rv_linear_fwd(P(x), S, P(W1), S, P(bH), B, H, S, 1, P(outH), H, st),
    "fc4 fwd+loss 128x128": lambda: Lb.rv_decode_out_loss_fwd(P(h), H, P(W4), H, P(bS), B, S, H, B, S, P(xf), S, None, S, P(outS), S, P(msep), P(cs), st),
    "PAIR fc4 bwd 256x256": lambda: Lb.rv_linear_dgrad_wgrad(P(dp4), S, P(W4), H, P(h), H, B, H, S, P(outH), H, P(cs), P(f32buf), H, PAIR_S, st),
    "wgrad fc1 TN split %d" % W1_S: lambda: Lb.rv_linear_wgrad(P(h), H, P(x), S, H, S, B, W1_S, P(f32buf), S, st),
}
e0, e1 = C.c_void_p(), C.c_void_p()
Lb.rv_event_create(C.byref(e0))
Lb.rv_event_create(C.byref(e1))


def timeit(fn, reps=20, rounds=5):
    for _ in range(3):
        fn()
    best = []
    for _ in range(rounds):
        Lb.rv_event_record(e0, st)
        for _ in range(reps):
            fn()
        Lb.rv_event_record(e1, st)
        ms = C.c_float()
        Lb.rv_event_elapsed_ms_sync(e0, e1, C.byref(ms))
        best.append(ms.value / reps * 1e3)
    best.sort()
    return best[len(best) // 2]


modes = [(0, "whole"), (8, "half slab bytes"), (1, "epi w/o global"), (2, "no epilogue"), (4, "no main loop")]
print("%-24s" % "us per launch" + "".join("%16s" % m[1] for m in modes))
for name, fn in cases.items():
    row = []
    for dbg, _ in modes:
        Lb.rv_gemm_force_tile(200 + dbg)
        row.append(timeit(fn))
    Lb.rv_gemm_force_tile(200)
    print("%-24s" % name + "".join("%16.1f" % v for v in row))
for only, nm in ((1, "PAIR dgrad blocks only"), (2, "PAIR wgrad blocks only")):
    Lb.rv_gemm_force_tile(300 + only)
    row = []
    for dbg, _ in modes:
        Lb.rv_gemm_force_tile(200 + dbg)
        row.append(timeit(cases["PAIR fc4 bwd 256x256"]))
    Lb.rv_gemm_force_tile(200)
    print("%-24s" % nm + "".join("%16.1f" % v for v in row))
Lb.rv_gemm_force_tile(300)
