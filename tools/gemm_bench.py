#!/usr/bin/env python3
"""Time every GEMM of the C2 training step on its own (HIP events, random bf16 operands,
interleaved rounds in one process).  Run on the GPU box:  python tools/gemm_bench.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rawaudiovae_kelsey_amd._lib import lib  # noqa: E402

B, S, H, L = 4096, 1024, 2048, 64
L2 = 2 * L
Lb = lib()
if os.environ.get('RV_TILE'):
    Lb.rv_gemm_force_tile(int(os.environ['RV_TILE']))
if os.environ.get('RV_PAIR_LOOP'):   # 2: two-slot ring, 8: ping-pong main loop of the paired 256x256 kernel
    Lb.rv_gemm_force_tile(100 + int(os.environ['RV_PAIR_LOOP']))
st = torch.cuda.current_stream().cuda_stream or None


PAD = int(os.environ.get("LDPAD", "0"))  # extra elements per row (breaks power-of-two strides)


def rnd(r, c):
    t = torch.randn(r, c + PAD, device="cuda").to(torch.bfloat16)
    return t


LD = lambda n: n + PAD


x, h, dp4, z, dmulv = rnd(B, S), rnd(B, H), rnd(B, S), rnd(B, L), rnd(B, L2)
W1, Wh, W3, W4 = rnd(H, S), rnd(L2, H), rnd(H, L), rnd(S, H)
bH, bS, bL2 = torch.randn(H, device="cuda"), torch.randn(S, device="cuda"), torch.randn(L2, device="cuda")
xf = torch.rand(B, S, device="cuda") * 2 - 1
outH, outS = torch.empty(B, H, dtype=torch.bfloat16, device="cuda"), torch.empty(B, S, dtype=torch.bfloat16, device="cuda")
f32buf = torch.empty(16 * 2048 * 2048, dtype=torch.float32, device="cuda")
cs = torch.empty(B // 128 * H, dtype=torch.float32, device="cuda")
msep = torch.empty(1024, dtype=torch.float32, device="cuda")
SPL = {k: int(os.environ.get("SPL_" + k, v)) for k, v in dict(heads=8, dz=4, w4=2, w3=8, wh=16, w1=2).items()}

P = lambda t: t.data_ptr()
from rawaudiovae_kelsey_amd._lib import dgrad_wgrad_pick  # noqa: E402
PAIR_S = dgrad_wgrad_pick(B, H, S)[2]
cases = {
    "fc1 fwd   4096x2048x1024 NT": (2 * B * H * S, lambda: Lb.rv_linear_fwd(P(x), LD(S), P(W1), LD(S), P(bH), B, H, S, 1, P(outH), H, st)),
    "heads fwd 4096x128x2048  NT": (2 * B * L2 * H, lambda: Lb.rv_linear_fwd_f32(P(h), H, P(Wh), H, P(bL2), B, L2, H, SPL["heads"], P(f32buf), L2, st)),
    "fc3 fwd   4096x2048x64   NT": (2 * B * H * L, lambda: Lb.rv_linear_fwd(P(z), L, P(W3), L, P(bH), B, H, L, 1, P(outH), H, st)),
    "fc4 fwd+loss 4096x1024x2048": (2 * B * S * H, lambda: Lb.rv_decode_out_loss_fwd(P(h), H, P(W4), H, P(bS), B, S, H, B, S, P(xf), S, None, S, P(outS), S, P(msep), P(cs), st)),
    "dgrad fc4 4096x2048x1024 NN": (2 * B * H * S, lambda: Lb.rv_linear_dgrad(P(dp4), S, P(W4), H, B, H, S, P(h), H, P(outH), H, P(cs), None, 0, 1, st)),
    "wgrad fc4 1024x2048x4096 TN": (2 * B * H * S, lambda: Lb.rv_linear_wgrad(P(dp4), LD(S), P(h), LD(H), S, H, B, SPL["w4"], -1, P(f32buf), H, 0, None, st)),
    "PAIR dgrad+wgrad fc4 (34.4GF)": (4 * B * H * S, lambda: Lb.rv_linear_dgrad_wgrad(P(dp4), LD(S), P(W4), LD(H), P(h), LD(H), B, H, S, P(outH), H, P(cs), P(f32buf), H, PAIR_S, 0, None, st)),
    "dz        4096x64x2048   NN": (2 * B * L * H, lambda: Lb.rv_linear_dgrad(P(h), H, P(W3), L, B, L, H, None, 0, None, 0, None, P(f32buf), L, SPL["dz"], st)),
    "wgrad fc3 2048x64x4096   TN": (2 * B * L * H, lambda: Lb.rv_linear_wgrad(P(h), H, P(z), L, H, L, B, SPL["w3"], -1, P(f32buf), L, 0, None, st)),
    "dgrad hd  4096x2048x128  NN": (2 * B * H * L2, lambda: Lb.rv_linear_dgrad(P(dmulv), L2, P(Wh), H, B, H, L2, P(h), H, P(outH), H, P(cs), None, 0, 1, st)),
    "wgrad hd  128x2048x4096  TN": (2 * B * H * L2, lambda: Lb.rv_linear_wgrad(P(dmulv), L2, P(h), H, L2, H, B, SPL["wh"], -1, P(f32buf), H, 0, None, st)),
    "pure NT f32 4096x2048x1024": (2 * B * H * S, lambda: Lb.rv_linear_fwd_f32(P(x), LD(S), P(W1), LD(S), None, B, H, S, 1, P(f32buf), H, st)),
    "pure NN f32 4096x2048x1024": (2 * B * H * S, lambda: Lb.rv_linear_dgrad(P(dp4), LD(S), P(W4), LD(H), B, H, S, None, 0, None, 0, None, P(f32buf), H, 1, st)),
    "pure NT f32 4096x1024x2048": (2 * B * H * S, lambda: Lb.rv_linear_fwd_f32(P(h), LD(H), P(W4), LD(H), None, B, S, H, 1, P(f32buf), S, st)),
    "wgrad fc1 2048x1024x4096 TN": (2 * B * H * S, lambda: Lb.rv_linear_wgrad(P(h), LD(H), P(x), LD(S), H, S, B, SPL["w1"], -1, P(f32buf), S, 0, None, st)),
}

e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)   # HIP events on the current stream
if PAD:
    cases = {k: v for k, v in cases.items() if k.startswith(('pure', 'fc1', 'wgrad fc4', 'wgrad fc1'))}
res = {k: [] for k in cases}
REPS, ROUNDS = 20, 5
for rnd_i in range(ROUNDS + 1):
    for k, (fl, fn) in cases.items():
        e0.record()
        for _ in range(REPS):
            fn()
        e1.record()
        e1.synchronize()
        ms = C.c_float(e0.elapsed_time(e1))
        if rnd_i:
            res[k].append(ms.value / REPS * 1e3)
tot = 0
for k, (fl, fn) in cases.items():
    v = sorted(res[k])
    med = v[len(v) // 2]
    tot += med
    print("%-30s median %7.1f us  min %7.1f us  %7.1f TFLOP/s" % (k, med, v[0], fl / med / 1e6))
print("sum of medians %.1f us" % tot)
