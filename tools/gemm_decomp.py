#!/usr/bin/env python3
"""Where the time of the step's big GEMM launches goes: fixed cost (launch, first tile, epilogue and its output stores)
against cost per 64-deep K tile, from a sweep of the contraction length through the PRODUCTION entry points and tiles
(fit t = a + b * K/64 over the three longest K), next to the per-CU LDS-fill ceiling of each tile.

The ceiling: a K tile of a BM x BN block tile stages (BM + BN) x 128 bytes through the CU's L2 -> LDS path, which
MI355X_MICROARCH.md ("Indexed rows: gather into LDS") measures at 66-73 GB/s per CU from the XCD's L2 and 33.5 GB/s
from beyond it; the MFMA floor of the same tile is BM x BN x 64 x 2 FLOP / (4 SIMDs x 1024 FLOP/clk x ~2.25 GHz).
    python tools/gemm_decomp.py        (GPU box; profiles/r04_gemm_decomp.txt)
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rawaudiovae_kelsey_amd._lib import dgrad_wgrad_pick, lib  # noqa: E402

Lb = lib()
st = torch.cuda.current_stream().cuda_stream or None
P = lambda t: t.data_ptr()  # noqa: E731
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)   # HIP events on the current stream


def rnd(r, c):
    return torch.randn(r, c, device="cuda").to(torch.bfloat16)


def best_us(fn, reps=10, rounds=5):
    for _ in range(3):
        fn()
    out = []
    for _ in range(rounds):
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        out.append(e0.elapsed_time(e1) / reps * 1e3)
    out.sort()
    return out[len(out) // 2]


def fit(pts):
    (k1, t1), (k2, t2) = pts[-3], pts[-1]
    b = (t2 - t1) / (k2 - k1)
    return t1 - b * k1, b


def ceilings(bm, bn):
    kb = (bm + bn) * 128 / 1024.0
    fill_lo, fill_hi = kb * 1024 / 73e3, kb * 1024 / 66e3             # us per K tile at 73 / 66 GB/s
    mfma = bm * bn * 64 * 2 / (4 * 1024 * 2.25e9) * 1e6                # us per K tile, one CU
    return kb, fill_lo, fill_hi, mfma


def report(name, bm, bn, n_blocks, pts, k_step, flops_step, in_step_us):
    a, b = fit(pts)
    kb, flo, fhi, mf = ceilings(bm, bn)
    print("%s" % name)
    print("   K sweep (K tiles: us): " + "  ".join("%d: %.1f" % p for p in pts))
    print("   fit: fixed %.1f us + %.3f us per K tile   (%d blocks of %d x %d = %.1f per CU)" % (a, b, n_blocks, bm, bn, n_blocks / 256.0))
    print("   one K tile stages %.0f KB per CU: fill ceiling %.2f-%.2f us (66-73 GB/s from L2), MFMA floor %.2f us; measured %.3f "
          "= %.0f GB/s per CU" % (kb, flo, fhi, mf, b, kb * 1024 / b / 1e3))
    t_loop = b * k_step
    print("   at the step's K (%d tiles): main loop %.1f us + fixed %.1f us = %.1f us alone (plain stores); in the step %s us" %
          (k_step, t_loop, a, t_loop + a, in_step_us))
    best = max(flo, mf) * k_step + a
    print("   with the main loop AT its ceiling and the same fixed cost: %.1f us = %.3f of the 2.5 PF peak (now %.3f alone)" %
          (best, flops_step / best / 1e6 / 2500.0, flops_step / (t_loop + a) / 1e6 / 2500.0))
    print()


def main():
    B, S, H = 4096, 1024, 2048
    in_step = dict(a.split("=") for a in sys.argv[1:])   # e.g. fc1=18.8 fc4=24.3 pair=31.0 (bench.py `kernels`)
    print(__doc__.split("\n    python")[0])
    print()
    # fc1 forward: 4096 x 2048 x K, 256 x 128 tiles (the picker's choice), bias + ReLU -> bf16
    pts = []
    for K in (256, 512, 1024, 2048, 4096):
        x, w, b, y = rnd(B, K), rnd(H, K), torch.randn(H, device="cuda"), torch.empty(B, H, dtype=torch.bfloat16, device="cuda")
        pts.append((K // 64, best_us(lambda: Lb.rv_linear_fwd(P(x), K, P(w), K, P(b), B, H, K, 1, P(y), H, st))))
    report("fc1 forward  relu(x W1^T + b1)  4096 x 2048 x K, tile 256 x 128 (8 waves 4 x 2), ring loop", 256, 128, 16 * 16, pts,
           S // 64, 2.0 * B * S * H, in_step.get("fc1", "?"))
    # fc4 forward + loss: 4096 x 1024 x K, 128 x 128 tiles
    pts = []
    xf = torch.rand(B, S, device="cuda") * 2 - 1
    for K in (512, 1024, 2048, 4096, 8192):
        h, w, b = rnd(B, K), rnd(S, K), torch.randn(S, device="cuda")
        dp4 = torch.empty(B, S, dtype=torch.bfloat16, device="cuda")
        mse, cs = torch.empty(1024, device="cuda"), torch.empty(B // 128 * S, device="cuda")
        pts.append((K // 64, best_us(lambda: Lb.rv_decode_out_loss_fwd(P(h), K, P(w), K, P(b), B, S, K, B, S, P(xf), S, None, S,
                                                                         P(dp4), S, P(mse), P(cs), st))))
    report("fc4 forward + tanh + MSE partials + dP4  4096 x 1024 x K, tile 128 x 128 (8 waves 2 x 4), ring loop", 128, 128, 32 * 8, pts,
           H // 64, 2.0 * B * S * H, in_step.get("fc4", "?"))
    # the paired fc4 backward's two halves on their own: 256 x 256 ping-pong tiles
    Lb.rv_gemm_force_tile(7)
    pts = []
    for K in (256, 512, 1024, 2048, 4096):   # dgrad dX = relu'(dY W): 4096 x 2048 x K  (NN), 128 blocks
        dy, w, m = rnd(B, K), rnd(K, H), rnd(B, H)
        dx, cs = torch.empty(B, H, dtype=torch.bfloat16, device="cuda"), torch.empty(B // 128 * H, device="cuda")
        pts.append((K // 64, best_us(lambda: Lb.rv_linear_dgrad(P(dy), K, P(w), H, B, H, K, P(m), H, P(dx), H, P(cs), None, 0, 1, st))))
    report("fc4 dgrad half of the pair  dX = relu'(dY W)  4096 x 2048 x K (NN), tile 256 x 256 ping-pong, 128 blocks", 256, 256, 128, pts,
           S // 64, 2.0 * B * S * H, "(pair) " + in_step.get("pair", "?"))
    pts = []
    slabs = torch.empty(4 * S * H, dtype=torch.float32, device="cuda")
    us = torch.empty(4 * (S // 32) * (H // 32), device="cuda")
    for Kb in (1024, 2048, 4096, 8192, 16384):   # wgrad dW = dY^T X: 1024 x 2048 x (Kb / 4 per split) (TN), 4 splits, 128 blocks
        dy, xx = rnd(Kb, S), rnd(Kb, H)
        pts.append((Kb // 4 // 64, best_us(lambda: Lb.rv_linear_wgrad(P(dy), S, P(xx), H, S, H, Kb, 4, 7, P(slabs), H, 1, P(us), st))))
    report("fc4 wgrad half of the pair  dW = dY^T X  1024 x 2048 x K per split (TN), 4 splits, fp16 slabs, tile 256 x 256 ping-pong, 128 blocks",
           256, 256, 128, pts, B // 4 // 64, 2.0 * B * S * H, "(pair) " + in_step.get("pair", "?"))
    Lb.rv_gemm_force_tile(-1)
    paired, bm, splits = dgrad_wgrad_pick(B, H, S)
    dp4, w4, h3 = rnd(B, S), rnd(S, H), rnd(B, H)
    dx, cs = torch.empty(B, H, dtype=torch.bfloat16, device="cuda"), torch.empty(B // 128 * H, device="cuda")
    t = best_us(lambda: Lb.rv_linear_dgrad_wgrad(P(dp4), S, P(w4), H, P(h3), H, B, H, S, P(dx), H, P(cs), P(slabs), H, splits, 1, P(us), st))
    print("the pair as one launch (128 + 128 blocks, every CU one block): %.1f us alone = %.3f of peak; in the step %s us"
          % (t, 4.0 * B * S * H / t / 1e6 / 2500.0, in_step.get("pair", "?")))


if __name__ == "__main__":
    main()
