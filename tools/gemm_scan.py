#!/usr/bin/env python3
"""Fixed cost vs per-K-tile cost of a GEMM tile configuration: times M x N x K NT GEMMs (bf16 output, bias+ReLU
epilogue) over a K sweep with a forced tile and fits t = a + b * (K / 64).  python tools/gemm_scan.py [tile] [M] [N]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rawaudiovae_kelsey_amd._lib import lib  # noqa: E402

tile = int(sys.argv[1]) if len(sys.argv) > 1 else 7
M = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
N = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
L = lib()
L.rv_gemm_force_tile(tile)
st = torch.cuda.current_stream().cuda_stream or None
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)   # HIP events on the current stream
pts = []
for K in (256, 512, 1024, 2048, 4096, 8192):
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    fn = lambda: L.rv_linear_fwd(x.data_ptr(), K, w.data_ptr(), K, b.data_ptr(), M, N, K, 1, y.data_ptr(), N, st)  # noqa: E731
    for _ in range(5):
        fn()
    best = 1e9
    for _ in range(5):
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        e1.synchronize()
        ms = C.c_float(e0.elapsed_time(e1))
        best = min(best, ms.value / 10 * 1e3)
    pts.append((K // 64, best))
    print("tile %d  %dx%dx%-5d %8.1f us  %7.1f TFLOP/s" % (tile, M, N, K, best, 2.0 * M * N * K / best / 1e6))
(k1, t1), (k2, t2) = pts[2], pts[-1]
b = (t2 - t1) / (k2 - k1)
print("per K tile %.3f us, fixed %.1f us (from K=1024 and K=8192)" % (b, t1 - b * k1))
