#!/usr/bin/env python3
"""Yardstick only (never the product path): what the vendor GEMM (hipBLASLt via torch.matmul) does on
the step's GEMM shapes, bf16 in / bf16 out, same box -- to tell how much headroom the hand-written
kernels have.  Run on the GPU box:  python tools/blas_yardstick.py"""
import torch

B, S, H, L = 4096, 1024, 2048, 64
dev = "cuda"
r = lambda a, b: torch.randn(a, b, device=dev).to(torch.bfloat16)  # noqa: E731
x, h, dp4, W1, W4 = r(B, S), r(B, H), r(B, S), r(H, S), r(S, H)
cases = {
    "fc1 fwd   NT 4096x2048x1024": (2 * B * H * S, lambda: x @ W1.t()),
    "fc4 fwd   NT 4096x1024x2048": (2 * B * H * S, lambda: h @ W4.t()),
    "dgrad fc4 NN 4096x2048x1024": (2 * B * H * S, lambda: dp4 @ W4),
    "wgrad fc4 TN 1024x2048x4096": (2 * B * H * S, lambda: dp4.t() @ h),
    "wgrad fc1 TN 2048x1024x4096": (2 * B * H * S, lambda: h.t() @ x),
    "big       NT 8192x8192x8192": (2 * 8192 ** 3, None),
}
a8, b8 = r(8192, 8192), r(8192, 8192)
cases["big       NT 8192x8192x8192"] = (2 * 8192 ** 3, lambda: a8 @ b8.t())
for k, (fl, fn) in cases.items():
    for _ in range(5):
        fn()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
    print("%-32s %8.1f us  %7.1f TFLOP/s" % (k, best, fl / best / 1e6))
