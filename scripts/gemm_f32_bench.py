#!/usr/bin/env python3
"""Micro-benchmark of the fp32 head GEMM forms on the hot-path shapes (run on the GPU box)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import ser_amd  # noqa: F401
from ser_amd import _lib as L, _ops as O

SH = [(3184, 256, 768), (3184, 768, 256), (3184, 256, 256), (512, 256, 768), (3184, 128, 768), (16, 512, 512), (16, 512, 1536)]


def t(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for bk in (0, 16, 32, 64):
    L.lib.ser_debug_set_f32_bk(bk)
    print("BK", bk)
    for M, N, K in SH:
        x, W, dy = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda"), torch.randn(M, N, device="cuda")
        b = torch.randn(N, device="cuda")
        dW, db = torch.zeros(N, K, device="cuda"), torch.zeros(N, device="cuda")
        f = t(lambda: O.linear_fwd(x, W, b))
        d = t(lambda: O.linear_dgrad(dy, W))
        w = t(lambda: O.linear_wgrad(dy, x, dW, db))
        fl = 2.0 * M * N * K / 1e6
        print(f"  M={M:5d} N={N:4d} K={K:4d}  fwd {f:7.1f}us {fl / f:6.1f}TF   dgrad {d:7.1f}us {fl / d:6.1f}TF   wgrad {w:7.1f}us {fl / w:6.1f}TF")
