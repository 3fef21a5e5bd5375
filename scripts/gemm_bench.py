#!/usr/bin/env python3
"""Micro-benchmark of the encoder GEMM (ser_gemm_bf16_nt) on the shapes of the hot path.
Run on the GPU box:  python scripts/gemm_bench.py [--plain]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import ser_amd  # noqa: F401
from ser_amd import _lib as L

SHAPES = [("ffn1", 3184, 3072, 768), ("ffn2", 3184, 768, 3072), ("qkv", 3184, 2304, 768), ("oproj", 3184, 768, 768),
          ("x_qkv", 512, 2304, 768), ("x_ffn1", 512, 3072, 768), ("x_ffn2", 512, 768, 3072), ("featproj", 3184, 768, 512),
          ("conv1-ish", 102384, 512, 1536), ("conv4-ish", 12784, 512, 1536), ("sq4096", 4096, 4096, 4096)]


def bench(M, N, K, x3, reps=20, split_out=False):
    a = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") / K ** 0.5
    ah, al = L.split_bf16(a, x3)
    wh, wl = L.split_bf16(w, x3)
    bias = torch.randn(N, device="cuda")
    c = torch.empty(M, N, device="cuda") if not split_out else None
    ch = torch.empty(M, N, dtype=torch.bfloat16, device="cuda") if split_out else None
    cl = torch.empty(M, N, dtype=torch.bfloat16, device="cuda") if split_out and x3 else None

    def run():
        L.check(L.lib.ser_gemm_bf16_nt(L.ptr(ah), L.ptr(al), K, L.ptr(wh), L.ptr(wl), K, M, N, K, L.ptr(bias), L.ACT_NONE, None, 0,
                                       L.ptr(c), L.ptr(ch), L.ptr(cl), N, L.stream_ptr()))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    return us, 2.0 * M * N * K / us / 1e6


if __name__ == "__main__":
    modes = [False] if "--plain" in sys.argv else ([True] if "--x3" in sys.argv else [True, False])
    for x3 in modes:
        print("mode", "bf16x3" if x3 else "bf16")
        for name, M, N, K in SHAPES:
            us, tf = bench(M, N, K, x3)
            us2, tf2 = bench(M, N, K, x3, split_out=True)
            print(f"  {name:10s} M={M:6d} N={N:5d} K={K:5d}  f32-out {us:9.1f} us {tf:7.1f} TF   split-out {us2:9.1f} us {tf2:7.1f} TF (algorithmic)")
