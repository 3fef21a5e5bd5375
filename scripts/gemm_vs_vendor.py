#!/usr/bin/env python3
"""Context for the roofline figure (MEASUREMENT ONLY - nothing here is on the product path): the encoder's GEMM shapes
through the vendor library (torch.matmul on bf16 = hipBLASLt / rocBLAS, one product per multiply) beside this repo's tile
kernel in its one-product and interleaved three-product modes, best tile configuration per shape.

    python scripts/gemm_vs_vendor.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import ser_amd  # noqa: F401
from ser_amd import _lib as L
from ser_amd._engines import TILE_CONFIGS_X3, TILE_HEIGHTS

SHAPES = [("qkv", 3696, 2304, 768), ("ffn1", 3696, 3072, 768), ("ffn2", 3696, 768, 3072), ("oproj", 3696, 768, 768),
          ("conv1", 102384, 512, 1536), ("conv2", 51184, 512, 1536), ("conv5", 6384, 512, 1024), ("sq4096", 4096, 4096, 4096)]
PEAK = 2500.0


def timed(fn, reps=20):
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


def ours(M, N, K, three):
    pm = 2 if three else 1
    g = torch.Generator().manual_seed(0)
    a = (torch.randn(M, K * pm, generator=g) * 0.5).to("cuda", torch.bfloat16)
    w = (torch.randn(N, K * pm, generator=g) * 0.05).to("cuda", torch.bfloat16)
    c = torch.empty(M, N * pm, dtype=torch.bfloat16, device="cuda")
    lo = (lambda t: t.data_ptr() + 2 * L.IL_GROUP) if three else (lambda t: None)
    best = (float("inf"), 0)
    for cfg in (TILE_CONFIGS_X3 if three else TILE_HEIGHTS):
        if cfg in (1192, 1256, 5128) and N < 256:
            continue
        L.lib.ser_debug_set_gemm_bm(cfg)
        try:
            us = timed(lambda: L.check(L.lib.ser_gemm_bf16_nt(a.data_ptr(), lo(a), K, w.data_ptr(), lo(w), K, M, N, K, None, L.ACT_NONE, None, 0, None,
                                                              c.data_ptr(), lo(c), N, L.stream_ptr()), "gemm"), reps=10)
        finally:
            L.lib.ser_debug_set_gemm_bm(0)
        best = min(best, (us, cfg))
    return best


def main():
    print(f"{'shape':8s} {'M':>7s} {'N':>5s} {'K':>5s} | vendor bf16 (1 product)   | this repo, 1 product          | this repo, 3 products (parity mode)")
    for name, M, N, K in SHAPES:
        g = torch.Generator().manual_seed(0)
        a = (torch.randn(M, K, generator=g) * 0.5).to("cuda", torch.bfloat16)
        w = (torch.randn(N, K, generator=g) * 0.05).to("cuda", torch.bfloat16)
        us_v = timed(lambda: torch.matmul(a, w.t()))
        fl = 2.0 * M * N * K
        u1, c1 = ours(M, N, K, False)
        u3, c3 = ours(M, N, K, True)
        print(f"{name:8s} {M:7d} {N:5d} {K:5d} | {us_v:7.1f} us {fl / us_v / 1e6:6.0f} TF {fl / us_v / 1e6 / PEAK:5.1%} | "
              f"{u1:7.1f} us {fl / u1 / 1e6:6.0f} TF {fl / u1 / 1e6 / PEAK:5.1%} cfg {c1:4d} | "
              f"{u3:7.1f} us {fl / u3 / 1e6:6.0f} TF alg, pipe {3 * fl / u3 / 1e6 / PEAK:5.1%} cfg {c3:4d}", flush=True)


if __name__ == "__main__":
    main()
