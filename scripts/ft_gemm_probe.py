#!/usr/bin/env python3
"""Tile configurations of the encoder NT GEMM on the fine-tuning step's shapes (batch 8: 1 592 audio rows, 256 text rows; fp32 out),
three-product and one-product operands.  usage: ft_gemm_probe.py [cfg ...]   (0 = the launcher's own choice)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ser_amd  # noqa: F401
from ser_amd import _lib as L

CFGS = [int(v) for v in sys.argv[1:]] or [0, 64, 96, 128, 7064, 7096, 7128, 3128]
SHAPES = [("audio qkv fwd", 1592, 2304, 768), ("audio o / dgrad", 1592, 768, 768), ("audio ffn1 fwd", 1592, 3072, 768), ("audio ffn2 fwd", 1592, 768, 3072),
          ("audio wgrad 768x768", 768, 768, 1600), ("audio wgrad 3072x768", 3072, 768, 1600), ("text qkv fwd", 256, 2304, 768), ("text ffn2 fwd", 256, 768, 3072)]


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


for three in (True, False):
    print("three-product operands" if three else "one-product operands", flush=True)
    for name, M, N, K in SHAPES:
        g = torch.Generator().manual_seed(0)
        a = (torch.randn(M, (2 if three else 1) * K, generator=g) * 0.5).to("cuda", torch.bfloat16)
        w = (torch.randn(N, (2 if three else 1) * K, generator=g) * 0.05).to("cuda", torch.bfloat16)
        c = torch.empty(M, N, dtype=torch.float32, device="cuda")
        lo = (lambda t: t.data_ptr() + 2 * L.IL_GROUP) if three else (lambda t: None)
        row = []
        for cfg in CFGS:
            L.lib.ser_debug_set_gemm_bm(cfg)
            run = lambda: L.check(L.lib.ser_gemm_bf16_nt(a.data_ptr(), lo(a), K, w.data_ptr(), lo(w), K, M, N, K, None, L.ACT_NONE, None, 0,
                                                         c.data_ptr(), None, None, N, L.stream_ptr()), "gemm")
            try:
                row.append(f"{cfg}: {timed(run):6.1f}")
            except Exception as e:  # noqa: BLE001
                row.append(f"{cfg}: n/a")
        L.lib.ser_debug_set_gemm_bm(0)
        print(f"  {name:22s} M={M:5d} N={N:5d} K={K:5d} us by cfg | " + " | ".join(row), flush=True)
