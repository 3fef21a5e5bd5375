#!/usr/bin/env python3
"""Tile configurations of the interleaved three-product encoder GEMM on the shapes of a grouped encoder pass
(`group` x 16 clips of 4 s + 32 tokens): time, algorithmic TFLOP/s, MFMA-pipe utilisation; bf16 planes out (as in the
encoders).  Correctness of every configuration against configuration 128 (bit-identical).
usage: gemm_cfg_probe.py [group] [cfg ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ser_amd  # noqa: F401
from ser_amd import _lib as L

G = int(sys.argv[1]) if len(sys.argv) > 1 else 4
if os.environ.get("SER_GEMM_GROUP_M"):
    L.lib.ser_debug_set_gemm_group_m(int(os.environ["SER_GEMM_GROUP_M"]))
CFGS = [int(v) for v in sys.argv[2:]] or [128, 192, 3128, 1256, 6256, 8256]
rows = G * (16 * 199 + 16 * 32)
SHAPES = [("qkv", rows, 2304, 768), ("oproj", rows, 768, 768), ("ffn1", rows, 3072, 768), ("ffn2", rows, 768, 3072),
          ("conv1", G * 16 * 6399, 512, 1536), ("conv3", G * 16 * 1599, 512, 1536), ("conv5", G * 16 * 399, 512, 1024), ("sq4096", 4096, 4096, 4096)]
PEAK = 2500.0


def timed(fn, reps=6):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for name, M, N, K in SHAPES:
    g = torch.Generator().manual_seed(0)
    a = (torch.randn(M, 2 * K, generator=g) * 0.5).to("cuda", torch.bfloat16)
    w = (torch.randn(N, 2 * K, generator=g) * 0.05).to("cuda", torch.bfloat16)
    c = torch.empty(M, 2 * N, dtype=torch.bfloat16, device="cuda")
    lo = lambda t: t.data_ptr() + 2 * L.IL_GROUP
    ref = None
    for cfg in CFGS:
        L.lib.ser_debug_set_gemm_bm(cfg)
        run = lambda: L.check(L.lib.ser_gemm_bf16_nt(a.data_ptr(), lo(a), K, w.data_ptr(), lo(w), K, M, N, K, None, L.ACT_NONE, None, 0, None,
                                                     c.data_ptr(), lo(c), N, L.stream_ptr()), "gemm")
        try:
            c.zero_()
            us = timed(run)
        except Exception as e:  # noqa: BLE001
            print(f"  {name:7s} cfg {cfg}: {e}")
            continue
        same = ""
        if ref is None:
            ref = c.clone()
        else:
            same = "identical" if torch.equal(ref, c) else f"DIFFERS ({(ref.float() - c.float()).abs().max().item():.3e})"
        tf = 2.0 * M * N * K / us / 1e6
        print(f"  {name:7s} M={M:7d} N={N:5d} K={K:5d} cfg {cfg:5d}: {us:8.1f} us  {tf:7.1f} TF  pipe {tf * 3 / PEAK:5.1%}  {same}", flush=True)
    L.lib.ser_debug_set_gemm_bm(0)
