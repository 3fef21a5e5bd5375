#!/usr/bin/env python3
"""Probe of the encoder GEMM's staging throughput (MI355X).  For each shape and tile height: time, algorithmic TFLOP/s,
MFMA-pipe utilisation and the global->LDS bytes the launch stages per second and CU.  `--resident` repeats a small
problem over a batch dimension with stride 0, so that every tile reads the same (L2-resident) operands: if the staged
GB/s per CU rises there, the product shapes are bound by L2 misses; if not, by the L2->LDS path itself.

    python scripts/gemm_il_probe.py [--plain] [--resident] [--stages]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import ser_amd  # noqa: F401
from ser_amd import _lib as L

L.lib.ser_debug_gemm_batched.restype = L.i32
L.lib.ser_debug_gemm_batched.argtypes = [L.vp, L.vp, L.i32, L.i32, L.i32, L.i32, L.i64, L.i64, L.vp, L.i64, L.i32, L.vp]
SHAPES = [("qkv", 3696, 2304, 768), ("ffn1", 3696, 3072, 768), ("ffn2", 3696, 768, 3072), ("oproj", 3696, 768, 768),
          ("conv1", 102384, 512, 1536), ("conv3", 25584, 512, 1536), ("sq4096", 4096, 4096, 4096)]
HEIGHTS = (64, 96, 128, 160, 192)
PEAK = 2500.0


def timed(fn, reps=10):
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


def run_shape(name, M, N, K, il, nb=1, stride0=False, heights=HEIGHTS):
    pm = 2 if il else 1
    g = torch.Generator().manual_seed(0)
    rows_a = M if stride0 else M * nb
    a = (torch.randn(rows_a, K * pm, generator=g) * 0.5).to("cuda", torch.bfloat16)
    w = (torch.randn(N, K * pm, generator=g) * 0.05).to("cuda", torch.bfloat16)
    c = torch.empty(M * nb, N * pm, dtype=torch.bfloat16, device="cuda")
    sa = 0 if stride0 else M * K
    for bm in heights:
        L.lib.ser_debug_set_gemm_bm(bm)
        us = timed(lambda: L.check(L.lib.ser_debug_gemm_batched(a.data_ptr(), w.data_ptr(), M, N, K, nb, sa, 0, c.data_ptr(), M * N, int(il), L.stream_ptr())))
        tiles = -(-M // bm) * -(-N // 128) * nb
        ktiles = K // (32 if il else 64)
        staged = tiles * ktiles * (bm + 128) * 128
        tf = 2.0 * M * N * K * nb / us / 1e6
        print(f"  {name:8s} M={M:6d}x{nb:<3d} N={N:5d} K={K:5d} BM={bm:3d}: {us:8.1f} us  {tf:7.1f} TF  mfma-util {tf * (3 if il else 1) / PEAK:5.1%}  "
              f"tiles {tiles:5d}  staged {staged / 1e6:8.1f} MB = {staged / us / 1e3 / 256:6.1f} GB/s/CU", flush=True)
    L.lib.ser_debug_set_gemm_bm(0)


def run_cfgs(name, M, N, K):
    """Interleaved mode: every tile configuration x split-K factor on one shape."""
    import ctypes as C
    fn = L.lib.ser_debug_gemm_il_cfg
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    g = torch.Generator().manual_seed(0)
    a = (torch.randn(M, 2 * K, generator=g) * 0.5).to("cuda", torch.bfloat16)
    w = (torch.randn(N, 2 * K, generator=g) * 0.05).to("cuda", torch.bfloat16)
    out = torch.empty(4, M, N, dtype=torch.float32, device="cuda")
    cfgs = [(64, 64, 128), (96, 96, 128), (128, 128, 128), (160, 160, 128), (192, 192, 128), (3064, 64, 128), (3096, 96, 128), (3128, 128, 128)]
    if "--wide" in sys.argv:
        cfgs += [(1128, 128, 256), (1192, 192, 256), (1256, 256, 256), (2256, 256, 128), (5128, 128, 256), (6256, 256, 128)]
    for cfg, bm, bn in cfgs:
        for ks in ((1, 2, 3, 4) if N <= 1024 and "--splitk" in sys.argv else (1,)):
            us = timed(lambda: L.check(fn(a.data_ptr(), w.data_ptr(), M, N, K, cfg, ks, out.data_ptr(), L.stream_ptr())))
            tiles = -(-M // bm) * -(-N // bn) * ks
            tf = 2.0 * M * N * K / us / 1e6
            print(f"  {name:8s} M={M:6d} N={N:5d} K={K:5d} tile {bm:3d}x{bn:3d} ksplit {ks}: {us:8.1f} us  {tf:7.1f} TF  mfma-util {tf * 3 / PEAK:5.1%}  workgroups {tiles:5d}",
                  flush=True)


if __name__ == "__main__":
    il = "--plain" not in sys.argv
    print("mode", "interleaved three-product" if il else "one product")
    if "--cfgs" in sys.argv:
        for name, M, N, K in SHAPES[:6]:
            run_cfgs(name, M, N, K)
    elif "--resident" in sys.argv:
        for nb in (16, 64):
            run_shape("resident", 512, 512, 768, il, nb=nb, stride0=True)
            run_shape("distinct", 512, 512, 768, il, nb=nb, stride0=False)
    elif "--stages" in sys.argv:
        for st in ((2, 2, 2, 2, 2, 2, 2), (2, 3, 2, 2, 2, 2, 2), (2, 4, 2, 2, 2, 2, 2), (3, 3, 2, 2, 3, 3, 3)):
            L.lib.ser_debug_set_gemm_stages(*st[:4])
            L.lib.ser_debug_set_gemm_stages_tall(*st[4:])
            print("stages [128x128, 64x128, 64x64, 128x64, 96, 160, 192] =", st)
            for name, M, N, K in SHAPES[:4]:
                run_shape(name, M, N, K, il)
    else:
        for name, M, N, K in SHAPES:
            run_shape(name, M, N, K, il)
