#!/usr/bin/env python3
"""Probe how CU masks map onto the chip: time one big GEMM on streams with different masks."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ser_amd  # noqa: F401
from ser_amd import _lib as L

M = N = K = 4096
a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / 64
ah, _ = L.split_bf16(a, False); wh, _ = L.split_bf16(w, False)
c = torch.empty(M, N, device="cuda")
torch.cuda.synchronize()


def run(stream, n=10):
    with torch.cuda.stream(stream):
        for _ in range(3):
            L.check(L.lib.ser_gemm_bf16_nt(L.ptr(ah), None, K, L.ptr(wh), None, K, M, N, K, None, 0, None, 0, L.ptr(c), None, None, N, L.stream_ptr()))
        stream.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            L.check(L.lib.ser_gemm_bf16_nt(L.ptr(ah), None, K, L.ptr(wh), None, K, M, N, K, None, 0, None, 0, L.ptr(c), None, None, N, L.stream_ptr()))
        stream.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


print("default stream          %.1f us" % run(torch.cuda.current_stream()))
for name, bits in (("all 256", range(256)), ("first 128", range(128)), ("even bits", range(0, 256, 2)),
                   ("all but bits%8==7", [b for b in range(256) if b % 8 != 7]), ("first 224", range(224)),
                   ("first 32", range(32)), ("bits%8==0", range(0, 256, 8))):
    s = L.cu_masked_stream(list(bits))
    print("%-22s %.1f us" % (name, run(s)))
