#!/usr/bin/env python3
"""Run a few launches of one GEMM shape (for PMC collection): gemm_one.py M N K [x3]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ser_amd  # noqa: F401
from ser_amd import _lib as L
M, N, K = (int(v) for v in sys.argv[1:4])
x3 = len(sys.argv) > 4
a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5
ah, al = L.split_bf16(a, x3); wh, wl = L.split_bf16(w, x3)
c = torch.empty(M, N, device="cuda")
for _ in range(5):
    L.check(L.lib.ser_gemm_bf16_nt(L.ptr(ah), L.ptr(al), K, L.ptr(wh), L.ptr(wl), K, M, N, K, None, 0, None, 0, L.ptr(c), None, None, N, L.stream_ptr()))
torch.cuda.synchronize()
