#!/usr/bin/env python3
"""Time the encoder-layer GEMM shapes (Wav2Vec2 + XLM-R rows in one problem) for every tile height."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ser_amd  # noqa: F401
from ser_amd import _lib as L
shapes = [(3696, 2304, 768, "QKV"), (3696, 768, 768, "out"), (3696, 3072, 768, "FFN1"), (3696, 768, 3072, "FFN2"),
          (102384, 512, 1536, "conv1"), (51184, 512, 1536, "conv2")]
for M, N, K, name in shapes:
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5
    ah, _ = L.split_bf16(a, False); wh, _ = L.split_bf16(w, False)
    bias = torch.randn(N, device="cuda")
    ch = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    res = {}
    for bm in (0, 64, 96, 128, 160, 192):
        L.lib.ser_debug_set_gemm_bm(bm)
        def run():
            L.check(L.lib.ser_gemm_bf16_nt(L.ptr(ah), None, K, L.ptr(wh), None, K, M, N, K, L.ptr(bias), L.ACT_GELU, None, 0, None,
                                           L.ptr(ch), None, N, L.stream_ptr()))
        for _ in range(3): run()
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        res[bm] = e0.elapsed_time(e1) / 20 * 1e3
    L.lib.ser_debug_set_gemm_bm(0)
    print(f"{name:6s} M={M} N={N} K={K}: " + "  ".join(f"bm{k}={v:.1f}us" for k, v in res.items()),
          f" best {min((v, k) for k, v in res.items() if k)[1]}  TF(best) {2*M*N*K/min(v for k, v in res.items() if k)/1e6:.0f}")
