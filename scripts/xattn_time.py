#!/usr/bin/env python3
"""Attention core timings (forward, backward) at the shapes of the path: encoder self-attention (B 8 / 16, 12 heads of 64, S 199 and 32)
and the head's cross-attention (8 heads of 32, 199 x 32 and 32 x 199).  SER_XATTN_MFMA=0 times the scalar kernels.  usage: xattn_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ser_amd
from ser_amd import _ops as O

dev = torch.device("cuda:0")


def t(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, B, Sq, Sk, heads, hd in (("audio self-attention, batch 8", 8, 199, 199, 12, 64), ("audio self-attention, batch 16", 16, 199, 199, 12, 64),
                                   ("text self-attention, batch 8", 8, 32, 32, 12, 64), ("cross-attention A<-T, batch 16", 16, 199, 32, 8, 32),
                                   ("cross-attention T<-A, batch 16", 16, 32, 199, 8, 32)):
    E = heads * hd
    q, k, v = (torch.randn(B * n_, E, device=dev) for n_ in (Sq, Sk, Sk))
    d = torch.randn(B * Sq, E, device=dev)
    ctx, P = O.xattn_fwd(q, k, v, None, B, Sq, Sk, heads)
    f = t(lambda: O.xattn_fwd(q, k, v, None, B, Sq, Sk, heads))
    bw = t(lambda: O.xattn_bwd(d, q, k, v, P, B, Sq, Sk, heads))
    fl = 4.0 * B * heads * Sq * Sk * hd
    print(f"{name:34s}: forward {f:7.1f} us ({fl / f / 1e6:6.1f} TFLOP/s)   backward {bw:7.1f} us ({2.5 * fl / bw / 1e6:6.1f} TFLOP/s)", flush=True)
