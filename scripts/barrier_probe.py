#!/usr/bin/env python3
"""Latency of the in-kernel grid hand-off (persist.hip) alone and beside chip-filling GEMMs:
mode 0 = plain accesses + agent-scope fences, mode 1 = coherent (sc1) accesses without cache-wide fences."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ser_amd  # noqa: F401
from ser_amd import _lib as L

f = L.lib.ser_debug_barrier_probe
f.restype = C.c_int
f.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
dev = torch.device("cuda:0")
flags = torch.zeros(128, dtype=torch.int32, device=dev)
data = torch.zeros(2 * 16 * 16 * 64, dtype=torch.float32, device=dev)
err = torch.zeros(1, dtype=torch.int32, device=dev)
A = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
Bm = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
Cc = torch.empty(8192, 8192, device=dev, dtype=torch.bfloat16)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
ROUNDS = 200


def run(G, mode, beside):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    if beside:
        with torch.cuda.stream(s2):
            for _ in range(6):
                torch.matmul(A, Bm, out=Cc)
    with torch.cuda.stream(s1):
        if beside:
            torch.cuda._sleep(200000)
        e0.record(s1)
        rc = f(G, ROUNDS, mode, flags.data_ptr(), data.data_ptr(), err.data_ptr(), s1.cuda_stream)
        assert rc == 0
        e1.record(s1)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / ROUNDS


for G in (4, 32):
    for mode in (0, 1):
        for beside in (False, True):
            run(G, mode, beside)
            us = run(G, mode, beside)
            print(f"G={G:2d} mode={'coherent' if mode else 'fenced  '} beside_gemm={beside!s:5}  {us:6.2f} us/round   "
                  f"errors={int(err.item())} abort={int(flags[64].item())}")
