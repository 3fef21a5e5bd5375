#!/usr/bin/env python3
"""What slows the head's ~150 dependent launches down beside the encoders?  Head-graph steps timed alone and beside three
synthetic partners on a second stream: a read-only stream (sum), a write-heavy stream (copy), and MFMA GEMMs whose outputs
are (a) written, (b) 64 columns wide (little output).  If only the writers hurt, the cost sits in the kernel-boundary cache
write-back (every boundary of the head flushes the L2 lines the partner dirtied), not in workgroup slots or the clock."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ser_amd.system import PipelinedStepper
from ser_amd import _lib as L

dev = torch.device("cuda:0")
sysm, wc, xc = bench.build_system("bf16x3", dev)
sysm.train()
opt = sysm.make_optimizer(1e-4)
st = PipelinedStepper(sysm, opt, group=1)
b = [x.to(dev) for x in bench.synth_batch(16, 4.0, 32, xc.vocab_size, 4, 1)]
for _ in range(st.prime):
    st.feed(*b)
for _ in range(4):
    st.step(*b)
torch.cuda.synchronize()
side = torch.cuda.Stream()
big = torch.randn(64 * 1024 * 1024, device=dev)            # 256 MB
big2 = torch.empty_like(big)
M, N, K = 16384, 4096, 1024
a = (torch.randn(M, 2 * K) * 0.5).to(dev, torch.bfloat16)
w = (torch.randn(N, 2 * K) * 0.05).to(dev, torch.bfloat16)
c = torch.empty(M, 2 * N, dtype=torch.bfloat16, device=dev)
lo = lambda t: t.data_ptr() + 2 * L.IL_GROUP


def gemm(n_cols):
    L.check(L.lib.ser_gemm_bf16_nt(a.data_ptr(), lo(a), K, w.data_ptr(), lo(w), K, M, n_cols, K, None, L.ACT_NONE, None, 0, None,
                                   c.data_ptr(), lo(c), N, L.stream_ptr()), "gemm")


partners = {
    "none": None,
    "read-only stream (sum of 256 MB)": lambda: big.sum(),
    "write-heavy stream (copy 256 MB)": lambda: big2.copy_(big),
    "GEMM 16384 x 4096 x 1024, output written (134 MB)": lambda: gemm(N),
    "GEMM 16384 x 128 x 1024 x 8 (little output)": lambda: [gemm(128) for _ in range(8)],
}
import ctypes as C
L.lib.ser_debug_clock_probe.restype = C.c_int
L.lib.ser_debug_clock_probe.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
probe_out = torch.zeros(2 * 8, dtype=torch.int64, device=dev)
probe_stream = torch.cuda.Stream()


def clock_ghz():
    """shader clock seen by a latency-bound wave right now (8 blocks, ~200 us of dependent FMAs each)"""
    with torch.cuda.stream(probe_stream):
        L.check(L.lib.ser_debug_clock_probe(probe_out.data_ptr(), 8, 100000, L.stream_ptr()), "clock probe")
    probe_stream.synchronize()
    v = probe_out.cpu().view(8, 2).double()
    return float((v[:, 0] / v[:, 1]).median() * 0.1)


cur = torch.cuda.current_stream()
NH = 6
for name, fn in partners.items():
    ts = []
    for rep in range(3):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        if fn is not None:
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                for _ in range(40):
                    fn()
        e0.record(cur)
        for _ in range(NH):
            st.g_head.replay()
            st.g_opt.replay()
        e1.record(cur)
        ghz = clock_ghz() if rep == 2 else None          # beside the partner (and the head)
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / NH)
    # clock beside the partner alone
    if fn is not None:
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(20):
                fn()
    ghz_partner = clock_ghz()
    torch.cuda.synchronize()
    print(f"{name:56s}: head step {min(ts):.3f} ms (runs {', '.join('%.3f' % t for t in ts)}); shader clock beside partner + head "
          f"{ghz:.2f} GHz, beside the partner alone {ghz_partner:.2f} GHz", flush=True)
