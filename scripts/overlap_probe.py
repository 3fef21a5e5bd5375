#!/usr/bin/env python3
"""Why do the encoder graph (3.2 ms alone) and the head graph (3.1 ms alone) take 4.8 ms together?
Replays each beside synthetic partners: a chain of N tiny dependent kernels (kernel boundaries, no work) and a few
large library GEMMs (work, few boundaries), and reports the completion time of each side."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ser_amd.system import PipelinedStepper

dev = torch.device("cuda:0")
sysm, wc, xc = bench.build_system("bf16", dev)
sysm.train()
opt = sysm.make_optimizer(1e-4)
st = PipelinedStepper(sysm, opt)
b = [t.to(dev) for t in bench.synth_batch(16, 4.0, 32, xc.vocab_size, 4, 1)]
for _ in range(st.prime):
    st.feed(*b)
for _ in range(3):
    st.step(*b)
torch.cuda.synchronize()

s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def capture(fn, stream):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=stream):
            fn()
    torch.cuda.synchronize()
    return g


tiny = torch.zeros(64, device=dev)
A = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
Bm = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
C = torch.empty(8192, 8192, device=dev, dtype=torch.bfloat16)
small = torch.randn(1 << 20, device=dev)


def chain(n):
    def f():
        for _ in range(n):
            tiny.add_(1.0)
    return f


def gemms(n):
    def f():
        for _ in range(n):
            torch.matmul(A, Bm, out=C)
    return f


def stream_copy(n):
    def f():
        for _ in range(n):
            small.mul_(1.0001)
    return f


g_chain330 = capture(chain(330), s2)
g_chain100 = capture(chain(100), s2)
g_gemm = capture(gemms(4), s2)
g_bw = capture(stream_copy(330), s2)


def run_pair(ga, gb, n=20):
    """ga on s1, gb on s2, launched together; returns (ms a, ms b, ms both)"""
    ta = tb = tt = 0.0
    for _ in range(n):
        torch.cuda.synchronize()
        e0, ea, eb = torch.cuda.Event(True), torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record(torch.cuda.current_stream())
        s1.wait_event(e0); s2.wait_event(e0)
        with torch.cuda.stream(s1):
            if ga is not None:
                ga.replay()
            ea.record(s1)
        with torch.cuda.stream(s2):
            if gb is not None:
                gb.replay()
            eb.record(s2)
        torch.cuda.synchronize()
        a, bb = e0.elapsed_time(ea), e0.elapsed_time(eb)
        ta += a; tb += bb; tt += max(a, bb)
    return ta / n, tb / n, tt / n


names = {"enc": st.g_enc, "head": st.g_head, "chain330": g_chain330, "chain100": g_chain100, "gemm4": g_gemm,
         "bw330": g_bw}
for k, g in names.items():
    print("%-10s alone: %.3f ms" % (k, run_pair(g, None)[0]))
for a, bb in (("enc", "head"), ("enc", "chain330"), ("enc", "chain100"), ("enc", "bw330"), ("head", "gemm4"),
              ("head", "chain330"), ("chain330", "gemm4"), ("chain330", "chain330")):
    ga, gb = names[a], names[bb]
    if a == bb:
        gb = capture(chain(330), s2)
    r = run_pair(ga, gb)
    print("%-10s || %-10s : %.3f | %.3f   (both %.3f)" % (a, bb, *r))
