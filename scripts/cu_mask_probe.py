#!/usr/bin/env python3
"""Does reserving compute units help the head's latency chain?  The encoder pass is replayed on a stream created with
hipExtStreamCreateWithCUMask (first `n` of 256 mask bits set) while the head graphs run on an unmasked stream.
(1) eager GEMM on masked streams: does the mask apply and how does the rate scale;  (2) the captured encoder graph replayed
on a masked stream: do graph launches honour the launch stream's mask;  (3) the pipelined step with the encoder stream
masked.   One masked stream per process is the clean experiment (each takes a hardware queue of its own).
usage: cu_mask_probe.py [cus ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
import bench
from ser_amd.system import PipelinedStepper
from ser_amd import _lib as L

G = 4
BITS = [int(v) for v in sys.argv[1:]] or [224]
dev = torch.device("cuda:0")
torch.cuda.init()
torch.zeros(1, device=dev)
hip = C.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.restype = C.c_int
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]


def masked_stream(bits_on, total=256, from_top=False):
    words = (total + 31) // 32
    m = (C.c_uint32 * words)()
    for i in range(total):
        on = (i >= total - bits_on) if from_top else (i < bits_on)
        if on:
            m[i // 32] |= (1 << (i % 32))
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), words, m)
    assert rc == 0, f"hipExtStreamCreateWithCUMask rc={rc}"
    return torch.cuda.ExternalStream(s.value, device=dev)


def timed(fn, stream, n=6):
    with torch.cuda.stream(stream):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record(stream)
        for _ in range(n):
            fn()
        e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


M, N, K = 12736, 3072, 768
a = (torch.randn(M, 2 * K) * 0.5).to(dev, torch.bfloat16)
w = (torch.randn(N, 2 * K) * 0.05).to(dev, torch.bfloat16)
c = torch.empty(M, 2 * N, dtype=torch.bfloat16, device=dev)
lo = lambda t: t.data_ptr() + 2 * L.IL_GROUP


def gemm():
    L.check(L.lib.ser_gemm_bf16_nt(a.data_ptr(), lo(a), K, w.data_ptr(), lo(w), K, M, N, K, None, L.ACT_NONE, None, 0, None,
                                   c.data_ptr(), lo(c), N, L.stream_ptr()), "gemm")



sysm, wc, xc = bench.build_system("bf16x3", dev)
sysm.train()
opt = sysm.make_optimizer(1e-4)
st = PipelinedStepper(sysm, opt, group=G)
b = [x.to(dev) for x in bench.synth_batch(16, 4.0, 32, xc.vocab_size, 4, 1)]
for _ in range(st.prime):
    st.feed(*b)
for _ in range(4 * G):
    st.step(*b)
torch.cuda.synchronize()


def step_ms(n=12 * G):
    for _ in range(2 * G):
        st.step(*b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        st.step(*b)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


plain = st.enc_stream
print(f"before any masked stream exists: encoder graph {timed(st.g_encs[0].replay, plain, 4):.3f} ms, pipelined step {step_ms():.3f} ms", flush=True)
for n in BITS:
    s = masked_stream(n)
    g = timed(gemm, s)
    e = timed(st.g_encs[0].replay, s, 4)
    torch.cuda.synchronize()
    st.enc_stream = s
    ms = step_ms()
    st.enc_stream = plain
    torch.cuda.synchronize()
    print(f"encoder stream on {n:3d} CUs: eager GEMM 12736 x 3072 x 768 {g:.4f} ms | encoder graph alone {e:.3f} ms | pipelined step {ms:.3f} ms | "
          f"back on the torch stream {step_ms():.3f} ms", flush=True)
