#!/usr/bin/env python3
"""Stand-alone durations of the captured graphs of the pipelined step (encoders | head fwd+bwd | AdamW)."""
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
st.feed(*b)
for _ in range(3):
    st.step(*b)
torch.cuda.synchronize()


def t(fn, n=20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print("encoder graph alone  %.3f ms" % t(st.g_enc.replay))
print("head graph alone     %.3f ms" % t(st.g_head.replay))
print("adamw graph alone    %.3f ms" % t(st.g_opt.replay))
print("pipelined step       %.3f ms" % t(lambda: st.step(*b)))

# who finishes last inside an overlapped step?
cur = torch.cuda.current_stream()
acc_e = acc_h = 0.0
N = 20
for _ in range(N):
    torch.cuda.synchronize()
    e0, ee, eh = torch.cuda.Event(True), torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record(cur)
    st.enc_stream.wait_stream(cur)
    with torch.cuda.stream(st.enc_stream):
        st.g_enc.replay()
        ee.record(st.enc_stream)
    st.g_head.replay()
    st.g_opt.replay()
    eh.record(cur)
    torch.cuda.synchronize()
    acc_e += e0.elapsed_time(ee); acc_h += e0.elapsed_time(eh)
print("overlapped: encoder graph done at %.3f ms, head+adamw done at %.3f ms" % (acc_e / N, acc_h / N))

# host cost of issuing one pipelined step (three graph launches + small copies), GPU idle at call time
tt = 0.0
for _ in range(20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st.step(*b)
    tt += time.perf_counter() - t0
torch.cuda.synchronize()
print("host time to issue one step: %.3f ms" % (tt / 20 * 1e3))
