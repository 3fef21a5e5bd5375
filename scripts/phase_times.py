#!/usr/bin/env python3
"""Stand-alone and overlapped durations of the captured graphs of the pipelined step, per group size:
encoder pass over `group` batches | head fwd+bwd | AdamW.   usage: phase_times.py [precision] [group ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ser_amd.system import PipelinedStepper

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
groups = [int(v) for v in sys.argv[2:]] or [1, 4]
dev = torch.device("cuda:0")


def t(fn, n=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for G in groups:
    sysm, wc, xc = bench.build_system(prec, dev)
    sysm.train()
    opt = sysm.make_optimizer(1e-4)
    st = PipelinedStepper(sysm, opt, group=G)
    b = [x.to(dev) for x in bench.synth_batch(16, 4.0, 32, xc.vocab_size, 4, 1)]
    for _ in range(st.prime):
        st.feed(*b)
    for _ in range(2 * G):
        st.step(*b)
    torch.cuda.synchronize()
    enc = t(st.g_encs[0].replay)
    head = t(st.g_head.replay)
    adam = t(st.g_opt.replay)

    def head_steps():
        for _ in range(G):
            st.g_head.replay()
            st.g_opt.replay()
    hs = t(head_steps)
    step = t(lambda: st.step(*b), n=4 * G)
    # who finishes last inside an overlapped group?
    cur = torch.cuda.current_stream()
    acc_e = acc_h = 0.0
    N = 8
    for _ in range(N):
        torch.cuda.synchronize()
        e0, ee, eh = torch.cuda.Event(True), torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record(cur)
        st.enc_stream.wait_stream(cur)
        with torch.cuda.stream(st.enc_stream):
            st.g_encs[0].replay()
            ee.record(st.enc_stream)
        head_steps()
        eh.record(cur)
        torch.cuda.synchronize()
        acc_e += e0.elapsed_time(ee); acc_h += e0.elapsed_time(eh)
    print(f"group {G}: encoder pass alone {enc:.3f} ms ({enc / G:.3f} / batch) | head graph alone {head:.3f} | adamw {adam:.3f} | "
          f"{G} head steps alone {hs:.3f} ({hs / G:.3f} / step) | pipelined step {step:.3f} ms | overlapped group: encoder done at "
          f"{acc_e / N:.3f} ms, {G} head steps done at {acc_h / N:.3f} ms ({acc_h / N / G:.3f} / step)", flush=True)
    del st, sysm, opt
    torch.cuda.empty_cache()
