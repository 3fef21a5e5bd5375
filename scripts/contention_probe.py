#!/usr/bin/env python3
"""Head-alone vs encoder-alone vs overlapped, for a per-kernel comparison under rocprofv3 --kernel-trace:
three phases separated by 50 ms of idle time.   usage: contention_probe.py [group]
Analyse the trace with scripts/contention_report.py <kernel_trace.csv>."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ser_amd.system import PipelinedStepper

G = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
sysm, wc, xc = bench.build_system("bf16x3", dev)
sysm.train()
opt = sysm.make_optimizer(1e-4)
st = PipelinedStepper(sysm, opt, group=G)
b = [x.to(dev) for x in bench.synth_batch(16, 4.0, 32, xc.vocab_size, 4, 1)]
for _ in range(st.prime):
    st.feed(*b)
for _ in range(2 * G):
    st.step(*b)
torch.cuda.synchronize()


def gap():
    torch.cuda.synchronize()
    time.sleep(0.05)


def head_steps(n):
    for _ in range(n):
        st.g_head.replay()
        st.g_opt.replay()


gap()
head_steps(2 * G)                      # phase 1: head alone
gap()
for _ in range(2):                     # phase 2: encoder pass alone
    st.g_encs[0].replay()
gap()
cur = torch.cuda.current_stream()     # phase 3: overlapped, as in a timed step
for _ in range(2):
    st.enc_stream.wait_stream(cur)
    with torch.cuda.stream(st.enc_stream):
        st.g_encs[0].replay()
    head_steps(G)
    cur.wait_stream(st.enc_stream)
gap()
print("done")
