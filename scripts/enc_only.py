#!/usr/bin/env python3
"""Replay only the encoder graph N times: for rocprofv3 kernel traces of the frozen encoders alone."""
import os, sys
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
st.step(*b)
torch.cuda.synchronize()
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    st.g_enc.replay()
torch.cuda.synchronize()
