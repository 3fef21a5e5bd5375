#!/usr/bin/env python3
"""BASELINE config 3 (full fine-tune, batch 8) stepped through TrainStepper for profiling: N captured steps, nothing else.
usage: finetune_steps.py [precision] [steps] [graph|eager]   (run under rocprofv3 --kernel-trace; summarise with burst_summary.py)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ser_amd.system import TrainStepper

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
graph = (sys.argv[3] if len(sys.argv) > 3 else "graph") == "graph"
dev = torch.device("cuda:0")
sysm, wc, xc = bench.build_system(prec, dev, unfreeze=True)
sysm.train()
for m in (sysm.audio_encoder, sysm.text_encoder):
    m.encoder_train_noise, m.noise_seed = True, 0
opt = sysm.make_optimizer(1e-4)
st = TrainStepper(sysm, opt, use_graph=graph)
batches = [[x.to(dev) for x in bench.synth_batch(8, 4.0, 32, xc.vocab_size, 4, 1 + j)] for j in range(4)]
for i in range(3):
    st.step(*batches[i % 4])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    st.step(*batches[i % 4])
torch.cuda.synchronize()
print(f"{prec} {'graph' if graph else 'eager'}: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms / step over {steps} steps", flush=True)
