#!/usr/bin/env python3
"""List the torch-side helper kernels (copies, fills, adds) an eager training step still launches,
grouped by python call site: candidates for removal from the captured graphs."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
dev = torch.device("cuda:0")
sysm, wc, xc = bench.build_system("bf16", dev)
sysm.train()
opt = sysm.make_optimizer(1e-4)
b = [t.to(dev) for t in bench.synth_batch(16, 4.0, 32, xc.vocab_size, 4, 1)]


def step():
    loss, _ = sysm.loss(*b)
    loss.backward()
    opt.step()
    opt.zero_grad()


step(); step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::add_", "aten::add", "aten::mul", "aten::cat",
                   "aten::clone", "aten::contiguous", "aten::sum", "aten::zeros"):
        st = [s for s in ev.stack if "ser" in s or "multilingual" in s or "bench" in s][:2]
        cnt[(ev.name, " <- ".join(st) if st else "(autograd engine)")] += 1
for (k, v) in sorted(cnt.items(), key=lambda kv: -kv[1]):
    print(v, k[0], k[1])
