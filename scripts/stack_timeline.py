#!/usr/bin/env python3
"""Phase timestamps of the persistent classifier-stack forward kernel (workgroup 0), averaged over blocks."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ser_amd  # noqa: F401
from ser_amd import _lib as L
from ser_amd.models.classifier import AdvancedOpenMaxClassifier

dev = torch.device("cuda:0")
Lb = 35
m = AdvancedOpenMaxClassifier(input_dim=512, num_labels=4, num_layers=Lb, base_dim=512).to(dev).train()
x = torch.randn(16, 512, device=dev, requires_grad=True)
buf = torch.zeros(2 * 8 * 64, dtype=torch.int64, device=dev)     # [fwd | bwd][block][8 marks], up to 64 blocks each
for it in range(3):
    if it == 2:
        L.lib.ser_debug_stack_timeline.argtypes = [C.c_void_p]
        assert L.lib.ser_debug_stack_timeline(buf.data_ptr()) == 0
    logits, unc, _ = m(x, use_openmax=False, return_uncertainty=True)
    (logits.sum() + unc.sum()).backward()
    torch.cuda.synchronize()
L.lib.ser_debug_stack_timeline(None)
t = buf.cpu()[:Lb * 8].view(Lb, 8).double() * 0.01          # us (100 MHz clock)
names = ["fetch h", "LN stats", "MFMA A + reduce", "finalize A (stores)", "fetch a", "MFMA B + reduce", "finalize B"]
d = t[1:, 1:] - t[1:, :-1]
for k, nme in enumerate(names):
    print(f"{nme:22s} {d[:, k].mean():6.2f} us  (min {d[:, k].min():.2f} max {d[:, k].max():.2f})")
nxt = t[2:, 0] - t[1:-1, 7]
print(f"{'to next block':22s} {nxt.mean():6.2f} us")
print(f"per block {(t[2:, 0] - t[1:-1, 0]).mean():.2f} us")
tb = buf.cpu()[8 * 64:8 * 64 + Lb * 8].view(Lb, 8).double() * 0.01      # backward: blocks run L-1 .. 0
names_b = ["MFMA da + reduce", "finalize da", "fetch da", "MFMA du + reduce", "finalize du", "fetch du", "LayerNorm backward"]
db = tb[1:-1, 1:] - tb[1:-1, :-1]
print("backward:")
for k, nme in enumerate(names_b):
    print(f"  {nme:20s} {db[:, k].mean():6.2f} us  (min {db[:, k].min():.2f} max {db[:, k].max():.2f})")
print(f"  to next block        {(tb[:-2, 0] - tb[1:-1, 7]).mean():6.2f} us")
print(f"  per block {(tb[:-2, 0] - tb[1:-1, 0]).mean():.2f} us")

# plain kernel durations without the timestamps (events around back-to-back launches)
from ser_amd import _ops as O
tab, gtab, scr = m._stack_tables()
h0 = torch.randn(16, 512, device=dev)
Hs, X1, U, A, ST = O.stack_fwd(h0, tab, Lb, scr[0])
DH = torch.randn(Lb + 1, 16, 512, device=dev)
torch.cuda.synchronize()
for name, fn in (("stack_fwd", lambda: O.stack_fwd(h0, tab, Lb, scr[0])),
                 ("stack_bwd", lambda: O.stack_bwd(tab, h0, Hs, X1, A, ST, DH, scr[1]))):
    fn()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch ({e0.elapsed_time(e1) / 20 * 1e3 / Lb:.2f} us per block)")
print("abort flags", int(scr[0][1]), int(scr[1][1]))
