#!/usr/bin/env python3
"""Shader clock seen by a latency-bound probe wave while the REAL schedule runs (not a synthetic partner): short probes
(~1.5 ms of dependent FMAs on 8 waves, one per XCD by round-robin placement) run on their own stream, each started by an event
behind every `every`-th pipelined step and never waited for, each into its own slot; read at the end.  Runs are long enough
(seconds) to show the clock settling: the chip holds its top clock for a while after an idle or light phase before the power
management pulls it down.  Three loads: the pipelined training
step, the encoder graph alone back to back, the head graph alone back to back.
usage: clock_under_load.py [group] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
import bench
from ser_amd.system import PipelinedStepper
from ser_amd import _lib as L

G = int(sys.argv[1]) if len(sys.argv) > 1 else 4
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 160
dev = torch.device("cuda:0")
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
fn = L.lib.ser_debug_clock_probe
fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
probe_stream = torch.cuda.Stream()


def run(name, body, n, every):
    """`n` calls of `body`; after every `every`-th call a probe is queued that starts when THAT call has finished on the device (an
    event on the main stream), i.e. it runs beside the calls that follow - probes are spread over the whole run in device time, not
    bunched at its start (the host runs far ahead of the device)."""
    slots = n // every + 1
    out = torch.zeros(slots, 8, 2, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    cur = torch.cuda.current_stream()
    t0 = time.perf_counter()
    k = 0
    for i in range(n):
        body()
        if i % every == every - 1:
            ev = torch.cuda.Event()
            ev.record(cur)
            probe_stream.wait_event(ev)
            with torch.cuda.stream(probe_stream):
                L.check(fn(out[k].data_ptr(), 8, 100000, probe_stream.cuda_stream), "probe")
            k += 1
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    v = out[:k].cpu().double()
    ghz = (v[..., 0] / v[..., 1].clamp(min=1) * 0.1)
    per_probe = ghz.median(dim=1).values
    q = max(1, k // 4)
    series = ", ".join(f"{float(per_probe[j * k // 8:(j + 1) * k // 8 or 1].median()):.2f}" for j in range(8)) if k >= 8 else ""
    print(f"{name:42s}: {ms:7.3f} ms per call | {k} probes: clock median {float(per_probe.median()):.3f} GHz, min {float(per_probe.min()):.3f}, "
          f"max {float(per_probe.max()):.3f} | first quarter {float(per_probe[:q].median()):.3f}, last quarter {float(per_probe[-q:].median()):.3f} | "
          f"medians of the eight eighths of the run: {series}", flush=True)


run("idle (probe only)", lambda: None, 40, 1)
run(f"pipelined training step, group {G}", lambda: st.step(*b), STEPS, 2)
run("encoder graph alone, back to back", lambda: st.g_encs[0].replay(), max(24, STEPS // 8), 1)


def head():
    st.g_head.replay()
    st.g_opt.replay()


run("head + AdamW graphs alone, back to back", head, STEPS, 2)
run(f"pipelined training step, group {G} (again)", lambda: st.step(*b), 2 * STEPS, 2)
