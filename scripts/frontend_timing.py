#!/usr/bin/env python3
"""Time the quality-gate + audio-conditioning front end: the device kernels on a batch (HIP events, median of several
runs) beside the numpy/scipy oracle per clip on the host (what the reference does inside AudioEncoder.forward).
Prints one JSON line; `python scripts/frontend_timing.py [--batch 16 --seconds 4]`."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ser_amd  # noqa: E402,F401
from ser_amd import _ops as O  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--batch", type=int, default=16)
    p.add_argument("--seconds", type=float, default=4.0)
    p.add_argument("--reps", type=int, default=20)
    p.add_argument("--cpu_clips", type=int, default=4)
    a = p.parse_args()
    T = int(16000 * a.seconds)
    rs = np.random.RandomState(0)
    t = np.arange(T) / 16000.0
    clips = []
    for b in range(a.batch):
        x = (1 + 0.8 * np.sin(2 * np.pi * 20 * t)) * sum(0.2 / (h + 1) * np.sin(2 * np.pi * 150 * (h + 1) * t + rs.uniform(0, 6)) for h in range(4))
        x[int(0.86 * T):] *= 0.05
        if b % 2:
            x = x + 0.2 * np.sin(2 * np.pi * 50 * t) * (t < 0.86 * a.seconds)
        clips.append((x + 1e-4 * rs.randn(T)).astype(np.float32))
    wave = torch.from_numpy(np.stack(clips)).cuda()
    lid = torch.tensor([[1.0, 0.0]] * a.batch).cuda()

    def run():
        raw, met, dec = O.quality_gates(wave, lid)
        return O.audio_conditioning(wave, dec), dec

    run()
    torch.cuda.synchronize()
    ms = []
    for _ in range(a.reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        (out, c_raw, meta), dec = run()
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    gpu_ms = float(np.median(ms))
    from oracle import dsp_oracle as D
    t0 = time.perf_counter()
    for c in clips[:a.cpu_clips]:
        D.front_end(c, None, None)
    cpu_ms = (time.perf_counter() - t0) / a.cpu_clips * 1e3
    # bytes each stage must move at least once (fp32 clip in, fp64 working copy written + read per filter / statistic pass)
    print(json.dumps({"what": "quality gates + audio conditioning front end", "batch": a.batch, "seconds": a.seconds,
                      "gpu_ms_per_batch": round(gpu_ms, 3), "gpu_clips_per_s": round(a.batch / gpu_ms * 1e3, 1),
                      "cpu_oracle_ms_per_clip": round(cpu_ms, 2), "cpu_clips_per_s": round(1e3 / cpu_ms, 2),
                      "accepted": int((dec == 2).sum()), "launches_per_batch": 20,
                      "note": "numpy/scipy oracle on one host core per clip; the reference runs the same per-clip CPU work inside forward"}))


if __name__ == "__main__":
    main()
