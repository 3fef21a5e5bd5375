#!/usr/bin/env python3
"""Inference throughput of the same path (eval.py's per-batch work: both encoders, head, OpenMax logits, softmax / arg-max /
energy consumers) at the benchmark configuration, hipGraph-replayed with device-resident inputs.  One JSON line.

    python scripts/infer_bench.py [--batch 16 --seconds 4 --tokens 32 --precision bf16x3 --steps 50]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
import ser_amd  # noqa: F401
from ser_amd import _ops as O


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--tokens", type=int, default=32)
    ap.add_argument("--precision", choices=["bf16x3", "bf16"], default="bf16x3")
    ap.add_argument("--steps", type=int, default=50)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    if os.environ.get("SER_POSCONV_GEMM"):
        from ser_amd import _lib as L
        L.lib.ser_debug_set_posconv_gemm(int(os.environ["SER_POSCONV_GEMM"]))
    sysm, wc, xc = bench.build_system(a.precision, dev)
    sysm.eval()
    batches = [[t.to(dev) for t in bench.synth_batch(a.batch, a.seconds, a.tokens, xc.vocab_size, sysm.num_labels, 99 + j)] for j in range(4)]
    wave, ids, mask = [t.clone() for t in batches[0][:3]]

    def fwd():
        logits = sysm(wave, ids, mask, use_openmax=True)
        return O.eval_consumers(logits, 1.0)

    with torch.no_grad():
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                fwd()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            probs, pred, energy = fwd()
        for j in range(5):
            for dst, src in zip((wave, ids, mask), batches[j % 4][:3]):
                dst.copy_(src)
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for j in range(a.steps):
            for dst, src in zip((wave, ids, mask), batches[j % 4][:3]):
                dst.copy_(src, non_blocking=True)
            g.replay()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    ms = dt / a.steps * 1e3
    print(json.dumps({"what": "inference (encoders + head + OpenMax + softmax/arg-max/energy), hipGraph replay, inputs resident",
                      "batch": a.batch, "seconds": a.seconds, "tokens": a.tokens, "precision": a.precision,
                      "ms_per_batch": round(ms, 3), "utt_per_s": round(a.batch / ms * 1e3, 1), "steps": a.steps,
                      "finite": bool(torch.isfinite(probs).all()), "pred_sample": pred[:4].tolist()}))


if __name__ == "__main__":
    main()
