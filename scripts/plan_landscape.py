#!/usr/bin/env python3
"""In-situ landscape of the GEMM tile plans (MI355X): for each of the four layer GEMM shapes of the benchmark configuration,
every interleaved three-product tile configuration is put into the plan table, the encoder graph is re-captured and the
overlapped replay (encoder graph beside the head graph) is timed; the other shapes keep the stepper's own choice.
Prints ms per configuration, so that stand-alone ties (scripts/gemm_vs_vendor.py) can be compared with what the step sees.

    python scripts/plan_landscape.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
import ser_amd  # noqa: F401
from ser_amd import _engines as E
from ser_amd.system import PipelinedStepper


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    sysm, wc, xc = bench.build_system("bf16x3", dev)
    sysm.train()
    opt = sysm.make_optimizer(lr=1e-4)
    st = PipelinedStepper(sysm, opt, None, None)
    st.refine_plans = False
    batch = [t.to(dev) for t in bench.synth_batch(16, 4.0, 32, xc.vocab_size, sysm.num_labels, 1234)]
    st.feed(*batch)
    st.step(*batch)
    torch.cuda.synchronize()
    measure = lambda: min(st._overlapped_ms(8) for _ in range(3))
    rows = 16 * 199 + 16 * 32
    H, F = 768, 3072
    print("baseline", round(measure(), 3), {k[:3]: v[0][1] for k, v in E._TUNE_RANKED.items() if k[0] == rows})
    for name, N, K in (("qkv", 3 * H, H), ("oproj", H, H), ("ffn1", F, H), ("ffn2", H, F)):
        key = (rows, N, K, True)
        ranked = E._TUNE_RANKED[key]
        best = ranked[0][1]
        alone = {cfg: ms for ms, cfg in ranked}
        res = []
        for cfg in sorted(alone, key=lambda c: alone[c]):
            E.set_plan(key, cfg)
            st._capture_encoders()
            st._overlapped_ms(2)
            res.append((measure(), cfg))
        E.set_plan(key, best)
        st._capture_encoders()
        print(name, "stand-alone best", best)
        for t, cfg in sorted(res):
            print(f"    cfg {cfg:5d}: overlapped {t:6.3f} ms   stand-alone {alone[cfg] / 8 * 1e3:7.1f} us")


if __name__ == "__main__":
    main()
