#!/usr/bin/env python3
"""HBM-side traffic per launch of the dominant kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE cannot
share a pass).  usage: pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>

Units and corrections per MI355X_MICROARCH.md (HBM / rocprofv3 section): both counters are in KiB; on gfx950
FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at 64 bytes, so it is doubled; WRITE_SIZE is exact.
Collect with, e.g.:
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out_f -o f -- python3 bench.py --steps 2 --warmup 1 --no-graph --no-pipeline --no-cpu-baseline
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d out_w -o w -- python3 bench.py --steps 2 --warmup 1 --no-graph --no-pipeline --no-cpu-baseline
"""
import csv, json, sys

GROUPS = {"gemm_bf16": ("gemm_bf16_nt_kernel", "gemm_bf16_pair_kernel"), "conv0_apply": ("conv0_apply_kernel", "conv0_kernel<true"),
          "adamw": ("adamw_kernel",), "stack_fwd": ("stack_fwd_kernel",), "stack_bwd": ("stack_bwd_kernel",)}


GEMMS_PER_PASS = 56        # encoder GEMM launches of one hot-path pass (7 conv/projection + 48 paired layer GEMMs + pos-conv)


def collect(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    passes = sum(1 for r in rows if "stack_fwd_kernel" in r["Kernel_Name"])
    acc = {k: [0, 0.0] for k in GROUPS}
    for k, pats in GROUPS.items():
        sel = [r for r in rows if any(p in r["Kernel_Name"] for p in pats)]
        if k == "gemm_bf16" and passes:      # drop the one-time tile-tuning launches (zero operands) at the start
            sel = sel[-GEMMS_PER_PASS * passes:]
        for r in sel:
            acc[k][0] += 1
            acc[k][1] += float(r["Counter_Value"]) * 1024.0
    return acc


f, w = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
out = {}
for k in GROUPS:
    if f[k][0] == 0 or w[k][0] == 0:
        continue
    fb, wb = 2.0 * f[k][1] / f[k][0], w[k][1] / w[k][0]
    out[k] = {"launches": f[k][0], "fetch_bytes_per_launch_corrected": fb, "write_bytes_per_launch": wb,
              "hbm_bytes_per_launch": fb + wb}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
