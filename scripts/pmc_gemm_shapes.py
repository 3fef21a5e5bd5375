#!/usr/bin/env python3
"""FETCH_SIZE per launch of the encoder GEMM by shape and tile configuration, from a rocprofv3 --pmc FETCH_SIZE counter csv of
scripts/gemm_cfg_probe.py (which runs each (shape, cfg) 2 + 6 times in a fixed order).
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d D -o f -- python3 scripts/gemm_cfg_probe.py 4 3128 1256 6256
    python3 scripts/pmc_gemm_shapes.py D/f_counter_collection.csv 4 3128 1256 6256"""
import csv, sys

G = int(sys.argv[2])
CFGS = [int(v) for v in sys.argv[3:]]
rows_ = G * (16 * 199 + 16 * 32)
SHAPES = [("qkv", rows_, 2304, 768), ("oproj", rows_, 768, 768), ("ffn1", rows_, 3072, 768), ("ffn2", rows_, 768, 3072),
          ("conv1", G * 16 * 6399, 512, 1536), ("conv3", G * 16 * 1599, 512, 1536), ("conv5", G * 16 * 399, 512, 1024), ("sq4096", 4096, 4096, 4096)]
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Counter_Name"] == "FETCH_SIZE" and "gemm_bf16" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
per = 8                                    # launches per (shape, cfg): 2 warm-up + 6 timed
i = 0
for name, M, N, K in SHAPES:
    alg = 4.0 * (M * K + N * K)
    for cfg in CFGS:
        seg = rows[i:i + per]
        i += per
        if len(seg) < per:
            break
        fetch = 2.0 * 1024.0 * sum(float(r["Counter_Value"]) for r in seg[2:]) / (per - 2)
        kn = seg[0]["Kernel_Name"].split("<")[1].split(">")[0] if "<" in seg[0]["Kernel_Name"] else "?"
        print(f"{name:7s} M={M:7d} N={N:5d} K={K:5d} cfg {cfg:5d} <{kn}>: fetched {fetch / 1e6:9.1f} MB per launch (corrected), operands {alg / 1e6:8.1f} MB -> {fetch / alg:5.2f}x")
