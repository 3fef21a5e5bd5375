#!/usr/bin/env python3
"""HBM-side traffic per launch of the dominant kernels from two rocprofv3 --pmc passes over the DRIVER's command
(FETCH_SIZE and WRITE_SIZE cannot share a pass):

  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out_f -o f -- python3 bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-inference
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d out_w -o w -- python3 bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-inference
  pmc_traffic.py out_f/f_counter_collection.csv out_w/w_counter_collection.csv out.json

Only launches of the timed schedule are counted: the encoder kernels of the LAST complete encoder passes (a pass starts at
its wave_stats_kernel dispatch; capture warm-ups and one-time set-up launches come earlier), the head kernels of the last
launches.  Units and corrections per MI355X_MICROARCH.md (HBM / rocprofv3 section): both counters are in KiB; on gfx950
FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at 64 bytes, so it is doubled; WRITE_SIZE is exact."""
import csv, json, sys

ENC = {"gemm_bf16": ("gemm_bf16_nt_kernel", "gemm_bf16_pair_kernel"), "conv0_apply": ("conv0_apply_kernel", "conv0_kernel<true"),
       "self_attention": ("self_attention",), "posconv": ("posconv_direct_kernel",), "layernorm": ("layernorm_pair_kernel", "layernorm_kernel")}
HEAD = {"adamw": ("adamw_multi_kernel", "adamw_kernel"), "stack_fwd": ("stack_fwd_kernel",), "stack_bwd": ("stack_bwd_kernel",),
        "gemm_x3": ("gemm_x3_kernel", "gemm_x3_group_kernel")}
PASSES = 3


def collect(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    starts = [i for i, r in enumerate(rows) if "wave_stats_kernel" in r["Kernel_Name"]]
    acc = {}
    if len(starts) >= PASSES + 1:
        seg = rows[starts[-PASSES - 1]:starts[-1]]           # the last PASSES complete passes
        for k, pats in ENC.items():
            sel = [r for r in seg if any(p in r["Kernel_Name"] for p in pats)]
            if sel:
                acc[k] = [len(sel), sum(float(r["Counter_Value"]) for r in sel) * 1024.0, len(sel) / PASSES]
    for k, pats in HEAD.items():
        sel = [r for r in rows if any(p in r["Kernel_Name"] for p in pats)][-8 * (21 if k == "gemm_x3" else 1):]
        if sel:
            acc[k] = [len(sel), sum(float(r["Counter_Value"]) for r in sel) * 1024.0, None]
    return acc


f, w = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
out = {}
for k in list(ENC) + list(HEAD):
    if k not in f or k not in w or f[k][0] == 0 or w[k][0] == 0:
        continue
    fb, wb = 2.0 * f[k][1] / f[k][0], w[k][1] / w[k][0]
    out[k] = {"launches_counted": f[k][0], "launches_per_encoder_pass": f[k][2], "fetch_bytes_per_launch_corrected": fb,
              "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
