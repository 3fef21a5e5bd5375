#!/usr/bin/env python3
"""Per-kernel summary of the TIMED schedule only, from a rocprofv3 --kernel-trace csv of `bench.py` (pipelined, graph-replayed):
windows between consecutive encoder-pass starts (= `group` steps) taken from the last third of the run, i.e. graph replays -
the capture warm-ups, the parity and the eager roofline legs are excluded (they come earlier / are cut off by the window
choice: the roofline leg's eager passes have no head-graph AdamW launches between them and are dropped).
usage: step_summary.py <kernel_trace.csv> <group> [windows]"""
import csv, re, sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
G = int(sys.argv[2])
NW = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0][:64]
ws = [i for i, r in enumerate(rows) if "wave_stats_kernel" in r["Kernel_Name"]]
# a replayed window holds exactly G AdamW launches; keep the last NW such windows
good = []
for a, b in zip(ws[:-1], ws[1:]):
    n_adam = sum(1 for r in rows[a:b] if "adamw_multi_kernel" in r["Kernel_Name"])
    if n_adam == G:
        good.append((a, b))
good = good[-NW - 1:-1] if len(good) > NW else good
assert good, "no replayed step window found"
agg = defaultdict(lambda: [0, 0.0])
span = 0.0
for a, b in good:
    span += (int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3
    for r in rows[a:b]:
        x = agg[name(r)]
        x[0] += 1
        x[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
steps = G * len(good)
tot = sum(v[1] for v in agg.values())
print(f"{len(good)} windows of {G} steps (graph replays only): {span / steps:.1f} us per step between encoder-pass starts (under the profiler), "
      f"kernel time {tot / steps:.1f} us per step over all queues\n")
print("| kernel | calls/step | avg us | ms/step | % of kernel time |")
print("|---|---|---|---|---|")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:36]:
    print(f"| {n} | {c / steps:.2f} | {t / c:.1f} | {t / steps / 1e3:.3f} | {100 * t / tot:.1f} |")
gemm = [(c, t) for n, (c, t) in agg.items() if n.startswith("gemm_bf16")]
gc, gt = sum(c for c, _ in gemm), sum(t for _, t in gemm)
print(f"\nencoder GEMM kernels (gemm_bf16_nt + gemm_bf16_pair): {gc / steps:.2f} launches / step, avg {gt / gc:.1f} us, {gt / steps / 1e3:.3f} ms / step")
