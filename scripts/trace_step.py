#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace csv of `bench.py` (pipelined, graph-replayed): for the middle timed steps, the span and
busy time of each hardware queue, and the kernel-time totals per kernel family on each queue.
usage: trace_step.py <kernel_trace.csv>"""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0]
# encoder passes start at wave_stats_kernel
ws = [i for i, r in enumerate(rows) if "wave_stats_kernel" in r["Kernel_Name"]]
print("encoder passes seen:", len(ws))
# graph-replayed passes: pick passes 6..9 (after tuning / warm-up), assuming >= 12
pick = ws[len(ws) // 2: len(ws) // 2 + 3]
for k, i0 in enumerate(pick[:-1]):
    i1 = pick[k + 1]
    seg = rows[i0:i1]
    t0 = int(seg[0]["Start_Timestamp"])
    t1 = int(rows[i1]["Start_Timestamp"])
    print(f"\n== step window {k}: {(t1 - t0) / 1e3:.1f} us between consecutive encoder-pass starts, {len(seg)} launches")
    byq = defaultdict(list)
    for r in seg:
        byq[r["Queue_Id"]].append(r)
    for q, rs in byq.items():
        s = min(int(r["Start_Timestamp"]) for r in rs)
        e = max(int(r["End_Timestamp"]) for r in rs)
        busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
        print(f"  queue {q}: {len(rs)} launches, span {(e - s) / 1e3:.1f} us (starts at +{(s - t0) / 1e3:.1f}), kernel time {busy / 1e3:.1f} us")
        agg = defaultdict(lambda: [0, 0.0])
        for r in rs:
            a = agg[name(r)]
            a[0] += 1
            a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
            print(f"      {n[:64]:64s} x{c:3d} {d:8.1f} us  avg {d / c:7.1f}")
