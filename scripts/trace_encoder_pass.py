#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace csv: the launches of the LAST encoder pass (from its wave_stats_kernel to the final
LayerNorm) in issue order, with durations, grid sizes and gaps — where one pass of the frozen encoders spends its time.
usage: trace_encoder_pass.py <kernel_trace.csv> [which-pass-from-the-end=1]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "wave_stats_kernel" in r["Kernel_Name"]]
i0 = starts[-back]
stream = rows[i0]["Stream_Id"] if "Stream_Id" in rows[i0] else None
queue = rows[i0]["Queue_Id"]
sel = []
for r in rows[i0:]:
    if r["Queue_Id"] != queue:
        continue
    if sel and "wave_stats_kernel" in r["Kernel_Name"]:
        break
    sel.append(r)
    if len(sel) > 140:
        break
t_prev = None
tot = 0.0
agg = {}
print(f"{'kernel':60s} {'grid':>8s} {'us':>8s} {'gap':>6s}")
for r in sel:
    name = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = 0.0 if t_prev is None else (s - t_prev) / 1e3
    t_prev = e
    d = (e - s) / 1e3
    tot += d
    agg.setdefault(name, [0, 0.0])
    agg[name][0] += 1
    agg[name][1] += d
    print(f"{name[:60]:60s} {int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])):8d} {d:8.1f} {gap:6.1f}")
span = (int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])) / 1e3
print(f"\n{len(sel)} launches, kernel time {tot:.1f} us, span {span:.1f} us")
for k, (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k[:70]:70s} x{n:3d} {d:9.1f} us")
