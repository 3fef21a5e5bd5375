#!/usr/bin/env python3
"""Per-kernel table over the last N steps of a rocprofv3 kernel trace, a step being delimited by its burst of optimizer launches
(`marker` kernel; bursts are separated by > `gap_ms`).   usage: burst_summary.py <kernel_trace.csv> [n_steps] [marker] [gap_ms]"""
import csv, sys, collections
f = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
marker = sys.argv[3] if len(sys.argv) > 3 else "adamw_multi_kernel"
gap = float(sys.argv[4]) if len(sys.argv) > 4 else 3.0
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
ends, last = [], None
for s, e, k in rows:
    if marker in k:
        if last is not None and s - last > gap * 1e6:
            ends.append(prev_end)
        last, prev_end = s, e
ends.append(prev_end)
assert len(ends) > n, f"only {len(ends)} steps in the trace"
w0, w1 = ends[-n - 1], ends[-1]
acc, cnt = collections.defaultdict(float), collections.defaultdict(int)
busy = 0.0
for s, e, k in rows:
    if s >= w0 and e <= w1:
        name = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:90]
        acc[name] += (e - s) / 1e6
        cnt[name] += 1
tot = sum(acc.values())
print(f"last {n} steps: wall {(w1 - w0) / 1e6 / n:.3f} ms / step, sum of kernel durations {tot / n:.3f} ms / step, {sum(cnt.values()) / n:.0f} launches / step\n")
print("| kernel | calls/step | avg us | ms/step | % |")
print("|---|---|---|---|---|")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1])[:60]:
    print(f"| {k} | {cnt[k] / n:.1f} | {v / cnt[k] * 1e3:.2f} | {v / n:.3f} | {100 * v / tot:.2f} |")
