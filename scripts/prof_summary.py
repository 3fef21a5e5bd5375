#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats run (csv) into a per-kernel table.
usage: prof_summary.py <dir-with-*_kernel_stats.csv> [steps]  -> markdown on stdout"""
import csv
import glob
import sys

d = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
f = glob.glob(d + "/*_kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total GPU kernel time {tot / 1e6:.3f} ms over {steps} steps = {tot / 1e6 / steps:.3f} ms/step\n")
print("| kernel | calls/step | avg us | ms/step | % |")
print("|---|---|---|---|---|")
for r in rows[:40]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    name = name.split("(")[0][:70]
    print(f"| {name} | {int(r['Calls']) / steps:.1f} | {float(r['AverageNs']) / 1e3:.2f} | "
          f"{float(r['TotalDurationNs']) / 1e6 / steps:.3f} | {float(r['Percentage']):.2f} |")
