#!/usr/bin/env python3
"""Per-kernel average duration in the three phases of scripts/contention_probe.py (head alone | encoder alone | overlapped),
and per-queue span / busy / gap totals.   usage: contention_report.py <kernel_trace.csv>"""
import csv, re, sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0][:56]
# phases = the last three bursts separated by >= 30 ms of idle time
bursts, cur, last_end = [], [], None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if last_end is not None and s - last_end > 30e6:
        bursts.append(cur); cur = []
    cur.append(r)
    last_end = e if last_end is None else max(last_end, e)
bursts.append(cur)
ph = bursts[-3:]
labels = ["head alone", "encoder alone", "overlapped"]
stats = [defaultdict(lambda: [0, 0.0]) for _ in ph]
for k, seg in enumerate(ph):
    t0 = min(int(r["Start_Timestamp"]) for r in seg); t1 = max(int(r["End_Timestamp"]) for r in seg)
    print(f"== {labels[k]}: {len(seg)} launches, span {(t1 - t0) / 1e3:.1f} us")
    byq = defaultdict(list)
    for r in seg:
        byq[r["Queue_Id"]].append(r)
        a = stats[k][name(r)]; a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for q, rs in byq.items():
        s = min(int(r["Start_Timestamp"]) for r in rs); e = max(int(r["End_Timestamp"]) for r in rs)
        busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
        print(f"   queue {q}: {len(rs)} launches, span {(e - s) / 1e3:.1f} us, kernel time {busy / 1e3:.1f} us, gaps {(e - s - busy) / 1e3:.1f} us")
print("\n| kernel | n alone | avg alone us | n overlapped | avg overlapped us | ratio | total overlapped us |")
print("|---|---|---|---|---|---|---|")
names = set(stats[0]) | set(stats[1])
out = []
for n in names:
    alone = stats[0].get(n) or stats[1].get(n)
    ov = stats[2].get(n)
    if not alone or not ov:
        continue
    a, o = alone[1] / alone[0], ov[1] / ov[0]
    out.append((ov[1], n, alone[0], a, ov[0], o))
for tot, n, na, a, no, o in sorted(out, reverse=True)[:40]:
    print(f"| {n} | {na} | {a:.1f} | {no} | {o:.1f} | {o / a:.2f} | {tot:.0f} |")

# ---- gap analysis of the busiest head queue in the overlapped phase ---------------------------------------------------
seg = ph[2]
byq = defaultdict(list)
for r in seg:
    byq[r["Queue_Id"]].append(r)
enc_q = max(byq, key=lambda q: sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in byq[q] if "gemm_bf16" in r["Kernel_Name"]))
head_q = max((q for q in byq if q != enc_q), key=lambda q: len(byq[q]))
hq = sorted(byq[head_q], key=lambda r: int(r["Start_Timestamp"]))
enc = sorted(byq[enc_q], key=lambda r: int(r["Start_Timestamp"]))


def enc_at(t):
    for r in enc:
        if int(r["Start_Timestamp"]) <= t <= int(r["End_Timestamp"]):
            return name(r)
    return "(encoder queue idle)"


gaps = []
for p, n in zip(hq[:-1], hq[1:]):
    g = (int(n["Start_Timestamp"]) - int(p["End_Timestamp"])) / 1e3
    gaps.append((g, name(p), name(n), enc_at(int(p["End_Timestamp"]))))
import statistics
gs = sorted(g for g, *_ in gaps)
print(f"\nhead queue {head_q}: {len(gaps)} gaps, total {sum(gs):.0f} us, median {statistics.median(gs):.1f}, p90 {gs[int(0.9 * len(gs))]:.1f}, max {gs[-1]:.1f}")
edges = [0, 2, 4, 8, 16, 32, 64, 128, 1e9]
for lo_, hi_ in zip(edges[:-1], edges[1:]):
    sel = [g for g in gs if lo_ <= g < hi_]
    print(f"   gaps in [{lo_:g}, {hi_:g}) us: {len(sel):4d}, total {sum(sel):8.0f} us")
by_enc = defaultdict(lambda: [0, 0.0])
for g, p, n, e in gaps:
    by_enc[e][0] += 1; by_enc[e][1] += g
print("   gap time by what the encoder queue was running when the previous head kernel ended:")
for e, (c_, t_) in sorted(by_enc.items(), key=lambda kv: -kv[1][1])[:10]:
    print(f"      {e:58s} x{c_:4d} total {t_:8.0f} us avg {t_ / c_:6.1f}")
by_next = defaultdict(lambda: [0, 0.0])
for g, p, n, e in gaps:
    by_next[n][0] += 1; by_next[n][1] += g
print("   gap time by the kernel that follows the gap:")
for e, (c_, t_) in sorted(by_next.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"      {e:58s} x{c_:4d} total {t_:8.0f} us avg {t_ / c_:6.1f}")
print("   largest gaps:")
for g, p, n, e in sorted(gaps, reverse=True)[:12]:
    print(f"      {g:7.1f} us  after {p}  before {n}  | encoder: {e}")
