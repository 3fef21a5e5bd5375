#!/usr/bin/env python3
"""Ordered kernel list of ONE head step (head graph + AdamW graph, alone on the chip) from a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d DIR -o k -- python3 scripts/head_sequence.py run
    python3 scripts/head_sequence.py report DIR/k_kernel_trace.csv"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if sys.argv[1] == "run":
    import torch
    import bench
    from ser_amd.system import PipelinedStepper
    dev = torch.device("cuda:0")
    sysm, wc, xc = bench.build_system("bf16x3", dev)
    sysm.train()
    opt = sysm.make_optimizer(1e-4)
    st = PipelinedStepper(sysm, opt, group=1)
    b = [x.to(dev) for x in bench.synth_batch(16, 4.0, 32, xc.vocab_size, 4, 1)]
    for _ in range(st.prime):
        st.feed(*b)
    for _ in range(3):
        st.step(*b)
    torch.cuda.synchronize()
    time.sleep(0.05)
    for _ in range(3):
        st.g_head.replay()
        st.g_opt.replay()
        torch.cuda.synchronize()
        time.sleep(0.05)
    print("done")
else:
    import csv, re
    rows = list(csv.DictReader(open(sys.argv[2])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    name = lambda r: re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0][:60]
    bursts, cur, last_end = [], [], None
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if last_end is not None and s - last_end > 30e6:
            bursts.append(cur); cur = []
        cur.append(r)
        last_end = e if last_end is None else max(last_end, e)
    bursts.append(cur)
    seg = bursts[-1]
    t0 = int(seg[0]["Start_Timestamp"])
    qs = sorted({r["Queue_Id"] for r in seg})
    print(f"{len(seg)} launches, span {(int(seg[-1]['End_Timestamp']) - t0) / 1e3:.1f} us, queues {qs}")
    prev_end = {}
    for r in seg:
        s, e, q = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]
        gap = (s - prev_end[q]) / 1e3 if q in prev_end else 0.0
        prev_end[q] = e
        print(f"  +{(s - t0) / 1e3:8.1f} us  q{q}  {(e - s) / 1e3:7.1f} us  gap {gap:6.1f}  grid {r.get('Grid_Size', '?'):>8} wg {r.get('Workgroup_Size', '?'):>4} lds {r.get('LDS_Block_Size', '?'):>6}  {name(r)}")
