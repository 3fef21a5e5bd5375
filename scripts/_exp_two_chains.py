import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import ser_amd
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
def cap(batch):
    sysm, wc, xc = bench.build_system("bf16x3", dev)
    sysm.eval()
    wave, ids, mask, _ = [t.to(dev) for t in bench.synth_batch(batch, 4.0, 32, xc.vocab_size, sysm.num_labels, 7)]
    with torch.no_grad():
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2): sysm.encode_frozen(wave, ids, mask)
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = sysm.encode_frozen(wave, ids, mask)
    return g, sysm, out
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/reps
g16a, *_ = cap(16); g16b, *_ = cap(16)
g8a, *_ = cap(8); g8b, *_ = cap(8)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
cur = torch.cuda.current_stream()
def seq(ga, gb):
    ga.replay(); gb.replay()
def par(ga, gb):
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1): ga.replay()
    with torch.cuda.stream(s2): gb.replay()
    cur.wait_stream(s1); cur.wait_stream(s2)
print("one batch-16 encoder graph:", round(t(lambda: g16a.replay()),3))
print("one batch-8 encoder graph:", round(t(lambda: g8a.replay()),3))
print("two batch-8 sequential:", round(t(lambda: seq(g8a,g8b)),3), " concurrent:", round(t(lambda: par(g8a,g8b)),3))
print("two batch-16 sequential:", round(t(lambda: seq(g16a,g16b)),3), " concurrent:", round(t(lambda: par(g16a,g16b)),3))
