import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ser_amd
from ser_amd import _engines as E, _lib as L
from oracle import ser_oracle as O
from tests.helpers import cfg_of, load_npz, split_fixture
from tests.test_gpu_encoders import _w2v_hf_cfg, _xlmr_hf_cfg
sda, _, ra = split_fixture(load_npz("audio_encoder.npz"))
sdt, _, rt = split_fixture(load_npz("text_encoder.npz"))
ca, ct = cfg_of(ra), cfg_of(rt)
masks = [0x7f, 0, 1, 2, 4, 8, 16, 32, 64]
for p in (L.PREC_BF16,):
    ea = E.Wav2Vec2Engine(_w2v_hf_cfg(ca), O.sub(sda, "encoder."), "cuda", p)
    et = E.XlmrEngine(_xlmr_hf_cfg(ct), O.sub(sdt, "encoder."), "cuda", p)
    g = torch.Generator().manual_seed(11)
    for B, T, St in ((3, 2400, 7), (2, 4000, 70), (5, 1700, 3)):
        wave = (0.1 * torch.randn(B, T, generator=g)).cuda()
        ids = torch.randint(4, ct["vocab"], (B, St), generator=g).cuda()
        mask = torch.ones(B, St).cuda()
        a1, t1 = ea.forward(wave), et.forward(ids, mask)
        a1b = ea.forward(wave)
        for mk in masks:
            L.lib.ser_debug_set_pair_mask(mk)
            a2, t2 = E.forward_pair(ea, et, wave, ids, mask)
            torch.cuda.synchronize()
            print(hex(mk), p, B, T, St, "nan sep", bool(a1.isnan().any()), bool(t1.isnan().any()), "nan pair", bool(a2.isnan().any()), bool(t2.isnan().any()),
                  "rerun equal", torch.equal(a1, a1b), "da", (a1 - a2).abs().max().item(), "dt", (t1 - t2).abs().max().item())
