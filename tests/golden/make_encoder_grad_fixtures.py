#!/usr/bin/env python3
"""Golden gradients of the UNFROZEN encoders (BASELINE config 3: `freeze_base=False`), from the reference modules.

Run in the build container only (needs /root/reference, read-only), after make_fixtures.py:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_encoder_grad_fixtures.py

The reference AudioEncoder / TextEncoder are built with `freeze_base=False`, loaded with the weights already recorded in
audio_encoder.npz / text_encoder.npz, run in eval mode (HF's SpecAugment, dropout and LayerDrop are training-mode
stochastic and not part of the pinned arithmetic) on the recorded inputs, and the gradient of sum(out * g) for a seeded g
is recorded for every parameter.  Only data is written: g and the gradients.
"""
import os
import sys
import tempfile

sys.dont_write_bytecode = True
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_fixtures as MF  # noqa: E402


def load(name):
    z = np.load(os.path.join(HERE, name), allow_pickle=False)
    return {k: z[k] for k in z.files}


def main():
    R = MF._import_reference()
    torch.set_grad_enabled(True)
    tmp = tempfile.mkdtemp(prefix="ser_fix_")
    da, dt = MF.make_local_models(tmp, MF.A_CFG, MF.T_CFG)

    # ---- audio ----
    z = load("audio_encoder.npz")
    ae = R["AudioEncoder"](model_name=da, adapter_dim=32, freeze_base=False, use_quality_gates=False,
                           use_audio_conditioning=False).eval()
    ae.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in z.items() if k.startswith("sd.")}, strict=True)
    assert all(p.requires_grad for p in ae.parameters())
    waves = [torch.from_numpy(z["wave0"]), torch.from_numpy(z["wave1"])]
    a_seq, _ = ae(waves, ["x", "y"])
    assert torch.allclose(a_seq.detach(), torch.from_numpy(z["a_seq"]), atol=1e-6), "forward differs from audio_encoder.npz"
    g = torch.randn(a_seq.shape, generator=torch.Generator().manual_seed(77))
    (a_seq * g).sum().backward()
    MF.save("audio_encoder_grads.npz", g_out=g.numpy(), **MF.grads_np(ae))

    # ---- text ----
    z = load("text_encoder.npz")
    te = R["TextEncoder"](model_name=dt, adapter_dim=32, freeze_base=False).eval()
    te.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in z.items() if k.startswith("sd.")}, strict=True)
    import json
    texts = json.loads(str(z["texts"]))
    t_seq, _ = te(texts)
    assert torch.allclose(t_seq.detach(), torch.from_numpy(z["t_seq"]), atol=1e-6), "forward differs from text_encoder.npz"
    g = torch.randn(t_seq.shape, generator=torch.Generator().manual_seed(78))
    (t_seq * g).sum().backward()
    MF.save("text_encoder_grads.npz", g_out=g.numpy(), **MF.grads_np(te))


if __name__ == "__main__":
    main()
