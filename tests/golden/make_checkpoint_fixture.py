#!/usr/bin/env python3
"""A checkpoint in the REFERENCE's own layout, written by the reference's modules and torch's AdamW / LambdaLR.

Run in the build container only (needs /root/reference, read-only), after make_fixtures.py:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_checkpoint_fixture.py

The seven reference modules are built exactly as src/train.py:54-69 builds them (small dimensions), the ten AdamW
parameter groups and the LambdaLR schedule as :72-83 / :114-121, two optimisation steps are taken on a seeded batch with
the reference's loss terms (:154-168), and the dictionary of :249-262 is written with torch.save.  Only data is
written: tensors and the optimizer / scheduler state dicts.

The reference's default AudioEncoder() also carries the quality-gate / conditioning sub-modules; their constructor needs
webrtcvad + librosa, which do not exist here, so the encoder is built with the gates off and the gate parameters are added
to the audio_encoder entry under the reference's key names (audio_encoder.py:25-52) with seeded values and, like the
reference (where they never receive a gradient on this path), with parameter indices but no optimizer state.  The index
shift this causes in the optimizer's `params` lists is part of what the loader has to handle.
"""
import os
import sys
import tempfile

sys.dont_write_bytecode = True
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_fixtures as MF  # noqa: E402

GATE_SHAPES = [  # (key, shape) in the registration order of ref audio_encoder.py:25-52 (quality first, then conditioning, then combined)
    ("quality_gates.quality_projection.0.weight", (32, 8)), ("quality_gates.quality_projection.0.bias", (32,)),
    ("quality_gates.quality_projection.3.weight", (8, 32)), ("quality_gates.quality_projection.3.bias", (8,)),
    ("quality_fusion.0.weight", (128, 136)), ("quality_fusion.0.bias", (128,)),
    ("audio_conditioning.conditioning_projection.0.weight", (32, 12)), ("audio_conditioning.conditioning_projection.0.bias", (32,)),
    ("audio_conditioning.conditioning_projection.3.weight", (12, 32)), ("audio_conditioning.conditioning_projection.3.bias", (12,)),
    ("conditioning_fusion.0.weight", (128, 140)), ("conditioning_fusion.0.bias", (128,)),
    ("combined_fusion.0.weight", (128, 148)), ("combined_fusion.0.bias", (128,)),
]


def main():
    R = MF._import_reference()
    sys.path.insert(0, MF.REF)
    from models.losses import LabelSmoothingCrossEntropy, ClassBalancedFocalLoss
    torch.set_grad_enabled(True)
    tmp = tempfile.mkdtemp(prefix="ser_fix_")
    da, dt = MF.make_local_models(tmp, MF.A_CFG, MF.T_CFG)
    torch.manual_seed(21)
    C = 4
    audio_encoder = R["AudioEncoder"](model_name=da, adapter_dim=32, use_quality_gates=False, use_audio_conditioning=False)
    text_encoder = R["TextEncoder"](model_name=dt, adapter_dim=32)
    cross = R["CrossModalAttention"](128, 128, shared_dim=64, num_heads=2)
    pool_a, pool_t = R["AttentiveStatsPooling"](128), R["AttentiveStatsPooling"](128)
    fusion = R["FusionLayer"](256, 256, 64)
    classifier = R["AdvancedOpenMaxClassifier"](input_dim=64, num_labels=C, num_layers=3, base_dim=64, dropout=0.15)
    prototypes = R["PrototypeMemory"](C, 64)
    lr = 1e-3
    optimizer = torch.optim.AdamW([
        {'params': audio_encoder.parameters(), 'lr': lr * 0.1, 'weight_decay': 0.025},
        {'params': text_encoder.parameters(), 'lr': lr * 0.1, 'weight_decay': 0.025},
        {'params': cross.parameters(), 'lr': lr, 'weight_decay': 0.05},
        {'params': pool_a.parameters(), 'lr': lr, 'weight_decay': 0.05},
        {'params': pool_t.parameters(), 'lr': lr, 'weight_decay': 0.05},
        {'params': fusion.parameters(), 'lr': lr, 'weight_decay': 0.05},
        {'params': classifier.deep_classifier.parameters(), 'lr': lr * 1.5, 'weight_decay': 0.06},
        {'params': classifier.anchor_clustering.parameters(), 'lr': lr * 2.0, 'weight_decay': 0.04},
        {'params': classifier.uncertainty_head.parameters(), 'lr': lr * 1.0, 'weight_decay': 0.05},
        {'params': prototypes.parameters(), 'lr': lr, 'weight_decay': 0.05},
    ], weight_decay=0.05)
    total_steps, warmup_steps = 10, 0

    def lr_lambda(step):
        if step < warmup_steps:
            return float(step) / max(1, warmup_steps)
        progress = (step - warmup_steps) / max(1, total_steps - warmup_steps)
        return 0.5 * (1.0 + torch.cos(torch.tensor(progress * 3.1415926535))).item()
    scheduler = torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda)
    ce, focal = LabelSmoothingCrossEntropy(0.1), ClassBalancedFocalLoss(beta=0.9999, gamma=2.0, num_classes=C)
    for m in (audio_encoder, text_encoder, cross, pool_a, pool_t, fusion, classifier):
        m.eval()                                        # deterministic steps (no dropout): the state is what is pinned
    g = torch.Generator().manual_seed(5)
    waves = [0.1 * torch.randn(4000, generator=g) for _ in range(4)]
    texts = [" ".join(f"w{int(i)}" for i in torch.randint(0, MF.VOCAB_WORDS, (6,), generator=g)) for _ in range(4)]
    labels = torch.tensor([0, 1, 2, 3])
    for _ in range(2):
        a_seq, a_mask = audio_encoder(waves, texts)
        t_seq, t_mask = text_encoder(texts)
        a_enh, t_enh = cross(a_seq, t_seq, a_mask, t_mask)
        fused = fusion(pool_a(a_enh, a_mask), pool_t(t_enh, t_mask))
        logits, unc, anchor = classifier(fused, use_openmax=False, return_uncertainty=True)
        correct = (labels == logits.argmax(dim=1)).float()
        loss = ce(logits, labels) + 0.3 * focal(logits, labels) + 0.1 * anchor + 0.05 * (unc * correct).mean() \
            + 0.01 * prototypes.prototype_loss(fused, labels)
        optimizer.zero_grad(set_to_none=True)
        loss.backward()
        optimizer.step()
        scheduler.step()
    ck = {'audio_encoder': audio_encoder.state_dict(), 'text_encoder': text_encoder.state_dict(), 'cross': cross.state_dict(),
          'pool_a': pool_a.state_dict(), 'pool_t': pool_t.state_dict(), 'fusion': fusion.state_dict(),
          'classifier': classifier.state_dict(), 'prototypes': prototypes.state_dict(), 'optimizer': optimizer.state_dict(),
          'scheduler': scheduler.state_dict(), 'epoch': 0, 'f1': 0.25}
    # gate sub-modules of the reference's default constructor (see the module docstring): keys appended, optimizer indices shifted
    n_gate = len(GATE_SHAPES)
    gg = torch.Generator().manual_seed(6)
    for k, shp in GATE_SHAPES:
        ck['audio_encoder'][k] = 0.05 * torch.randn(shp, generator=gg)
    n_audio = len(list(audio_encoder.parameters()))
    opt = ck['optimizer']
    opt['state'] = {(i if i < n_audio else i + n_gate): v for i, v in opt['state'].items()}
    for gi, grp in enumerate(opt['param_groups']):
        grp['params'] = [(i if i < n_audio else i + n_gate) for i in grp['params']]
    opt['param_groups'][0]['params'] = list(range(n_audio + n_gate))
    # expected values for the loader test: a few optimizer moments by parameter NAME
    probe = {}
    names = {id(p): ("cross." + n) for n, p in cross.named_parameters()}
    names.update({id(p): ("classifier." + n) for n, p in classifier.named_parameters()})
    names.update({id(p): ("audio_encoder." + n) for n, p in audio_encoder.named_parameters()})
    for p, st in optimizer.state.items():
        nm = names.get(id(p))
        if nm in ("cross.q_a.weight", "classifier.deep_classifier.residual_layers.1.block.1.weight", "audio_encoder.adapter.0.weight"):
            probe[nm] = dict(exp_avg=st['exp_avg'].clone(), exp_avg_sq=st['exp_avg_sq'].clone(), step=float(st['step']))
    ck['_probe'] = probe
    path = os.path.join(HERE, "ref_checkpoint_small.pt")
    torch.save(ck, path)
    print(f"wrote {path}: {os.path.getsize(path) / 1e6:.2f} MB; optimizer state entries {len(opt['state'])}, scheduler {ck['scheduler']['last_epoch']}")


if __name__ == "__main__":
    main()
