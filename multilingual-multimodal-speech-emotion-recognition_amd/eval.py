#!/usr/bin/env python3
"""Evaluation CLI — flags of the reference's src/eval.py:72-78 (--manifest --checkpoint --batch_size
--use_tta --num_tta --calibrate --val_manifest) on the HIP hot path; the number of classes is read from
the checkpoint instead of being hard-coded to 6 (ref :102)."""
import argparse
import os
import sys

import numpy as np
import torch
from torch.utils.data import DataLoader

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
import ser_amd  # noqa: E402,F401
from ser_amd.data.dataset import SERDataset  # noqa: E402
from ser_amd.data.preprocess import add_noise_snr, speed_perturb  # noqa: E402
from ser_amd.models import AudioEncoder, TextEncoder  # noqa: E402
from ser_amd.system import SERSystem  # noqa: E402
from ser_amd.utils import energy_score, weighted_f1  # noqa: E402

EMOTIONS = ['angry', 'happy', 'sad', 'neutral', 'disgust', 'fear']


def collate_fn(batch):
    audios, texts, labels = zip(*batch)
    return list(audios), list(texts), torch.tensor(labels, dtype=torch.long)


def tta_variants(audio, num_augs=5):
    """original, speed 0.95 / 1.05, noise 15 / 20 dB (ref :23-41), truncated to num_augs."""
    v = [audio, speed_perturb(audio, 0.95), speed_perturb(audio, 1.05), add_noise_snr(audio, 15), add_noise_snr(audio, 20)]
    return v[:num_augs]


def find_optimal_temperature(val_logits, val_labels):
    """grid search over logspace(-1, 2, 100) of the reference's one-bin |confidence - accuracy| proxy (:49-67): the 100
    objectives in one launch (`ser_temperature_grid`), then the first strict minimum, as the reference's loop picks it."""
    from ser_amd import _ops as O
    temps = torch.logspace(-1, 2, 100, device=val_logits.device)
    ece = O.temperature_grid(val_logits, val_labels, temps).cpu()
    best_t, best = 1.0, float('inf')
    for t_, e in zip(temps.cpu().tolist(), ece.tolist()):
        if e < best:
            best, best_t = e, t_
    return best_t


def O_axpby(x, acc, a):
    """acc (+)= a * x on the device (ser_axpby)."""
    from ser_amd import _ops as O
    if acc is None:
        acc = torch.zeros_like(x)
    return O.axpby(x.contiguous(), acc, a, 1.0)


def logits_for(sysm, audio_list, text_list, use_openmax):
    a_seq, a_mask = sysm.audio_encoder(audio_list, text_list)
    t_seq, t_mask = sysm.text_encoder(text_list)
    return sysm.classifier(sysm.head(a_seq, a_mask, t_seq, t_mask), use_openmax=use_openmax)


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('--manifest', type=str, required=True)
    p.add_argument('--checkpoint', type=str, required=True)
    p.add_argument('--batch_size', type=int, default=8)
    p.add_argument('--use_tta', action='store_true')
    p.add_argument('--num_tta', type=int, default=5)
    p.add_argument('--calibrate', action='store_true')
    p.add_argument('--val_manifest', type=str)
    p.add_argument('--audio_model', type=str, default='facebook/wav2vec2-base')
    p.add_argument('--text_model', type=str, default='xlm-roberta-base')
    p.add_argument('--precision', choices=['bf16x3', 'bf16'], default='bf16x3')
    p.add_argument('--use_quality_gates', action='store_true', help="quality gates of the reference's default AudioEncoder() (device kernels)")
    p.add_argument('--use_audio_conditioning', action='store_true', help="audio conditioning of the reference's default AudioEncoder() (device kernels)")
    p.add_argument('--vad_method', type=str, default='webrtc')
    args = p.parse_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit("the HIP hot path needs an MI355X; there is no CPU fallback")
    device = torch.device("cuda")
    ck = torch.load(args.checkpoint, map_location=device, weights_only=False)
    num_labels = ck['classifier']['weibull_alpha'].numel()
    ae = AudioEncoder(args.audio_model, use_quality_gates=args.use_quality_gates, vad_method=args.vad_method,
                      use_audio_conditioning=args.use_audio_conditioning, precision=args.precision)
    te = TextEncoder(args.text_model, precision=args.precision)
    sysm = SERSystem(ae, te, num_labels=num_labels).to(device)
    sysm.load_checkpoint_dict(ck)
    sysm.eval()

    temp = 1.0
    if args.calibrate and args.val_manifest:
        vl, vy = [], []
        with torch.no_grad():
            for audio_list, text_list, labels in DataLoader(SERDataset(args.val_manifest), batch_size=args.batch_size,
                                                            collate_fn=collate_fn):
                vl.append(logits_for(sysm, audio_list, text_list, use_openmax=False))
                vy.append(labels.to(device))
        temp = find_optimal_temperature(torch.cat(vl), torch.cat(vy))
        print(f"Optimal temperature: {temp:.3f}")

    preds, gold, energies, probs = [], [], [], []
    with torch.no_grad():
        for audio_list, text_list, labels in DataLoader(SERDataset(args.manifest), batch_size=args.batch_size,
                                                        collate_fn=collate_fn):
            if args.use_tta:   # variant-major batches: every variant of the batch is one batched forward; the mean over
                variants = [tta_variants(a, args.num_tta) for a in audio_list]            # variants is an axpby chain on the device
                nv = len(variants[0])
                lg = None
                for k in range(nv):
                    lk = logits_for(sysm, [v[k] for v in variants], text_list, True)
                    lg = O_axpby(lk, lg, 1.0 / nv)
            else:
                lg = logits_for(sysm, audio_list, text_list, True)
            from ser_amd import _ops as O
            pr, pd, en = O.eval_consumers(lg, temp if args.calibrate else 1.0)     # /T, softmax, arg-max, -logsumexp in one launch
            probs.append(pr.cpu())
            preds.append(pd.cpu())
            energies.append(en.cpu())
            gold.append(labels)
    preds, gold = torch.cat(preds), torch.cat(gold)
    energies, probs = torch.cat(energies).numpy(), torch.cat(probs).numpy()
    from sklearn.metrics import classification_report, confusion_matrix
    names = EMOTIONS[:num_labels] if num_labels <= len(EMOTIONS) else [str(i) for i in range(num_labels)]
    print("=" * 50 + "\nEVALUATION RESULTS\n" + "=" * 50)
    print(f"Weighted F1 Score: {weighted_f1(preds, gold):.4f}")
    print(f"Energy Score - Mean: {energies.mean():.3f}, Std: {energies.std():.3f}")
    print(f"Temperature: {temp:.3f}")
    print(classification_report(gold.numpy(), preds.numpy(), labels=list(range(num_labels)), target_names=names, zero_division=0))
    print(confusion_matrix(gold.numpy(), preds.numpy(), labels=list(range(num_labels))))
    conf = probs.max(axis=1)
    print(f"Mean confidence: {conf.mean():.3f}  Std: {conf.std():.3f}  >0.8: {(conf > 0.8).mean():.3f}  <0.5: {(conf < 0.5).mean():.3f}")


if __name__ == "__main__":
    main()
