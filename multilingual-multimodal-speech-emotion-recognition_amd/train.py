#!/usr/bin/env python3
"""Training CLI — same flags as the reference's src/train.py:28-38, same loop (:123-263), same
checkpoint layout (:249-262), on the HIP hot path.

    python -m ser_amd.train --train_manifest train_70.jsonl --val_manifest val_20.jsonl --epochs 5 --batch_size 16

Additions (all optional): --num_labels, --audio_model / --text_model (HF names or local directories),
--precision {bf16,bf16x3}, --synthetic N (seeded synthetic clips instead of manifests), --graph
(hipGraph-captured steps for fixed-length batches).  Multi-GPU: launch with torch.distributed.run; each
rank takes every world_size-th batch and gradients are averaged over RCCL.

Reference defects deliberately not reproduced: autocast() without device_type (:151),
--resume_from using the scheduler before it exists (:108), cross/pool modules left in train mode during
validation (:181).
"""
import argparse
import os
import sys

import torch
from torch.utils.data import DataLoader

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
import ser_amd  # noqa: E402,F401
from ser_amd.data.dataset import SERDataset, SyntheticSERDataset  # noqa: E402
from ser_amd.data.preprocess import add_noise_snr, speed_perturb  # noqa: E402
from ser_amd.models import AudioEncoder, TextEncoder  # noqa: E402
from ser_amd.optim import WarmupCosine  # noqa: E402
from ser_amd.system import GradReducer, SERSystem, TrainStepper  # noqa: E402
from ser_amd.utils import weighted_f1  # noqa: E402

NUM_LABELS = 4


def collate_fn(batch):
    audios, texts, labels = zip(*batch)
    return list(audios), list(texts), torch.tensor(labels, dtype=torch.long)


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument('--train_manifest', type=str, default='train_70.jsonl')
    p.add_argument('--val_manifest', type=str, default='val_20.jsonl')
    p.add_argument('--epochs', type=int, default=5)
    p.add_argument('--batch_size', type=int, default=4)
    p.add_argument('--lr', type=float, default=1e-4)
    p.add_argument('--warmup_ratio', type=float, default=0.1)
    p.add_argument('--use_amp', action='store_true', help='accepted for compatibility: the encoders already run bf16 MFMA')
    p.add_argument('--augment', action='store_true')
    p.add_argument('--proto_weight', type=float, default=0.05)
    p.add_argument('--save_dir', type=str, default='checkpoints')
    p.add_argument('--resume_from', type=str, help='Path to checkpoint to resume from')
    p.add_argument('--num_labels', type=int, default=NUM_LABELS)
    p.add_argument('--audio_model', type=str, default='facebook/wav2vec2-base')
    p.add_argument('--text_model', type=str, default='xlm-roberta-base')
    p.add_argument('--precision', choices=['bf16', 'bf16x3'], default='bf16')
    p.add_argument('--synthetic', type=int, default=0, help='train on N seeded synthetic utterances (no manifests needed)')
    p.add_argument('--graph', action='store_true')
    return p


def augment(audio_list):
    out = []
    for w in audio_list:
        if torch.rand(1).item() < 0.5:
            w = speed_perturb(w, 0.9 + 0.2 * torch.rand(1).item())
        if torch.rand(1).item() < 0.5:
            w = add_noise_snr(w, 10 + 10 * torch.rand(1).item())
        out.append(w)
    return out


def augment_device(wave, seed):
    """The same augmentation policy (ref train.py:129-142, host RNG draws in the same order) for an equal-length batch
    already on the device: resampling and noise run in HIP (data/gpu_augment.py)."""
    from ser_amd.data import gpu_augment as G
    rows, noisy, snrs = [], [], []
    for b in range(wave.shape[0]):
        w = wave[b:b + 1]
        if torch.rand(1).item() < 0.5:
            w = G.speed_perturb(w, 0.9 + 0.2 * torch.rand(1).item())
        if torch.rand(1).item() < 0.5:
            noisy.append(b)
            snrs.append(10 + 10 * torch.rand(1).item())
        rows.append(w)
    out = torch.cat(rows, dim=0)
    if noisy:
        idx = torch.tensor(noisy, device=wave.device)
        out[idx] = G.add_noise_snr(out[idx], torch.tensor(snrs), seed)
    return out


def tokenise(te, text_list, device):
    enc = te.tokenizer(text_list, padding=True, truncation=True, return_tensors="pt")
    return enc["input_ids"].to(device), enc["attention_mask"].to(device)


def main(argv=None):
    args = build_parser().parse_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit("the HIP hot path needs an MI355X; there is no CPU fallback")
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", 1), ("RANK", 0), ("LOCAL_RANK", 0)))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)

    if args.synthetic:
        train_ds = SyntheticSERDataset(args.synthetic, num_labels=args.num_labels, seed=0)
        val_ds = SyntheticSERDataset(max(args.batch_size, args.synthetic // 4), num_labels=args.num_labels, seed=1)
    else:
        train_ds, val_ds = SERDataset(args.train_manifest), SERDataset(args.val_manifest)
    train_loader = DataLoader(train_ds, batch_size=args.batch_size, shuffle=True, collate_fn=collate_fn)
    val_loader = DataLoader(val_ds, batch_size=args.batch_size, shuffle=False, collate_fn=collate_fn)

    torch.manual_seed(0)
    ae = AudioEncoder(args.audio_model, use_quality_gates=False, use_audio_conditioning=False, precision=args.precision)
    te = TextEncoder(args.text_model, precision=args.precision)
    sysm = SERSystem(ae, te, num_labels=args.num_labels).to(device)
    sysm.dropout_seed += rank                      # every data-parallel rank draws its own dropout masks
    opt = sysm.make_optimizer(lr=args.lr)
    total_steps = len(train_loader) * args.epochs
    sched = WarmupCosine(opt, total_steps, args.warmup_ratio)
    start_epoch = 0
    if args.resume_from and os.path.exists(args.resume_from):
        ck = torch.load(args.resume_from, map_location=device, weights_only=False)
        sysm.load_checkpoint_dict(ck)
        opt.load_state_dict(ck['optimizer'])
        sched.load_state_dict(ck['scheduler'])
        start_epoch = ck['epoch'] + 1
        print(f"Resuming from epoch {start_epoch}")
    reducer = GradReducer(sysm) if world > 1 else None
    stepper = TrainStepper(sysm, opt, sched, reducer, use_graph=args.graph, use_proto=args.proto_weight > 0)

    f1 = 0.0
    for epoch in range(start_epoch, args.epochs):
        sysm.train()
        for bi, (audio_list, text_list, labels) in enumerate(train_loader):
            if bi % world != rank:
                continue
            lens = {w.numel() for w in audio_list}
            if args.augment and len(lens) != 1:
                audio_list = augment(audio_list)
            ids, mask = tokenise(te, text_list, device)
            if len(lens) == 1:
                wave = torch.stack(audio_list).to(device)
                if args.augment:
                    wave = augment_device(wave, seed=epoch * 1000003 + bi)
                loss = stepper.step(wave, ids, mask, labels.to(device))
            else:   # ragged clips: the reference's pad-to-longest semantics, eager launches
                opt.zero_grad(set_to_none=True)
                a_seq, a_mask = sysm.audio_encoder(audio_list, text_list)
                t_seq, t_mask = sysm.text_encoder.forward_ids(ids, mask)
                fused = sysm.head(a_seq, a_mask, t_seq, t_mask)
                logits, unc, _ = sysm.classifier(fused, use_openmax=False, return_uncertainty=True)
                loss = sysm.criterion(logits, unc, fused, sysm.prototypes.prototypes, labels.to(device), args.proto_weight > 0)
                loss.backward()
                if reducer:
                    reducer.finish()
                opt.step()
                sched.step()
        sysm.eval()
        preds, gold, feats = [], [], []
        with torch.no_grad():
            for audio_list, text_list, labels in val_loader:
                a_seq, a_mask = sysm.audio_encoder(audio_list, text_list)
                t_seq, t_mask = sysm.text_encoder(text_list)
                fused = sysm.head(a_seq, a_mask, t_seq, t_mask)
                preds.append(torch.argmax(sysm.classifier(fused), dim=1).cpu())
                gold.append(labels)
                if epoch == args.epochs - 1:
                    feats.append(sysm.classifier.penultimate_features(fused))
        f1 = weighted_f1(torch.cat(preds), torch.cat(gold))
        if rank == 0:
            print(f"Epoch {epoch} F1: {f1}")
        if epoch == args.epochs - 1:
            sysm.classifier.fit_weibull(torch.cat(feats), torch.cat(gold))
        if rank == 0:
            os.makedirs(args.save_dir, exist_ok=True)
            ck = sysm.checkpoint_dict()
            ck.update(optimizer=opt.state_dict(), scheduler=sched.state_dict(), epoch=epoch, f1=f1)
            torch.save(ck, os.path.join(args.save_dir, f'epoch_{epoch}_f1_{f1:.4f}.pt'))
    if world > 1:
        torch.distributed.destroy_process_group()
    return f1


if __name__ == "__main__":
    main()
