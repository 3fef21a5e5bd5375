#!/usr/bin/env python3
"""Training CLI — same flags as the reference's src/train.py:28-38, same loop (:123-263), same
checkpoint layout (:249-262), on the HIP hot path.

    python -m ser_amd.train --train_manifest train_70.jsonl --val_manifest val_20.jsonl --epochs 5 --batch_size 16

Additions (all optional): --num_labels, --audio_model / --text_model (HF names or local directories),
--precision {bf16x3,bf16}, --synthetic N (seeded synthetic clips instead of manifests), --graph
(hipGraph-captured steps, one graph per input shape), --seed, --no_bucketing, --unfreeze_encoders (BASELINE config 3).

Multi-GPU: launch with torch.distributed.run, one process per GPU.  `ShardedBucketBatchSampler` gives every rank the
SAME number of batches per epoch (the tail is padded by wrapping around), a rank decodes only its own batches, the
shuffle comes from a generator seeded by (seed, epoch) on every rank alike, and the augmentation draws come from a
per-rank generator, so ranks can never disagree about the epoch's batch list or wait in an all-reduce nobody joins.
The learning-rate schedule counts the steps ONE rank takes.

Reference defects deliberately not reproduced: autocast() without device_type (:151),
--resume_from using the scheduler before it exists (:108), cross/pool modules left in train mode during
validation (:181).
"""
import argparse
import math
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
from ser_amd.data.sampler import ShardedBucketBatchSampler  # noqa: E402
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
    p.add_argument('--use_amp', action='store_true',
                   help="the reference's mixed-precision switch (ref train.py:34,88,151: bf16 autocast): selects --precision bf16, i.e. one bf16 "
                        "MFMA product per multiply with fp32 accumulation in the encoders (frozen or fine-tuned) and in the head's backward "
                        "products; the head's forward stays fp32-equivalent.  No loss scaling is needed for bf16 (GradScaler is a no-op there)")
    p.add_argument('--augment', action='store_true')
    p.add_argument('--proto_weight', type=float, default=0.05)
    p.add_argument('--save_dir', type=str, default='checkpoints')
    p.add_argument('--resume_from', type=str, help='Path to checkpoint to resume from')
    p.add_argument('--num_labels', type=int, default=NUM_LABELS)
    p.add_argument('--audio_model', type=str, default='facebook/wav2vec2-base')
    p.add_argument('--text_model', type=str, default='xlm-roberta-base')
    p.add_argument('--precision', choices=['bf16x3', 'bf16'], default='bf16x3',
                   help='bf16x3: three bf16 MFMA products per multiply (meets the 1e-3 logit tolerance); bf16: one product (fast mode)')
    p.add_argument('--synthetic', type=int, default=0, help='train on N seeded synthetic utterances (no manifests needed)')
    p.add_argument('--synthetic_seconds', type=str, default='4', help='duration(s) of the synthetic clips, comma separated (ragged corpus)')
    p.add_argument('--graph', action='store_true')
    p.add_argument('--seed', type=int, default=0)
    p.add_argument('--use_quality_gates', action='store_true',
                   help="the reference AudioEncoder()'s quality gates (VAD / SNR / clipping / spectral checks, ref quality_gates.py) as device kernels")
    p.add_argument('--use_audio_conditioning', action='store_true',
                   help="the reference AudioEncoder()'s audio conditioning (hum notch, high-pass, Wiener, loudness, ref audio_conditioning.py) as device kernels")
    p.add_argument('--vad_method', type=str, default='webrtc', help="'webrtc' (reference default; the energy VAD is substituted) or 'librosa'")
    p.add_argument('--no_bucketing', action='store_true', help='plain shuffled batches instead of length-bucketed ones')
    p.add_argument('--unfreeze_encoders', action='store_true',
                   help='BASELINE config 3: full fine-tune, every Wav2Vec2 / XLM-R parameter trained (reference freeze_base=False)')
    p.add_argument('--encoder_train_noise', action='store_true',
                   help="frozen encoders: run them with HF's training-mode noise (dropout sites, LayerDrop, SpecAugment), as the "
                        "reference does by calling .train() on them (train.py:124); slower: fp32 three-product operators, no graph")
    p.add_argument('--no_encoder_noise', action='store_true',
                   help='with --unfreeze_encoders: leave out the encoders\' own training-mode noise (HF dropout sites, LayerDrop, '
                        'SpecAugment), which is on by default as in the reference (.train() on both encoders, src/train.py:124)')
    return p


class AugmentRng:
    """The augmentation policy of ref train.py:129-142 — 50 % speed perturbation 0.9-1.1, 50 % noise at 10-20 dB — with
    its draws taken from a generator owned by the rank (seed, rank, epoch): the global RNG is left alone."""

    def __init__(self, seed, rank):
        self.seed, self.rank = int(seed), int(rank)
        self.gen = torch.Generator()
        self.set_epoch(0)

    def set_epoch(self, epoch):
        self.gen.manual_seed((self.seed * 7919 + self.rank) * 1000003 + int(epoch))

    def u(self):
        return torch.rand(1, generator=self.gen).item()

    def host(self, audio_list):
        out = []
        for w in audio_list:
            if self.u() < 0.5:
                w = speed_perturb(w, 0.9 + 0.2 * self.u())
            if self.u() < 0.5:
                w = add_noise_snr(w, 10 + 10 * self.u(), generator=self.gen)
            out.append(w)
        return out

    def device(self, wave, noise_seed):
        """The same policy (host draws in the same order) for an equal-length batch already on the device: resampling
        and noise run in HIP (data/gpu_augment.py)."""
        from ser_amd.data import gpu_augment as G
        rows, noisy, snrs = [], [], []
        for b in range(wave.shape[0]):
            w = wave[b:b + 1]
            if self.u() < 0.5:
                w = G.speed_perturb(w, 0.9 + 0.2 * self.u())
            if self.u() < 0.5:
                noisy.append(b)
                snrs.append(10 + 10 * self.u())
            rows.append(w)
        out = torch.cat(rows, dim=0)
        if noisy:
            idx = torch.tensor(noisy, device=wave.device)
            out[idx] = G.add_noise_snr(out[idx], torch.tensor(snrs), noise_seed)
        return out


class HipEngine:
    """The seven modules on the HIP path + optimizer, scheduler, gradient reducer and stepper: what `run` drives."""

    def __init__(self, args, device, rank, world, steps_per_epoch):
        from ser_amd.models import AudioEncoder, TextEncoder
        from ser_amd.optim import WarmupCosine
        from ser_amd.system import GradReducer, SERSystem, TrainStepper
        torch.manual_seed(args.seed)               # identical initial replicas on every rank
        self.args, self.device, self.rank = args, device, rank
        frozen = not args.unfreeze_encoders
        # the reference's train.py builds AudioEncoder() with its defaults (quality gates + audio conditioning on, ref :54);
        # here they are opt-in flags because they silence every clip that comes with a transcript
        ae = AudioEncoder(args.audio_model, use_quality_gates=args.use_quality_gates, vad_method=args.vad_method,
                          use_audio_conditioning=args.use_audio_conditioning, precision=args.precision, freeze_base=frozen)
        self.gated = args.use_quality_gates or args.use_audio_conditioning
        self.te = TextEncoder(args.text_model, precision=args.precision, freeze_base=frozen)
        self.sys = SERSystem(ae, self.te, num_labels=args.num_labels).to(device)
        self.sys.dropout_seed += rank              # every data-parallel rank draws its own dropout masks
        noisy = (args.unfreeze_encoders and not args.no_encoder_noise) or args.encoder_train_noise
        if noisy:
            for m in (ae, self.te):
                m.encoder_train_noise, m.noise_seed = True, args.seed * 64 + rank
        self.opt = self.sys.make_optimizer(lr=args.lr)
        self.sched = WarmupCosine(self.opt, steps_per_epoch * args.epochs, args.warmup_ratio)
        self.reducer = GradReducer(self.sys) if world > 1 else None
        # LayerDrop / SpecAugment are per-step host decisions: a captured step takes them as device words staged before each replay
        # (models/_finetune.py Noise.stage; a dropped layer's AdamW update is gated by the same word).  Under data parallelism the
        # ranks drop different layers and the reduced gradient decides, so that combination steps eagerly
        self.stepper = TrainStepper(self.sys, self.opt, self.sched, self.reducer, use_graph=args.graph and (not noisy or world == 1),
                                    use_proto=args.proto_weight > 0)
        self.aug = AugmentRng(args.seed, rank) if args.augment else None
        self.start_epoch = 0
        if args.resume_from and os.path.exists(args.resume_from):
            ck = torch.load(args.resume_from, map_location=device, weights_only=False)
            self.sys.load_checkpoint_dict(ck)
            by_name = {(k, n): p for k in self.sys.CKPT_KEYS for n, p in getattr(self.sys, k).named_parameters()}
            self.opt.load_state_dict(ck['optimizer'], order=self.sys.torch_param_order(ck), params_by_name=by_name)
            self.sched.load_state_dict(ck['scheduler'])
            self.start_epoch = ck['epoch'] + 1
            print(f"Resuming from epoch {self.start_epoch}")

    def tokenise(self, text_list):
        enc = self.te.tokenizer(text_list, padding=True, truncation=True, return_tensors="pt")
        return enc["input_ids"].to(self.device), enc["attention_mask"].to(self.device)

    def begin_epoch(self, epoch):
        self.sys.train()
        if self.aug:
            self.aug.set_epoch(epoch)

    def train_step(self, audio_list, text_list, labels, epoch, step):
        s, dev = self.sys, self.device
        lens = {w.numel() for w in audio_list}
        ids, mask = self.tokenise(text_list)
        if len(lens) == 1:
            wave = torch.stack(audio_list).to(dev)
            if self.aug:
                wave = self.aug.device(wave, noise_seed=(epoch * 1000003 + step) * 64 + self.rank)
            # equal-length batch: one (graph-captured, with --graph) step; with the gate flags on the front end's kernels head
            # it - the transcripts enter as the per-clip language features (host: a text operation)
            lid = s.language_features(text_list, wave.shape[0]).to(dev) if self.gated else None
            return self.stepper.step(wave, ids, mask, labels.to(dev), lid)
        elif self.aug:
            # ragged clips: the reference's pad-to-longest semantics (one encoder pass per distinct length), eager launches
            audio_list = self.aug.host(audio_list)
        self.opt.zero_grad(set_to_none=True)
        for _, n in s.encoder_noises():          # eager forward: LayerDrop / SpecAugment act as host decisions (a captured step left the
            n.static = False                     # device-word mode on), and a dropped layer's parameters simply have no gradient
        gates, self.opt.gates = self.opt.gates, {}
        if self.reducer:
            self.reducer.arm(overlap=False)      # one adapter backward per clip length: buckets are complete only after backward
        a_seq, a_mask = s.audio_encoder(audio_list, text_list)
        t_seq, t_mask = s.text_encoder.forward_ids(ids, mask)
        s._set_precision()
        with s._dropout_scope():
            fused = s.head(a_seq, a_mask, t_seq, t_mask)
            logits, unc, _ = s.classifier(fused, use_openmax=False, return_uncertainty=True)
        loss = s.criterion(logits, unc, fused, s.prototypes.prototypes, labels.to(dev), self.args.proto_weight > 0)
        loss.backward()
        if self.reducer:
            self.reducer.finish()
        self.opt.step()
        self.opt.gates = gates
        self.sched.step()
        return loss.detach()

    def end_epoch_health(self, last_loss):
        """Fail loudly instead of training on: a non-finite loss, or a hand-off wait that the persistent classifier
        kernels abandoned (sticky word 1 of their scratch areas, csrc/persist.hip)."""
        if last_loss is not None and not math.isfinite(float(last_loss)):
            raise RuntimeError(f"non-finite training loss: {float(last_loss)}")
        self.sys.check_persistent_kernels()

    def begin_eval(self):
        self.sys.eval()

    @torch.no_grad()
    def predict(self, audio_list, text_list, want_features):
        s = self.sys
        a_seq, a_mask = s.audio_encoder(audio_list, text_list)
        t_seq, t_mask = s.text_encoder(text_list)
        fused = s.head(a_seq, a_mask, t_seq, t_mask)
        pred = torch.argmax(s.classifier(fused), dim=1).cpu()
        return pred, (s.classifier.penultimate_features(fused) if want_features else None)

    def fit_weibull(self, feats, gold):
        self.sys.classifier.fit_weibull(feats, gold)

    def checkpoint(self, epoch, f1):
        """The reference's checkpoint dict (ref train.py:249-262).  `optimizer` / `scheduler` are written in the formats of
        torch.optim.AdamW / LambdaLR, so the reference's own --resume_from (`optimizer.load_state_dict`, ref :100-108) reads
        a checkpoint written here, and this package reads both (FlatAdamW.load_state_dict, WarmupCosine.load_state_dict)."""
        ck = self.sys.checkpoint_dict()
        by_name = {(k, n): p for k in self.sys.CKPT_KEYS for n, p in getattr(self.sys, k).named_parameters()}
        lrs = [g["lr"] for g in self.opt.param_groups]
        base = [self.opt.base_lr * g["lr_mult"] for g in self.opt.groups]
        sched = dict(self.sched.state_dict(), base_lrs=base, _last_lr=lrs, _step_count=self.sched.last_epoch + 1,
                     _get_lr_called_within_step=False, lr_lambdas=[None] * len(base))
        ck.update(optimizer=self.opt.torch_state_dict(self.sys.torch_param_order(), by_name), scheduler=sched, epoch=epoch, f1=f1)
        return ck


def make_loaders(args, train_ds, val_ds, rank, world):
    lengths = None if args.no_bucketing or not hasattr(train_ds, "lengths") else train_ds.lengths()
    sampler = ShardedBucketBatchSampler(lengths if lengths is not None else len(train_ds), args.batch_size, world, rank, seed=args.seed)
    train_loader = DataLoader(train_ds, batch_sampler=sampler, collate_fn=collate_fn)
    val_loader = DataLoader(val_ds, batch_size=args.batch_size, shuffle=False, collate_fn=collate_fn)
    return sampler, train_loader, val_loader


def run(args, engine, sampler, train_loader, val_loader, rank=0, world=1, log=print):
    """The epoch loop of ref train.py:123-263 over an engine (HipEngine in production; the data-parallel CPU tests drive the
    same loop with a stand-in engine).  Returns the last validation F1."""
    f1 = 0.0
    for epoch in range(engine.start_epoch, args.epochs):
        sampler.set_epoch(epoch)
        engine.begin_epoch(epoch)
        loss = None
        for step, (audio_list, text_list, labels) in enumerate(train_loader):
            loss = engine.train_step(audio_list, text_list, labels, epoch, step)
        engine.end_epoch_health(loss)
        engine.begin_eval()
        preds, gold, feats = [], [], []
        last = epoch == args.epochs - 1
        for audio_list, text_list, labels in val_loader:
            p, f = engine.predict(audio_list, text_list, want_features=last)
            preds.append(p)
            gold.append(labels)
            if f is not None:
                feats.append(f)
        f1 = weighted_f1(torch.cat(preds), torch.cat(gold))
        if rank == 0:
            log(f"Epoch {epoch} F1: {f1}")
        if last and feats:
            engine.fit_weibull(torch.cat(feats), torch.cat(gold))
        if rank == 0:
            os.makedirs(args.save_dir, exist_ok=True)
            torch.save(engine.checkpoint(epoch, f1), os.path.join(args.save_dir, f'epoch_{epoch}_f1_{f1:.4f}.pt'))
    return f1


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.use_amp:
        args.precision = 'bf16'
    if not torch.cuda.is_available():
        raise SystemExit("the HIP hot path needs an MI355X; there is no CPU fallback")
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", 1), ("RANK", 0), ("LOCAL_RANK", 0)))
    if os.environ.get("SER_SINGLE_DEVICE"):      # rehearsal on a one-GPU box: every rank shares GPU 0 (gloo backend)
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("SER_DIST_BACKEND", "nccl")      # "nccl" == RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    if args.synthetic:
        secs = [float(v) for v in args.synthetic_seconds.split(",")]
        train_ds = SyntheticSERDataset(args.synthetic, seconds=secs, num_labels=args.num_labels, seed=0)
        val_ds = SyntheticSERDataset(max(args.batch_size, args.synthetic // 4), seconds=secs, num_labels=args.num_labels, seed=1)
    else:
        train_ds, val_ds = SERDataset(args.train_manifest), SERDataset(args.val_manifest)
    sampler, train_loader, val_loader = make_loaders(args, train_ds, val_ds, rank, world)
    engine = HipEngine(args, device, rank, world, steps_per_epoch=len(sampler))
    f1 = run(args, engine, sampler, train_loader, val_loader, rank, world)
    if world > 1:
        torch.distributed.destroy_process_group()
    return f1


if __name__ == "__main__":
    main()
