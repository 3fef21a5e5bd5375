"""CPU training step built on the oracle (TEST / BASELINE INFRASTRUCTURE, not the product).

Used by `bench.py`'s `cpu_baseline` leg (kind "port": the build's own CPU restatement of the
reference path, timed on the GPU box's host cores) and by the parity report.  One step =
forward of the whole path (encoders frozen, no grad — the stance DESIGN.md records), the
reference's training loss, torch-autograd backward over the trainable head, and an AdamW update of
every grad-bearing tensor with the reference's group multipliers (train.py:72-83).
"""
import time

import torch

from . import ser_oracle as O

GROUPS = dict(audio_encoder=(0.1, .025), text_encoder=(0.1, .025), cross=(1.0, .05), pool_a=(1.0, .05), pool_t=(1.0, .05),
              fusion=(1.0, .05), prototypes=(1.0, .05))


def _classifier_group(name):
    if name.startswith("deep_classifier."):
        return 1.5, .06
    if name.startswith("anchor_clustering."):
        return 2.0, .04
    return 1.0, .05


class OracleTrainer:
    def __init__(self, sds, a_cfg, t_cfg, num_layers=35, heads=8, num_labels=4, lr=1e-4, dropout_seed=None,
                 p_cross=0.1, p_fusion=0.1, p_classifier=0.15, train_encoders=False):
        """dropout_seed: when given, every step applies the head's training-mode dropout with the masks of the build's
        generator (state = seed + step), i.e. the CPU baseline times the same train-mode step the HIP path runs."""
        self.a_cfg, self.t_cfg, self.L, self.heads, self.C, self.lr = a_cfg, t_cfg, num_layers, heads, num_labels, lr
        self.dropout_seed, self.p = dropout_seed, (p_cross, p_fusion, p_classifier)
        # BASELINE config 3 (reference `freeze_base=False`): the encoders receive gradients and AdamW updates in their
        # groups (lr x 0.1); their own training-mode noise (HF dropout / SpecAugment / LayerDrop) is not modelled
        self.train_encoders = train_encoders
        self.sds = {k: {n: v.detach().clone().float() if v.dtype.is_floating_point else v.clone() for n, v in sd.items()}
                    for k, sd in sds.items()}
        self.state = {}
        self.t = 0

    def _leafs(self):
        for k, sd in self.sds.items():
            for n, v in sd.items():
                train = v.dtype.is_floating_point and (self.train_encoders or not n.startswith("encoder.")) \
                    and not n.startswith("weibull") and n != "activation_vectors"
                v.requires_grad_(train)
                v.grad = None

    def forward(self, waves, ids, mask, use_openmax=False, training=True):
        with torch.no_grad():
            return O.full_forward(self.sds, waves, ids, mask, self.a_cfg, self.t_cfg, self.L, self.heads, use_openmax, training)

    def step(self, waves, ids, mask, labels):
        self._leafs()
        with torch.set_grad_enabled(self.train_encoders):   # frozen encoders unless config 3
            enc_a = O.wav2vec2_forward(O.sub(self.sds["audio_encoder"], "encoder."),
                                       torch.stack([O.normalise_waveform(w) for w in waves]), self.a_cfg)
            enc_t = O.xlmr_forward(O.sub(self.sds["text_encoder"], "encoder."), ids, mask, self.t_cfg)
        a_seq = O.adapter(enc_a, O.sub(self.sds["audio_encoder"], "adapter."))
        t_seq = O.adapter(enc_t, O.sub(self.sds["text_encoder"], "adapter."))
        a_mask, t_mask = torch.ones(a_seq.shape[:2]), mask.float()
        drop = None
        if self.dropout_seed is not None:
            drop = O.DropoutPlan(self.dropout_seed + self.t + 1, *self.p)
        a_enh, t_enh = O.cross_attention_forward(self.sds["cross"], a_seq, t_seq, a_mask, t_mask, self.heads, drop)
        fused = O.fusion_forward(self.sds["fusion"], O.pooling_forward(self.sds["pool_a"], a_enh, a_mask),
                                 O.pooling_forward(self.sds["pool_t"], t_enh, t_mask), drop)
        logits, unc, _, _ = O.classifier_forward(self.sds["classifier"], fused, self.L, False, True, drop)
        loss = O.train_loss(logits, unc, fused, self.sds["prototypes"]["prototypes"], labels, self.C)
        loss.backward()
        self.t += 1
        with torch.no_grad():
            for k, sd in self.sds.items():
                for n, v in sd.items():
                    if v.grad is None:
                        continue
                    mult, wd = _classifier_group(n) if k == "classifier" else GROUPS[k]
                    m, s = self.state.setdefault((k, n), (torch.zeros_like(v), torch.zeros_like(v)))
                    p, m2, s2 = O.adamw_step(v, v.grad, m, s, self.t, self.lr * mult, wd)
                    v.copy_(p); m.copy_(m2); s.copy_(s2)
        return loss.item(), logits.detach()


def time_steps(trainer, waves, ids, mask, labels, warmup=2, steps=5, budget_s=None):
    """Per-step wall times (seconds) of `steps` training steps after `warmup` untimed ones (BASELINE.md section 2:
    2 + >= 5, report the median).  With `budget_s` the timed loop stops early once that much time has been spent,
    but never before three steps."""
    t_all = time.perf_counter()
    for _ in range(warmup):
        trainer.step(waves, ids, mask, labels)
    out = []
    for _ in range(steps):
        t0 = time.perf_counter()
        trainer.step(waves, ids, mask, labels)
        out.append(time.perf_counter() - t0)
        if budget_s is not None and len(out) >= 3 and time.perf_counter() - t_all > budget_s:
            break
    return out
