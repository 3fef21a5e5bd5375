"""The hot path assembled: the seven modules of ref src/train.py:54-69 and its training / inference
step (:145-177, :181-201), plus the data-parallel gradient reduction the reference does not have.

`SERSystem.train_step` is what `bench.py` times and what `train.py` loops over.  With
`use_graph=True` the forward+backward and the optimizer launches are captured into two hipGraphs
(graph replay removes the per-kernel host launch cost of the ~2k small head kernels); the gradient
all-reduce runs between them.
"""
import contextlib
import math

import torch
import torch.nn as nn

from .models import AudioEncoder, TextEncoder, FusionLayer
from .models.classifier import AdvancedOpenMaxClassifier
from .models.cross_attention import CrossModalAttention
from .models.pooling import AttentiveStatsPooling
from .models.losses import TrainLoss
from .models.prototypes import PrototypeMemory
from . import _ops
from .optim import FlatAdamW, WarmupCosine


class SERSystem(nn.Module):
    def __init__(self, audio_encoder, text_encoder, num_labels=4, shared_dim=256, num_heads=8, proj_dim=512,
                 num_layers=35, base_dim=512, dropout=0.15):
        super().__init__()
        self.audio_encoder, self.text_encoder = audio_encoder, text_encoder
        ah, th = audio_encoder.encoder.config.hidden_size, text_encoder.encoder.config.hidden_size
        self.cross = CrossModalAttention(ah, th, shared_dim=shared_dim, num_heads=num_heads)
        self.pool_a = AttentiveStatsPooling(ah)
        self.pool_t = AttentiveStatsPooling(th)
        self.fusion = FusionLayer(ah * 2, th * 2, proj_dim)
        self.classifier = AdvancedOpenMaxClassifier(input_dim=proj_dim, num_labels=num_labels, num_layers=num_layers,
                                                    base_dim=base_dim, dropout=dropout)
        self.prototypes = PrototypeMemory(num_labels, proj_dim)
        self.criterion = TrainLoss(num_labels)
        self.num_labels = num_labels
        self._graph = None
        self._side = None
        # `bf16` precision mode: the backward token-level head GEMMs round their operands to bf16 (one MFMA product
        # per multiply, fp32 accumulation), as the frozen encoders do; `bf16x3` keeps the fp32-equivalent 3-product
        # form everywhere.  Forward products of the head always use the 3-product form (1e-3 logit budget).
        self.head_backward_products = 1 if getattr(audio_encoder, "precision", "bf16x3") == "bf16" else 3
        # Dropout of the trainable head in training mode (cross-attention, fusion, classifier: the nn.Dropout layers of the
        # reference).  The masks come from a counter-based generator keyed by (state, layer, element); `state` is a device
        # word advanced once per training step (inside the captured graph), so every replay draws new masks.  Parity with
        # the reference's golden vectors is defined with dropout off (`train_dropout = False`, or `.eval()`).
        self.train_dropout = True
        self.dropout_seed = 0x5EED
        self._drop_state = None
        nl = num_layers
        self.cross._drop_sites = (1, 2, 3, 4)
        self.fusion._drop_sites = (5, 6)
        self.classifier._drop_sites = (7, 8, 9, 16)               # 16 .. 16 + 2 * num_layers - 1: the residual blocks
        assert 16 + 2 * nl < 1 << 16

    # ---- reference checkpoint layout (train.py:249-262) ---------------------------------------------------------
    CKPT_KEYS = ("audio_encoder", "text_encoder", "cross", "pool_a", "pool_t", "fusion", "classifier", "prototypes")

    def checkpoint_dict(self):
        return {k: getattr(self, k).state_dict() for k in self.CKPT_KEYS}

    # parameters of the reference's default AudioEncoder() (quality gates / audio conditioning on, ref audio_encoder.py:25-52)
    # that a module built with the gates off does not have
    GATE_PREFIXES = ("quality_gates.", "quality_fusion.", "audio_conditioning.", "conditioning_fusion.", "combined_fusion.")

    def load_checkpoint_dict(self, ck, log=None):
        """Load the eight module entries of a checkpoint in the reference's layout (ref train.py:249-262), written by
        this package or by the reference itself.  Strict, with one explicit exception: a reference checkpoint carries the
        gate / conditioning parameters of its default AudioEncoder(); when this system's audio encoder was built without
        them, exactly those keys (GATE_PREFIXES) are skipped and reported."""
        for k in self.CKPT_KEYS:
            mod, sd = getattr(self, k), ck[k]
            if k == "audio_encoder":
                own = set(mod.state_dict().keys())
                skipped = [n for n in sd if n not in own and n.startswith(self.GATE_PREFIXES)]
                if skipped:
                    sd = {n: v for n, v in sd.items() if n not in skipped}
                    (log or print)(f"load_checkpoint_dict: audio encoder built without quality gates / conditioning; "
                                   f"skipping {len(skipped)} checkpoint entries ({skipped[0]} ...)")
            mod.load_state_dict(sd)

    # ---- optimizer state interchange with torch.optim.AdamW (the reference's optimizer, ref train.py:72-83) --------------
    OPT_GROUPS = (("audio_encoder", ""), ("text_encoder", ""), ("cross", ""), ("pool_a", ""), ("pool_t", ""), ("fusion", ""),
                  ("classifier", "deep_classifier."), ("classifier", "anchor_clustering."), ("classifier", "uncertainty_head."),
                  ("prototypes", ""))
    BUFFER_KEYS = ("weibull_alpha", "weibull_beta", "weibull_tau", "activation_vectors")

    def torch_param_order(self, ck=None):
        """[(module key, parameter name)] in the index order of the reference's AdamW: its ten groups in order, inside a
        group the registration order of the parameters, which is the order of the parameter keys of the module's state
        dict.  With `ck` the order is read from the checkpoint's own state dicts (so parameters this build does not
        hold, e.g. the gate modules, keep their indices); without it, from this system's modules."""
        order = []
        for key, prefix in self.OPT_GROUPS:
            names = list(ck[key].keys()) if ck is not None else [n for n, _ in getattr(self, key).named_parameters()]
            for n in names:
                if n.startswith(prefix) and not any(n.endswith(b) for b in self.BUFFER_KEYS):
                    order.append((key, n))
        return order

    def make_optimizer(self, lr=1e-4):
        """The ten AdamW parameter groups of ref train.py:72-83."""
        c = self.classifier
        return FlatAdamW([
            {'params': self.audio_encoder.parameters(), 'lr': lr * 0.1, 'weight_decay': 0.025},
            {'params': self.text_encoder.parameters(), 'lr': lr * 0.1, 'weight_decay': 0.025},
            {'params': self.cross.parameters(), 'lr': lr, 'weight_decay': 0.05},
            {'params': self.pool_a.parameters(), 'lr': lr, 'weight_decay': 0.05},
            {'params': self.pool_t.parameters(), 'lr': lr, 'weight_decay': 0.05},
            {'params': self.fusion.parameters(), 'lr': lr, 'weight_decay': 0.05},
            {'params': c.deep_classifier.parameters(), 'lr': lr * 1.5, 'weight_decay': 0.06},
            {'params': c.anchor_clustering.parameters(), 'lr': lr * 2.0, 'weight_decay': 0.04},
            {'params': c.uncertainty_head.parameters(), 'lr': lr * 1.0, 'weight_decay': 0.05},
            {'params': self.prototypes.parameters(), 'lr': lr, 'weight_decay': 0.05},
        ], lr=lr, weight_decay=0.05)

    # ---- forward pieces ------------------------------------------------------------------------------------------
    def encode(self, wave, ids, attn_mask, lid=None):
        """Frozen encoders (one paired call) + the two trainable adapters (independent: the text one on a second
        stream).  With the audio encoder's gate flags on (the reference's default AudioEncoder()), the front end runs first
        and the projected quality / conditioning features are fused into the audio sequence (ref audio_encoder.py:65-132)."""
        from .models.adapter import adapter_apply
        self.prepare()
        noisy = self.training and (getattr(self.audio_encoder, "encoder_train_noise", False) or getattr(self.text_encoder, "encoder_train_noise", False))
        if noisy or not (self.audio_encoder.freeze_base and self.text_encoder.freeze_base):
            # BASELINE config 3 (reference freeze_base=False): the fine-tuning form of the encoders, with gradients; also the
            # frozen encoders when their training-mode noise is requested (forward only, same operators)
            # the two encoders are independent until the cross-attention: the text encoder (a few hundred small launches) runs on a
            # second stream beside the audio encoder; autograd runs each backward node on the stream of its forward, so the
            # backward passes overlap the same way
            ids_d, mask_d = ids.to(wave.device), attn_mask.to(wave.device)
            if getattr(self, "_enc_side", None) is None:
                self._enc_side = torch.cuda.Stream()
            with _ops.fork(self._enc_side) as f:
                t_seq, t_mask = self.text_encoder.forward_ids(ids_d, mask_d)
            a_seq = self.audio_encoder.encode(wave.to(torch.float32))
            f.join(produced=[t_seq, t_mask], consumed=[ids_d, mask_d])
            a_mask = torch.ones(a_seq.shape[0], a_seq.shape[1], dtype=torch.float32, device=a_seq.device)
            return a_seq, a_mask, t_seq, t_mask
        q_raw = c_raw = None
        if self.gates_on():
            wave, q_raw, c_raw = self.front_end(wave, lid)
        a_enc, t_enc = self.encode_frozen(wave, ids.to(wave.device), attn_mask.to(wave.device))
        a_seq, t_seq = self._adapters(a_enc, t_enc)
        if q_raw is not None or c_raw is not None:
            a_seq = self.audio_encoder.fuse_gate_features(a_seq, q_raw, c_raw)
        a_mask = self._ones_mask(a_seq)
        return a_seq, a_mask, t_seq, attn_mask.to(device=a_seq.device, dtype=torch.float32)

    # ---- the encoders' training-mode noise under a captured step (models/_finetune.py Noise) -------------------------
    def encoder_noises(self):
        """[(encoder module, its Noise)] for the encoders whose training-mode noise is on (created on first use)."""
        from .models._finetune import Noise
        out = []
        for idx, m in enumerate((self.audio_encoder, self.text_encoder)):
            if self.training and getattr(m, "encoder_train_noise", False):
                if getattr(m, "_noise", None) is None:
                    m._noise = Noise(m.encoder.config, idx, seed=getattr(m, "noise_seed", 0))
                out.append((m, m._noise))
        return out

    def stage_encoder_noise(self, wave_shape, ids_shape, device):
        """Draw this batch's LayerDrop / SpecAugment decisions on the host (same generator, same order as the eager forward) and
        stage them as device words for a captured forward.  Call once per step, before the replay."""
        from .models._finetune import wav2vec2_frames
        for m, noise in self.encoder_noises():
            cfg = m.encoder.config
            if noise.enc == 0:
                noise.plan(cfg.num_hidden_layers, int(wave_shape[0]), wav2vec2_frames(cfg, wave_shape[1]))
            else:
                noise.plan(cfg.num_hidden_layers, int(ids_shape[0]), int(ids_shape[1]))
            noise.stage(device)

    def layerdrop_gates(self):
        """{id(parameter): device word} for FlatAdamW.set_gates: the parameters of transformer layer l of an encoder with
        LayerDrop follow `skip_dev[l]` (a dropped layer's parameters take no update, as with torch's `grad is None`)."""
        import re
        gates = {}
        for m, noise in self.encoder_noises():
            if noise.layerdrop <= 0 or noise.skip_dev is None or getattr(m, "freeze_base", True):
                continue
            for name, p_ in m.encoder.named_parameters():
                mt = re.match(r"encoder\.layers?\.(\d+)\.", name)
                if mt:
                    gates[id(p_)] = noise.skip_dev[int(mt.group(1)):int(mt.group(1)) + 1]
        return gates

    def _ones_mask(self, seq):
        """[B, S] of ones (HF returns no attention mask for wav2vec2-base: ref audio_encoder.py:162-163), cached per shape so a
        step does not launch a fill kernel for it."""
        key = (seq.shape[0], seq.shape[1], str(seq.device))
        cache = self.__dict__.setdefault("_mask_cache", {})
        if key not in cache:          # one tensor per shape, kept: a captured graph of another shape still reads its own
            cache[key] = torch.ones(seq.shape[0], seq.shape[1], dtype=torch.float32, device=seq.device)
        return cache[key]

    def _adapters(self, a_enc, t_enc):
        """Both residual adapters (independent): the text one on the side stream, or - GROUPED_HEAD - one grouped launch per level."""
        from .models.adapter import adapter_apply, adapters_apply
        if _ops.GROUPED_HEAD:
            return adapters_apply(self.audio_encoder, self.text_encoder, a_enc, t_enc)
        with _ops.fork(self._side_stream()) as f:
            t_seq = adapter_apply(self.text_encoder, t_enc)
        a_seq = adapter_apply(self.audio_encoder, a_enc)
        f.join(produced=[t_seq], consumed=[t_enc])
        return a_seq, t_seq

    def _side_stream(self):
        if self._side is None:
            self._side = torch.cuda.Stream()
        return self._side

    # ---- the reference's default front end (quality gates + audio conditioning, ref audio_encoder.py:65-132) ----------
    def gates_on(self):
        ae = self.audio_encoder
        return bool(getattr(ae, "use_quality_gates", False) or getattr(ae, "use_audio_conditioning", False))

    def language_features(self, texts, n):
        """[n, 2] host tensor for `front_end` (ref quality_gates.py:252-301,514-517); texts=None: the no-transcript branch."""
        from .models.frontend import language_features
        qg = getattr(self.audio_encoder, "quality_gates", None)
        return language_features(texts, n, getattr(qg, "language_detector", None), getattr(qg, "enable_language_detection", True))

    @torch.no_grad()
    def front_end(self, wave, lid=None):
        """Quality gates -> clips that are not accepted become silence -> audio conditioning, for a whole batch of
        equal-length clips on the device (ref audio_encoder.py:67-86; ~20 launches, no host decision, capturable):
        wave [B,T], lid [B,2] (language_features; default = no transcript) -> (wave for the encoder, q_raw [B,8] | None,
        c_raw [B,12] | None)."""
        ae = self.audio_encoder
        q_raw = c_raw = dec = None
        wave = wave.to(torch.float32)
        if ae.use_quality_gates:
            if lid is None:
                lid = self.language_features(None, wave.shape[0])
            qg = ae.quality_gates
            q_raw, _, dec = _ops.quality_gates(wave, lid.to(wave.device, non_blocking=True), qg.pad_mode, qg.sample_rate)
            if not ae.use_audio_conditioning:
                wave = wave * (dec == 2).to(wave.dtype)[:, None]                     # ref :74-77
        if ae.use_audio_conditioning:
            wave, c_raw, _ = _ops.audio_conditioning(wave, dec, ae.audio_conditioning.sample_rate)   # not accepted -> silence
        return wave, q_raw, c_raw

    @torch.no_grad()
    def encode_frozen_gated(self, wave, ids, attn_mask, lid=None, slot=0):
        """front_end + the frozen encoders -> (a_enc, t_enc, q_raw, c_raw); the raw features are fused into the audio
        sequence by the trainable gate modules in the head (`loss_from_encoded(gate_features=...)`)."""
        wave, q_raw, c_raw = self.front_end(wave, lid)
        a_enc, t_enc = self.encode_frozen(wave, ids, attn_mask, slot)
        return a_enc, t_enc, q_raw, c_raw

    @torch.no_grad()
    def encode_frozen(self, wave, ids, attn_mask, slot=0):
        """Only the frozen part (no adapters): Wav2Vec2 and XLM-R forward -> (a_enc, t_enc).  One C call walks both
        models; with equal depth their layers share launches (ser_encoders_forward)."""
        from ._engines import forward_pair
        return forward_pair(self.audio_encoder.engine(), self.text_encoder.engine(), wave, ids, attn_mask, slot)

    def _set_precision(self):
        from . import _lib as L
        L.check(L.lib.ser_set_head_backward_products(self.head_backward_products), "ser_set_head_backward_products")

    def _dropout_scope(self):
        """Context for the training forward: dropout on (and the generator state advanced by one step) when the system
        is in training mode with `train_dropout`; otherwise the identity."""
        if not (self.training and self.train_dropout):
            return _ops.dropout_scope(None)
        self._dropout_word().add_(1)                              # one launch; captured with the step, so replays advance it
        return _ops.dropout_scope(self._drop_state)

    def _dropout_word(self):
        """The device word of the dropout generator (None when training-mode dropout is off)."""
        if not (self.training and self.train_dropout):
            return None
        dev = next(self.classifier.parameters()).device
        if self._drop_state is None or self._drop_state.device != dev:
            self._drop_state = torch.full((1,), int(self.dropout_seed), dtype=torch.int64, device=dev)
        return self._drop_state

    def loss_from_encoded(self, a_enc, t_enc, attn_mask, labels, use_proto=True, split=False, gate_features=None):
        """Everything trainable: adapters -> cross-attention -> pooling -> fusion -> classifier -> loss.
        split=True: the classifier and the loss hang off a detached copy of `fused`; returns (loss, logits, fused, leaf),
        so that the caller can run the classifier's backward first (loss.backward(): its gradient bucket is then
        complete and can travel over xGMI) and the rest afterwards (fused.backward(leaf.grad))."""
        from .models.adapter import adapter_apply
        self.prepare()
        self._set_precision()
        with self._dropout_scope():
            a_seq, t_seq = self._adapters(a_enc, t_enc)
            if gate_features is not None and (gate_features[0] is not None or gate_features[1] is not None):
                a_seq = self.audio_encoder.fuse_gate_features(a_seq, gate_features[0], gate_features[1])
            a_mask = self._ones_mask(a_seq)
            fused = self.head(a_seq, a_mask, t_seq, attn_mask.to(torch.float32))
            leaf = fused.detach().requires_grad_() if split else fused
            logits, unc, _ = self.classifier(leaf, use_openmax=False, return_uncertainty=True)
        total = self.criterion(logits, unc, leaf, self.prototypes.prototypes, labels, use_proto=use_proto)
        if split:
            return total, logits, fused, leaf
        return total, logits

    def head(self, a_seq, a_mask, t_seq, t_mask):
        self.prepare()
        a_enh, t_enh = self.cross(a_seq, t_seq, a_mask, t_mask)
        # the two poolings are independent: text pooling (forward, and therefore its backward, which autograd runs
        # on the forward's stream) goes to the side stream
        if _ops.GROUPED_HEAD:
            from .models.pooling import pools_apply
            a_vec, t_vec = pools_apply(self.pool_a, self.pool_t, a_enh, a_mask, t_enh, t_mask)      # grouped launches, one stream
            return self.fusion(a_vec, t_vec)
        # the two poolings are independent: text pooling (forward, and therefore its backward, which autograd runs on the
        # forward's stream) goes to the side stream
        with _ops.fork(self._side_stream()) as f:
            t_vec = self.pool_t(t_enh, t_mask)
        a_vec = self.pool_a(a_enh, a_mask)
        f.join(produced=[t_vec], consumed=[t_enh, t_mask])
        return self.fusion(a_vec, t_vec)

    def forward(self, wave, ids, attn_mask, use_openmax=True, lid=None):
        fused = self.head(*self.encode(wave, ids, attn_mask, lid))
        return self.classifier(fused, use_openmax=use_openmax)

    def loss(self, wave, ids, attn_mask, labels, use_proto=True, lid=None):
        self._set_precision()
        with self._dropout_scope():
            fused = self.head(*self.encode(wave, ids, attn_mask, lid))
            logits, unc, anchor = self.classifier(fused, use_openmax=False, return_uncertainty=True)
        total = self.criterion(logits, unc, fused, self.prototypes.prototypes, labels, use_proto=use_proto)
        return total, logits

    def check_persistent_kernels(self):
        """Raise if a bounded hand-off wait inside the persistent classifier kernels was ever abandoned (sticky word 1 of
        their scratch areas, csrc/persist.hip): from then on the stack's outputs are not trustworthy."""
        cache = getattr(self.classifier, "_stack_cache", None)
        if cache is not None and any(int(sc[1]) != 0 for sc in cache[3]):
            raise RuntimeError("a hand-off wait inside the persistent classifier kernels was abandoned")

    # ---- data parallel -------------------------------------------------------------------------------------------
    def buckets(self):
        """Flat gradient buckets in the order backward completes them (classifier first)."""
        out = []
        for m in (self.classifier, self.fusion, self.pool_a, self.pool_t, self.cross):
            out.append(m._flat)
        gate = getattr(self.audio_encoder, "_gate_flat", None)      # quality / conditioning projections + their fusion Linear
        if gate is not None:
            out.append(gate)
        for m in (self.audio_encoder, self.text_encoder):
            if hasattr(m, "_adapter_flat"):
                out.append(m._adapter_flat)
        return out

    def prepare(self):
        """Flatten every trainable bucket NOW, on the current stream, before the forward forks onto side streams.
        A bucket that is first flattened in the middle of a forward is built on whatever stream is current there (the
        text adapter and the text pooling run on the side stream), which is how a freed parameter block came to be reused
        by the main stream under a still-queued copy (models/_flat.py).  Cheap when nothing moved
        (pointer comparisons); called at the top of every forward entry."""
        for b in self.buckets():
            b.ensure()


class GradReducer:
    """Data-parallel mean of the flat gradient buckets over RCCL (torch.distributed backend 'nccl').

    Each bucket is reduced with ONE all-reduce, issued on a side stream the moment the module's
    backward has written it (`grad_ready_hook`), so the 76 MB classifier bucket travels over xGMI
    while fusion / pooling / cross-attention / adapter backward still run.  `finish()` makes the
    compute stream wait for the collectives before the optimizer reads the buckets.
    """

    def __init__(self, system, process_group=None, overlap=True):
        import torch.distributed as dist
        self.dist, self.pg = dist, process_group
        self.world = dist.get_world_size(process_group)
        self.system, self.overlap = system, overlap
        self.loose = [p for p in system.prototypes.parameters()]
        # fine-tuning (BASELINE config 3): the encoders' own parameters get gradients from autograd; they are reduced as
        # ONE coalesced buffer per encoder in finish()
        self.coalesced = [[p for p in m.encoder.parameters() if p.requires_grad]
                          for m in (system.audio_encoder, system.text_encoder) if not getattr(m, "freeze_base", True)] \
            if hasattr(system, "audio_encoder") else []
        self.cuda = torch.cuda.is_available() and next(system.parameters()).is_cuda
        self.stream = torch.cuda.Stream() if self.cuda else None
        self.pending = []
        self._seq, self._next, self._ready = None, 0, set()
        self.hold = False
        # issuing a collective while a compute graph replays is the point of the overlap; it is only done on RCCL
        # (gloo stages device tensors through the host and serialises against the replay) unless forced by env
        import os
        self.early = os.environ.get("SER_DP_EARLY", "1" if dist.get_backend(process_group) == "nccl" else "0") == "1"

    def _reduce(self, t):
        # SUM over ranks; the 1/world scale is applied by the library's own axpby kernel in finish()
        return self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.pg, async_op=True)

    def _issue(self, t):
        if self.cuda:
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                self.pending.append((self._reduce(t), t))
        else:
            self.pending.append((self._reduce(t), t))

    # The collectives of one step are issued in ONE fixed sequence on every rank, whatever path a rank's step takes
    # (eager with per-bucket hooks on a ragged batch, a captured graph on an equal-length one, the two-piece graph):
    # the buckets in `system.buckets()` order, then the loose parameters, then the fine-tuning buffers.  A hook only
    # marks its bucket ready; a bucket is issued once every bucket before it has been.  (Issuing in hook order would let
    # two ranks pair different buckets of equal size - pool_a with pool_t - in one all-reduce.)
    def _advance(self, upto=None):
        seq = self._seq
        while self._next < len(seq) and (upto is None or self._next < upto) and (upto is not None or id(seq[self._next]) in self._ready):
            b = seq[self._next]
            self._next += 1
            if b.gflat is not None:
                self._issue(b.gflat)

    def _hook(self, bucket):
        # never from inside a graph capture or its eager warm-up (`hold`): a collective issued there would reduce
        # gradients the captured replay is about to overwrite
        if self.hold or (self.cuda and torch.cuda.is_current_stream_capturing()):
            return
        self._ready.add(id(bucket))
        self._advance()

    def start(self, buckets, loose=()):
        """Issue the all-reduce of the given (already complete) buckets now - as far as the fixed sequence allows -
        without waiting for them.  (`loose` parameters always travel in finish(), after the buckets.)"""
        if self._seq is None:
            self._begin()
        for b in buckets:
            self._ready.add(id(b))
        self._advance()

    def _begin(self):
        self._seq = list(self.system.buckets())
        self._next = 0
        self._ready = set()

    def arm(self, overlap=None):
        """Eager stepping only: fire each bucket's all-reduce from its module's backward.  The hooks live for ONE step —
        finish() removes them — so a later graph capture / replay on the same modules never runs them.
        overlap=False: no hooks, everything travels in finish() - REQUIRED when a module's backward runs more than once
        per step (a ragged batch encodes each clip length separately, so the audio adapter publishes its bucket once per
        length group: the first publish would send a partial gradient)."""
        self._begin()
        if self.overlap if overlap is None else overlap:
            for b in self.system.buckets():
                b.grad_ready_hook = self._hook

    def disarm(self):
        for b in self.system.buckets():
            b.grad_ready_hook = None

    def quiet(self):
        """Context for graph capture and its eager warm-up passes: hooks removed, nothing is reduced, nothing marked done."""
        import contextlib

        @contextlib.contextmanager
        def scope():
            prev, self.hold = self.hold, True
            self.disarm()
            try:
                yield
            finally:
                self.hold = prev
                self._seq = None
                self.pending.clear()
        return scope()

    def finish(self):
        if self._seq is None:
            self._begin()
        self._advance(upto=len(self._seq))    # every bucket the hooks / start() have not issued yet, in sequence
        for p in self.loose:
            if p.grad is not None:
                self._issue(p.grad)
        back = []
        for group in self.coalesced:
            live = [p for p in group if p.grad is not None]
            if live:
                flat = torch.cat([p.grad.reshape(-1) for p in live])
                self._issue(flat)
                back.append((flat, live))
        inv = 1.0 / self.world
        if self.cuda:
            from . import _ops as O
            with torch.cuda.stream(self.stream):
                for w, t in self.pending:
                    w.wait()                      # orders the side stream after the collective
                    O.axpby(t, t, inv, 0.0)       # mean = sum / world, on the device
            torch.cuda.current_stream().wait_stream(self.stream)
        else:
            for w, t in self.pending:
                w.wait()
                t.mul_(inv)                       # CPU tensors only occur in the gloo unit tests
        for flat, live in back:                   # scaled means back into the per-parameter gradients
            off = 0
            for p in live:
                n = p.grad.numel()
                p.grad.copy_(flat[off:off + n].view_as(p.grad))
                off += n
        self.pending.clear()
        self._seq = None
        self.disarm()


class TrainStepper:
    """One optimizer step of ref train.py:145-177 on static device buffers, optionally graph-captured.

    Graph mode captures the step as hipGraphs.  With a gradient reducer the backward is captured in two pieces —
    A: forward + loss + classifier backward, B: fusion / pooling / cross-attention / adapter backward — so that the
    76 MB classifier bucket (3/4 of all gradient bytes) is all-reduced over xGMI while graph B runs; the remaining
    buckets follow, then the optimizer graph.
    """

    def __init__(self, system, optimizer, scheduler=None, reducer=None, use_graph=False, use_proto=True, split_backward=None):
        self.sys, self.opt, self.sched, self.reducer = system, optimizer, scheduler, reducer
        self.use_graph, self.use_proto = use_graph, use_proto
        # opt-in: the single-graph path is the one rehearsed end to end (tests/test_gpu_dp.py, 2-rank bench rehearsal);
        # the two-piece capture is bit-identical (tests/test_gpu_system.py) but its comm overlap cannot be measured on a
        # one-GPU box, so it is not the default
        self.split = False if split_backward is None else split_backward
        self.g_fb = self.g_b = self.g_opt = None
        self.static = None
        self.loss = None
        # one captured graph set per input shape (waveform length, token count, batch size): real manifests produce a few
        # distinct shapes (length buckets, the last partial batch); beyond `max_graphs` a new shape runs eagerly
        self._graphs = {}
        self.max_graphs = 8
        # Full fine-tune on one GPU: the text encoder's backward (on its own stream) ends well before the audio encoder's, so its
        # 278 M parameters take their AdamW update right there - from a hook behind the last of their gradients - instead of
        # at the end of the step behind everything else (1.5 ms of HBM-bound work off the critical chain)
        self._early = None
        te = self.sys.text_encoder
        if reducer is None and not getattr(te, "freeze_base", True) and hasattr(torch.Tensor, "register_post_accumulate_grad_hook"):
            params = [p_ for p_ in te.encoder.parameters() if p_.requires_grad]
            self._early = dict(ids={id(p_) for p_ in params}, n=None, seen=0, armed=False, done=False)
            for p_ in params:
                p_.register_post_accumulate_grad_hook(self._early_hook)

    def _early_hook(self, p_):
        e = self._early
        e["seen"] += 1
        if e["armed"] and e["n"] is not None and e["seen"] == e["n"]:
            self.opt.launch(only=e["ids"])              # on the stream the text encoder's backward runs on
            e["done"] = True

    def _early_begin(self, armed):
        """Before a backward pass: count the gradient hooks; armed: the last one launches the text encoder's update."""
        e = self._early
        if e is not None:
            e["seen"], e["armed"], e["done"] = 0, bool(armed), False

    def _early_end(self):
        """After a backward pass: remember how many text-encoder parameters receive a gradient (the pooler does not); returns the
        ids the regular optimizer launch must skip."""
        e = self._early
        if e is None:
            return None
        if e["n"] is None and e["seen"] > 0:
            e["n"] = e["seen"]
        e["armed"] = False
        return e["ids"] if e["done"] else None

    def _fwd_bwd(self, wave, ids, mask, labels, lid=None):
        loss, logits = self.sys.loss(wave, ids, mask, labels, self.use_proto, lid=lid)
        # eager data parallelism reduces a bucket from its module's backward hook, so gradients cannot be deferred there
        with _ops.defer_wgrads(self.use_graph or self.reducer is None), _ops.unit_loss_grad():
            loss.backward()
        _ops.wgrad_join()
        return loss.detach(), logits.detach()

    # ---- split form: the classifier + loss hang off a detached copy of `fused`
    def _fwd_bwd_a(self, wave, ids, mask, labels, lid=None):
        s = self.sys
        s._set_precision()
        with s._dropout_scope():              # the same precision mode and training-mode dropout as SERSystem.loss
            fused = s.head(*s.encode(wave, ids, mask, lid))
            leaf = fused.detach().requires_grad_()
            logits, unc, _ = s.classifier(leaf, use_openmax=False, return_uncertainty=True)
        loss = s.criterion(logits, unc, leaf, s.prototypes.prototypes, labels, use_proto=self.use_proto)
        with _ops.defer_wgrads(), _ops.unit_loss_grad():
            loss.backward()
        _ops.wgrad_join()
        self._fused, self._dfused = fused, leaf.grad
        return loss.detach(), logits.detach()

    def _bwd_b(self):
        with _ops.defer_wgrads():
            self._fused.backward(self._dfused)
        _ops.wgrad_join()
        self._fused = self._dfused = None

    def step(self, wave, ids, mask, labels, lid=None):
        """lid: [B, 2] language features for the quality gates (SERSystem.language_features) when the audio encoder's gate
        flags are on; default: the no-transcript branch."""
        dev = wave.device
        if self.sys.gates_on() and lid is None:
            lid = self.sys.language_features(None, wave.shape[0]).to(dev)
        if not self.use_graph:
            for _, n in self.sys.encoder_noises():                # host decisions, taken inside the forward
                n.static = False
            gates, self.opt.gates = self.opt.gates, {}            # a dropped layer's parameters simply have no gradient here
            self.opt.zero_grad(set_to_none=True)
            if self.reducer:
                self.reducer.arm()
            self.opt.prepare_step(dev)                            # before backward: the early update reads the step's scalars
            self._early_begin(True)
            self.loss, self.logits = self._fwd_bwd(wave, ids, mask, labels, lid)
            skip = self._early_end()
            if self.reducer:
                self.reducer.finish()
            self.opt.launch(skip=skip)
            self.opt.gates = gates
        else:
            key = (tuple(wave.shape), tuple(ids.shape))
            if key not in self._graphs:
                if len(self._graphs) >= self.max_graphs:      # too many shapes to keep a graph for each: eager step
                    self.use_graph = False
                    try:
                        return self.step(wave, ids, mask, labels, lid)
                    finally:
                        self.use_graph = True
                g_opt = self.g_opt
                self._capture(wave, ids, mask, labels, lid)
                if g_opt is not None:
                    self.g_opt = g_opt                          # the optimizer launch does not depend on the input shape
                self._graphs[key] = (self.static, self.g_fb, self.g_b, self.loss, self.logits)
            self.static, self.g_fb, self.g_b, self.loss, self.logits = self._graphs[key]
            for p_, g_ in getattr(self, "_loose_grads", ()):      # a caller's zero_grad(set_to_none=True) must not detach them
                p_.grad = g_
            for _, n in self.sys.encoder_noises():
                n.static = True
            self.sys.stage_encoder_noise(wave.shape, ids.shape, dev)   # this step's LayerDrop / SpecAugment draws (no-op without noise)
            for s, t in zip(self.static, (wave, ids, mask, labels) + ((lid,) if lid is not None else ())):
                s.copy_(t, non_blocking=True)
            self.opt.prepare_step(dev)                            # before the replay: the text encoder's early update sits inside it
            self.g_fb.replay()
            if self.split:
                if self.reducer and self.reducer.early:
                    self.reducer.start([self.sys.classifier._flat], [self.sys.prototypes.prototypes])
                self.g_b.replay()
            if self.reducer:
                self.reducer.finish()
            self.g_opt.replay()
        if self.sched:
            self.sched.step()
        return self.loss

    def _capture(self, wave, ids, mask, labels, lid=None):
        # a reducer armed by an earlier eager step (ragged batch) must not fire from the warm-up passes or the capture
        with (self.reducer.quiet() if self.reducer else contextlib.nullcontext()):
            self._capture_impl(wave, ids, mask, labels, lid)

    def _capture_impl(self, wave, ids, mask, labels, lid=None):
        dev = wave.device
        self.sys.prepare()
        noises = self.sys.encoder_noises()
        if not noises:
            return self._capture_body(wave, ids, mask, labels, lid)
        # the encoders' own noise: host decisions become device words (Noise.static).  The warm-up passes and the capture must
        # not use up draws, or a captured run would see other masks than eager stepping with the same seeds: the host generators
        # and the dropout word are put back afterwards
        drop = self.sys._dropout_word()
        saved = [n.state() for _, n in noises], (drop.clone() if drop is not None else None)
        for _, n in noises:
            n.static = True
        self.sys.stage_encoder_noise(wave.shape, ids.shape, dev)
        # before the optimizer launch is captured; not under data parallelism: ranks drop different layers, every replica applies
        # the reduced gradient (a rank's zero contribution included)
        self.opt.set_gates(self.sys.layerdrop_gates() if self.reducer is None else {})
        try:
            self._capture_body(wave, ids, mask, labels, lid)
        finally:
            for (_, n), st in zip(noises, saved[0]):
                n.set_state(st)
            if drop is not None:
                drop.copy_(saved[1])

    def _capture_body(self, wave, ids, mask, labels, lid=None):
        dev = wave.device
        self.static = [wave.clone(), ids.clone(), mask.clone(), labels.clone()] + ([lid.clone()] if lid is not None else [])
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):       # warm-up outside capture: workspaces, flat buckets, optimizer state
            for _ in range(2):
                self.opt.zero_grad(set_to_none=True)
                self._early_begin(False)       # counts the text encoder's gradient hooks, launches nothing
                self._fwd_bwd(*self.static)
                self._early_end()
            self.opt.prepare_step(dev)
            self.opt.t -= 1                  # the warm-up must not count as a step
            if self.opt._plan is None:
                self.opt._build_plan()
            for grp, segs, loose in self.opt._plan:   # allocate m/v before capture
                for b, s, e in segs:
                    self.opt._mv(id(b), b.flat)
                for p in loose:
                    self.opt._mv(id(p), p.data)
        torch.cuda.current_stream().wait_stream(side)
        self.opt.zero_grad(set_to_none=True)
        # Parameters outside the flat buckets (the prototypes) get their gradient as a tensor of the captured graph.  The
        # optimizer graph is captured once and the reducer reads `p.grad`, so every later graph set (another input shape)
        # must write the SAME tensor: it is handed back to autograd, zeroed inside the capture and accumulated into.
        keep = getattr(self, "_loose_grads", None)
        if keep:
            for p_, g_ in keep:
                p_.grad = g_

        def zero_kept():
            for _, g_ in keep or ():
                g_.zero_()
        self.g_fb = torch.cuda.CUDAGraph()
        if self.split:
            with torch.cuda.graph(self.g_fb):
                zero_kept()
                self.loss, self.logits = self._fwd_bwd_a(*self.static)
            self.g_b = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_b, pool=self.g_fb.pool()):
                self._bwd_b()
        else:
            with torch.cuda.graph(self.g_fb):
                zero_kept()
                self._early_begin(True)        # (a post-accumulate hook sees the complete gradient, fresh or accumulated into)
                self.loss, self.logits = self._fwd_bwd(*self.static)
                skip = self._early_end()
            assert getattr(self, "_opt_skip", skip) == skip or not self._graphs, "every graph set must make the same early update"
            self._opt_skip = skip
        if not keep:
            self._loose_grads = [(p_, p_.grad) for grp, segs, loose in self.opt._plan for p_ in loose if p_.grad is not None]
        if self.g_opt is None or not self._graphs:
            self.g_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_opt):
                self.opt.launch(skip=getattr(self, "_opt_skip", None))


class PipelinedStepper:
    """Training steps with the frozen-encoder forward issued ahead of, and beside, the trainable part.

    With `freeze_base=True` (the reference default, BASELINE config 2) the encoder outputs of a batch do not depend on
    any parameter the optimizer touches, so computing them early changes nothing numerically
    (tests/test_gpu_system.py checks bit-identity with sequential stepping).  Two things are bought with that:

    * overlap - the head is ~150 small, latency-bound launches that occupy a handful of CUs each, while the encoder
      GEMMs fill the chip; on two streams they run side by side instead of back to back;
    * GEMM shape - ONE encoder pass covers `group` consecutive batches (rows of all of them in every launch).  At batch
      16 a layer GEMM is 3 696 rows = 174-700 tiles on 256 CUs behind a ~4 us launch floor and an ~8 us epilogue (26-38 %
      MFMA-pipe utilisation); at group 4 it is 14 784 rows and sits with the conv GEMMs at 45-59 %.  Rows are independent
      (every normalisation of the encoders is per clip, every tile configuration sums k in the same order), so a clip's
      features are bit-identical whatever it is batched with.

    Every `step` still executes one head forward / backward / AdamW update on a fresh batch, and every batch goes through
    exactly one encoder pass - nothing is cached or skipped, the encoder work of `group` steps is merely issued as one
    pass, `group`..`2 group` steps early.  Schedule (two slots, each = input staging + encoder outputs of one group):

        steps of epoch e :   head consumes slot e%2 (encoded during epoch e-1), one batch per step
                             encoder pass over slot (e+1)%2 runs on the encoder stream (its inputs were fed during e-1)
                             each step's new batch is staged into slot e%2's INPUT area (the pass that read it is complete)
        end of epoch e   :   slot e%2's input area is full again -> its encoder pass is issued (after the last head copy)

    `feed(batch)` stages a batch; `prime` = 2 * group feeds are needed before the first `step`; `step(next_batch)` trains on
    the oldest batch in flight and stages `next_batch`.  `drain()` trains on what is still in flight without new input.
    """

    def __init__(self, system, optimizer, scheduler=None, reducer=None, use_proto=True, split_backward=None, group=None, depth=None):
        self.sys, self.opt, self.sched, self.reducer, self.use_proto = system, optimizer, scheduler, reducer, use_proto
        if group is None:
            group = depth if depth is not None else 4       # `depth` (round 2: encoder passes in flight) is accepted as an alias
        assert group >= 1
        self.group = int(group)
        self.prime = 2 * self.group
        self.enc_stream = torch.cuda.Stream()
        self.g_encs = [None, None]
        self.enc_done = [torch.cuda.Event(), torch.cuda.Event()]
        self.g_head = self.g_head_b = self.g_opt = None
        # Data parallel: the head graph is captured in two pieces - A: forward + loss + classifier backward, B: the
        # backward of fusion / pooling / cross-attention / adapters - and the all-reduce of the classifier bucket (76 MB of
        # the 100 MB of gradients) is issued between them, so it travels over xGMI while B runs.  Same arithmetic as the
        # single graph (the two consumers of `fused` add their gradients in the detached leaf).
        self.split = (reducer is not None and reducer.early) if split_backward is None else bool(split_backward)
        self.loss = None
        self.fill_k, self.fill_n = 0, 0          # slot whose input area is being filled / batches staged in it
        self.launched = []                       # (slot, batches in it) whose encoder pass has been issued, not yet consumed (oldest first)
        self.cons_k, self.cons_j, self.cons_n = None, 0, 0       # slot the head is consuming / next batch of it / batches in it

    @property
    def g_enc(self):
        return self.g_encs[0]

    @property
    def pending(self):
        """Batches staged or encoded but not trained on yet."""
        n = self.fill_n + sum(nv for _, nv in self.launched)
        if self.cons_k is not None:
            n += self.cons_n - self.cons_j
        return n

    def _alloc(self, wave, ids, mask, labels):
        G = self.group
        rep = lambda t: t.repeat(*([G] + [1] * (t.dim() - 1))).contiguous()
        self.B = wave.shape[0]
        # per slot: input staging of a whole group (valid contents from the start: the capture warm-up encodes them)
        self.in_slots = [[rep(wave), rep(ids), rep(mask), rep(labels)] for _ in range(2)]
        self.cur_mask, self.cur_labels = mask.clone(), labels.clone()                # inputs of the head graph
        # the reference's default front end (gate flags of the audio encoder on): its 20 launches head the encoder graph; the
        # per-clip language features are one more staged input, the raw quality / conditioning features two more outputs
        self.gated = self.sys.gates_on()
        if self.gated:
            lid = self.sys.language_features(None, self.B).to(wave.device)
            for sl in self.in_slots:
                sl.append(rep(lid))

    def _capture(self, wave, ids, mask, labels):
        with (self.reducer.quiet() if self.reducer else contextlib.nullcontext()):
            self._capture_impl(wave, ids, mask, labels)

    def _capture_impl(self, wave, ids, mask, labels):
        s, dev = self.sys, wave.device
        s.prepare()                              # flat buckets on the caller's stream, before anything forks
        self._alloc(wave, ids, mask, labels)
        B = self.B
        from . import _engines as E
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):            # warm-up outside capture: engines, workspaces, tile plans, optimizer state
            # every rank captures here with the same shapes: the one point where the tile timing pass can be a collective
            # (rank 0's plans everywhere); the lazy pass inside the forward is off under data parallelism
            E.tune_pair(s.audio_encoder.engine(), s.text_encoder.engine(), self.in_slots[0][0].shape[0], self.in_slots[0][0].shape[1],
                        self.in_slots[0][1].shape[0], self.in_slots[0][1].shape[1], collective=True)
            a, t, *gf = self._encode_slot(0)
            # per slot: encoder outputs of the group + the mask / labels (+ raw gate features) that travel with them
            self.enc_slots = [[a.clone(), t.clone(), self.in_slots[k][2].clone(), self.in_slots[k][3].clone()] + [x.clone() for x in gf]
                              for k in range(2)]
            self.enc_cur = [a[:B].clone(), t[:B].clone()] + [x[:B].clone() for x in gf]
            del a, t, gf
            for _ in range(2):
                self.opt.zero_grad(set_to_none=True)
                self._head_fwd_bwd()
            self.opt.prepare_step(dev)
            self.opt.t -= 1
            if self.opt._plan is None:
                self.opt._build_plan()
            for grp, segs, loose in self.opt._plan:
                for b, st, e in segs:
                    self.opt._mv(id(b), b.flat)
                for p in loose:
                    self.opt._mv(id(p), p.data)
        torch.cuda.current_stream().wait_stream(side)
        self.opt.zero_grad(set_to_none=True)
        self._capture_encoders()
        self.g_head = torch.cuda.CUDAGraph()
        if self.split:
            with torch.cuda.graph(self.g_head):
                self.loss, self.logits = self._head_a()
            self.g_head_b = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_head_b, pool=self.g_head.pool()):
                self._head_b()
        else:
            with torch.cuda.graph(self.g_head):
                self.loss, self.logits = self._head_fwd_bwd()
        self.g_opt = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_opt):
            self.opt.launch()
        # gradients of the parameters outside the flat buckets (the prototypes) live in tensors the captured head graph writes
        # on every replay; `zero_grad(set_to_none=True)` below would drop the references, and a reducer that finds
        # `p.grad is None` skips the parameter - its replicas then drift apart (tests/test_gpu_dp.py)
        self._loose_grads = [(p, p.grad) for grp, segs, loose in self.opt._plan for p in loose if p.grad is not None]
        self._pick_encoder_stream()

    def _encode_slot(self, k):
        """(front end +) frozen encoders over slot k's staged group -> [a_enc, t_enc (, q_raw, c_raw: the gate features present)]."""
        sl = self.in_slots[k]
        if not self.gated:
            return list(self.sys.encode_frozen(sl[0], sl[1], sl[2]))
        a, t, q, c = self.sys.encode_frozen_gated(sl[0], sl[1], sl[2], sl[4])
        self._gate_idx = [i for i, x in enumerate((q, c)) if x is not None]          # which of (quality, conditioning) travel
        return [a, t] + [x for x in (q, c) if x is not None]

    def _gate_features(self):
        if not self.gated:
            return None
        gf = [None, None]
        for j, i in enumerate(self._gate_idx):
            gf[i] = self.enc_cur[2 + j]
        return tuple(gf)

    def _capture_encoders(self):
        """One graph per slot: (front end +) encoders over the slot's staged group -> the slot's output area (+ its mask /
        labels).  Both graphs run on the one encoder stream, never at the same time, and share the engines' workspace."""
        for k in range(2):
            self.g_encs[k] = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_encs[k]):
                outs = self._encode_slot(k)
                self.enc_slots[k][0].copy_(outs[0])
                self.enc_slots[k][1].copy_(outs[1])
                self.enc_slots[k][2].copy_(self.in_slots[k][2])
                self.enc_slots[k][3].copy_(self.in_slots[k][3])
                for j, x in enumerate(outs[2:]):
                    self.enc_slots[k][4 + j].copy_(x)

    def _replay_all(self, stream=None):
        """One encoder pass on the encoder stream beside `group` head replays, joined (idempotent: no optimizer step)."""
        cur = torch.cuda.current_stream()
        es = stream or self.enc_stream
        es.wait_stream(cur)
        with torch.cuda.stream(es):
            self.g_encs[0].replay()
        for _ in range(self.group):
            self.g_head.replay()
            if self.g_head_b is not None:
                self.g_head_b.replay()
        cur.wait_stream(es)

    def _overlapped_ms(self, reps=3, stream=None):
        cur = torch.cuda.current_stream()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(cur)
        for _ in range(reps):
            self._replay_all(stream)
        e1.record(cur)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    def _head_a(self):
        s = self.sys
        loss, logits, fused, leaf = s.loss_from_encoded(self.enc_cur[0], self.enc_cur[1], self.cur_mask, self.cur_labels,
                                                        self.use_proto, split=True, gate_features=self._gate_features())
        with _ops.defer_wgrads(), _ops.unit_loss_grad():
            loss.backward()
        _ops.wgrad_join()
        self._fused, self._dfused = fused, leaf.grad
        return loss.detach(), logits.detach()

    def _head_b(self):
        with _ops.defer_wgrads():
            self._fused.backward(self._dfused)
        _ops.wgrad_join()
        self._fused = self._dfused = None

    def _head_fwd_bwd(self):
        if self.split:
            out = self._head_a()
            self._head_b()
            return out
        loss, logits = self.sys.loss_from_encoded(self.enc_cur[0], self.enc_cur[1], self.cur_mask, self.cur_labels, self.use_proto,
                                                  gate_features=self._gate_features())
        with _ops.defer_wgrads(), _ops.unit_loss_grad():
            loss.backward()
        _ops.wgrad_join()
        return loss.detach(), logits.detach()

    def _pick_encoder_stream(self, candidates=5, reps=2):
        """HIP multiplexes streams onto a few hardware queues; two streams that land on the same queue run their
        graphs back to back.  Replay the captured graphs beside each other on a few fresh streams and keep the stream on
        which they actually overlap (replays are idempotent: no optimizer step is involved)."""
        best, best_ms = self.enc_stream, float("inf")
        for es in [self.enc_stream] + [torch.cuda.Stream() for _ in range(candidates - 1)]:
            ms = self._overlapped_ms(reps, es)
            if ms < best_ms * 0.97:
                best, best_ms = es, ms
        self.enc_stream = best
        self.overlap_ms = best_ms / self.group           # per step
        self._reset_grads_after_idle_replays()

    def profile_encoder_passes(self, n=3):
        """`n` encoder passes over slot 0's staged group issued EAGERLY on the encoder stream - the same launches the
        slot's graph replays, but outside a graph, so that per-launch HIP events (ser_prof_gemm_start / _stop) can
        bracket them - each beside `group` head-graph replays on the current stream, as in the timed schedule, back to back
        without a host synchronisation in between (the chip's clock management reacts to the load of the last milliseconds: an
        idle gap between passes would let it recover and flatter the kernels).  Probe waves started with each pass report the
        shader clock they saw (`clock_ghz_per_pass`, `clock_ghz_under_load` = their median).  Idempotent for training state: no optimizer step,
        gradients reset afterwards."""
        from . import _lib as L
        import ctypes as C
        cur, es = torch.cuda.current_stream(), self.enc_stream
        probe_stream = torch.cuda.Stream()
        probe = torch.zeros(n, 8, 2, dtype=torch.int64, device=self.enc_cur[0].device)
        fn = L.lib.ser_debug_clock_probe
        fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        torch.cuda.synchronize()
        keep = []
        for i in range(n):
            es.wait_stream(cur)
            with torch.cuda.stream(es):
                started = torch.cuda.Event()
                started.record(es)
                outs = self._encode_slot(0)
            probe_stream.wait_event(started)             # the probe starts with this pass and runs beside its first ~6 ms
            with torch.cuda.stream(probe_stream):        # dependent FMAs on 8 waves (one per XCD by round-robin placement)
                L.check(fn(probe[i].data_ptr(), 8, 400000, probe_stream.cuda_stream), "ser_debug_clock_probe")
            for _ in range(self.group):
                self.g_head.replay()
                if self.g_head_b is not None:
                    self.g_head_b.replay()
            cur.wait_stream(es)
            keep.append(outs)
            for x in outs:
                x.record_stream(cur)
        torch.cuda.synchronize()
        v = probe.cpu().double()
        ghz = (v[..., 0] / v[..., 1].clamp(min=1) * 0.1).median(dim=1).values            # per pass
        self.clock_ghz_per_pass = [round(float(g), 3) for g in ghz]
        self.clock_ghz_under_load = float(ghz.median())
        self._reset_grads_after_idle_replays()

    def _reset_grads_after_idle_replays(self):
        self.opt.zero_grad(set_to_none=True)
        for p, g in self._loose_grads:           # keep pointing at the tensors the captured graphs write (see _capture)
            p.grad = g

    def feed(self, wave, ids, mask, labels, lid=None):
        """Stage a batch into the slot being filled (encoder stream); the slot's encoder pass is issued when it is full.
        lid: [B, 2] language features of the clips' transcripts for the quality gates (SERSystem.language_features;
        default: the no-transcript branch) - only read when the audio encoder's gate flags are on."""
        if self.g_encs[0] is None:
            self._capture(wave, ids, mask, labels)
        k, n, B = self.fill_k, self.fill_n, self.B
        assert wave.shape[0] == B, "every batch of a pipelined run has the batch size of the first"
        assert all(k != lk for lk, _ in self.launched), "both slots are in flight: call step() before feeding again"
        es, cur = self.enc_stream, torch.cuda.current_stream()
        es.wait_stream(cur)           # the batch's producer, and every head-side copy out of this slot, are on `cur`
        with torch.cuda.stream(es):
            srcs = [wave, ids, mask, labels]
            if self.gated:
                srcs.append(lid if lid is not None else self.sys.language_features(None, B))
            for dst, src in zip(self.in_slots[k], srcs):
                dst[n * B:(n + 1) * B].copy_(src, non_blocking=True)
                if src.is_cuda:
                    src.record_stream(es)
            n += 1
            if n == self.group:
                self.g_encs[k].replay()
                self.enc_done[k].record(es)
        if n == self.group:
            self.launched.append((k, n))
            self.fill_k, self.fill_n = k ^ 1, 0
        else:
            self.fill_n = n

    def _head_step(self):
        """Head forward / backward / update on the next encoded batch (no staging)."""
        dev = self.enc_cur[0].device
        cur = torch.cuda.current_stream()
        if self.cons_k is None:
            assert self.launched, "nothing in flight: feed `prime` batches before the first step"
            (self.cons_k, self.cons_n), self.cons_j = self.launched.pop(0), 0
            cur.wait_event(self.enc_done[self.cons_k])       # the slot's encoder pass is complete
        k, j, B = self.cons_k, self.cons_j, self.B
        for p_, g_ in self._loose_grads:          # a caller's zero_grad(set_to_none=True) must not detach them (see _capture)
            p_.grad = g_
        sl = slice(j * B, (j + 1) * B)
        self.enc_cur[0].copy_(self.enc_slots[k][0][sl], non_blocking=True)
        self.enc_cur[1].copy_(self.enc_slots[k][1][sl], non_blocking=True)
        self.cur_mask.copy_(self.enc_slots[k][2][sl], non_blocking=True)
        self.cur_labels.copy_(self.enc_slots[k][3][sl], non_blocking=True)
        for j in range(2, len(self.enc_cur)):                     # raw gate features of the batch
            self.enc_cur[j].copy_(self.enc_slots[k][2 + j][sl], non_blocking=True)
        self.cons_j += 1
        if self.cons_j == self.cons_n:
            self.cons_k = None                    # fully copied out (on `cur`): the slot's next pass waits for `cur` in feed()
        return dev

    def _head_launch(self, dev):
        self.g_head.replay()
        if self.g_head_b is not None:
            if self.reducer:                      # classifier bucket + prototypes on their way while the rest of backward runs
                self.reducer.start([self.sys.classifier._flat], [self.sys.prototypes.prototypes])
            self.g_head_b.replay()
        if self.reducer:
            self.reducer.finish()
        self.opt.prepare_step(dev)
        self.g_opt.replay()
        if self.sched:
            self.sched.step()
        return self.loss

    def step(self, wave, ids, mask, labels, lid=None):
        """Head step on the oldest batch in flight; `wave, ...` is staged for a later encoder pass alongside it."""
        assert self.pending >= self.prime, f"call feed(batch) {self.prime} times before the first step"
        dev = self._head_step()
        self.feed(wave, ids, mask, labels, lid)   # staged (and, every `group` steps, encoded) on the other stream, beside the head
        return self._head_launch(dev)

    def drain(self):
        """Train on every batch that is still in flight without staging new ones (end of an epoch); a partly filled slot
        is encoded as it stands and only its staged batches are trained on.  Yields the loss of each step."""
        if self.fill_n:
            k, n = self.fill_k, self.fill_n
            es, cur = self.enc_stream, torch.cuda.current_stream()
            es.wait_stream(cur)
            with torch.cuda.stream(es):           # the rest of the slot still holds older batches: encoded, never consumed
                self.g_encs[k].replay()
                self.enc_done[k].record(es)
            self.launched.append((k, n))
            self.fill_k, self.fill_n = k ^ 1, 0
        while self.launched or self.cons_k is not None:
            yield self._head_launch(self._head_step())
