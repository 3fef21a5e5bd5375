"""The hot path assembled: the seven modules of ref src/train.py:54-69 and its training / inference
step (:145-177, :181-201), plus the data-parallel gradient reduction the reference does not have.

`SERSystem.train_step` is what `bench.py` times and what `train.py` loops over.  With
`use_graph=True` the forward+backward and the optimizer launches are captured into two hipGraphs
(graph replay removes the per-kernel host launch cost of the ~2k small head kernels); the gradient
all-reduce runs between them.
"""
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
    def encode(self, wave, ids, attn_mask):
        """Frozen encoders (one paired call) + the two trainable adapters (independent: the text one on a second
        stream)."""
        from .models.adapter import adapter_apply
        noisy = self.training and (getattr(self.audio_encoder, "encoder_train_noise", False) or getattr(self.text_encoder, "encoder_train_noise", False))
        if noisy or not (self.audio_encoder.freeze_base and self.text_encoder.freeze_base):
            # BASELINE config 3 (reference freeze_base=False): the fine-tuning form of the encoders, with gradients; also the
            # frozen encoders when their training-mode noise is requested (forward only, same operators)
            a_seq = self.audio_encoder.encode(wave.to(torch.float32))
            t_seq, t_mask = self.text_encoder.forward_ids(ids.to(wave.device), attn_mask.to(wave.device))
            a_mask = torch.ones(a_seq.shape[0], a_seq.shape[1], dtype=torch.float32, device=a_seq.device)
            return a_seq, a_mask, t_seq, t_mask
        a_enc, t_enc = self.encode_frozen(wave, ids.to(wave.device), attn_mask.to(wave.device))
        cur = torch.cuda.current_stream()
        if self._side is None:
            self._side = torch.cuda.Stream()
        side = self._side
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            t_seq = adapter_apply(self.text_encoder, t_enc)
        t_enc.record_stream(side)
        a_seq = adapter_apply(self.audio_encoder, a_enc)
        a_mask = torch.ones(a_seq.shape[0], a_seq.shape[1], dtype=torch.float32, device=a_seq.device)
        cur.wait_stream(side)
        t_seq.record_stream(cur)
        return a_seq, a_mask, t_seq, attn_mask.to(device=a_seq.device, dtype=torch.float32)

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
        dev = next(self.classifier.parameters()).device
        if self._drop_state is None or self._drop_state.device != dev:
            self._drop_state = torch.full((1,), int(self.dropout_seed), dtype=torch.int64, device=dev)
        self._drop_state.add_(1)                                  # one launch; captured with the step, so replays advance it
        return _ops.dropout_scope(self._drop_state)

    def loss_from_encoded(self, a_enc, t_enc, attn_mask, labels, use_proto=True, split=False):
        """Everything trainable: adapters -> cross-attention -> pooling -> fusion -> classifier -> loss.
        split=True: the classifier and the loss hang off a detached copy of `fused`; returns (loss, logits, fused, leaf),
        so that the caller can run the classifier's backward first (loss.backward(): its gradient bucket is then
        complete and can travel over xGMI) and the rest afterwards (fused.backward(leaf.grad))."""
        from .models.adapter import adapter_apply
        self._set_precision()
        cur = torch.cuda.current_stream()
        if self._side is None:
            self._side = torch.cuda.Stream()
        side = self._side
        with self._dropout_scope():
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                t_seq = adapter_apply(self.text_encoder, t_enc)
            a_seq = adapter_apply(self.audio_encoder, a_enc)
            cur.wait_stream(side)
            t_seq.record_stream(cur)
            a_mask = torch.ones(a_seq.shape[0], a_seq.shape[1], dtype=torch.float32, device=a_seq.device)
            fused = self.head(a_seq, a_mask, t_seq, attn_mask.to(torch.float32))
            leaf = fused.detach().requires_grad_() if split else fused
            logits, unc, _ = self.classifier(leaf, use_openmax=False, return_uncertainty=True)
        total = self.criterion(logits, unc, leaf, self.prototypes.prototypes, labels, use_proto=use_proto)
        if split:
            return total, logits, fused, leaf
        return total, logits

    def head(self, a_seq, a_mask, t_seq, t_mask):
        a_enh, t_enh = self.cross(a_seq, t_seq, a_mask, t_mask)
        # the two poolings are independent: text pooling (forward, and therefore its backward, which autograd runs
        # on the forward's stream) goes to the side stream
        cur = torch.cuda.current_stream()
        if self._side is None:
            self._side = torch.cuda.Stream()
        side = self._side
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            t_vec = self.pool_t(t_enh, t_mask)
        t_enh.record_stream(side)
        a_vec = self.pool_a(a_enh, a_mask)
        cur.wait_stream(side)
        t_vec.record_stream(cur)
        return self.fusion(a_vec, t_vec)

    def forward(self, wave, ids, attn_mask, use_openmax=True):
        fused = self.head(*self.encode(wave, ids, attn_mask))
        return self.classifier(fused, use_openmax=use_openmax)

    def loss(self, wave, ids, attn_mask, labels, use_proto=True):
        self._set_precision()
        with self._dropout_scope():
            fused = self.head(*self.encode(wave, ids, attn_mask))
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
        for m in (self.audio_encoder, self.text_encoder):
            if hasattr(m, "_adapter_flat"):
                out.append(m._adapter_flat)
        return out


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
        self._done = set()
        # issuing a collective while a compute graph replays is the point of the overlap; it is only done on RCCL
        # (gloo stages device tensors through the host and serialises against the replay) unless forced by env
        import os
        self.early = os.environ.get("SER_DP_EARLY", "1" if dist.get_backend(process_group) == "nccl" else "0") == "1"

    def _reduce(self, t):
        # SUM over ranks; the 1/world scale is applied by the library's own axpby kernel in finish()
        return self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.pg, async_op=True)

    def _hook(self, bucket):
        if id(bucket) in self._done:
            return
        self._done.add(id(bucket))
        if self.cuda:
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                self.pending.append((self._reduce(bucket.gflat), bucket.gflat))
        else:
            self.pending.append((self._reduce(bucket.gflat), bucket.gflat))

    def start(self, buckets, loose=()):
        """Issue the all-reduce of the given (already complete) buckets now, without waiting for them."""
        for b in buckets:
            if b.gflat is not None:
                self._hook(b)
        for p in loose:
            if p.grad is not None and id(p) not in self._done:
                self._done.add(id(p))
                if self.cuda:
                    self.stream.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(self.stream):
                        self.pending.append((self._reduce(p.grad), p.grad))
                else:
                    self.pending.append((self._reduce(p.grad), p.grad))

    def arm(self):
        self._done.clear()
        if self.overlap:
            for b in self.system.buckets():
                b.grad_ready_hook = self._hook

    def finish(self):
        for b in self.system.buckets():       # anything the hooks did not catch (overlap off, or first step)
            if b.gflat is not None:
                self._hook(b)
        for p in self.loose:
            if p.grad is not None and id(p) not in self._done:
                self.pending.append((self._reduce(p.grad), p.grad))
        back = []
        for group in self.coalesced:
            live = [p for p in group if p.grad is not None]
            if live:
                flat = torch.cat([p.grad.reshape(-1) for p in live])
                self.pending.append((self._reduce(flat), flat))
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
        self._done.clear()


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

    def _fwd_bwd(self, wave, ids, mask, labels):
        loss, logits = self.sys.loss(wave, ids, mask, labels, self.use_proto)
        # eager data parallelism reduces a bucket from its module's backward hook, so gradients cannot be deferred there
        with _ops.defer_wgrads(self.use_graph or self.reducer is None):
            loss.backward()
        _ops.wgrad_join()
        return loss.detach(), logits.detach()

    # ---- split form: the classifier + loss hang off a detached copy of `fused`
    def _fwd_bwd_a(self, wave, ids, mask, labels):
        s = self.sys
        s._set_precision()
        with s._dropout_scope():              # the same precision mode and training-mode dropout as SERSystem.loss
            fused = s.head(*s.encode(wave, ids, mask))
            leaf = fused.detach().requires_grad_()
            logits, unc, _ = s.classifier(leaf, use_openmax=False, return_uncertainty=True)
        loss = s.criterion(logits, unc, leaf, s.prototypes.prototypes, labels, use_proto=self.use_proto)
        with _ops.defer_wgrads():
            loss.backward()
        _ops.wgrad_join()
        self._fused, self._dfused = fused, leaf.grad
        return loss.detach(), logits.detach()

    def _bwd_b(self):
        with _ops.defer_wgrads():
            self._fused.backward(self._dfused)
        _ops.wgrad_join()
        self._fused = self._dfused = None

    def step(self, wave, ids, mask, labels):
        dev = wave.device
        if not self.use_graph:
            self.opt.zero_grad(set_to_none=True)
            if self.reducer:
                self.reducer.arm()
            self.loss, self.logits = self._fwd_bwd(wave, ids, mask, labels)
            if self.reducer:
                self.reducer.finish()
            self.opt.prepare_step(dev)
            self.opt.launch()
        else:
            key = (tuple(wave.shape), tuple(ids.shape))
            if key not in self._graphs:
                if len(self._graphs) >= self.max_graphs:      # too many shapes to keep a graph for each: eager step
                    self.use_graph = False
                    try:
                        return self.step(wave, ids, mask, labels)
                    finally:
                        self.use_graph = True
                g_opt = self.g_opt
                self._capture(wave, ids, mask, labels)
                if g_opt is not None:
                    self.g_opt = g_opt                          # the optimizer launch does not depend on the input shape
                self._graphs[key] = (self.static, self.g_fb, self.g_b, self.loss, self.logits)
            self.static, self.g_fb, self.g_b, self.loss, self.logits = self._graphs[key]
            for p_, g_ in getattr(self, "_loose_grads", ()):      # a caller's zero_grad(set_to_none=True) must not detach them
                p_.grad = g_
            for s, t in zip(self.static, (wave, ids, mask, labels)):
                s.copy_(t, non_blocking=True)
            self.g_fb.replay()
            if self.split:
                if self.reducer and self.reducer.early:
                    self.reducer.start([self.sys.classifier._flat], [self.sys.prototypes.prototypes])
                self.g_b.replay()
            if self.reducer:
                self.reducer.finish()
            self.opt.prepare_step(dev)
            self.g_opt.replay()
        if self.sched:
            self.sched.step()
        return self.loss

    def _capture(self, wave, ids, mask, labels):
        dev = wave.device
        self.static = [wave.clone(), ids.clone(), mask.clone(), labels.clone()]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):       # warm-up outside capture: workspaces, flat buckets, optimizer state
            for _ in range(2):
                self.opt.zero_grad(set_to_none=True)
                self._fwd_bwd(*self.static)
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
                self.loss, self.logits = self._fwd_bwd(*self.static)
        if not keep:
            self._loose_grads = [(p_, p_.grad) for grp, segs, loose in self.opt._plan for p_ in loose if p_.grad is not None]
        if self.g_opt is None or not self._graphs:
            self.g_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_opt):
                self.opt.launch()


class PipelinedStepper:
    """Training steps with the frozen-encoder forward of batch t+1 overlapped with the trainable part of batch t.

    With `freeze_base=True` (the reference default, BASELINE config 2) the encoder outputs of a batch do not depend
    on any parameter the optimizer touches, so computing them one step early changes nothing numerically
    (tests/test_gpu_system.py checks bit-identity with sequential stepping).  What it buys: the head is ~600 small,
    latency-bound launches that occupy a handful of CUs each, while the encoder GEMMs fill the chip; on two streams
    they run side by side instead of back to back.  Every call still executes exactly one encoder forward and one
    head forward/backward/AdamW update — nothing is cached or skipped, the encoder work is merely issued earlier.

    `feed(batch)` must be called once before the first `step`; `step(next_batch)` trains on the batch fed
    previously and starts the encoders on `next_batch`.
    """

    def __init__(self, system, optimizer, scheduler=None, reducer=None, use_proto=True, split_backward=None, depth=1):
        self.sys, self.opt, self.sched, self.reducer, self.use_proto = system, optimizer, scheduler, reducer, use_proto
        # depth = encoder passes in flight: 1 = the encoders of batch t+1 beside the head of batch t; 2 = those of t+1 and
        # t+2 beside it (two encoder graphs on two streams with their own workspaces: one chain's small kernels - attention,
        # LayerNorm, tile tails - run under the other's GEMMs).  Either way every step runs exactly one encoder pass and
        # one update; `feed` must be called `depth` times before the first `step`.
        assert depth in (1, 2)
        self.depth = depth
        self.enc_streams = [torch.cuda.Stream() for _ in range(depth)]
        self.g_encs = [None] * depth
        self.g_head = self.g_head_b = self.g_opt = None
        # Data parallel: the head graph is captured in two pieces — A: forward + loss + classifier backward, B: the
        # backward of fusion / pooling / cross-attention / adapters — and the all-reduce of the classifier bucket (76 MB of
        # the 100 MB of gradients) is issued between them, so it travels over xGMI while B runs.  Same arithmetic as the
        # single graph (the two consumers of `fused` add their gradients in the detached leaf).
        self.split = (reducer is not None and reducer.early) if split_backward is None else bool(split_backward)
        self.refine_plans = True          # in-situ choice between near-tied GEMM tile configurations (refine_gemm_plans)
        self.loss = None
        self.queue, self.free = [], list(range(depth))       # slots whose encoder pass is in flight (oldest first) / unused

    # single-slot names kept for the measurement scripts
    @property
    def g_enc(self):
        return self.g_encs[0]

    @property
    def enc_stream(self):
        return self.enc_streams[0]

    @enc_stream.setter
    def enc_stream(self, st):
        self.enc_streams[0] = st

    @property
    def pending(self):
        return len(self.queue) > 0

    def _alloc(self, wave, ids, mask, labels):
        self.in_slots = [[wave.clone(), ids.clone(), mask.clone(), labels.clone()] for _ in range(self.depth)]   # encoder graph inputs
        self.in_next = self.in_slots[0]
        self.cur_mask, self.cur_labels = mask.clone(), labels.clone()                # inputs of the head graph
        self.nxt_labels = labels.clone()

    def _capture(self, wave, ids, mask, labels):
        s, dev = self.sys, wave.device
        self._alloc(wave, ids, mask, labels)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):            # warm-up outside capture
            self.enc_slots = []
            for k in range(self.depth):
                a, t = s.encode_frozen(self.in_slots[k][0], self.in_slots[k][1], self.in_slots[k][2], slot=k)
                self.enc_slots.append([a.clone(), t.clone()])
            self.enc_next = self.enc_slots[0]
            self.enc_cur = [a.clone(), t.clone()]
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
        if self.refine_plans:
            try:
                self.refine_gemm_plans()
            except Exception as e:      # noqa: BLE001 - a tuning extra must not take the step down: keep the stand-alone plans
                import sys
                sys.stderr.write(f"PipelinedStepper: in-situ plan refinement skipped ({type(e).__name__}: {e})\n")
                self._capture_encoders()
                self._reset_grads_after_idle_replays()

    def _capture_encoders(self):
        s = self.sys
        for k in range(self.depth):
            self.g_encs[k] = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_encs[k]):
                a, t = s.encode_frozen(self.in_slots[k][0], self.in_slots[k][1], self.in_slots[k][2], slot=k)
                self.enc_slots[k][0].copy_(a)
                self.enc_slots[k][1].copy_(t)

    def _replay_all(self, streams=None):
        """Every encoder graph on its stream beside the head graph(s), joined (idempotent: no optimizer step involved)."""
        cur = torch.cuda.current_stream()
        streams = streams or self.enc_streams
        for k, es in enumerate(streams):
            es.wait_stream(cur)
            with torch.cuda.stream(es):
                self.g_encs[k].replay()
        self.g_head.replay()
        if self.g_head_b is not None:
            self.g_head_b.replay()
        for es in streams:
            cur.wait_stream(es)

    def _overlapped_ms(self, reps=6, streams=None):
        cur = torch.cuda.current_stream()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(cur)
        for _ in range(reps):
            self._replay_all(streams)
        e1.record(cur)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    def refine_gemm_plans(self, gain=0.988):
        """The engines pick a tile configuration per GEMM shape from stand-alone timings; where the runner-up is within a
        few percent, which of the two is faster BESIDE the head graph is a different question (a 32 KB single-buffer tile
        and an 86 KB three-stage tile tie alone and differ by 3 % of the step here).  For each such shape: switch the plan
        to each close alternative, re-capture the encoder graph, time the overlapped replay, keep the fastest.  Results are
        bit-identical either way (same products, same order per accumulator)."""
        from . import _engines as E
        self.plan_refinements = []
        cands = E.close_runner_ups()
        if not cands:
            return
        measure = lambda: min(self._overlapped_ms(8) for _ in range(3))
        self._overlapped_ms(2)
        base = measure()
        for key, best, alts in cands:
            keep = best
            for alt in alts:
                E.set_plan(key, alt)
                self._capture_encoders()
                self._overlapped_ms(2)
                t = measure()
                if t < base * gain:
                    self.plan_refinements.append((key[:3], keep, alt, round(base, 3), round(t, 3)))
                    base, keep = t, alt
            E.set_plan(key, keep)
            self._capture_encoders()
        self._reset_grads_after_idle_replays()

    def _head_a(self):
        s = self.sys
        loss, logits, fused, leaf = s.loss_from_encoded(self.enc_cur[0], self.enc_cur[1], self.cur_mask, self.cur_labels,
                                                        self.use_proto, split=True)
        with _ops.defer_wgrads():
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
        loss, logits = self.sys.loss_from_encoded(self.enc_cur[0], self.enc_cur[1], self.cur_mask, self.cur_labels, self.use_proto)
        with _ops.defer_wgrads():
            loss.backward()
        _ops.wgrad_join()
        return loss.detach(), logits.detach()

    def _pick_encoder_stream(self, candidates=6, reps=3):
        """HIP multiplexes streams onto a few hardware queues; two streams that land on the same queue run their
        graphs back to back.  Replay the captured graphs beside each other on a few fresh streams and keep, slot by
        slot, the stream on which they actually overlap (replays are idempotent: no optimizer step is involved)."""
        for k in range(self.depth):
            best, best_ms = self.enc_streams[k], float("inf")
            for es in [self.enc_streams[k]] + [torch.cuda.Stream() for _ in range(candidates - 1)]:
                ms = self._overlapped_ms(reps, self.enc_streams[:k] + [es])
                if ms < best_ms * 0.97:
                    best, best_ms = es, ms
            self.enc_streams[k] = best
            self.overlap_ms = best_ms
        self._reset_grads_after_idle_replays()

    def _reset_grads_after_idle_replays(self):
        self.opt.zero_grad(set_to_none=True)
        for p, g in self._loose_grads:           # keep pointing at the tensors the captured graphs write (see _capture)
            p.grad = g

    def feed(self, wave, ids, mask, labels):
        """Start the encoders on a batch (a free slot's stream); its head step happens `depth` `step` calls later."""
        if self.g_encs[0] is None:
            self._capture(wave, ids, mask, labels)
        assert self.free, "every encoder slot is in flight: call step() before feeding again"
        k = self.free.pop(0)
        es, cur = self.enc_streams[k], torch.cuda.current_stream()
        es.wait_stream(cur)                       # previous users of this slot's buffers on the main stream are done
        with torch.cuda.stream(es):
            for dst, src in zip(self.in_slots[k], (wave, ids, mask, labels)):
                dst.copy_(src, non_blocking=True)
            self.g_encs[k].replay()
        self.queue.append(k)

    def step(self, wave, ids, mask, labels):
        """Head step on the oldest batch in flight, encoders of this batch alongside it."""
        assert len(self.queue) == self.depth, f"call feed(batch) {self.depth} time(s) before the first step"
        dev = wave.device
        cur = torch.cuda.current_stream()
        k = self.queue.pop(0)
        for p_, g_ in self._loose_grads:          # a caller's zero_grad(set_to_none=True) must not detach them (see _capture)
            p_.grad = g_
        cur.wait_stream(self.enc_streams[k])      # encoder outputs of the batch to train on are ready
        self.enc_cur[0].copy_(self.enc_slots[k][0], non_blocking=True)
        self.enc_cur[1].copy_(self.enc_slots[k][1], non_blocking=True)
        self.cur_mask.copy_(self.in_slots[k][2], non_blocking=True)
        self.cur_labels.copy_(self.in_slots[k][3], non_blocking=True)
        self.free.append(k)
        self.feed(wave, ids, mask, labels)        # encoders of the NEXT batch: other stream, runs beside the head
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
