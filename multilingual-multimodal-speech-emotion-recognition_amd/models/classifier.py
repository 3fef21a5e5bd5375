"""Classifiers — drop-in for ref src/models/classifier.py on HIP kernels.

`AdvancedOpenMaxClassifier` (ref :157-305) keeps the reference's module tree so that its 304
state_dict entries (parameters + Weibull buffers) interchange, but its forward/backward is one
autograd node that walks the 35 residual blocks with the fp32 MFMA GEMM and the LayerNorm kernels
(the reference's own forward at :200-238, not DeepClassifier.forward, is what is restated).  The
class-anchor branch (:8-70) contributes an identically-zero loss with zero gradient in the
reference; here its parameters are kept (and weight-decayed, as there) but nothing is computed.
Dropout is the identity (parity definition).
"""
from typing import Tuple

import numpy as np
import torch
import torch.nn as nn

from .. import _ops as O
from ._flat import FlatParams


class ClassAnchorClustering(nn.Module):
    """Parameter holder for ref classifier.py:8-70 (class_anchors, anchor_projection.{0,1}, temperature)."""

    def __init__(self, feature_dim: int, num_classes: int, anchor_dim: int = 128):
        super().__init__()
        self.feature_dim, self.num_classes, self.anchor_dim = feature_dim, num_classes, anchor_dim
        self.class_anchors = nn.Parameter(torch.randn(num_classes, anchor_dim))
        self.anchor_projection = nn.Sequential(nn.Linear(feature_dim, anchor_dim), nn.LayerNorm(anchor_dim), nn.ReLU(),
                                               nn.Dropout(0.1))
        self.temperature = nn.Parameter(torch.tensor(1.0))

    def forward(self, features):
        # compute_clustering_loss == mean(clamp(s - max(s), min=0)) == 0 for every input (ref :58-70)
        return None, torch.zeros((), dtype=features.dtype, device=features.device)


class DeepResidualBlock(nn.Module):
    def __init__(self, dim: int, dropout: float = 0.1):
        super().__init__()
        self.block = nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, dim), nn.ReLU(), nn.Dropout(dropout),
                                   nn.Linear(dim, dim), nn.Dropout(dropout))


class DeepClassifier(nn.Module):
    """Module tree of ref classifier.py:92-143 (weights Xavier-uniform, zero biases :131-138)."""

    def __init__(self, input_dim: int, num_classes: int, num_layers: int = 35, base_dim: int = 512, dropout: float = 0.1):
        super().__init__()
        self.input_dim, self.num_classes, self.num_layers, self.base_dim = input_dim, num_classes, num_layers, base_dim
        self.input_projection = nn.Sequential(nn.Linear(input_dim, base_dim), nn.LayerNorm(base_dim), nn.ReLU(),
                                              nn.Dropout(dropout))
        self.residual_layers = nn.ModuleList([DeepResidualBlock(base_dim, dropout) for _ in range(num_layers)])
        self.layer_norms = nn.ModuleList([nn.LayerNorm(base_dim) for _ in range(num_layers)])
        self.output_projection = nn.Sequential(nn.Linear(base_dim, base_dim // 2), nn.LayerNorm(base_dim // 2), nn.ReLU(),
                                               nn.Dropout(dropout), nn.Linear(base_dim // 2, num_classes))
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(module):
        if isinstance(module, nn.Linear):
            nn.init.xavier_uniform_(module.weight)
            if module.bias is not None:
                nn.init.zeros_(module.bias)


class _ClassifierFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, m, x, *params):
        dc, uh_ = m.deep_classifier, m.uncertainty_head
        x = x.contiguous()
        ip, op = dc.input_projection, dc.output_projection
        t0 = O.linear_fwd(x, ip[0].weight, ip[0].bias)
        y0, ln0 = O.ln_fwd(t0, ip[1].weight, ip[1].bias)
        sites = m._drop_sites
        d_in, d_out, d_unc = O.dropout_ctx(ip[3].p), O.dropout_ctx(op[3].p), O.dropout_ctx(uh_[2].p)
        h = O.act_fwd(y0, O.ACT_RELU, d_in, sites[0])            # ReLU + Dropout (ref classifier.py:108-109)
        h0 = h
        blocks = []
        stack = None
        nblk = len(dc.residual_layers)
        if nblk and O.stack_supported(nblk, h.shape[0], h.shape[1]):
            # the whole residual stack in one persistent launch
            tab, _, flags = m._stack_tables()
            d_blk = O.dropout_ctx(dc.residual_layers[0].block[3].p)             # ref classifier.py:83,85
            Hs, X1, U, A, ST = O.stack_fwd(h, tab, nblk, flags[0], dc.layer_norms[0].eps, d_blk, sites[3])
            stack = (Hs, X1, U, A, ST, d_blk)
            h = Hs[nblk - 1]
        fused_ln = h.shape[1] <= 512
        for blk, lno in (() if stack is not None else zip(dc.residual_layers, dc.layer_norms)):
            b = blk.block
            hin = h
            if fused_ln and hin.shape[0] <= 16 and hin.shape[1] % 16 == 0:
                # LN_out + LN_in + Linear + ReLU of the block in one launch
                a, x1, u, stats = O.linear_fwd_ln2(hin, b[1].weight, b[1].bias, O.ACT_RELU, lno.weight, lno.bias,
                                                   b[0].weight, b[0].bias)
                lnA = lnB = None
            elif fused_ln:
                x1, u, stats = O.ln2_fwd(hin, lno.weight, lno.bias, b[0].weight, b[0].bias)
                lnA = lnB = None
                a = O.linear_fwd(u, b[1].weight, b[1].bias, O.ACT_RELU)
            else:
                x1, lnA = O.ln_fwd(hin, lno.weight, lno.bias)
                u, lnB = O.ln_fwd(x1, b[0].weight, b[0].bias)
                stats = None
                a = O.linear_fwd(u, b[1].weight, b[1].bias, O.ACT_RELU)
            d_b = O.dropout_ctx(b[3].p)                            # per-Linear path (M > 16): standalone dropout launches
            if d_b is None:
                h = O.linear_fwd(a, b[4].weight, b[4].bias, residual=x1)
            else:
                bi = len(blocks)
                O.dropout_(a, d_b, sites[3] + 2 * bi)
                h = O.linear_fwd(a, b[4].weight, b[4].bias)
                O.dropout_(h, d_b, sites[3] + 2 * bi + 1)
                O.axpby(x1, h, 1.0, 1.0)                         # h = x1 + drop(t)
            blocks.append((lnA, lnB, u, a, hin, x1, stats, d_b))
        tf = O.linear_fwd(h, op[0].weight, op[0].bias)
        yf, lnF = O.ln_fwd(tf, op[1].weight, op[1].bias)
        f = O.act_fwd(yf, O.ACT_RELU, d_out, sites[1])           # ReLU + Dropout (ref classifier.py:126-127,218)
        logits = O.linear_fwd(f, op[4].weight, op[4].bias)
        uh = O.linear_fwd(f, uh_[0].weight, uh_[0].bias, O.ACT_RELU)
        O.dropout_(uh, d_unc, sites[2])                          # ref classifier.py:195
        unc = O.linear_fwd(uh, uh_[3].weight, uh_[3].bias, O.ACT_SIGMOID)
        ctx.m = m
        ctx.saved = (x, ln0, h0, blocks, h, lnF, f, uh, unc, stack)
        ctx.drop = (d_in, d_out, d_unc)
        ctx.need_dx = x.requires_grad
        ctx.mark_non_differentiable(f)
        return logits, unc, f

    @staticmethod
    def backward(ctx, dlogits, dunc, _df_unused):
        m = ctx.m
        dc, uh_ = m.deep_classifier, m.uncertainty_head
        ip, op = dc.input_projection, dc.output_projection
        x, ln0, h0, blocks, h_last, lnF, f, uh, unc, stack = ctx.saved
        d_in, d_out, d_unc = ctx.drop
        sites = m._drop_sites
        fp = m._flat
        acc = fp.accumulating()
        g = fp.gview
        dev = x.device
        B = x.shape[0]
        if dlogits is None:
            dlogits = torch.zeros(B, op[4].weight.shape[0], dtype=torch.float32, device=dev)
        if dunc is None:
            dunc = torch.zeros(B, 1, dtype=torch.float32, device=dev)
        dlogits, dunc = dlogits.contiguous(), dunc.contiguous()
        # heads
        O.linear_wgrad(dlogits, f, g(op[4].weight), g(op[4].bias), acc)
        df = O.linear_dgrad(dlogits, op[4].weight)
        du2 = O.act_bwd(dunc, unc, O.ACT_SIGMOID, inplace=False)
        O.linear_wgrad(du2, uh, g(uh_[3].weight), g(uh_[3].bias), acc)
        duh = O.linear_dgrad(du2, uh_[3].weight)
        O.act_bwd(duh, uh, O.ACT_RELU, dctx=d_unc, site=sites[2])
        O.linear_wgrad(duh, f, g(uh_[0].weight), g(uh_[0].bias), acc)
        O.linear_dgrad(duh, uh_[0].weight, out=df, accumulate=True)
        # output projection
        O.act_bwd(df, f, O.ACT_RELU, dctx=d_out, site=sites[1])
        dtf = O.ln_bwd(df, lnF, op[1].weight, g(op[1].weight), g(op[1].bias), acc)
        O.linear_wgrad(dtf, h_last, g(op[0].weight), g(op[0].bias), acc)
        wg = []
        if stack is not None:
            Hs, X1, U, A, ST, d_blk = stack
            nblk, Mr, Dd = Hs.shape
            tab, gtab, flags = m._stack_tables()
            DH = torch.empty(nblk + 1, Mr, Dd, dtype=torch.float32, device=dev)
            O.linear_dgrad(dtf, op[0].weight, out=DH[nblk])
            DA, DU, DX1, DT = O.stack_bwd(tab, h0, Hs, X1, A, ST, DH, flags[1], d_blk, sites[3])
            O.stack_ln_param_bwd(gtab, h0, Hs, X1, ST, DU, DX1, acc)
            Gout = DT if DT is not None else DH          # gradient at the (dropped) second Linear's output
            for i in range(nblk - 1, -1, -1):
                b = dc.residual_layers[i].block
                wg.append((Gout[i + 1], A[i], g(b[4].weight), g(b[4].bias)))
                wg.append((DA[i], U[i], g(b[1].weight), g(b[1].bias)))
            dh = DH[0]
        else:
            dh = O.linear_dgrad(dtf, op[0].weight)
        # residual stack, last block first.  The weight gradients are only collected here and issued in ONE batched
        # launch after the loop: they are off the dgrad chain, which is the critical path of this backward.
        for i in range(len(blocks) - 1, -1, -1):
            lnA, lnB, u, a, hin, x1, stats, d_b = blocks[i]
            b, lno = dc.residual_layers[i].block, dc.layer_norms[i]
            dt = dh
            if d_b is not None:                                            # gradient through the second dropout
                dt = O.dropout_(dh.clone(), d_b, sites[3] + 2 * i + 1)
            da = O.linear_dgrad(dt, b[4].weight, relu_mask=a)              # dgrad with ReLU' fused (a > 0: active and kept)
            O.dropout_(da, d_b, sites[3] + 2 * i)                          # the first dropout's 1 / (1 - p) on the kept entries
            wg.append((dt, a, g(b[4].weight), g(b[4].bias)))
            wg.append((da, u, g(b[1].weight), g(b[1].bias)))
            du = O.linear_dgrad(da, b[1].weight)
            if stats is not None:
                dh = O.ln2_bwd(du, dh, hin, x1, stats, lno.weight, b[0].weight, g(lno.weight), g(lno.bias), g(b[0].weight),
                               g(b[0].bias), acc)
            else:
                dx1 = O.ln_bwd(du, lnB, b[0].weight, g(b[0].weight), g(b[0].bias), acc, dx_add=dh)
                dh = O.ln_bwd(dx1, lnA, lno.weight, g(lno.weight), g(lno.bias), acc)
        if wg:
            O.linear_wgrad_batch(wg, acc)
        # input projection
        O.act_bwd(dh, h0, O.ACT_RELU, dctx=d_in, site=sites[0])
        dt0 = O.ln_bwd(dh, ln0, ip[1].weight, g(ip[1].weight), g(ip[1].bias), acc)
        O.linear_wgrad(dt0, x, g(ip[0].weight), g(ip[0].bias), acc)
        dx = O.linear_dgrad(dt0, ip[0].weight) if ctx.need_dx else None
        # anchor branch: defined-but-zero gradients in the reference.  Nothing ever writes those views of the flat gradient
        # bucket, which is created zero-filled (models/_flat.py), so they are zeros without a fill launch per parameter and step
        fp.publish()
        ctx.saved = None
        return (None, dx) + (None,) * len(fp.params)


class AdvancedOpenMaxClassifier(nn.Module):
    def __init__(self, input_dim: int, num_labels: int, num_layers: int = 35, base_dim: int = 512, dropout: float = 0.1,
                 alpha: float = 20.0):
        super().__init__()
        self.num_labels = num_labels
        self.alpha = alpha
        self.deep_classifier = DeepClassifier(input_dim=input_dim, num_classes=num_labels, num_layers=num_layers,
                                              base_dim=base_dim, dropout=dropout)
        self.anchor_clustering = ClassAnchorClustering(feature_dim=base_dim // 2, num_classes=num_labels, anchor_dim=128)
        self.register_buffer('weibull_alpha', torch.ones(num_labels))
        self.register_buffer('weibull_beta', torch.ones(num_labels))
        self.register_buffer('weibull_tau', torch.zeros(num_labels))
        self.register_buffer('activation_vectors', torch.zeros(num_labels, base_dim // 2))
        self.uncertainty_head = nn.Sequential(nn.Linear(base_dim // 2, 64), nn.ReLU(), nn.Dropout(dropout), nn.Linear(64, 1),
                                              nn.Sigmoid())
        ac = self.anchor_clustering
        self._anchor_params = [ac.class_anchors] + list(ac.anchor_projection.parameters())
        grad_params = (list(self.deep_classifier.parameters()) + self._anchor_params +
                       list(self.uncertainty_head.parameters()))       # anchor `temperature` never gets a gradient
        self._flat = FlatParams(grad_params)
        # input projection, output projection, uncertainty head, then two per residual block (first id of the run)
        self._drop_sites = (O.new_dropout_site(), O.new_dropout_site(), O.new_dropout_site(), O.new_dropout_site(2 * num_layers))

    def _run(self, x):
        self._flat.ensure()
        return _ClassifierFn.apply(self, x, *self._flat.params)

    def _stack_tables(self):
        """Device tables of parameter / gradient pointers of the residual stack for the persistent kernels
        (csrc/persist.hip: StackBlockPtrs, StackGradPtrs) + their hand-off flag words.  Rebuilt when the flat
        buckets move (`.to(device)`, re-flattening)."""
        fp, dc = self._flat, self.deep_classifier
        key = (fp.flat.data_ptr(), fp.gflat.data_ptr())
        if getattr(self, "_stack_cache", None) is None or self._stack_cache[0] != key:
            rows, grows = [], []
            for blk, lno in zip(dc.residual_layers, dc.layer_norms):
                b = blk.block
                ps = (lno.weight, lno.bias, b[0].weight, b[0].bias, b[1].weight, b[1].bias, b[4].weight, b[4].bias)
                rows.append([p.data_ptr() for p in ps])
                grows.append([fp.gview(p).data_ptr() for p in ps[:4]])
            dev = fp.flat.device
            tab = torch.tensor(rows, dtype=torch.int64).to(dev)
            gtab = torch.tensor(grows, dtype=torch.int64).to(dev)
            nb = int(O.L.lib.ser_stack_scratch_bytes(dc.base_dim))
            flags = [torch.zeros(nb // 4, dtype=torch.int32, device=dev) for _ in range(2)]   # launch counter, abort word, ring
            self._stack_cache = (key, tab, gtab, flags)
        return self._stack_cache[1:]

    def penultimate_features(self, x: torch.Tensor) -> torch.Tensor:
        """The 256-d features train.py:222-236 recomputes layer by layer for fit_weibull."""
        return self._run(x)[2]

    def forward(self, x: torch.Tensor, use_openmax: bool = True, return_uncertainty: bool = False):
        logits, unc, feats = self._run(x)
        z = getattr(self, "_zero_scalar", None)           # the constant 0. of the anchor branch (ref :58-70), one fill per device
        if z is None or z.device != x.device:
            self._zero_scalar = z = torch.zeros((), dtype=torch.float32, device=x.device)
        anchor_loss = z
        if use_openmax and not self.training:
            logits = self.openmax_forward(feats, logits)
        if return_uncertainty:
            return logits, unc, anchor_loss
        return logits

    def openmax_forward(self, features: torch.Tensor, logits: torch.Tensor) -> torch.Tensor:
        out = logits.detach().clone()
        return O.openmax_(features.detach().contiguous(), self.activation_vectors, self.weibull_alpha, self.weibull_beta,
                          self.weibull_tau, out, 0.3, 0.8)

    def fit_weibull(self, features: torch.Tensor, labels: torch.Tensor):
        """ref :277-305 — end-of-training calibration, off the hot path (host arithmetic as in the reference)."""
        print("Fitting enhanced Weibull distributions for OpenMax...")
        feats = features.detach().float().cpu()
        labs = labels.detach().cpu()
        for c in range(self.num_labels):
            msk = labs == c
            if int(msk.sum()) == 0:
                continue
            cf = feats[msk]
            mu = cf.mean(dim=0)
            d = torch.norm(cf - mu, dim=1).numpy()
            self.activation_vectors[c] = mu.to(self.activation_vectors.device)
            self.weibull_alpha[c] = 2.5
            self.weibull_beta[c] = float(d.std() * 1.5)
            self.weibull_tau[c] = float(d.min() * 0.8)
        print("Enhanced Weibull fitting completed!")


class _MlpFn(torch.autograd.Function):
    """Linear-ReLU-Linear-ReLU-Linear stack of the legacy classifiers (ref :309-436)."""

    @staticmethod
    def forward(ctx, m, x, *params):
        lin = [l for l in m.net if isinstance(l, nn.Linear)]
        x = x.contiguous()
        h1 = O.linear_fwd(x, lin[0].weight, lin[0].bias, O.ACT_RELU)
        h2 = O.linear_fwd(h1, lin[1].weight, lin[1].bias, O.ACT_RELU)
        y = O.linear_fwd(h2, lin[2].weight, lin[2].bias)
        ctx.m, ctx.saved, ctx.need_dx = m, (x, h1, h2), x.requires_grad
        ctx.mark_non_differentiable(h2)
        return y, h2

    @staticmethod
    def backward(ctx, dy, _):
        m = ctx.m
        lin = [l for l in m.net if isinstance(l, nn.Linear)]
        x, h1, h2 = ctx.saved
        fp = m._flat
        acc, g = fp.accumulating(), fp.gview
        dy = dy.contiguous()
        O.linear_wgrad(dy, h2, g(lin[2].weight), g(lin[2].bias), acc)
        d2 = O.linear_dgrad(dy, lin[2].weight)
        O.act_bwd(d2, h2, O.ACT_RELU)
        O.linear_wgrad(d2, h1, g(lin[1].weight), g(lin[1].bias), acc)
        d1 = O.linear_dgrad(d2, lin[1].weight)
        O.act_bwd(d1, h1, O.ACT_RELU)
        O.linear_wgrad(d1, x, g(lin[0].weight), g(lin[0].bias), acc)
        dx = O.linear_dgrad(d1, lin[0].weight) if ctx.need_dx else None
        fp.publish()
        return (None, dx) + (None,) * len(fp.params)


def _legacy_net(input_dim, hidden, num_labels, p):
    return nn.Sequential(nn.Linear(input_dim, 256), nn.ReLU(), nn.Dropout(p), nn.Linear(256, hidden), nn.ReLU(),
                         nn.Dropout(p), nn.Linear(hidden, num_labels))


class OpenMaxClassifier(nn.Module):
    """ref classifier.py:309-419."""

    def __init__(self, input_dim: int, num_labels: int, hidden: int = 128, p: float = 0.1, alpha: float = 20.0):
        super().__init__()
        self.num_labels, self.alpha = num_labels, alpha
        self.net = _legacy_net(input_dim, hidden, num_labels, p)
        self.register_buffer('weibull_alpha', torch.ones(num_labels))
        self.register_buffer('weibull_beta', torch.ones(num_labels))
        self.register_buffer('weibull_tau', torch.zeros(num_labels))
        self.register_buffer('activation_vectors', torch.zeros(num_labels, hidden))
        self._flat = FlatParams(list(self.parameters()))

    def forward(self, x: torch.Tensor, use_openmax: bool = True) -> torch.Tensor:
        self._flat.ensure()
        logits, act = _MlpFn.apply(self, x, *self._flat.params)
        if use_openmax and not self.training:
            out = logits.detach().clone()
            # legacy rule: threshold 0.5, scale (1 - p)   (ref :381-386)
            return O.openmax_(act.detach().contiguous(), self.activation_vectors, self.weibull_alpha, self.weibull_beta,
                              self.weibull_tau, out, 0.5, 1.0)
        return logits

    def fit_weibull(self, activations: torch.Tensor, labels: torch.Tensor):
        feats, labs = activations.detach().float().cpu(), labels.detach().cpu()
        for c in range(self.num_labels):
            msk = labs == c
            if int(msk.sum()) == 0:
                continue
            cf = feats[msk]
            mu = cf.mean(dim=0)
            d = torch.norm(cf - mu, dim=1).numpy()
            self.activation_vectors[c] = mu.to(self.activation_vectors.device)
            self.weibull_alpha[c] = 2.0
            self.weibull_beta[c] = float(d.std())
            self.weibull_tau[c] = float(d.min())


class Classifier(nn.Module):
    """ref classifier.py:422-436."""

    def __init__(self, input_dim: int, num_labels: int, hidden: int = 128, p: float = 0.1):
        super().__init__()
        self.net = _legacy_net(input_dim, hidden, num_labels, p)
        self._flat = FlatParams(list(self.parameters()))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        self._flat.ensure()
        return _MlpFn.apply(self, x, *self._flat.params)[0]
