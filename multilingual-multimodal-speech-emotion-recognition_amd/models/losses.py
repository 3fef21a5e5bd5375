"""Losses — ref src/models/losses.py and the combination at src/train.py:154-168, on one HIP kernel.

`TrainLoss` is the fused criterion the build's train.py uses: value and all gradients come out of a
single launch of `ser_train_loss`.  `LabelSmoothingCrossEntropy` and `ClassBalancedFocalLoss` keep
the reference's class API (constructor + forward(logits, target)) by evaluating the same kernel
with the other terms' weights at zero.
"""
from typing import Optional

import torch
import torch.nn as nn

from .. import _ops as O


class _TrainLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, unc, fused, protos, labels, cfg):
        losses, dl, du, df, dp = O.train_loss(logits.contiguous(), unc.contiguous(), fused.contiguous(), protos.contiguous(),
                                              labels.contiguous(), **cfg)
        ctx.save_for_backward(dl, du, df, dp)
        ctx.mark_non_differentiable(losses)
        return losses[0], losses

    @staticmethod
    def backward(ctx, g, _):
        dl, du, df, dp = ctx.saved_tensors
        # upstream scale stays on the device (no .item()): the step can be captured into a hipGraph
        if not O._UNIT_LOSS_GRAD[0]:             # steppers: the seed is ones by construction (see _ops.unit_loss_grad)
            for t in (dl, du, df, dp):
                O.scale_dev_(t, g)
        return dl, du, df, dp, None, None


class TrainLoss(nn.Module):
    """CE_ls + 0.3 focal + 0.05 uncertainty term + 0.01 prototype loss (ref train.py:154-168)."""

    def __init__(self, num_classes: int, smoothing: float = 0.1, cb_beta: float = 0.9999, gamma: float = 2.0,
                 w_focal: float = 0.3, w_unc: float = 0.05, w_proto: float = 0.01, margin: float = 0.5):
        super().__init__()
        self.num_classes = num_classes
        self.cfg = dict(smoothing=smoothing, cb_beta=cb_beta, gamma=gamma, w_focal=w_focal, w_unc=w_unc, w_proto=w_proto,
                        margin=margin)

    def forward(self, logits, uncertainty, fused, prototypes, labels, use_proto: bool = True):
        cfg = dict(self.cfg, use_proto=use_proto)
        total, parts = _TrainLossFn.apply(logits, uncertainty, fused, prototypes, labels, cfg)
        self.last_parts = parts
        return total


def _single_term(logits, target, **kw):
    B, C = logits.shape
    dev = logits.device
    z1 = torch.zeros(B, 1, dtype=torch.float32, device=dev)
    zf = torch.zeros(B, 8, dtype=torch.float32, device=dev)
    zp = torch.zeros(C, 8, dtype=torch.float32, device=dev)
    cfg = dict(smoothing=0.0, cb_beta=0.9999, gamma=2.0, w_focal=0.0, w_unc=0.0, w_proto=0.0, margin=0.5, use_proto=False)
    cfg.update(kw)
    total, parts = _TrainLossFn.apply(logits, z1, zf, zp, target.long(), cfg)
    return total, parts


class LabelSmoothingCrossEntropy(nn.Module):
    def __init__(self, smoothing: float = 0.1):
        super().__init__()
        self.smoothing = smoothing

    def forward(self, logits: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        return _single_term(logits, target, smoothing=self.smoothing)[0]


class _FocalOnlyFn(torch.autograd.Function):
    """focal = (total - ce) with w_focal = 1; both pieces come from the kernel, the subtraction of the CE
    gradient is done by running the kernel twice (w_focal = 1 and w_focal = 0) and one axpby."""

    @staticmethod
    def forward(ctx, logits, target, beta, gamma):
        B, C = logits.shape
        dev = logits.device
        z1 = torch.zeros(B, 1, dtype=torch.float32, device=dev)
        zf = torch.zeros(B, 8, dtype=torch.float32, device=dev)
        zp = torch.zeros(C, 8, dtype=torch.float32, device=dev)
        kw = dict(smoothing=0.0, cb_beta=beta, gamma=gamma, w_unc=0.0, w_proto=0.0, margin=0.5, use_proto=False)
        l1, d1, *_ = O.train_loss(logits.contiguous(), z1, zf, zp, target.long().contiguous(), w_focal=1.0, **kw)
        l0, d0, *_ = O.train_loss(logits.contiguous(), z1, zf, zp, target.long().contiguous(), w_focal=0.0, **kw)
        O.axpby(d0, d1, -1.0, 1.0)
        ctx.save_for_backward(d1)
        return l1[2].clone()

    @staticmethod
    def backward(ctx, g):
        (d1,) = ctx.saved_tensors
        O.scale_dev_(d1, g)
        return d1, None, None, None


class ClassBalancedFocalLoss(nn.Module):
    def __init__(self, beta: float = 0.9999, gamma: float = 2.0, num_classes: Optional[int] = None):
        super().__init__()
        self.beta, self.gamma, self.num_classes = beta, gamma, num_classes
        self.register_buffer("effective_num", torch.tensor(1.0))

    def forward(self, logits: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
        if self.num_classes is None:
            raise NotImplementedError("un-weighted focal variant (num_classes=None) is not on the hot path")
        return _FocalOnlyFn.apply(logits, targets, self.beta, self.gamma)


class SupConLoss(nn.Module):
    """Constructed but never called by the reference's train.py (:86); kept for API completeness."""

    def __init__(self, temperature: float = 0.07):
        super().__init__()
        self.temperature = temperature

    def forward(self, features, labels):
        raise NotImplementedError("SupConLoss is never called on the reference's training path (train.py:86)")
