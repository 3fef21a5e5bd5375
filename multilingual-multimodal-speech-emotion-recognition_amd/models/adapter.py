"""Residual bottleneck adapter  x + W2 relu(W1 x + b1) + b2  (ref src/models/audio_encoder.py:19-21,112;
text_encoder.py:17-19,57), forward and backward on the fp32 MFMA GEMM."""
import torch

from .. import _ops as O
from ._flat import FlatParams


class _AdapterFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, owner, x, *params):
        w1, b1, w2, b2 = owner.adapter[0].weight, owner.adapter[0].bias, owner.adapter[2].weight, owner.adapter[2].bias
        shp = x.shape
        x2 = x.reshape(-1, shp[-1]).contiguous()
        h = O.linear_fwd(x2, w1, b1, O.ACT_RELU)
        y = O.linear_fwd(h, w2, b2, O.ACT_NONE, residual=x2)
        ctx.owner, ctx.shape = owner, shp
        ctx.save_for_backward(x2, h)
        ctx.need_dx = x.requires_grad
        return y.view(shp)

    @staticmethod
    def backward(ctx, dy):
        owner = ctx.owner
        x2, h = ctx.saved_tensors
        fp = owner._adapter_flat
        acc = fp.accumulating()
        w1, w2 = owner.adapter[0].weight, owner.adapter[2].weight
        dy2 = dy.reshape(-1, dy.shape[-1]).contiguous()
        O.linear_wgrad(dy2, h, fp.gview(w2), fp.gview(owner.adapter[2].bias), acc)
        dh = O.linear_dgrad(dy2, w2)
        O.act_bwd(dh, h, O.ACT_RELU)
        O.linear_wgrad(dh, x2, fp.gview(w1), fp.gview(owner.adapter[0].bias), acc)
        dx = None
        if ctx.need_dx:
            dx = dy2.clone()
            O.linear_dgrad(dh, w1, out=dx, accumulate=True)
            dx = dx.view(ctx.shape)
        fp.publish()
        return (None, dx) + (None,) * 4


def adapter_apply(owner, x):
    """owner has `.adapter` = Sequential(Linear, ReLU, Linear)."""
    if not hasattr(owner, "_adapter_flat"):
        owner._adapter_flat = FlatParams(list(owner.adapter.parameters()))
    owner._adapter_flat.ensure()
    return _AdapterFn.apply(owner, x, *owner.adapter.parameters())
