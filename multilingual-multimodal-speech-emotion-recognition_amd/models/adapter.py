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


class _AdapterPairFn(torch.autograd.Function):
    """The audio and the text adapter (independent: ref audio_encoder.py:112, text_encoder.py:57) as one autograd node, each
    of their two levels ONE grouped launch (`O.linear_fwd_group`).  Same kernels and results as two `_AdapterFn` calls."""

    @staticmethod
    def forward(ctx, oa, ot, xa, xt, *params):
        shp_a, shp_t = xa.shape, xt.shape
        xa2, xt2 = xa.reshape(-1, shp_a[-1]).contiguous(), xt.reshape(-1, shp_t[-1]).contiguous()
        A, T = oa.adapter, ot.adapter
        ha, ht = O.linear_fwd_group([(xa2, A[0].weight, A[0].bias, O.ACT_RELU, None), (xt2, T[0].weight, T[0].bias, O.ACT_RELU, None)])
        ya, yt = O.linear_fwd_group([(ha, A[2].weight, A[2].bias, O.ACT_NONE, xa2), (ht, T[2].weight, T[2].bias, O.ACT_NONE, xt2)])
        ctx.owners, ctx.shapes = (oa, ot), (shp_a, shp_t)
        ctx.save_for_backward(xa2, ha, xt2, ht)
        ctx.need_dx = (xa.requires_grad, xt.requires_grad)
        return ya.view(shp_a), yt.view(shp_t)

    @staticmethod
    def backward(ctx, dya, dyt):
        oa, ot = ctx.owners
        xa2, ha, xt2, ht = ctx.saved_tensors
        outs = []
        dys, dhs = [], []
        for owner, dy in ((oa, dya), (ot, dyt)):
            dys.append(dy.reshape(-1, dy.shape[-1]).contiguous())
        for owner, dy2, h in ((oa, dys[0], ha), (ot, dys[1], ht)):
            fp = owner._adapter_flat
            O.linear_wgrad(dy2, h, fp.gview(owner.adapter[2].weight), fp.gview(owner.adapter[2].bias), fp.accumulating())
        dha, dht = O.linear_dgrad_group([(dys[0], oa.adapter[2].weight, None, False), (dys[1], ot.adapter[2].weight, None, False)])
        for owner, dh, h, x2 in ((oa, dha, ha, xa2), (ot, dht, ht, xt2)):
            fp = owner._adapter_flat
            acc = fp.accumulating()
            O.act_bwd(dh, h, O.ACT_RELU)
            O.linear_wgrad(dh, x2, fp.gview(owner.adapter[0].weight), fp.gview(owner.adapter[0].bias), acc)
        for k, (owner, dy2, dh, shp) in enumerate(((oa, dys[0], dha, ctx.shapes[0]), (ot, dys[1], dht, ctx.shapes[1]))):
            dx = None
            if ctx.need_dx[k]:
                dx = dy2.clone()
                O.linear_dgrad(dh, owner.adapter[0].weight, out=dx, accumulate=True)
                dx = dx.view(shp)
            outs.append(dx)
        oa._adapter_flat.publish()
        ot._adapter_flat.publish()
        return (None, None, outs[0], outs[1]) + (None,) * 8


def adapters_apply(audio_owner, text_owner, a_enc, t_enc):
    """Both residual bottleneck adapters -> (a_seq, t_seq)."""
    for owner in (audio_owner, text_owner):
        if not hasattr(owner, "_adapter_flat"):
            owner._adapter_flat = FlatParams(list(owner.adapter.parameters()))
        owner._adapter_flat.ensure()
    return _AdapterPairFn.apply(audio_owner, text_owner, a_enc, t_enc, *audio_owner.adapter.parameters(), *text_owner.adapter.parameters())
