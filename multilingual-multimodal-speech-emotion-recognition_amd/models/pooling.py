"""AttentiveStatsPooling — drop-in for ref src/models/pooling.py:6-28 on HIP kernels."""
from typing import Optional

import torch
import torch.nn as nn

from .. import _ops as O
from ._flat import FlatParams


class _PoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, m, x, mask, *params):
        B, S, D = x.shape
        xc = x.contiguous()
        x2 = xc.view(B * S, D)
        w1, b1, w2, b2 = m.attention[0].weight, m.attention[0].bias, m.attention[2].weight, m.attention[2].bias
        h = O.linear_fwd(x2, w1, b1, O.ACT_TANH)
        logit = O.linear_fwd(h, w2, b2)
        mk = mask.to(torch.float32).contiguous() if mask is not None else None
        out, alpha = O.pool_fwd(xc, logit, mk)
        ctx.m = m
        ctx.save_for_backward(xc, h, alpha, out)
        return out

    @staticmethod
    def backward(ctx, dout):
        m = ctx.m
        xc, h, alpha, out = ctx.saved_tensors
        B, S, D = xc.shape
        fp = m._flat
        acc = fp.accumulating()
        w1, w2 = m.attention[0].weight, m.attention[2].weight
        dx, dlogit = O.pool_bwd(dout.contiguous(), xc, alpha, out)
        dl = dlogit.view(B * S, 1)
        O.linear_wgrad(dl, h, fp.gview(w2), fp.gview(m.attention[2].bias), acc)
        dh = O.linear_dgrad(dl, w2)
        O.act_bwd(dh, h, O.ACT_TANH)
        x2 = xc.view(B * S, D)
        O.linear_wgrad(dh, x2, fp.gview(w1), fp.gview(m.attention[0].bias), acc)
        O.linear_dgrad(dh, w1, out=dx.view(B * S, D), accumulate=True)
        fp.publish()
        return (None, dx, None) + (None,) * len(fp.params)


class _PoolPairFn(torch.autograd.Function):
    """pool_a and pool_t (independent: ref src/train.py:148-149) as one autograd node with grouped launches per level; same
    kernels and results as two `_PoolFn` calls."""

    @staticmethod
    def forward(ctx, ma, mt, xa, mka, xt, mkt, *params):
        xs, hs, alphas, outs = [], [], [], []
        for x in (xa, xt):
            xs.append(x.contiguous())
        (Ba, Sa, Da), (Bt, St, Dt) = xs[0].shape, xs[1].shape
        x2a, x2t = xs[0].view(Ba * Sa, Da), xs[1].view(Bt * St, Dt)
        Aa, At = ma.attention, mt.attention
        ha, ht = O.linear_fwd_group([(x2a, Aa[0].weight, Aa[0].bias, O.ACT_TANH, None), (x2t, At[0].weight, At[0].bias, O.ACT_TANH, None)])
        la, lt = O.linear_fwd_group([(ha, Aa[2].weight, Aa[2].bias, O.ACT_NONE, None), (ht, At[2].weight, At[2].bias, O.ACT_NONE, None)])
        res = []
        for xc, logit, mask in ((xs[0], la, mka), (xs[1], lt, mkt)):
            mk = mask.to(torch.float32).contiguous() if mask is not None else None
            out, alpha = O.pool_fwd(xc, logit, mk)
            res.append((out, alpha))
        ctx.ms = (ma, mt)
        ctx.save_for_backward(xs[0], ha, res[0][1], res[0][0], xs[1], ht, res[1][1], res[1][0])
        return res[0][0], res[1][0]

    @staticmethod
    def backward(ctx, douta, doutt):
        ma, mt = ctx.ms
        xa, ha, alpha_a, out_a, xt, ht, alpha_t, out_t = ctx.saved_tensors
        items = []
        for m, dout, xc, h, alpha, out in ((ma, douta, xa, ha, alpha_a, out_a), (mt, doutt, xt, ht, alpha_t, out_t)):
            B, S, D = xc.shape
            fp = m._flat
            acc = fp.accumulating()
            dx, dlogit = O.pool_bwd(dout.contiguous(), xc, alpha, out)
            dl = dlogit.view(B * S, 1)
            O.linear_wgrad(dl, h, fp.gview(m.attention[2].weight), fp.gview(m.attention[2].bias), acc)
            items.append((m, fp, acc, xc, h, dx, dl))
        dha, dht = O.linear_dgrad_group([(items[0][6], ma.attention[2].weight, None, False), (items[1][6], mt.attention[2].weight, None, False)])
        grp = []
        for (m, fp, acc, xc, h, dx, dl), dh in zip(items, (dha, dht)):
            B, S, D = xc.shape
            O.act_bwd(dh, h, O.ACT_TANH)
            O.linear_wgrad(dh, xc.view(B * S, D), fp.gview(m.attention[0].weight), fp.gview(m.attention[0].bias), acc)
            grp.append((dh, m.attention[0].weight, dx.view(B * S, D), True))
        O.linear_dgrad_group(grp)
        ma._flat.publish()
        mt._flat.publish()
        return (None, None, items[0][5], None, items[1][5], None) + (None,) * 8


def pools_apply(pool_a, pool_t, a_enh, a_mask, t_enh, t_mask):
    """Both attentive-statistics poolings -> (a_vec, t_vec)."""
    pool_a._flat.ensure()
    pool_t._flat.ensure()
    return _PoolPairFn.apply(pool_a, pool_t, a_enh, a_mask, t_enh, t_mask, *pool_a._flat.params, *pool_t._flat.params)


class AttentiveStatsPooling(nn.Module):
    def __init__(self, input_dim: int, hidden_dim: int = 128):
        super().__init__()
        self.attention = nn.Sequential(nn.Linear(input_dim, hidden_dim), nn.Tanh(), nn.Linear(hidden_dim, 1))
        self._flat = FlatParams(list(self.parameters()))

    def forward(self, x: torch.Tensor, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        self._flat.ensure()
        return _PoolFn.apply(self, x, mask, *self._flat.params)
