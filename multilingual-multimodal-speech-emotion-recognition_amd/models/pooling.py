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


class AttentiveStatsPooling(nn.Module):
    def __init__(self, input_dim: int, hidden_dim: int = 128):
        super().__init__()
        self.attention = nn.Sequential(nn.Linear(input_dim, hidden_dim), nn.Tanh(), nn.Linear(hidden_dim, 1))
        self._flat = FlatParams(list(self.parameters()))

    def forward(self, x: torch.Tensor, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        self._flat.ensure()
        return _PoolFn.apply(self, x, mask, *self._flat.params)
