"""FusionLayer (gated fusion) — drop-in for ref src/models/fusion.py:5-25 on HIP kernels."""
import torch
import torch.nn as nn

from .. import _ops as O
from ._flat import FlatParams


def _branch_fwd(x, proj, gate, site):
    h = O.linear_fwd(x, proj[0].weight, proj[0].bias, O.ACT_RELU)
    dctx = O.dropout_ctx(proj[2].p)                               # ref fusion.py:9,12 (training mode only)
    O.dropout_(h, dctx, site)
    a = O.linear_fwd(h, proj[3].weight, proj[3].bias)
    gh = O.linear_fwd(a, gate[0].weight, gate[0].bias, O.ACT_RELU)
    gl = O.linear_fwd(gh, gate[2].weight, gate[2].bias)          # [B,1] gate logit
    return a, gl, (x, h, gh, dctx, site)


def _branch_bwd(da, dgl, a, saved, proj, gate, g, acc, need_dx):
    x, h, gh, dctx, site = saved
    O.linear_wgrad(dgl, gh, g(gate[2].weight), g(gate[2].bias), acc)
    dgh = O.linear_dgrad(dgl, gate[2].weight)
    O.act_bwd(dgh, gh, O.ACT_RELU)
    O.linear_wgrad(dgh, a, g(gate[0].weight), g(gate[0].bias), acc)
    O.linear_dgrad(dgh, gate[0].weight, out=da, accumulate=True)       # a feeds both the mix and its own gate
    O.linear_wgrad(da, h, g(proj[3].weight), g(proj[3].bias), acc)
    dh = O.linear_dgrad(da, proj[3].weight)
    O.act_bwd(dh, h, O.ACT_RELU, dctx=dctx, site=site)
    O.linear_wgrad(dh, x, g(proj[0].weight), g(proj[0].bias), acc)
    return O.linear_dgrad(dh, proj[0].weight) if need_dx else None


class _FusionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, m, av, tv, *params):
        av, tv = av.contiguous(), tv.contiguous()
        a, ga, sa = _branch_fwd(av, m.proj_a, m.gate_a, m._drop_sites[0])
        t, gt, st = _branch_fwd(tv, m.proj_t, m.gate_t, m._drop_sites[1])
        out = O.fusion_mix_fwd(a, t, ga, gt)
        ctx.m, ctx.saved = m, (a, t, ga, gt, sa, st)
        ctx.need = (av.requires_grad, tv.requires_grad)
        return out

    @staticmethod
    def backward(ctx, dout):
        m = ctx.m
        a, t, ga, gt, sa, st = ctx.saved
        fp = m._flat
        acc = fp.accumulating()
        da, dt, dga, dgt = O.fusion_mix_bwd(dout.contiguous(), a, t, ga, gt)
        dav = _branch_bwd(da, dga, a, sa, m.proj_a, m.gate_a, fp.gview, acc, ctx.need[0])
        dtv = _branch_bwd(dt, dgt, t, st, m.proj_t, m.gate_t, fp.gview, acc, ctx.need[1])
        fp.publish()
        ctx.saved = None
        return (None, dav, dtv) + (None,) * len(fp.params)


class FusionLayer(nn.Module):
    def __init__(self, audio_dim: int, text_dim: int, proj_dim: int):
        super().__init__()
        self.proj_a = nn.Sequential(nn.Linear(audio_dim, proj_dim), nn.ReLU(), nn.Dropout(0.1), nn.Linear(proj_dim, proj_dim))
        self.proj_t = nn.Sequential(nn.Linear(text_dim, proj_dim), nn.ReLU(), nn.Dropout(0.1), nn.Linear(proj_dim, proj_dim))
        gate_hidden = max(32, proj_dim // 2)
        self.gate_a = nn.Sequential(nn.Linear(proj_dim, gate_hidden), nn.ReLU(), nn.Linear(gate_hidden, 1))
        self.gate_t = nn.Sequential(nn.Linear(proj_dim, gate_hidden), nn.ReLU(), nn.Linear(gate_hidden, 1))
        self._flat = FlatParams(list(self.parameters()))
        self._drop_sites = (O.new_dropout_site(), O.new_dropout_site())

    def forward(self, audio_vec: torch.Tensor, text_vec: torch.Tensor) -> torch.Tensor:
        self._flat.ensure()
        return _FusionFn.apply(self, audio_vec, text_vec, *self._flat.params)
