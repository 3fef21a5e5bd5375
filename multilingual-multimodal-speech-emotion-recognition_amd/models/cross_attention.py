"""CrossModalAttention — drop-in for ref src/models/cross_attention.py:6-53 on HIP kernels.

Same constructor, attributes and state_dict keys (q_a, k_t, v_t, attn_a.{in_proj_weight,in_proj_bias,
out_proj.*}, out_a, q_t, k_a, v_a, attn_t.*, out_t, norm_a, norm_t).  `attn_a` / `attn_t` stay
nn.MultiheadAttention objects so that checkpoints interchange, but they are never called: the
projections run on the fp32 MFMA GEMM and softmax(QK^T)V on the xattn kernels.  Dropout (attention
probabilities and block outputs) is active inside `_ops.dropout_scope` (training steps of SERSystem) and
the identity otherwise, which is the parity definition of the build.
"""
from typing import Optional

import torch
import torch.nn as nn

from .. import _ops as O
from ._flat import FlatParams


_SIDE = {}


def _side_stream():
    dev = torch.cuda.current_device()
    if dev not in _SIDE:
        _SIDE[dev] = torch.cuda.Stream()
    return _SIDE[dev]


def _dir_fwd(x_q, x_kv, kv_mask, B, Sq, Sk, Wq, bq, Wk, bk, Wv, bv, Wi, bi, Wo, bo, Wout, bout, ng, nb, heads, p_out=0.0, site=0, p_attn=0.0, site_attn=0):
    E = Wq.shape[0]
    q1 = O.linear_fwd(x_q, Wq, bq)
    k1 = O.linear_fwd(x_kv, Wk, bk)
    v1 = O.linear_fwd(x_kv, Wv, bv)
    Q = O.linear_fwd(q1, Wi[:E], bi[:E])
    K = O.linear_fwd(k1, Wi[E:2 * E], bi[E:2 * E])
    V = O.linear_fwd(v1, Wi[2 * E:], bi[2 * E:])
    dattn = O.dropout_ctx(p_attn)                               # nn.MultiheadAttention(dropout=p), ref cross_attention.py:18,25
    ctx, P = O.xattn_fwd(Q, K, V, kv_mask, B, Sq, Sk, heads, dattn, site_attn)
    c2 = O.linear_fwd(ctx, Wo, bo)
    o = O.linear_fwd(c2, Wout, bout)
    dout = O.dropout_ctx(p_out)                                  # ref cross_attention.py:43,51: self.dropout(out)
    O.dropout_(o, dout, site)
    y, ln = O.ln_fwd(x_q, ng, nb, 1e-5, x2=o)
    return y, (q1, k1, v1, Q, K, V, P, ctx, c2, ln, dout, site, dattn, site_attn)


def _dir_bwd(dy, saved, x_q, x_kv, B, Sq, Sk, Wq, Wk, Wv, Wi, Wo, Wout, ng, g, acc, heads, dx_q, dx_kv):
    """Writes the input gradients into dx_q / dx_kv (uninitialised on entry) and the parameter grads via g(param)."""
    q1, k1, v1, Q, K, V, P, ctx, c2, ln, dout, site, dattn, site_attn = saved
    E = Wq["w"].shape[0]
    dz = O.ln_bwd(dy, ln, ng["w"], g(ng["w"]), g(ng["b"]), acc)
    O.axpby(dz, dx_q, 1.0, 0.0)                                  # residual branch: first write of dx_q (no zero fill needed)
    O.dropout_(dz, dout, site)                                   # from here on dz is the gradient at the dropped branch
    O.linear_wgrad(dz, c2, g(Wout["w"]), g(Wout["b"]), acc)
    dc2 = O.linear_dgrad(dz, Wout["w"])
    O.linear_wgrad(dc2, ctx, g(Wo["w"]), g(Wo["b"]), acc)
    dctx = O.linear_dgrad(dc2, Wo["w"])
    dQ, dK, dV = O.xattn_bwd(dctx, Q, K, V, P, B, Sq, Sk, heads, dattn, site_attn)
    gWi, gbi = g(Wi["w"]), g(Wi["b"])
    O.linear_wgrad(dQ, q1, gWi[:E], gbi[:E], acc)
    O.linear_wgrad(dK, k1, gWi[E:2 * E], gbi[E:2 * E], acc)
    O.linear_wgrad(dV, v1, gWi[2 * E:], gbi[2 * E:], acc)
    dq1 = O.linear_dgrad(dQ, Wi["w"][:E])
    dk1 = O.linear_dgrad(dK, Wi["w"][E:2 * E])
    dv1 = O.linear_dgrad(dV, Wi["w"][2 * E:])
    O.linear_wgrad(dq1, x_q, g(Wq["w"]), g(Wq["b"]), acc)
    O.linear_wgrad(dk1, x_kv, g(Wk["w"]), g(Wk["b"]), acc)
    O.linear_wgrad(dv1, x_kv, g(Wv["w"]), g(Wv["b"]), acc)
    O.linear_dgrad(dq1, Wq["w"], out=dx_q, accumulate=True)
    O.linear_dgrad(dk1, Wk["w"], out=dx_kv, accumulate=False)          # first write of dx_kv
    O.linear_dgrad(dv1, Wv["w"], out=dx_kv, accumulate=True)


class _CrossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, m, a, t, a_mask, t_mask, *params):
        B, Sa, Da = a.shape
        St, Dt = t.shape[1], t.shape[2]
        a2, t2 = a.reshape(B * Sa, Da).contiguous(), t.reshape(B * St, Dt).contiguous()
        am = a_mask.to(torch.float32).contiguous() if a_mask is not None else None
        tm = t_mask.to(torch.float32).contiguous() if t_mask is not None else None
        with O.fork(_side_stream()) as f:     # T <- A is independent of A <- T
            yt, st = _dir_fwd(t2, a2, am, B, St, Sa, m.q_t.weight, m.q_t.bias, m.k_a.weight, m.k_a.bias, m.v_a.weight,
                              m.v_a.bias, m.attn_t.in_proj_weight, m.attn_t.in_proj_bias, m.attn_t.out_proj.weight,
                              m.attn_t.out_proj.bias, m.out_t.weight, m.out_t.bias, m.norm_t.weight, m.norm_t.bias,
                              m.num_heads, m.dropout.p, m._drop_sites[1], m.attn_t.dropout, m._drop_sites[3])
        ya, sa = _dir_fwd(a2, t2, tm, B, Sa, St, m.q_a.weight, m.q_a.bias, m.k_t.weight, m.k_t.bias, m.v_t.weight,
                          m.v_t.bias, m.attn_a.in_proj_weight, m.attn_a.in_proj_bias, m.attn_a.out_proj.weight,
                          m.attn_a.out_proj.bias, m.out_a.weight, m.out_a.bias, m.norm_a.weight, m.norm_a.bias, m.num_heads,
                          m.dropout.p, m._drop_sites[0], m.attn_a.dropout, m._drop_sites[2])
        f.join(produced=[yt], consumed=[a2, t2, am])
        ctx.m, ctx.dims = m, (B, Sa, St, Da, Dt)
        ctx.sa, ctx.st, ctx.a2, ctx.t2 = sa, st, a2, t2
        return ya.view(B, Sa, Da), yt.view(B, St, Dt)

    @staticmethod
    def backward(ctx, dya, dyt):
        m = ctx.m
        B, Sa, St, Da, Dt = ctx.dims
        fp = m._flat
        acc = fp.accumulating()
        g = fp.gview
        P = lambda mod: {"w": mod.weight, "b": mod.bias}
        PI = lambda mha: {"w": mha.in_proj_weight, "b": mha.in_proj_bias}
        dev = dya.device
        da = torch.empty(B * Sa, Da, dtype=torch.float32, device=dev)       # every buffer is fully written by its first use
        dt = torch.empty(B * St, Dt, dtype=torch.float32, device=dev)
        da2 = torch.empty(B * Sa, Da, dtype=torch.float32, device=dev)      # contributions of the other direction
        dt2 = torch.empty(B * St, Dt, dtype=torch.float32, device=dev)
        dya2, dyt2 = dya.reshape(B * Sa, Da).contiguous(), dyt.reshape(B * St, Dt).contiguous()
        with O.fork(_side_stream()) as f:     # the two directions touch disjoint parameters and disjoint gradient buffers
            _dir_bwd(dyt2, ctx.st, ctx.t2, ctx.a2, B, St, Sa, P(m.q_t), P(m.k_a), P(m.v_a),
                     PI(m.attn_t), P(m.attn_t.out_proj), P(m.out_t), P(m.norm_t), g, acc, m.num_heads, dt, da2)
        _dir_bwd(dya2, ctx.sa, ctx.a2, ctx.t2, B, Sa, St, P(m.q_a), P(m.k_t), P(m.v_t),
                 PI(m.attn_a), P(m.attn_a.out_proj), P(m.out_a), P(m.norm_a), g, acc, m.num_heads, da, dt2)
        f.join(consumed=[dyt2, dt, da2])
        O.axpby(da2, da, 1.0, 1.0)
        O.axpby(dt2, dt, 1.0, 1.0)
        fp.publish()
        ctx.sa = ctx.st = None
        return (None, da.view(B, Sa, Da), dt.view(B, St, Dt), None, None) + (None,) * len(fp.params)


class CrossModalAttention(nn.Module):
    def __init__(self, audio_dim: int, text_dim: int, shared_dim: int = 256, num_heads: int = 8, dropout: float = 0.1):
        super().__init__()
        self.shared_dim = shared_dim
        self.num_heads = num_heads
        assert shared_dim % num_heads == 0, f"shared_dim {shared_dim} must be divisible by num_heads {num_heads}"
        self.q_a = nn.Linear(audio_dim, shared_dim)
        self.k_t = nn.Linear(text_dim, shared_dim)
        self.v_t = nn.Linear(text_dim, shared_dim)
        self.attn_a = nn.MultiheadAttention(shared_dim, num_heads, dropout=dropout, batch_first=True)
        self.out_a = nn.Linear(shared_dim, audio_dim)
        self.q_t = nn.Linear(text_dim, shared_dim)
        self.k_a = nn.Linear(audio_dim, shared_dim)
        self.v_a = nn.Linear(audio_dim, shared_dim)
        self.attn_t = nn.MultiheadAttention(shared_dim, num_heads, dropout=dropout, batch_first=True)
        self.out_t = nn.Linear(shared_dim, text_dim)
        self.dropout = nn.Dropout(dropout)
        self.norm_a = nn.LayerNorm(audio_dim)
        self.norm_t = nn.LayerNorm(text_dim)
        self._flat = FlatParams(list(self.parameters()))
        self._drop_sites = tuple(O.new_dropout_site() for _ in range(4))   # output a, output t, attention a, attention t

    def forward(self, audio_seq: torch.Tensor, text_seq: torch.Tensor, audio_mask: Optional[torch.Tensor] = None,
                text_mask: Optional[torch.Tensor] = None):
        self._flat.ensure()
        return _CrossFn.apply(self, audio_seq, text_seq, audio_mask, text_mask, *self._flat.params)
