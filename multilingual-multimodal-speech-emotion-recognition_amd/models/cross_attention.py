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


def _dirs(m):
    """Per direction (A <- T, T <- A): the Linear / attention / norm modules of ref cross_attention.py:15-26."""
    return ((m.q_a, m.k_t, m.v_t, m.attn_a, m.out_a, m.norm_a), (m.q_t, m.k_a, m.v_a, m.attn_t, m.out_t, m.norm_t))


class _CrossGroupedFn(torch.autograd.Function):
    """Both directions of ref cross_attention.py:32-53 as ONE autograd node whose independent products go out level by level
    in grouped launches (`O.linear_fwd_group` / `O.linear_dgrad_group`): 6 + 6 + 2 + 2 Linear layers forward are four launches,
    not sixteen on two streams.  (Round 2 ran the two directions on two streams; a join between two hardware queues costs
    50-200 us under load, more than the kernels it overlapped: profiles/r03_a_head_step_kernel_sequence.txt.)"""

    @staticmethod
    def forward(ctx, m, a, t, a_mask, t_mask, *params):
        B, Sa, Da = a.shape
        St, Dt = t.shape[1], t.shape[2]
        a2, t2 = a.reshape(B * Sa, Da).contiguous(), t.reshape(B * St, Dt).contiguous()
        am = a_mask.to(torch.float32).contiguous() if a_mask is not None else None
        tm = t_mask.to(torch.float32).contiguous() if t_mask is not None else None
        (qa, kt, vt, mha_a, out_a, norm_a), (qt, ka, va, mha_t, out_t, norm_t) = _dirs(m)
        E = qa.weight.shape[0]
        N_ = O.ACT_NONE
        # level 1: the six projections into the shared space (ref :15-17, :22-24)
        q1a, k1t, v1t, q1t, k1a, v1a = O.linear_fwd_group([
            (a2, qa.weight, qa.bias, N_, None), (t2, kt.weight, kt.bias, N_, None), (t2, vt.weight, vt.bias, N_, None),
            (t2, qt.weight, qt.bias, N_, None), (a2, ka.weight, ka.bias, N_, None), (a2, va.weight, va.bias, N_, None)])
        # level 2: nn.MultiheadAttention's packed in-projection (torch nn/functional.py:6576-6603)
        Wa, ba, Wt, bt = mha_a.in_proj_weight, mha_a.in_proj_bias, mha_t.in_proj_weight, mha_t.in_proj_bias
        Qa, Ka, Va, Qt, Kt, Vt = O.linear_fwd_group([
            (q1a, Wa[:E], ba[:E], N_, None), (k1t, Wa[E:2 * E], ba[E:2 * E], N_, None), (v1t, Wa[2 * E:], ba[2 * E:], N_, None),
            (q1t, Wt[:E], bt[:E], N_, None), (k1a, Wt[E:2 * E], bt[E:2 * E], N_, None), (v1a, Wt[2 * E:], bt[2 * E:], N_, None)])
        sites = m._drop_sites                       # output a, output t, attention a, attention t
        d_attn_a, d_attn_t = O.dropout_ctx(mha_a.dropout), O.dropout_ctx(mha_t.dropout)      # ref :18,25
        ctx_a, Pa = O.xattn_fwd(Qa, Ka, Va, tm, B, Sa, St, m.num_heads, d_attn_a, sites[2])
        ctx_t, Pt = O.xattn_fwd(Qt, Kt, Vt, am, B, St, Sa, m.num_heads, d_attn_t, sites[3])
        c2a, c2t = O.linear_fwd_group([(ctx_a, mha_a.out_proj.weight, mha_a.out_proj.bias, N_, None),
                                       (ctx_t, mha_t.out_proj.weight, mha_t.out_proj.bias, N_, None)])
        oa, ot = O.linear_fwd_group([(c2a, out_a.weight, out_a.bias, N_, None), (c2t, out_t.weight, out_t.bias, N_, None)])
        d_out = O.dropout_ctx(m.dropout.p)                                                   # ref :43,51: self.dropout(out)
        O.dropout_(oa, d_out, sites[0])
        O.dropout_(ot, d_out, sites[1])
        ya, lna = O.ln_fwd(a2, norm_a.weight, norm_a.bias, 1e-5, x2=oa)
        yt, lnt = O.ln_fwd(t2, norm_t.weight, norm_t.bias, 1e-5, x2=ot)
        ctx.m, ctx.dims = m, (B, Sa, St, Da, Dt)
        ctx.saved = (a2, t2, q1a, k1t, v1t, q1t, k1a, v1a, Qa, Ka, Va, Qt, Kt, Vt, Pa, Pt, ctx_a, ctx_t, c2a, c2t, lna, lnt,
                     d_attn_a, d_attn_t, d_out)
        return ya.view(B, Sa, Da), yt.view(B, St, Dt)

    @staticmethod
    def backward(ctx, dya, dyt):
        m = ctx.m
        B, Sa, St, Da, Dt = ctx.dims
        (a2, t2, q1a, k1t, v1t, q1t, k1a, v1a, Qa, Ka, Va, Qt, Kt, Vt, Pa, Pt, ctx_a, ctx_t, c2a, c2t, lna, lnt,
         d_attn_a, d_attn_t, d_out) = ctx.saved
        (qa, kt, vt, mha_a, out_a, norm_a), (qt, ka, va, mha_t, out_t, norm_t) = _dirs(m)
        fp = m._flat
        acc = fp.accumulating()
        g = fp.gview
        E = qa.weight.shape[0]
        sites = m._drop_sites
        dya2, dyt2 = dya.reshape(B * Sa, Da).contiguous(), dyt.reshape(B * St, Dt).contiguous()
        # LayerNorm backward; its input gradient is both the residual branch (first term of da / dt) and, after the output
        # dropout, the gradient of the block output
        dza = O.ln_bwd(dya2, lna, norm_a.weight, g(norm_a.weight), g(norm_a.bias), acc)
        dzt = O.ln_bwd(dyt2, lnt, norm_t.weight, g(norm_t.weight), g(norm_t.bias), acc)
        # (a copy, always: dza / dzt stay operands of weight-gradient launches that may be deferred to the end of backward,
        # while da / dt are accumulated into below)
        da, dt = dza.clone(), dzt.clone()
        O.dropout_(dza, d_out, sites[0])
        O.dropout_(dzt, d_out, sites[1])
        O.linear_wgrad(dza, c2a, g(out_a.weight), g(out_a.bias), acc)
        O.linear_wgrad(dzt, c2t, g(out_t.weight), g(out_t.bias), acc)
        dc2a, dc2t = O.linear_dgrad_group([(dza, out_a.weight, None, False), (dzt, out_t.weight, None, False)])
        O.linear_wgrad(dc2a, ctx_a, g(mha_a.out_proj.weight), g(mha_a.out_proj.bias), acc)
        O.linear_wgrad(dc2t, ctx_t, g(mha_t.out_proj.weight), g(mha_t.out_proj.bias), acc)
        dctx_a, dctx_t = O.linear_dgrad_group([(dc2a, mha_a.out_proj.weight, None, False), (dc2t, mha_t.out_proj.weight, None, False)])
        dQa, dKa, dVa = O.xattn_bwd(dctx_a, Qa, Ka, Va, Pa, B, Sa, St, m.num_heads, d_attn_a, sites[2])
        dQt, dKt, dVt = O.xattn_bwd(dctx_t, Qt, Kt, Vt, Pt, B, St, Sa, m.num_heads, d_attn_t, sites[3])
        Wa, Wt = mha_a.in_proj_weight, mha_t.in_proj_weight
        gWa, gba, gWt, gbt = g(Wa), g(mha_a.in_proj_bias), g(Wt), g(mha_t.in_proj_bias)
        for dX, x1, gW, gb, lo in ((dQa, q1a, gWa, gba, 0), (dKa, k1t, gWa, gba, E), (dVa, v1t, gWa, gba, 2 * E),
                                   (dQt, q1t, gWt, gbt, 0), (dKt, k1a, gWt, gbt, E), (dVt, v1a, gWt, gbt, 2 * E)):
            O.linear_wgrad(dX, x1, gW[lo:lo + E], gb[lo:lo + E], acc)
        dq1a, dk1t, dv1t, dq1t, dk1a, dv1a = O.linear_dgrad_group([
            (dQa, Wa[:E], None, False), (dKa, Wa[E:2 * E], None, False), (dVa, Wa[2 * E:], None, False),
            (dQt, Wt[:E], None, False), (dKt, Wt[E:2 * E], None, False), (dVt, Wt[2 * E:], None, False)])
        for dX, x, lin in ((dq1a, a2, qa), (dk1t, t2, kt), (dv1t, t2, vt), (dq1t, t2, qt), (dk1a, a2, ka), (dv1a, a2, va)):
            O.linear_wgrad(dX, x, g(lin.weight), g(lin.bias), acc)
        # input gradients: three contributions each for the audio and the text sequence, accumulated into the residual term
        # (two problems of one launch never write the same buffer)
        O.linear_dgrad_group([(dq1a, qa.weight, da, True), (dq1t, qt.weight, dt, True)])
        O.linear_dgrad_group([(dk1a, ka.weight, da, True), (dk1t, kt.weight, dt, True)])
        O.linear_dgrad_group([(dv1a, va.weight, da, True), (dv1t, vt.weight, dt, True)])
        fp.publish()
        ctx.saved = None
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
        fn = _CrossGroupedFn if O.GROUPED_HEAD else _CrossFn
        return fn.apply(self, audio_seq, text_seq, audio_mask, text_mask, *self._flat.params)
