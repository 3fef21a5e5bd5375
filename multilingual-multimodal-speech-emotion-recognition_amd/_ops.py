"""fp32 head operators: thin Python wrappers that allocate outputs (torch = device memory only)
and launch the HIP kernels of libser_hip.so.  No arithmetic happens in torch here."""
import torch

from . import _lib as L

ACT_NONE, ACT_GELU, ACT_RELU, ACT_TANH, ACT_SIGMOID = L.ACT_NONE, L.ACT_GELU, L.ACT_RELU, L.ACT_TANH, L.ACT_SIGMOID


import contextlib

# (Experiment, off by default.)  Weight gradients feed nothing but the optimizer, so under hipGraph capture they can be issued on their own stream
# and leave the dgrad chain (the critical path of backward).  Operands are kept alive until `wgrad_join()` so the
# capture-time allocator cannot hand their memory to a later main-stream tensor.  Eager mode stays single-stream
# (the data-parallel bucket hooks rely on a module's gradients being complete when its backward returns).
_WG = {"stream": {}, "keep": [], "dirty": False, "enabled": False}   # measured: hundreds of fork/join edges make hipGraph replay 50 % slower


def _wgrad_scope(*tensors):
    if not (_WG["enabled"] and torch.cuda.is_current_stream_capturing()):
        return contextlib.nullcontext()
    dev = torch.cuda.current_device()
    if dev not in _WG["stream"]:
        _WG["stream"][dev] = torch.cuda.Stream()
    side = _WG["stream"][dev]
    side.wait_stream(torch.cuda.current_stream())
    _WG["keep"].extend(t for t in tensors if t is not None)
    _WG["dirty"] = True
    return torch.cuda.stream(side)


# Deferred weight gradients: inside `defer_wgrads()` every linear_wgrad / linear_wgrad_batch request is only recorded;
# leaving the scope issues them as a few grouped launches (ser_linear_wgrad_group for token-level problems,
# ser_linear_wgrad_batch for M <= 16).  Weight gradients feed nothing but the optimizer, so this takes ~120 launches
# off the dgrad chains of backward.  Not used with the eager data-parallel hooks (a bucket must be complete when its
# module's backward returns).
_DEFER = {"active": False, "tok": [], "skinny": []}


@contextlib.contextmanager
def defer_wgrads(enable=True):
    if not enable or _DEFER["active"]:
        yield
        return
    _DEFER["active"] = True
    try:
        yield
    finally:
        _DEFER["active"] = False
        flush_deferred_wgrads()


def flush_deferred_wgrads():
    import ctypes as C
    tok, skinny = _DEFER["tok"], _DEFER["skinny"]
    _DEFER["tok"], _DEFER["skinny"] = [], []
    for acc in (False, True):
        grp = [p for p in tok if p[4] == acc]
        for s0 in range(0, len(grp), 32):
            chunk = grp[s0:s0 + 32]
            ptrs = (C.c_void_p * (4 * len(chunk)))()
            dims = (C.c_int * (3 * len(chunk)))()
            for i, (dy, x, dW, db, _) in enumerate(chunk):
                ptrs[4 * i], ptrs[4 * i + 1], ptrs[4 * i + 2], ptrs[4 * i + 3] = L.ptr(dy), L.ptr(x), L.ptr(dW), L.ptr(db)
                dims[3 * i], dims[3 * i + 1], dims[3 * i + 2] = dy.shape[0], dy.shape[1], x.shape[1]
            nb = L.lib.ser_linear_wgrad_group_workspace_bytes(dims, len(chunk))
            ws = torch.empty(nb, dtype=torch.uint8, device=chunk[0][0].device)
            L.check(L.lib.ser_linear_wgrad_group(ptrs, dims, len(chunk), 1 if acc else 0, L.ptr(ws), nb, L.stream_ptr()),
                    "ser_linear_wgrad_group")
        sk = [p[:4] for p in skinny if p[4] == acc]
        by_m = {}
        for p in sk:
            by_m.setdefault(p[0].shape[0], []).append(p)
        for probs in by_m.values():
            _wgrad_batch_now(probs, acc)


def _groupable(dy, x, dW):
    M, N = dy.shape
    K = x.shape[1]
    return (M > 16 and N >= 4 and K >= 4 and N % 4 == 0 and K % 4 == 0 and dy.data_ptr() % 16 == 0
            and x.data_ptr() % 16 == 0 and dy.is_contiguous() and x.is_contiguous())


def wgrad_join():
    """Make the current stream wait for every weight-gradient launch issued since the last join (capture only)."""
    if _WG["dirty"]:
        dev = torch.cuda.current_device()
        torch.cuda.current_stream().wait_stream(_WG["stream"][dev])
        _WG["keep"].clear()
        _WG["dirty"] = False


# Independent branches of the head (text adapter, T<-A cross-attention direction, text pooling) may run on a second stream.
# SIDE_STREAMS = False runs them inline on the caller's stream (same arithmetic either way).
import os as _os
SIDE_STREAMS = _os.environ.get("SER_SIDE_STREAMS", "1") == "1"
# GROUPED_HEAD: the alternative to the side stream - the independent products of one dependency level (both cross-attention
# directions, both adapters, both poolings) as ONE grouped launch each (ser_linear_fwd_group / ser_linear_dgrad_group) on one
# stream.  Measured (profiles/r03_b_grouped_head.txt): 2 % faster with the head alone on the chip, 4-6 % SLOWER beside the
# encoder pass - two queues get the head a larger share of a contended chip than one queue with wider launches - so it is
# off by default.
GROUPED_HEAD = _os.environ.get("SER_HEAD_GROUPED", "0") == "1"


class fork:
    """with fork(side_stream) as f: <independent work>; then f.join(produced=[...], consumed=[...]).
    On entry the side stream waits for the current one; `join` makes the current stream wait for the side stream and
    records the cross-stream uses with the allocator (tensors produced on the side stream and used on the current one,
    tensors of the current stream consumed on the side stream).  With SIDE_STREAMS off, or side=None, everything runs
    inline and join is a no-op."""

    def __init__(self, side):
        self.side = side if SIDE_STREAMS else None

    def __enter__(self):
        if self.side is not None:
            self.cur = torch.cuda.current_stream()
            self.side.wait_stream(self.cur)
            self.ctx = torch.cuda.stream(self.side)
            self.ctx.__enter__()
        return self

    def __exit__(self, *a):
        if self.side is not None:
            self.ctx.__exit__(*a)
        return False

    def join(self, produced=(), consumed=()):
        if self.side is None:
            return
        self.cur.wait_stream(self.side)
        for t in produced:
            if t is not None:
                t.record_stream(self.cur)
        for t in consumed:
            if t is not None:
                t.record_stream(self.side)


L._sig("ser_set_linear_forward_products", L.i32, L.i32)
L._sig("ser_get_linear_forward_products", L.i32)


@contextlib.contextmanager
def linear_forward_products(n):
    """Token-level forward Linear products inside the scope use `n` MFMA products per multiply (3: fp32-equivalent, 1: bf16
    operands - the reference's --use_amp arithmetic).  The fine-tuning encoders select 1 in the `bf16` precision mode."""
    prev = L.lib.ser_get_linear_forward_products()
    L.check(L.lib.ser_set_linear_forward_products(int(n)), "ser_set_linear_forward_products")
    try:
        yield
    finally:
        L.lib.ser_set_linear_forward_products(prev)


# `loss.backward()` without arguments seeds the graph with ones: the fused loss kernel has already written d(loss)/d(inputs)
# for that seed, so scaling its four gradient tensors by the upstream value (four launches, plus autograd's fill of the
# seed) is only needed when a caller backpropagates something else through the loss.  The steppers, which always call
# `loss.backward()` on the loss itself, declare that with this scope.
_UNIT_LOSS_GRAD = [False]


@contextlib.contextmanager
def unit_loss_grad():
    prev, _UNIT_LOSS_GRAD[0] = _UNIT_LOSS_GRAD[0], True
    try:
        yield
    finally:
        _UNIT_LOSS_GRAD[0] = prev


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def empty(*shape, like):
    return torch.empty(*shape, dtype=torch.float32, device=like.device)


def linear_fwd(x, W, b=None, act=ACT_NONE, residual=None, out=None):
    """y[M,N] = act(x[M,K] W[N,K]^T + b) + residual."""
    M, K = x.shape
    N = W.shape[0]
    y = out if out is not None else empty(M, N, like=x)
    L.check(L.lib.ser_linear_fwd(L.ptr(x), L.ptr(W), L.ptr(b), act, L.ptr(residual), N, L.ptr(y), M, N, K, L.stream_ptr()),
            "ser_linear_fwd")
    return y


L._sig("ser_linear_fwd_group", L.i32, L.vp, L.vp, L.i32, L.vp)
L._sig("ser_linear_dgrad_group", L.i32, L.vp, L.vp, L.i32, L.vp)


def linear_fwd_group(problems):
    """Independent Linear layers of one dependency level in ONE launch.  problems: list of (x, W, b | None, act, residual | None)
    -> list of y.  Bit-identical to linear_fwd called per problem."""
    import ctypes as C
    n = len(problems)
    ptrs = (C.c_void_p * (5 * n))()
    dims = (C.c_int * (5 * n))()
    outs = []
    for i, (x, W, b, act, res) in enumerate(problems):
        M, K = x.shape
        N = W.shape[0]
        assert x.is_contiguous() and W.is_contiguous() and W.shape[1] == K
        y = empty(M, N, like=x)
        outs.append(y)
        ptrs[5 * i], ptrs[5 * i + 1], ptrs[5 * i + 2], ptrs[5 * i + 3], ptrs[5 * i + 4] = L.ptr(x), L.ptr(W), L.ptr(b), L.ptr(res), L.ptr(y)
        dims[5 * i], dims[5 * i + 1], dims[5 * i + 2], dims[5 * i + 3], dims[5 * i + 4] = M, N, K, int(act), N
    L.check(L.lib.ser_linear_fwd_group(ptrs, dims, n, L.stream_ptr()), "ser_linear_fwd_group")
    return outs


def linear_dgrad_group(problems):
    """Input gradients of independent Linear layers in ONE launch.  problems: list of (dy, W, out | None, accumulate) ->
    list of dx (`out` is written / accumulated into when given).  Bit-identical to linear_dgrad called per problem."""
    import ctypes as C
    n = len(problems)
    ptrs = (C.c_void_p * (3 * n))()
    dims = (C.c_int * (4 * n))()
    outs = []
    for i, (dy, W, out, acc) in enumerate(problems):
        M, N = dy.shape
        K = W.shape[1]
        assert dy.is_contiguous() and W.shape[0] == N
        dx = out if out is not None else empty(M, K, like=dy)
        assert dx.is_contiguous() or dx.stride(-1) == 1
        outs.append(dx)
        ptrs[3 * i], ptrs[3 * i + 1], ptrs[3 * i + 2] = L.ptr(dy), L.ptr(W), L.ptr(dx)
        dims[4 * i], dims[4 * i + 1], dims[4 * i + 2], dims[4 * i + 3] = M, N, K, 1 if acc else 0
    L.check(L.lib.ser_linear_dgrad_group(ptrs, dims, n, L.stream_ptr()), "ser_linear_dgrad_group")
    return outs


def linear_dgrad(dy, W, out=None, accumulate=False, relu_mask=None):
    """dx[M,K] (+)= (dy[M,N] W[N,K]) * relu'(relu_mask)."""
    M, N = dy.shape
    K = W.shape[1]
    dx = out if out is not None else empty(M, K, like=dy)
    L.check(L.lib.ser_linear_dgrad(L.ptr(dy), L.ptr(W), L.ptr(relu_mask), L.ptr(dx), M, N, K, 1 if accumulate else 0,
                                   L.stream_ptr()), "ser_linear_dgrad")
    return dx


def linear_wgrad_pair(dya, xa, dWa, dba, dyb, xb, dWb, dbb, accumulate=False):
    """Two skinny (M <= 16) weight gradients + bias gradients in one launch."""
    M = dya.shape[0]
    if M > 16 or _DEFER["active"]:
        linear_wgrad(dya, xa, dWa, dba, accumulate)
        linear_wgrad(dyb, xb, dWb, dbb, accumulate)
        return
    with _wgrad_scope(dya, xa, dyb, xb):
        L.check(L.lib.ser_linear_wgrad_pair(L.ptr(dya), L.ptr(xa), L.ptr(dWa), L.ptr(dba), dya.shape[1], xa.shape[1],
                                            L.ptr(dyb), L.ptr(xb), L.ptr(dWb), L.ptr(dbb), dyb.shape[1], xb.shape[1], M,
                                            1 if accumulate else 0, L.stream_ptr()), "ser_linear_wgrad_pair")


def linear_wgrad_batch(problems, accumulate=False):
    """problems: list of (dy[M,N], x[M,K], dW[N,K], db[N] or None), all with the same M <= 16 -> one launch per 80."""
    M = problems[0][0].shape[0]
    if M > 16 or _DEFER["active"]:
        for dy, x, dW, db in problems:
            linear_wgrad(dy, x, dW, db, accumulate)
        return
    _wgrad_batch_now(problems, accumulate)


def _wgrad_batch_now(problems, accumulate):
    import ctypes as C
    M = problems[0][0].shape[0]
    for s0 in range(0, len(problems), 80):
        chunk = problems[s0:s0 + 80]
        ptrs = (C.c_void_p * (4 * len(chunk)))()
        dims = (C.c_int * (2 * len(chunk)))()
        for i, (dy, x, dW, db) in enumerate(chunk):
            ptrs[4 * i], ptrs[4 * i + 1], ptrs[4 * i + 2], ptrs[4 * i + 3] = L.ptr(dy), L.ptr(x), L.ptr(dW), L.ptr(db)
            dims[2 * i], dims[2 * i + 1] = dy.shape[1], x.shape[1]
        L.check(L.lib.ser_linear_wgrad_batch(ptrs, dims, len(chunk), M, 1 if accumulate else 0, L.stream_ptr()),
                "ser_linear_wgrad_batch")


def ln2_fwd(x, g1, b1, g2, b2, eps=1e-5):
    """y1 = LN(x; g1,b1), y2 = LN(y1; g2,b2) in one launch -> y1, y2, stats[4, rows]."""
    rows, D = x.shape
    y1, y2 = torch.empty_like(x), torch.empty_like(x)
    stats = empty(4, rows, like=x)
    L.check(L.lib.ser_layernorm2_fwd(L.ptr(x), L.ptr(g1), L.ptr(b1), L.ptr(g2), L.ptr(b2), eps, rows, D, L.ptr(y1), L.ptr(y2),
                                     L.ptr(stats), L.stream_ptr()), "ser_layernorm2_fwd")
    return y1, y2, stats


def linear_fwd_ln2(x, W, b, act, g1, b1, g2, b2, eps=1e-5):
    """y = act(LN(LN(x;g1,b1);g2,b2) W^T + b) in one launch (M <= 16, K <= 512) -> y, x1, u, stats[4, M]."""
    M, K = x.shape
    N = W.shape[0]
    y = empty(M, N, like=x)
    y1, y2 = torch.empty_like(x), torch.empty_like(x)
    stats = empty(4, M, like=x)
    L.check(L.lib.ser_linear_fwd_ln2(L.ptr(x), L.ptr(W), L.ptr(b), act, L.ptr(g1), L.ptr(b1), L.ptr(g2), L.ptr(b2), eps,
                                     L.ptr(y1), L.ptr(y2), L.ptr(stats), L.ptr(y), M, N, K, L.stream_ptr()),
            "ser_linear_fwd_ln2")
    return y, y1, y2, stats


# ---- persistent walk of the classifier's residual stack (csrc/persist.hip) ---------------------------------------------
USE_STACK = True     # one launch per direction for the whole stack when the shape allows it (M <= 16, D <= 512)


def stack_supported(L_, M, D):
    return USE_STACK and bool(L.lib.ser_stack_supported(L_, M, D))


def stack_fwd(x0, table, L_, flags, eps=1e-5, dctx=None, site=0):
    """All L residual blocks forward -> (Hs[L,M,D] block outputs, X1, U, A [L,M,D], ST[L,4,M]).  dctx / site: the two
    dropout layers of every block in training mode (sites site + 2 i, site + 2 i + 1)."""
    M, D = x0.shape
    Hs, X1, U, A = (empty(L_, M, D, like=x0) for _ in range(4))
    ST = empty(L_, 4, M, like=x0)
    state, p = dctx if dctx is not None else (None, 0.0)
    L.check(L.lib.ser_stack_fwd(L.ptr(table), L.ptr(x0), L.ptr(Hs), L.ptr(X1), L.ptr(U), L.ptr(A), L.ptr(ST), L_, M, D, eps,
                                L.ptr(flags), L.ptr(state), int(site), p, L.stream_ptr()), "ser_stack_fwd")
    return Hs, X1, U, A, ST


def stack_bwd(table, x0, Hs, X1, A, ST, DH, flags, dctx=None, site=0):
    """Backward dgrad chain of the stack.  DH[L] holds the gradient at the stack output on entry; on return DH[i] is
    the gradient at the input of block i (DH[0]: at x0).  -> DA, DU, DX1 [L,M,D] for the batched parameter gradients,
    and DT [L+1,M,D] (the dropped block-output gradients) when dropout is active, else None."""
    L_, M, D = Hs.shape
    DA, DU, DX1 = (empty(L_, M, D, like=x0) for _ in range(3))
    state, p = dctx if dctx is not None else (None, 0.0)
    DT = empty(L_ + 1, M, D, like=x0) if dctx is not None else None
    L.check(L.lib.ser_stack_bwd(L.ptr(table), L.ptr(x0), L.ptr(Hs), L.ptr(X1), L.ptr(A), L.ptr(ST), L.ptr(DH), L.ptr(DA),
                                L.ptr(DU), L.ptr(DX1), L_, M, D, L.ptr(flags), L.ptr(state), int(site), p, L.ptr(DT),
                                L.stream_ptr()), "ser_stack_bwd")
    return DA, DU, DX1, DT


def stack_ln_param_bwd(gtable, x0, Hs, X1, ST, DU, DX1, accumulate=False):
    L_, M, D = Hs.shape
    L.check(L.lib.ser_stack_ln_param_bwd(L.ptr(gtable), L.ptr(x0), L.ptr(Hs), L.ptr(X1), L.ptr(ST), L.ptr(DU), L.ptr(DX1),
                                         L_, M, D, 1 if accumulate else 0, L.stream_ptr()), "ser_stack_ln_param_bwd")


def ln2_bwd(du, dres, x, y1, stats, g1, g2, dg1, db1, dg2, db2, accumulate=False):
    """dx = LN1'(LN2'(du) + dres) and the four parameter gradients in one launch."""
    rows, D = du.shape
    dx = torch.empty_like(du)
    L.check(L.lib.ser_layernorm2_bwd(L.ptr(du), L.ptr(dres), L.ptr(x), L.ptr(y1), L.ptr(stats), L.ptr(g1), L.ptr(g2), rows, D,
                                     L.ptr(dx), L.ptr(dg1), L.ptr(db1), L.ptr(dg2), L.ptr(db2), 1 if accumulate else 0,
                                     L.stream_ptr()), "ser_layernorm2_bwd")
    return dx


def linear_wgrad(dy, x, dW, db=None, accumulate=False):
    """dW[N,K] (+)= dy[M,N]^T x[M,K];  db[N] (+)= colsum(dy) in the same pass."""
    M, N = dy.shape
    K = x.shape[1]
    if _DEFER["active"]:
        if M <= 16 and K % 4 == 0:
            _DEFER["skinny"].append((dy, x, dW, db, bool(accumulate)))
            return
        if _groupable(dy, x, dW):
            _DEFER["tok"].append((dy, x, dW, db, bool(accumulate)))
            return
    nbytes = L.lib.ser_linear_wgrad_workspace_bytes(M, N, K)
    with _wgrad_scope(dy, x):
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dy.device) if nbytes else None
        if ws is not None and _WG["dirty"]:
            _WG["keep"].append(ws)
        L.check(L.lib.ser_linear_wgrad(L.ptr(dy), L.ptr(x), L.ptr(dW), L.ptr(db), M, N, K, 1 if accumulate else 0, L.ptr(ws),
                                       nbytes, L.stream_ptr()), "ser_linear_wgrad")


def act_fwd(x, act, dctx=None, site=0):
    """y = act(x), followed by dropout in the same pass when dctx is given."""
    y = torch.empty_like(x)
    if dctx is None:
        L.check(L.lib.ser_act_fwd(L.ptr(x), act, x.numel(), L.ptr(y), L.stream_ptr()), "ser_act_fwd")
    else:
        L.check(L.lib.ser_act_drop_fwd(L.ptr(x), act, x.numel(), L.ptr(y), L.ptr(dctx[0]), int(site), dctx[1], L.stream_ptr()),
                "ser_act_drop_fwd")
    return y


def act_bwd(dy, y, act, inplace=True, dctx=None, site=0):
    """dx = dy * act'(y); with dctx, y = dropout(act(x)) and the dropout mask is applied to dy in the same pass."""
    dx = dy if inplace else torch.empty_like(dy)
    if dctx is None:
        L.check(L.lib.ser_act_bwd(L.ptr(dy), L.ptr(y), act, dy.numel(), L.ptr(dx), L.stream_ptr()), "ser_act_bwd")
    else:
        L.check(L.lib.ser_act_drop_bwd(L.ptr(dy), L.ptr(y), act, dy.numel(), L.ptr(dx), L.ptr(dctx[0]), int(site), dctx[1],
                                       L.stream_ptr()), "ser_act_drop_bwd")
    return dx


def axpby(x, y, a=1.0, b=1.0):
    """y = a*x + b*y in place."""
    L.check(L.lib.ser_axpby(L.ptr(x), a, b, x.numel(), L.ptr(y), L.stream_ptr()), "ser_axpby")
    return y


def scale_dev_(x, s):
    """x *= s[0] with s a device scalar (no host synchronisation)."""
    L.check(L.lib.ser_scale_dev(L.ptr(x), L.ptr(s.reshape(1).contiguous()), x.numel(), L.stream_ptr()), "ser_scale_dev")
    return x


def ln_fwd(x, gamma, beta, eps=1e-5, x2=None, keep_z=None):
    """-> y, saved=(z, mean, rstd).  z is x itself unless a second addend is given."""
    rows, D = x.shape
    y = torch.empty_like(x)
    need_z = x2 is not None if keep_z is None else keep_z
    z = torch.empty_like(x) if need_z else None
    mean, rstd = empty(rows, like=x), empty(rows, like=x)
    L.check(L.lib.ser_layernorm_fwd(L.ptr(x), L.ptr(x2), L.ptr(gamma), L.ptr(beta), eps, rows, D, L.ptr(y), L.ptr(z),
                                    L.ptr(mean), L.ptr(rstd), L.stream_ptr()), "ser_layernorm_fwd")
    return y, (z if need_z else x, mean, rstd)


def ln_bwd(dy, saved, gamma, dgamma=None, dbeta=None, accumulate=False, dx_add=None, need_dx=True):
    z, mean, rstd = saved
    rows, D = dy.shape
    dx = torch.empty_like(dy) if need_dx else None
    nb = L.lib.ser_layernorm_bwd_workspace_bytes(rows, D)
    ws = torch.empty(nb, dtype=torch.uint8, device=dy.device) if nb else None
    L.check(L.lib.ser_layernorm_bwd(L.ptr(dy), L.ptr(z), L.ptr(mean), L.ptr(rstd), L.ptr(gamma), L.ptr(dx_add), rows, D,
                                    L.ptr(dx), L.ptr(dgamma), L.ptr(dbeta), 1 if accumulate else 0, L.ptr(ws),
                                    L.stream_ptr()), "ser_layernorm_bwd")
    return dx


def xattn_fwd(q, k, v, key_mask, B, Sq, Sk, heads, dctx=None, site=0):
    """q [B*Sq,E], k,v [B*Sk,E] (may be column slices given as (tensor, ld) through .stride) -> ctx, P.
    dctx / site: attention-probability dropout (training mode)."""
    E = q.shape[1]
    hd = E // heads
    P = empty(B, heads, Sq, Sk, like=q)
    ctx = empty(B * Sq, E, like=q)
    state, p = dctx if dctx is not None else (None, 0.0)
    Pd = torch.empty_like(P) if dctx is not None else None          # dropped probabilities (dv in backward)
    L.check(L.lib.ser_xattn_fwd(q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(), v.stride(0),
                                L.ptr(key_mask), B, Sq, Sk, heads, hd, L.ptr(P), L.ptr(ctx), E, L.ptr(state), int(site), p,
                                L.ptr(Pd), L.stream_ptr()), "ser_xattn_fwd")
    return ctx, (P if Pd is None else (P, Pd))


def xattn_bwd(dctx, q, k, v, P, B, Sq, Sk, heads, drop=None, site=0, out=None):
    """P: what xattn_fwd returned (the softmax output, or (softmax, dropped) with dropout).  out: (dq, dk, dv) to write into -
    column blocks of a fused buffer are fine (row strides are passed on)."""
    P, Pd = P if isinstance(P, tuple) else (P, None)
    E = q.shape[1]
    hd = E // heads
    dS = torch.empty_like(P)
    if out is None:
        dq, dk, dv = empty(B * Sq, E, like=q), empty(B * Sk, E, like=q), empty(B * Sk, E, like=q)
    else:
        dq, dk, dv = out
        assert all(t.stride(1) == 1 and t.dtype == torch.float32 for t in out)
    L.check(L.lib.ser_xattn_bwd(L.ptr(dctx), E, q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(),
                                v.stride(0), L.ptr(P), B, Sq, Sk, heads, hd, L.ptr(dS), dq.data_ptr(), dq.stride(0), dk.data_ptr(),
                                dk.stride(0), dv.data_ptr(), dv.stride(0), L.ptr(drop[0] if drop is not None else None), int(site),
                                drop[1] if drop is not None else 0.0, L.ptr(Pd), L.stream_ptr()), "ser_xattn_bwd")
    return dq, dk, dv


def pool_fwd(x, logits, mask):
    B, S, D = x.shape
    alpha, out = empty(B, S, like=x), empty(B, 2 * D, like=x)
    L.check(L.lib.ser_pool_fwd(L.ptr(x), L.ptr(logits), L.ptr(mask), B, S, D, L.ptr(alpha), L.ptr(out), L.stream_ptr()),
            "ser_pool_fwd")
    return out, alpha


def pool_bwd(dout, x, alpha, out):
    B, S, D = x.shape
    dx, scratch, dlogits = torch.empty_like(x), empty(B, S, like=x), empty(B, S, like=x)
    L.check(L.lib.ser_pool_bwd(L.ptr(dout), L.ptr(x), L.ptr(alpha), L.ptr(out), B, S, D, L.ptr(dx), L.ptr(scratch),
                               L.ptr(dlogits), L.stream_ptr()), "ser_pool_bwd")
    return dx, dlogits


def fusion_mix_fwd(a, t, ga, gt):
    B, P = a.shape
    out = torch.empty_like(a)
    L.check(L.lib.ser_fusion_mix_fwd(L.ptr(a), L.ptr(t), L.ptr(ga), L.ptr(gt), B, P, L.ptr(out), L.stream_ptr()),
            "ser_fusion_mix_fwd")
    return out


def fusion_mix_bwd(dout, a, t, ga, gt):
    B, P = a.shape
    da, dt = torch.empty_like(a), torch.empty_like(t)
    dga, dgt = empty(B, 1, like=a), empty(B, 1, like=a)
    L.check(L.lib.ser_fusion_mix_bwd(L.ptr(dout), L.ptr(a), L.ptr(t), L.ptr(ga), L.ptr(gt), B, P, L.ptr(da), L.ptr(dt),
                                     L.ptr(dga), L.ptr(dgt), L.stream_ptr()), "ser_fusion_mix_bwd")
    return da, dt, dga, dgt


def train_loss(logits, unc, fused, protos, labels, smoothing=0.1, cb_beta=0.9999, gamma=2.0, w_focal=0.3, w_unc=0.05,
               w_proto=0.01, margin=0.5, use_proto=True, grad_scale=None):
    B, C = logits.shape
    D = fused.shape[1]
    losses = empty(5, like=logits)
    dlogits, dunc = torch.empty_like(logits), torch.empty_like(unc)
    dfused, dprotos = torch.empty_like(fused), torch.empty_like(protos)
    L.check(L.lib.ser_train_loss(L.ptr(logits), L.ptr(unc), L.ptr(fused), L.ptr(protos), L.ptr(labels), B, C, D, smoothing,
                                 cb_beta, gamma, w_focal, w_unc, w_proto, margin, 1 if use_proto else 0,
                                 L.ptr(grad_scale), L.ptr(losses), L.ptr(dlogits), L.ptr(dunc), L.ptr(dfused),
                                 L.ptr(dprotos), L.stream_ptr()), "ser_train_loss")
    return losses, dlogits, dunc, dfused, dprotos


def openmax_(feats, act_vec, walpha, wbeta, wtau, logits, thresh=0.3, reduce=0.8):
    B, F = feats.shape
    C = logits.shape[1]
    L.check(L.lib.ser_openmax(L.ptr(feats), L.ptr(act_vec), L.ptr(walpha), L.ptr(wbeta), L.ptr(wtau), B, C, F, thresh,
                              reduce, L.ptr(logits), L.stream_ptr()), "ser_openmax")
    return logits


# ---- dropout (training mode) ----------------------------------------------------------------------------------------
# Off by default: module-level parity is defined with dropout as the identity (the golden vectors of the reference were
# captured that way).  `SERSystem` turns it on around its training forward (`dropout_scope`); every autograd node reads the
# state at forward time and keeps what it needs for backward, so the switch only has to cover the forward call.
_DROP = {"on": False, "state": None}
_SITES = [0]


def new_dropout_site(n=1):
    """A process-unique id for one dropout layer (part of the mask generator's key); n > 1 reserves a run of ids and
    returns the first."""
    first = _SITES[0] + 1
    _SITES[0] += n
    return first


class dropout_scope:
    def __init__(self, state):
        self.state = state          # device int64[1]: the generator state, advanced once per step by the owner

    def __enter__(self):
        self.prev = dict(_DROP)
        _DROP["on"], _DROP["state"] = self.state is not None, self.state

    def __exit__(self, *a):
        _DROP.update(self.prev)


def dropout_ctx(p):
    """-> (state tensor, p) when dropout is active, else None."""
    if _DROP["on"] and p > 0.0:
        return (_DROP["state"], float(p))
    return None


def dropout_(x, dctx, site):
    """In place: x *= mask / (1 - p).  The same call on a gradient is the backward of the layer."""
    if dctx is None:
        return x
    state, p = dctx
    L.check(L.lib.ser_dropout(L.ptr(x), x.numel(), L.ptr(state), int(site), p, L.ptr(x), L.stream_ptr()), "ser_dropout")
    return x


def dropout(x, dctx, site):
    """Out of place: y = x * mask / (1 - p) (one launch; `dropout_` on a clone costs a copy more)."""
    x = _c(x)
    if dctx is None:
        return x.clone()
    state, p = dctx
    y = torch.empty_like(x)
    L.check(L.lib.ser_dropout(L.ptr(x), x.numel(), L.ptr(state), int(site), p, L.ptr(y), L.stream_ptr()), "ser_dropout")
    return y


def adamw_multi_(segments, hyper, b1, b2, eps):
    """segments: list of (p, g, m, v, lr_mult, weight_decay[, gate]) flat fp32 tensors -> one launch per 16 segments.  gate: an
    int32 device tensor (one element) or None; a segment whose gate is non-zero when the kernel runs is left untouched."""
    import ctypes as C
    n = len(segments)
    ptrs = (C.c_void_p * (4 * n))()
    gates = (C.c_void_p * n)()
    cnt = (C.c_longlong * n)()
    lrm = (C.c_float * n)()
    wd = (C.c_float * n)()
    gated = False
    for i, sg in enumerate(segments):
        p, g, m, v, lr_mult, weight_decay = sg[:6]
        ptrs[4 * i], ptrs[4 * i + 1], ptrs[4 * i + 2], ptrs[4 * i + 3] = L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v)
        cnt[i], lrm[i], wd[i] = p.numel(), lr_mult, weight_decay
        if len(sg) > 6 and sg[6] is not None:
            assert sg[6].dtype == torch.int32 and sg[6].numel() == 1
            gates[i] = sg[6].data_ptr()
            gated = True
    L.check(L.lib.ser_adamw_multi_gated(ptrs, cnt, lrm, wd, gates if gated else None, n, L.ptr(hyper), b1, b2, eps, L.stream_ptr()),
            "ser_adamw_multi")


def adamw_(p, g, m, v, hyper, lr_mult, weight_decay, beta1=0.9, beta2=0.999, eps=1e-8):
    L.check(L.lib.ser_adamw(L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), p.numel(), L.ptr(hyper), lr_mult, weight_decay, beta1,
                            beta2, eps, L.stream_ptr()), "ser_adamw")


# ---- eval-side consumers (csrc/evalops.hip) --------------------------------------------------------------------------------
L._sig("ser_eval_consumers", L.i32, L.vp, L.i32, L.i32, L.f32, L.vp, L.vp, L.vp, L.vp)
L._sig("ser_temperature_grid", L.i32, L.vp, L.vp, L.i32, L.i32, L.vp, L.i32, L.vp, L.vp)


def eval_consumers(logits, temperature=1.0):
    """logits [B, C] -> (probs = softmax(logits / T), pred int64 [B], energy = -logsumexp(logits / T))  (ref eval.py:192-206)."""
    logits = _c(logits)
    B, Cc = logits.shape
    probs = torch.empty_like(logits)
    pred = torch.empty(B, dtype=torch.int64, device=logits.device)
    energy = empty(B, like=logits)
    L.check(L.lib.ser_eval_consumers(L.ptr(logits), B, Cc, float(temperature), L.ptr(probs), L.ptr(pred), L.ptr(energy), L.stream_ptr()),
            "ser_eval_consumers")
    return probs, pred, energy


def temperature_grid(logits, labels, temps):
    """ece[g] of the reference's one-bin |confidence - accuracy| proxy for every temperature of the grid (ref eval.py:49-67)."""
    logits, labels, temps = _c(logits), _c(labels.to(torch.int64)), _c(temps.to(torch.float32))
    ece = empty(temps.numel(), like=logits)
    L.check(L.lib.ser_temperature_grid(L.ptr(logits), L.ptr(labels), logits.shape[0], logits.shape[1], L.ptr(temps), temps.numel(),
                                       L.ptr(ece), L.stream_ptr()), "ser_temperature_grid")
    return ece


# ---- quality-gate / audio-conditioning front end (csrc/frontend.hip) ----------------------------------------------------
L._sig("ser_frontend_workspace_bytes", L.sz, L.i32, L.i32)
L._sig("ser_frontend_init", L.i32)
L._sig("ser_quality_gates", L.i32, L.vp, L.i32, L.i32, L.i32, L.vp, L.i32, L.vp, L.vp, L.vp, L.vp, L.sz, L.vp)
L._sig("ser_audio_conditioning", L.i32, L.vp, L.vp, L.i32, L.i32, L.i32, L.vp, L.vp, L.vp, L.vp, L.sz, L.vp)
_FE_WS = {}      # device -> workspace, grow-only; the two calls below are meant for ONE stream at a time per device (like the engines' workspaces)


def _frontend_ws(B, T, dev):
    need = L.lib.ser_frontend_workspace_bytes(B, T)
    ws = _FE_WS.get(dev)
    if ws is None or ws.numel() < need:
        L.check(L.lib.ser_frontend_init(), "ser_frontend_init")
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        _FE_WS[dev] = ws
    return ws, need


def quality_gates(wave, lid, pad_mode="constant", sample_rate=16000):
    """wave [B, T] f32, lid [B, 2] (language entropy, confidence) -> (q_raw [B, 8], metrics [B, 8], decision int32 [B])
    (ref quality_gates.py:497-560, energy VAD)."""
    wave, lid = _c(wave), _c(lid.to(torch.float32))
    B, T = wave.shape
    assert lid.shape == (B, 2)
    raw, met = empty(B, 8, like=wave), empty(B, 8, like=wave)
    dec = torch.empty(B, dtype=torch.int32, device=wave.device)
    ws, need = _frontend_ws(B, T, wave.device)
    L.check(L.lib.ser_quality_gates(L.ptr(wave), B, T, int(sample_rate), L.ptr(lid), 1 if pad_mode == "reflect" else 0, L.ptr(raw),
                                    L.ptr(met), L.ptr(dec), L.ptr(ws), need, L.stream_ptr()), "ser_quality_gates")
    return raw, met, dec


def audio_conditioning(wave, decision=None, sample_rate=16000):
    """wave [B, T] f32 (+ decision int32 [B]: clips not marked 2 = 'accept' are conditioned as silence) ->
    (conditioned [B, T], c_raw [B, 12], meta [B, 12])  (ref audio_conditioning.py:503-584)."""
    wave = _c(wave)
    B, T = wave.shape
    out = torch.empty_like(wave)
    raw, meta = empty(B, 12, like=wave), empty(B, 12, like=wave)
    if decision is not None:
        decision = _c(decision.to(torch.int32))
    ws, need = _frontend_ws(B, T, wave.device)
    L.check(L.lib.ser_audio_conditioning(L.ptr(wave), L.ptr(decision) if decision is not None else None, B, T, int(sample_rate),
                                         L.ptr(out), L.ptr(raw), L.ptr(meta), L.ptr(ws), need, L.stream_ptr()),
            "ser_audio_conditioning")
    return out, raw, meta
