"""Fine-tuning form of the two encoders — BASELINE config 3: the reference's `freeze_base=False`
(src/models/audio_encoder.py:15-17, src/models/text_encoder.py:13-15, src/train_two_phase.py:166-172), where every
Wav2Vec2 / XLM-R parameter receives a gradient.

The frozen path (csrc/encoders.hip) keeps the encoders as packed split-bf16 planes and has no backward.  This path runs
the same arithmetic on fp32 tensors with torch as the autograd tape and the view / permute plumbing:
  * encoder-sized Linear layers and the 512-channel conv layers run forward, input-gradient and weight-gradient products on the
    frozen encoders' MFMA tile kernel (`ser_gemm_bf16_nt`) through split operand planes - straight and transposed from ONE pass
    per operand (`ser_split_bf16_both`), the conv's window view read straight out of the planes of its input, its input gradient
    folded back from per-window gradients by a gather (`ser_conv_col2im`);
  * what does not fit the tile kernel (conv0 with one input channel, the 16-group positional conv with 48 channels per group,
    small test shapes) is `ser_gemm_f32` (fp32 in / out, operands split on the fly, any strides: a Conv1d is a GEMM over a
    strided window view);
  * attention is the head's `ser_xattn_fwd,bwd` on the fused q | k | v buffer (fp32 matrix pipe, csrc/xattn_mfma.hip); LayerNorm,
    GELU and the dropout sites are the head's kernels, fused where they follow each other (dropout + residual + LayerNorm, GELU +
    dropout); csrc/finetune.hip adds the pieces the head never needed (GELU backward, GroupNorm over time, the positional conv's
    overlap-add, embedding gather / owner-computes scatter-add);
  * products per multiply: 3 (fp32-equivalent) or, in the `bf16` precision mode (train.py --use_amp), 1;
  * the step is capturable: LayerDrop / SpecAugment decisions can be staged as device words (`Noise.stage`).
The one place torch arithmetic is used is the weight-norm of the positional conv's 4.7 M-element weight (g * v / ||v||, a parameter
transform, hf modeling_wav2vec2.py:326-349).

Semantics: eval-mode HuggingFace forward (no encoder dropout / SpecAugment / LayerDrop), i.e. what the golden gradients
of tests/golden/{audio,text}_encoder_grads.npz were captured with; `masked_spec_embed` and the XLM-R pooler take no part
and get no gradient.
"""
import ctypes as C

import torch

from .. import _lib as L
from .. import _ops as O

_sig = L._sig
_sig("ser_gelu_bwd", L.i32, L.vp, L.vp, L.i64, L.vp, L.vp)
_sig("ser_gelu_drop_bwd", L.i32, L.vp, L.vp, L.i64, L.vp, L.vp, C.c_uint, L.f32, L.vp)
_sig("ser_colnorm_workspace_bytes", L.sz, L.i32, L.i32)
_sig("ser_colnorm_fwd", L.i32, L.vp, L.i32, L.i32, L.i32, L.i32, L.vp, L.vp, L.f32, L.vp, L.vp, L.vp, L.vp, L.vp)
_sig("ser_colnorm_bwd", L.i32, L.vp, L.vp, L.vp, L.vp, L.vp, L.i32, L.i32, L.i32, L.i32, L.vp, L.vp, L.vp, L.i32, L.vp, L.vp)
_sig("ser_toeplitz_add", L.i32, L.vp, L.i32, L.i32, L.i32, L.i32, L.vp, L.vp)
_sig("ser_conv_col2im", L.i32, L.vp, L.i32, L.i32, L.i32, L.i32, L.i64, L.vp, L.vp)
_sig("ser_posconv_direct_supported", L.i32, L.i32, L.i32, L.i32, L.i32)
_sig("ser_posconv_pack", L.i32, L.vp, L.i32, L.i32, L.i32, L.i32, L.vp, L.vp)
_sig("ser_posconv_fwd", L.i32, L.vp, L.vp, L.vp, L.i32, L.i32, L.i32, L.i32, L.i32, L.vp, L.vp, L.vp)
_sig("ser_posconv_dgrad", L.i32, L.vp, L.vp, L.vp, L.i32, L.i32, L.i32, L.i32, L.i32, L.vp, L.vp)
_sig("ser_embed_fwd", L.i32, L.vp, L.vp, L.vp, L.vp, L.vp, L.i32, L.i32, L.i32, L.i32, L.vp, L.vp)
_sig("ser_embed_bwd", L.i32, L.vp, L.vp, L.vp, L.i32, L.i32, L.i32, L.i32, L.i32, L.vp, L.vp, L.vp, L.vp)
_sig("ser_wave_normalize", L.i32, L.vp, L.i32, L.i32, L.vp, L.vp, L.vp)


def _gemm(a_ptr, sam, sak, b_ptr, sbk, sbn, M, N, K, c, ldc, bias=None, accumulate=False, products=3):
    """c[M,N] (+)= A . B (+ bias) with element strides (ser_gemm_f32): A[m,k] = a + m*sam + k*sak, B[k,n] = b + k*sbk + n*sbn.
    products: bf16 MFMA products per multiply (3: fp32-equivalent; 1: the --use_amp arithmetic)."""
    L.check(L.lib.ser_gemm_f32_np(a_ptr, int(sam), int(sak), b_ptr, int(sbk), int(sbn), int(M), int(N), int(K), L.ptr(bias), L.ACT_NONE,
                                  None, 0, c.data_ptr() if isinstance(c, torch.Tensor) else c, int(ldc), 1 if accumulate else 0,
                                  int(products), L.stream_ptr()), "ser_gemm_f32")


def _wgrad_now(dy, x, dW, db):
    """Weight gradient written before backward returns: the buffers handed back to autograd become the parameters' .grad,
    so the head's deferred (grouped, end-of-backward) weight-gradient launches must not be used for them."""
    prev = O._DEFER["active"]
    O._DEFER["active"] = False
    try:
        O.linear_wgrad(dy, x, dW, db)
    finally:
        O._DEFER["active"] = prev


_sig("ser_colsum_tall_workspace_bytes", L.sz, L.i32)
_sig("ser_colsum_tall", L.i32, L.vp, L.i32, L.i32, L.i32, L.vp, L.vp, L.vp)


def _planes(x, three):
    """fp32 [R, C] (C % 64 == 0) -> kernel operand of the encoder NT GEMM: interleaved hi/lo planes, or the hi plane alone."""
    if three:
        t = L.split_bf16_il(x)
        return t, t.data_ptr(), t.data_ptr() + 2 * L.IL_GROUP
    hi, _ = L.split_bf16(x, False)
    return hi, hi.data_ptr(), None


def _planes_t(x, three):
    """fp32 [R, C] -> operand planes of x^T [C, Rp] (R zero-padded to a multiple of 64): (keep-alive, hi, lo, Rp)."""
    t, Rp = L.split_bf16_t(x, three)
    return t, t.data_ptr(), (t.data_ptr() + 2 * L.IL_GROUP) if three else None, Rp


# Note (round 3): running a layer's weight-gradient products on a helper stream beside its input-gradient products (a fork / join
# per backward node) was built and had to be taken out again: with a few hundred such diamonds in one capture, hipStreamEndCapture of
# ROCm 7.2 recurses without bound (stack overflow; with an unlimited stack > 270 GB of host memory).  The two encoders on two
# streams (SERSystem.encode) and the head's own forks - a dozen diamonds - are fine.  The fork-only variant (helper stream that is
# joined once per backward pass) captures, but is slower: 28.9 instead of 24.6 ms - every cross-stream edge of a replayed graph
# costs more than the launch it takes off the chain.


def _ptrs(t, three):
    """(hi, lo) pointers of a planes tensor (interleaved when `three`, else the hi plane alone)."""
    return t.data_ptr(), (t.data_ptr() + 2 * L.IL_GROUP) if three else None


def _nt(a, lda, w, ldw, M, N, K, out, bias=None):
    """out[M,N] fp32 = A[M,K] . W[N,K]^T (+ bias) on the encoder tile kernel (csrc/gemm_bf16.hip); a / w = (hi, lo) pointers."""
    L.check(L.lib.ser_gemm_bf16_nt(a[0], a[1], int(lda), w[0], w[1], int(ldw), int(M), int(N), int(K), L.ptr(bias), L.ACT_NONE, None, 0,
                                   out.data_ptr(), None, None, int(N), L.stream_ptr()), "ser_gemm_bf16_nt")


L._sig("ser_gemm_bf16_nt_splitk", L.i32, L.vp, L.vp, L.i32, L.vp, L.vp, L.i32, L.i32, L.i32, L.i32, L.i32, L.vp, L.vp)
L._sig("ser_sum_slabs_bias", L.i32, L.vp, L.i32, L.i64, L.i32, L.vp, L.vp, L.vp)


def _nt_wgrad(a, lda, w, ldw, M, N, K, out, three, kmin=1024, bias=None):
    """out[M,N] = A[M,K] . W[N,K]^T for a product whose K is long and whose output is small - a weight gradient (K = tokens of the
    batch, or conv frames), or the input gradient of a wide layer (K = 2 304 / 3 072 output features, kmin = 2048).  In the
    three-product mode the K range is cut into slices on different workgroups (`ser_gemm_bf16_nt_splitk`) and the partial sums are
    added in slice order by one column-sum launch; a hundred-odd output tiles would otherwise walk the whole range alone."""
    ks = 1
    if three and K >= kmin:
        ks = 8 if K >= 3200 else 4
    if ks == 1:
        return _nt(a, lda, w, ldw, M, N, K, out, bias)
    slabs = torch.empty(ks, M * N, dtype=torch.float32, device=out.device)
    L.check(L.lib.ser_gemm_bf16_nt_splitk(a[0], a[1], int(lda), w[0], w[1], int(ldw), int(M), int(N), int(K), ks, slabs.data_ptr(),
                                          L.stream_ptr()), "ser_gemm_bf16_nt_splitk")
    if bias is not None:                                                      # a forward product: slices + bias in one pass
        L.check(L.lib.ser_sum_slabs_bias(slabs.data_ptr(), ks, M * N, int(N), L.ptr(bias), out.data_ptr(), L.stream_ptr()), "ser_sum_slabs_bias")
    else:
        L.check(L.lib.ser_colsum(slabs.data_ptr(), ks, M * N, M * N, out.data_ptr(), 0, L.stream_ptr()), "ser_colsum")


def _tile_ok(M, N, K):
    return M >= 64 and N % 64 == 0 and K % 64 == 0


class WeightPlanes:
    """Operand planes of MANY weight matrices refreshed by ONE launch per step (`ser_split_bf16_both_multi`): for every entry the
    planes of W (forward product) and of W^T (input-gradient product) in persistent buffers - instead of a pass per weight inside
    every Linear's forward, and instead of concatenating q | k | v weights first: an entry may stack several matrices vertically
    (their planes land in row blocks of one fused operand, their biases in one fused bias vector).

    entries: {key: [(W, b | None), ...]} with all W of an entry [N_i, K] (same K, N_i % 32 == 0 when stacked).  three_f / three_b:
    interleaved hi / lo planes (three-product mode) for the forward / backward operand.  get(key) -> (ws, wt, Np, bias | None)."""

    def __init__(self, entries, three_f, three_b, device):
        import numpy as np
        self.mode = (bool(three_f), bool(three_b))
        self.out, rows = {}, []
        self.ptrs = []
        blk = 0
        for key, mats in entries.items():
            K = mats[0][0].shape[1]
            N = sum(int(W.shape[0]) for W, _ in mats)
            Np = (N + 63) // 64 * 64
            ws = torch.empty(N, (2 if three_f else 1) * K, dtype=torch.bfloat16, device=device)
            wt = torch.zeros(K, (2 if three_b else 1) * Np, dtype=torch.bfloat16, device=device)
            fused_bias = None
            if len(mats) > 1 and all(b is not None for _, b in mats):
                fused_bias = torch.empty(N, dtype=torch.float32, device=device)
            elif len(mats) == 1:
                fused_bias = mats[0][1]                         # the parameter itself: no copy
            r0 = 0
            for W, b in mats:
                R = int(W.shape[0])
                assert W.dim() == 2 and W.shape[1] == K and W.is_contiguous() and K % 32 == 0 and (len(mats) == 1 or R % 32 == 0)
                cover = Np if len(mats) == 1 else R
                s_hi = ws.data_ptr() + 2 * r0 * ws.shape[1]
                copy = len(mats) > 1 and b is not None and fused_bias is not None
                rows.append([W.data_ptr(), s_hi, s_hi + 2 * L.IL_GROUP if three_f else 0, wt.data_ptr(),
                             wt.data_ptr() + 2 * L.IL_GROUP if three_b else 0, R, K, Np, r0, K, cover,
                             b.data_ptr() if copy else 0, fused_bias.data_ptr() + 4 * r0 if copy else 0, R if copy else 0, blk, K // 32])
                self.ptrs.append((W, W.data_ptr()))
                blk += (cover // 32) * (K // 32)
                r0 += R
            self.out[key] = (ws, wt, Np, fused_bias)
        self.n, self.blocks = len(rows), blk
        self.table = torch.from_numpy(np.asarray(rows, dtype=np.int64)).to(device)

    def valid(self, mode):
        return self.mode == mode and all(W.data_ptr() == ptr for W, ptr in self.ptrs)

    def refresh(self):
        L.check(L.lib.ser_split_bf16_both_multi(self.table.data_ptr(), self.n, self.blocks, L.stream_ptr()), "ser_split_bf16_both_multi")
        return self

    def get(self, key):
        return self.out.get(key)


def weight_planes(model, entries_fn, rows):
    """The encoder's WeightPlanes for this step - built once (persistent buffers, device descriptor table), refreshed by one launch -
    or None when the layers are too small for the tile path (`_tile_ok`)."""
    three_f = L.lib.ser_get_linear_forward_products() == 3
    three_b = L.lib.ser_get_head_backward_products() == 3
    wp = getattr(model, "_ft_weight_planes", None)
    if wp is None or not wp.valid((three_f, three_b)):
        entries = entries_fn()
        if not entries or not all(_tile_ok(rows, sum(W.shape[0] for W, _ in mats), mats[0][0].shape[1]) for mats in entries.values()):
            model._ft_weight_planes = None
            return None
        dev = next(iter(entries.values()))[0][0].device
        wp = model._ft_weight_planes = WeightPlanes(entries, three_f, three_b, dev)
    return wp.refresh()


class _Linear(torch.autograd.Function):
    """y = x W^T + b on contiguous fp32 [M, K].

    Encoder-sized layers (M >= 64, N and K multiples of 64) run all three products on the MFMA tile kernel of the frozen
    encoders (csrc/gemm_bf16.hip, K-contiguous NT form): the operands are split into bf16 planes by one elementwise pass each
    - transposed where the product needs it (dx = dy W reads W^T; dW = dy^T x reads dy^T and x^T, the token axis zero-padded
    to a multiple of 64) - which costs a few MB of traffic per layer against a 5x faster product than the head's 64 x 64
    on-the-fly-split kernel.  Products per multiply: forward `ser_get_linear_forward_products` (3, or 1 under --use_amp),
    backward `ser_get_head_backward_products`.  Smaller layers keep the head's kernels."""

    @staticmethod
    def forward(ctx, x, W, b, prepared=None):
        """prepared: (ws, wt, Np, bias) of W from the step's WeightPlanes - the planes of W and W^T already exist."""
        ctx.save_for_backward(x, W)
        ctx.has_b = b is not None
        M, K = x.shape
        N = W.shape[0]
        ctx.tile = _tile_ok(M, N, K)
        if not ctx.tile:
            return O.linear_fwd(x, W, b)
        three = L.lib.ser_get_linear_forward_products() == 3
        y = torch.empty(M, N, dtype=torch.float32, device=x.device)
        if any(ctx.needs_input_grad[:3]):                         # (grad mode is off inside forward: this is the "will backward run" test)
            # one pass over x and one over W produce the forward operands AND the transposed ones of the backward products
            three_b = L.lib.ser_get_head_backward_products() == 3
            xs, xt, Mp = L.split_bf16_both(x, three, three_b)
            if prepared is not None:
                ws, wt, Np = prepared[:3]
            else:
                ws, wt, Np = L.split_bf16_both(W, three, three_b)
            ctx.planes_t = (xt, Mp, wt, Np, three_b)
            _nt_wgrad(_ptrs(xs, three), K, _ptrs(ws, three), K, M, N, K, y, three, kmin=2048, bias=b)   # (K = 3072: the FFN's second Linear)
            return y
        ctx.planes_t = None
        xs, ws = _planes(x, three), _planes(W, three)
        _nt(xs[1:], K, ws[1:], K, M, N, K, y, b)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        dy = dy.contiguous()
        M, K = x.shape
        N = W.shape[0]
        dW = torch.empty_like(W)
        db = torch.empty(N, dtype=torch.float32, device=W.device) if ctx.has_b else None
        if not ctx.tile:
            dx = O.linear_dgrad(dy, W) if ctx.needs_input_grad[0] else None
            _wgrad_now(dy, x, dW, db)
            return dx, dW, db, None
        three = L.lib.ser_get_head_backward_products() == 3
        saved = ctx.planes_t if (ctx.planes_t is not None and ctx.planes_t[4] == three) else None
        ctx.planes_t = None
        if saved is not None:
            xt, Mp, wt, Np, _ = saved
        else:
            (xt, Mp), (wt, Np) = L.split_bf16_t(x, three), L.split_bf16_t(W, three)
        if db is not None:                                       # the bias gradient's first stage rides on the operand pass over dy
            dys, dyt, Mp2, part = L.split_bf16_both(dy, three, three, colpart=True)
        else:
            dys, dyt, Mp2 = L.split_bf16_both(dy, three, three)
        assert Mp2 == Mp
        dx = torch.empty(M, K, dtype=torch.float32, device=x.device) if ctx.needs_input_grad[0] else None
        if dx is not None:                                       # dx[M,K] = dy[M,N] . (W^T)[K,N]^T
            _nt_wgrad(_ptrs(dys, three), N, _ptrs(wt, three), Np, M, K, N, dx, three, kmin=2048)
        _nt_wgrad(_ptrs(dyt, three), Mp, _ptrs(xt, three), Mp, N, K, Mp, dW, three)   # dW[N,K] = (dy^T)[N,Mp] . (x^T)[K,Mp]^T
        if db is not None:                                       # second stage: the Mp / 32 block sums, in block order
            L.check(L.lib.ser_colsum(L.ptr(part), part.shape[0], N, N, L.ptr(db), 0, L.stream_ptr()), "ser_colsum")
        return dx, dW, db, None


class _LinearQKV(torch.autograd.Function):
    """[q | k | v] = x [Wq; Wk; Wv]^T + [bq | bk | bv] as ONE product per direction on the fused operand planes of the step's
    WeightPlanes (no concatenation of the weights): forward, dx = dy W, dW = dy^T x; the three weight / bias gradients are the row
    blocks of one [3 H, K] / [3 H] result."""

    @staticmethod
    def forward(ctx, x, Wq, Wk, Wv, bq, bk, bv, prepared):
        ws, wt, Np, bias = prepared
        M, K = x.shape
        N = ws.shape[0]
        three = L.lib.ser_get_linear_forward_products() == 3
        three_b = L.lib.ser_get_head_backward_products() == 3
        xs, xt, Mp = L.split_bf16_both(x, three, three_b)
        ctx.planes_t = (xt, Mp, wt, Np, three_b)
        ctx.dims = (M, N, K, [int(Wq.shape[0]), int(Wk.shape[0]), int(Wv.shape[0])])
        y = torch.empty(M, N, dtype=torch.float32, device=x.device)
        _nt(_ptrs(xs, three), K, _ptrs(ws, three), K, M, N, K, y, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        xt, Mp, wt, Np, three = ctx.planes_t
        ctx.planes_t = None
        M, N, K, parts = ctx.dims
        assert three == (L.lib.ser_get_head_backward_products() == 3), "the backward product count changed between forward and backward"
        dy = dy.contiguous()
        dys, dyt, Mp2, part = L.split_bf16_both(dy, three, three, colpart=True)
        assert Mp2 == Mp
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(M, K, dtype=torch.float32, device=dy.device)
            _nt_wgrad(_ptrs(dys, three), N, _ptrs(wt, three), Np, M, K, N, dx, three, kmin=2048)
        dW = torch.empty(N, K, dtype=torch.float32, device=dy.device)
        _nt_wgrad(_ptrs(dyt, three), Mp, _ptrs(xt, three), Mp, N, K, Mp, dW, three)
        db = torch.empty(N, dtype=torch.float32, device=dy.device)
        L.check(L.lib.ser_colsum(L.ptr(part), part.shape[0], N, N, L.ptr(db), 0, L.stream_ptr()), "ser_colsum")
        a, b_ = parts[0], parts[0] + parts[1]
        return dx, dW[:a], dW[a:b_], dW[b_:], db[:a], db[a:b_], db[b_:], None


def linear(x, W, b=None, prepared=None):
    shp = x.shape
    y = _Linear.apply(x.reshape(-1, shp[-1]).contiguous(), W, b, prepared)
    return y.reshape(*shp[:-1], W.shape[0])


SLACK = 16      # zero rows behind the last clip of a padded activation buffer (the last windows read past it)


def padded_lengths(T, kernels, strides):
    """Frame counts L[i] of the conv stack and padded per-clip row counts Lp[i] with Lp[i-1] == stride_i * Lp[i]: with every
    clip of layer i-1 stored on Lp[i-1] rows, window row t of clip b of layer i starts at row s (b Lp[i] + t) of the flat
    [B * Lp[i-1], C] buffer — ONE strided view for the whole batch, so a conv layer (forward, weight gradient, each tap of
    the input gradient) is one GEMM instead of one per clip.  The rows t >= L[i] of a clip are padding: they hold finite
    junk going forward (never read by a valid window: Lp[i-1] >= s (L[i] - 1) + k) and zeros going backward."""
    L = [int(T)]
    for k, s_ in zip(kernels, strides):
        L.append((L[-1] - k) // s_ + 1)
    n = len(kernels)
    pad = 0
    while True:
        Lp = [0] * (n + 1)
        Lp[n] = L[n] + pad
        for i in range(n, 0, -1):
            Lp[i - 1] = strides[i - 1] * Lp[i]
        if all(Lp[i - 1] >= strides[i - 1] * (L[i] - 1) + kernels[i - 1] for i in range(1, n + 1)):
            return L, Lp
        pad += 1


class _ConvPad(torch.autograd.Function):
    """Conv1d(C_in -> C_out, kernel k, stride s, no padding, no bias), channels-last, whole batch at once on the padded
    row stride (see padded_lengths): x [rows_in + SLACK, C_in] -> y [rows_out + SLACK, C_out], rows_in = s * rows_out.
    W2 [C_out, k*C_in] holds the taps in (tap, channel) order.

    Channel counts that are multiples of 64 (the 512-channel layers of the Base front end) run on the MFMA tile kernel of the
    frozen encoders, like `_Linear`: forward = one NT product over the window view of the split planes of x (row step s*C_in,
    k*C_in contiguous values: no im2col); dW2 = dy^T . windows with the transposed planes of the k strided row subsets x[j::s]
    stacked under each other; dx = col2im(dy . W2) - the per-window gradients as one product, folded back by a gather kernel.
    Other shapes keep the head's strided fp32 GEMM: forward over the window view, dW2 one GEMM, dx one accumulating GEMM per tap."""

    @staticmethod
    def forward(ctx, x, W2, k, s, rows_out):
        Cin, Cout = x.shape[1], W2.shape[0]
        assert x.shape[0] >= s * (rows_out - 1) + k
        ctx.tile = Cin % 64 == 0 and Cout % 64 == 0 and rows_out >= 64 and x.is_contiguous()
        ctx.save_for_backward(x, W2)
        ctx.cfg = (k, s, rows_out)
        if ctx.tile:
            three = L.lib.ser_get_linear_forward_products() == 3
            xs, ws = _planes(x, three), _planes(W2.contiguous(), three)
            y = torch.empty(rows_out + SLACK, Cout, dtype=torch.float32, device=x.device)
            y[rows_out:].zero_()
            _nt(xs[1:], s * Cin, ws[1:], k * Cin, rows_out, Cout, k * Cin, y)
            return y
        y = torch.zeros(rows_out + SLACK, Cout, dtype=torch.float32, device=x.device)
        _gemm(x.data_ptr(), s * Cin, 1, W2.data_ptr(), 1, k * Cin, rows_out, Cout, k * Cin, y, Cout)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W2 = ctx.saved_tensors
        k, s, rows_out = ctx.cfg
        Cin, Cout = x.shape[1], W2.shape[0]
        dy = dy.contiguous()
        dW2 = torch.empty_like(W2)
        if ctx.tile:
            three = L.lib.ser_get_head_backward_products() == 3
            dyv = dy[:rows_out]
            Mp = (rows_out + 63) // 64 * 64
            winT = torch.empty(k * Cin, (2 if three else 1) * Mp, dtype=torch.bfloat16, device=x.device)
            dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
            dwin = torch.empty(rows_out, k * Cin, dtype=torch.float32, device=x.device) if dx is not None else None
            dys, dyt, Mp2 = L.split_bf16_both(dyv, three, three)                     # dy and (dy^T)[Cout, Mp] in one pass
            assert Mp2 == Mp
            for j in range(k):                                                       # rows j*Cin.. of windows^T = (x[j::s])^T
                L.split_bf16_t(x[j::s][:rows_out], three, out=winT[j * Cin:(j + 1) * Cin])
            _nt_wgrad(_ptrs(dyt, three), Mp, _ptrs(winT, three), Mp, Cout, k * Cin, Mp, dW2, three)
            if dx is not None:
                w2t = _planes_t(W2.contiguous(), three)                              # dwin[M, k Cin] = dy . W2
                _nt(_ptrs(dys, three), Cout, w2t[1:3], w2t[3], rows_out, k * Cin, Cout, dwin)
                L.check(L.lib.ser_conv_col2im(L.ptr(dwin), rows_out, k, s, Cin, x.shape[0], L.ptr(dx), L.stream_ptr()), "ser_conv_col2im")
            return dx, dW2, None, None, None
        _gemm(dy.data_ptr(), 1, Cout, x.data_ptr(), s * Cin, 1, Cout, k * Cin, rows_out, dW2, k * Cin)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.zeros_like(x)
            for j in range(k):
                _gemm(dy.data_ptr(), Cout, 1, W2.data_ptr() + 4 * j * Cin, k * Cin, 1, rows_out, Cin, Cout,
                      dx.data_ptr() + 4 * j * Cin, s * Cin, accumulate=True)
        return dx, dW2, None, None, None


class _Gelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        ctx.save_for_backward(x)
        return O.act_fwd(x, O.ACT_GELU)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        L.check(L.lib.ser_gelu_bwd(L.ptr(dy), L.ptr(x), x.numel(), L.ptr(dx), L.stream_ptr()), "ser_gelu_bwd")
        return dx


gelu = _Gelu.apply


class _GeluDrop(torch.autograd.Function):
    """dropout(gelu(x)) - the FFN activation followed by its activation dropout - as one pass forward and one backward."""

    @staticmethod
    def forward(ctx, x, state, p, site):
        x = x.contiguous()
        ctx.save_for_backward(x)
        ctx.key = (state, p, site)
        return O.act_fwd(x, O.ACT_GELU, (state, p), site)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        state, p, site = ctx.key
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        L.check(L.lib.ser_gelu_drop_bwd(L.ptr(dy), L.ptr(x), x.numel(), L.ptr(dx), L.ptr(state), int(site), float(p), L.stream_ptr()),
                "ser_gelu_drop_bwd")
        return dx, None, None, None


def gelu_drop(x, noise, p, site):
    """gelu(x), then the dropout site `site` of an encoder when its noise is on and a dropout scope is active."""
    d = O.dropout_ctx(p) if (noise is not None and p > 0.0) else None
    if d is None:
        return gelu(x)
    return _GeluDrop.apply(x, d[0], d[1], site)


class _LayerNorm(torch.autograd.Function):
    """LN(x (+ x2)) over the last dim on [rows, D]: the head's kernels (saved: the sum, mean, rstd)."""

    @staticmethod
    def forward(ctx, x, x2, gamma, beta, eps):
        y, saved = O.ln_fwd(x.contiguous(), gamma, beta, eps, x2=None if x2 is None else x2.contiguous())
        ctx.save_for_backward(saved[0], saved[1], saved[2], gamma)
        ctx.two = x2 is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        z, mean, rstd, gamma = ctx.saved_tensors
        dg, db = torch.empty_like(gamma), torch.empty_like(gamma)
        dx = O.ln_bwd(dy.contiguous(), (z, mean, rstd), gamma, dg, db)
        return dx, (dx if ctx.two else None), dg, db, None


_sig("ser_layernorm_drop_fwd", L.i32, L.vp, L.vp, L.vp, L.vp, L.f32, L.i32, L.i32, L.vp, L.vp, L.vp, L.vp, L.vp, C.c_uint, L.f32, L.vp)
_sig("ser_layernorm_drop_bwd", L.i32, L.vp, L.vp, L.vp, L.vp, L.vp, L.i32, L.i32, L.vp, L.vp, L.vp, L.vp, L.i32, L.vp, L.vp, C.c_uint,
     L.f32, L.vp)
_sig("ser_layernorm_bwd_workspace_bytes", L.sz, L.i32, L.i32)


class _LayerNormDrop(torch.autograd.Function):
    """LN(dropout(x) + x2) on [rows, D]: hidden dropout, residual add and LayerNorm of a post-LN block as one pass forward and one
    pass (+ the parameter-gradient reduction) backward - the same values as `_Dropout` followed by `_LayerNorm`."""

    @staticmethod
    def forward(ctx, x, x2, gamma, beta, eps, state, p, site):
        x, x2 = x.contiguous(), x2.contiguous()
        rows, D = x.shape
        y, z = torch.empty_like(x), torch.empty_like(x)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        L.check(L.lib.ser_layernorm_drop_fwd(L.ptr(x), L.ptr(x2), L.ptr(gamma), L.ptr(beta), eps, rows, D, L.ptr(y), L.ptr(z), L.ptr(mean),
                                             L.ptr(rstd), L.ptr(state), int(site), float(p), L.stream_ptr()), "ser_layernorm_drop_fwd")
        ctx.save_for_backward(z, mean, rstd, gamma)
        ctx.key = (state, p, site)
        return y

    @staticmethod
    def backward(ctx, dy):
        z, mean, rstd, gamma = ctx.saved_tensors
        state, p, site = ctx.key
        dy = dy.contiguous()
        rows, D = dy.shape
        dx, dx2 = torch.empty_like(dy), torch.empty_like(dy)
        dg, db = torch.empty_like(gamma), torch.empty_like(gamma)
        nb = L.lib.ser_layernorm_bwd_workspace_bytes(rows, D)
        ws = torch.empty(nb, dtype=torch.uint8, device=dy.device) if nb else None
        L.check(L.lib.ser_layernorm_drop_bwd(L.ptr(dy), L.ptr(z), L.ptr(mean), L.ptr(rstd), L.ptr(gamma), rows, D, L.ptr(dx), L.ptr(dx2),
                                             L.ptr(dg), L.ptr(db), 0, L.ptr(ws), L.ptr(state), int(site), float(p), L.stream_ptr()),
                "ser_layernorm_drop_bwd")
        return dx, dx2, dg, db, None, None, None, None


def layer_norm_drop(x, gamma, beta, eps, residual, noise, p, site):
    """LN(dropout_site(x) + residual); plain LN(x + residual) when the encoder's noise is off or no dropout scope is active."""
    d = O.dropout_ctx(p) if (noise is not None and p > 0.0) else None
    if d is None:
        return layer_norm(x, gamma, beta, eps, residual=residual)
    shp = x.shape
    y = _LayerNormDrop.apply(x.reshape(-1, shp[-1]), residual.reshape(-1, shp[-1]), gamma, beta, eps, d[0], d[1], site)
    return y.reshape(shp)


def layer_norm(x, gamma, beta, eps, residual=None):
    shp = x.shape
    y = _LayerNorm.apply(x.reshape(-1, shp[-1]), None if residual is None else residual.reshape(-1, shp[-1]), gamma, beta, eps)
    return y.reshape(shp)


class _ColNorm(torch.autograd.Function):
    """GroupNorm(C, C) on channels-last [B, L, C]: per (clip, channel) statistics over time (hf :302-323)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, B, Ln, Ls):
        """x [>= B*Ls rows, C]: clip b owns rows b*Ls .. b*Ls+Ls-1, the first Ln of them are its frames."""
        x = x.contiguous()
        Cc = x.shape[1]
        y = torch.zeros_like(x)
        mean = torch.empty(B, Cc, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        ws = torch.empty(int(L.lib.ser_colnorm_workspace_bytes(B, Cc)), dtype=torch.uint8, device=x.device)
        L.check(L.lib.ser_colnorm_fwd(L.ptr(x), B, Ln, Ls, Cc, L.ptr(gamma), L.ptr(beta), eps, L.ptr(y), L.ptr(mean), L.ptr(rstd), L.ptr(ws),
                                      L.stream_ptr()), "ser_colnorm_fwd")
        ctx.save_for_backward(x, mean, rstd, gamma)
        ctx.dims = (B, Ln, Ls)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd, gamma = ctx.saved_tensors
        B, Ln, Ls = ctx.dims
        Cc = x.shape[1]
        dy = dy.contiguous()
        dx = torch.zeros_like(x) if ctx.needs_input_grad[0] else None
        dg, db = torch.empty_like(gamma), torch.empty_like(gamma)
        ws = torch.empty(int(L.lib.ser_colnorm_workspace_bytes(B, Cc)), dtype=torch.uint8, device=x.device)
        L.check(L.lib.ser_colnorm_bwd(L.ptr(dy), L.ptr(x), L.ptr(mean), L.ptr(rstd), L.ptr(gamma), B, Ln, Ls, Cc, L.ptr(dx), L.ptr(dg),
                                      L.ptr(db), 0, L.ptr(ws), L.stream_ptr()), "ser_colnorm_bwd")
        return dx, dg, db, None, None, None, None


class _AttentionQKV(torch.autograd.Function):
    """softmax(q k^T / sqrt(d) + key mask) v per head (hf wav2vec2 :438-463, xlm_roberta :211-250) on the fused projection output
    qkv [B*S, 3 H] (q | k | v column blocks): the head's attention kernels read the blocks through their row stride and backward
    writes dq | dk | dv into one [B*S, 3 H] gradient - no slice copies going in and no zero-fill + copy + add per block coming back."""

    @staticmethod
    def forward(ctx, qkv, key_mask, B, S, heads, drop, site):
        qkv = qkv.contiguous()
        H = qkv.shape[1] // 3
        out, P = O.xattn_fwd(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], key_mask, B, S, S, heads, drop, site)
        ctx.P = P
        ctx.save_for_backward(qkv)
        ctx.dims = (B, S, heads, drop, site)
        return out

    @staticmethod
    def backward(ctx, dctx):
        (qkv,) = ctx.saved_tensors
        B, S, heads, drop, site = ctx.dims
        H = qkv.shape[1] // 3
        d = torch.empty_like(qkv)
        O.xattn_bwd(dctx.contiguous(), qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], ctx.P, B, S, S, heads, drop, site,
                    out=(d[:, :H], d[:, H:2 * H], d[:, 2 * H:]))
        return d, None, None, None, None, None, None


class _PosConv(torch.autograd.Function):
    """The whole positional conv (hf :326-368: grouped Conv1d, kernel K, padding K // 2, last frame dropped when K is even) on
    z [B*S, H] with the weight-normed weight Wp [H, Cg, K].  One group = one product over a window view: its slab [B*R + K, Cg] holds
    per clip R = S + K - 1 zero-padded rows, clips back to back (+ K rows of slack), so the window rows of every clip are ONE strided
    view with row step Cg, out[m] = slab[m : m + K].flatten() . W2^T for m in [0, B*R) - the rows m = b R + t, t >= S, straddle two
    clips and are never used (their upstream gradient is zero).  The zero
    padding, the group split of z and of the weight, and the reassembly of the output happen ONCE for all groups (a padded
    [G, B*R + K, Cg] slab buffer, one [H, K*Cg] weight matrix, one [G, B*R, Cg] output), in forward and in backward, instead of a
    pad / cat / slice chain per group on the autograd tape.  Products per multiply follow the Linear layers' setting."""

    @staticmethod
    def forward(ctx, z, Wp, bias, B, S, K, G):
        H = z.shape[1]
        Cg, R = H // G, S + K - 1
        rows = B * R
        slabs = torch.zeros(G, rows + K, Cg, dtype=torch.float32, device=z.device)
        slabs[:, :rows].view(G, B, R, Cg)[:, :, K // 2:K // 2 + S] = z.view(B, S, G, Cg).permute(2, 0, 1, 3)
        W2 = Wp.permute(0, 2, 1).reshape(H, K * Cg).contiguous()               # taps in (tap, channel) order; rows g*Cg.. = group g
        bias = bias.contiguous()
        y = torch.empty(G, rows, Cg, dtype=torch.float32, device=z.device)
        np_ = L.lib.ser_get_linear_forward_products()
        for g in range(G):
            _gemm(slabs[g].data_ptr(), Cg, 1, W2[g * Cg:].data_ptr(), 1, K * Cg, rows, Cg, K * Cg, y[g].data_ptr(), Cg,
                  bias=bias[g * Cg:(g + 1) * Cg], products=np_)
        ctx.save_for_backward(slabs, W2)
        ctx.dims = (B, S, K, G)
        return y.view(G, B, R, Cg)[:, :, :S].permute(1, 2, 0, 3).reshape(B * S, H)

    @staticmethod
    def backward(ctx, dout):
        slabs, W2 = ctx.saved_tensors
        B, S, K, G = ctx.dims
        H = W2.shape[0]
        Cg, R = H // G, S + K - 1
        rows = B * R
        dout = dout.contiguous()
        # dy of group g = rows K-1 .. K-1+rows of a buffer with K-1 zero rows in front and K+1 behind: the input gradient is then
        # the SAME window-view product as forward, over this buffer with the taps reversed and the channel roles swapped -
        # dslab[r, ci] = sum_{k', co} dyp[r + k', co] Wt[ci, k' Cg + co], Wt[ci, k', co] = W2[co, K-1-k', ci] - one GEMM per group
        # instead of a [rows, K Cg] buffer of per-window gradients and an overlap-add pass.  Rows t >= S of a clip straddle two
        # clips in forward: no gradient.
        dyp = torch.zeros(G, rows + 2 * K, Cg, dtype=torch.float32, device=dout.device)
        dyp[:, K - 1:K - 1 + rows].view(G, B, R, Cg)[:, :, :S] = dout.view(B, S, G, Cg).permute(2, 0, 1, 3)
        dW2 = torch.empty_like(W2)
        db = torch.empty(H, dtype=torch.float32, device=dout.device)
        # the bias reaches every output frame once: its gradient is the column sum of dout itself (one two-stage sum for all groups)
        ws = torch.empty(int(L.lib.ser_colsum_tall_workspace_bytes(H)), dtype=torch.uint8, device=dout.device)
        L.check(L.lib.ser_colsum_tall(L.ptr(dout), B * S, H, H, L.ptr(db), L.ptr(ws), L.stream_ptr()), "ser_colsum_tall")
        np_ = L.lib.ser_get_head_backward_products()
        esz = 4 * Cg                                                           # bytes per row
        for g in range(G):
            dy_g = dyp[g].data_ptr() + (K - 1) * esz
            _gemm(dy_g, 1, Cg, slabs[g].data_ptr(), Cg, 1, Cg, K * Cg, rows, dW2[g * Cg:].data_ptr(), K * Cg, products=np_)
        dz = None
        if ctx.needs_input_grad[0]:
            Wt = W2.view(G, Cg, K, Cg).flip(2).permute(0, 3, 2, 1).contiguous()   # [G, ci, k', co]
            dslabs = torch.empty(G, rows + K, Cg, dtype=torch.float32, device=dout.device)
            for g in range(G):
                _gemm(dyp[g].data_ptr(), Cg, 1, Wt[g].data_ptr(), 1, K * Cg, rows + K, Cg, K * Cg, dslabs[g].data_ptr(), Cg, products=np_)
            dz = dslabs[:, :rows].view(G, B, R, Cg)[:, :, K // 2:K // 2 + S].permute(1, 2, 0, 3).reshape(B * S, H)
        return dz, dW2.view(H, K, Cg).permute(0, 2, 1), db, None, None, None, None


class _PosConvDirect(torch.autograd.Function):
    """GELU(posconv(z) + bias) + z - the positional conv, its activation and the residual add (hf :326-368, :706-708) - on the frozen
    encoders' resident-slab kernel at the Base geometry (K = 128, 16 groups of 48; `ser_posconv_direct_supported`): the S + 127
    rows of a (clip, group) live in LDS as split planes, only the weights stream.  Forward keeps the pre-activation; backward is
    dpre = dout . GELU'(pre), then the SAME kernel as a correlation over dpre with the taps reversed and the channel roles swapped
    (+ dout for the residual branch) for dz, and the window-view products of `_PosConv` for the weight gradient.  Three products per
    multiply in both precision modes."""

    @staticmethod
    def forward(ctx, z, Wp, bias, B, S, K, G):
        z, Wp, bias = z.contiguous(), Wp.contiguous(), bias.contiguous()
        H = z.shape[1]
        packed = torch.empty(H * K * 128, dtype=torch.bfloat16, device=z.device)
        L.check(L.lib.ser_posconv_pack(L.ptr(Wp), H, G, K, 0, packed.data_ptr(), L.stream_ptr()), "ser_posconv_pack")
        out, pre = torch.empty_like(z), torch.empty_like(z)
        L.check(L.lib.ser_posconv_fwd(L.ptr(z), packed.data_ptr(), L.ptr(bias), B, S, H, G, K, L.ptr(out), L.ptr(pre), L.stream_ptr()),
                "ser_posconv_fwd")
        ctx.save_for_backward(z, Wp, pre)
        ctx.dims = (B, S, K, G)
        return out

    @staticmethod
    def backward(ctx, dout):
        z, Wp, pre = ctx.saved_tensors
        B, S, K, G = ctx.dims
        H = z.shape[1]
        Cg, R = H // G, S + K - 1
        rows = B * R
        dout = dout.contiguous()
        dpre = torch.empty_like(dout)
        L.check(L.lib.ser_gelu_bwd(L.ptr(dout), L.ptr(pre), dout.numel(), L.ptr(dpre), L.stream_ptr()), "ser_gelu_bwd")
        dz = None
        if ctx.needs_input_grad[0]:
            packed = torch.empty(H * K * 128, dtype=torch.bfloat16, device=z.device)
            L.check(L.lib.ser_posconv_pack(L.ptr(Wp), H, G, K, 1, packed.data_ptr(), L.stream_ptr()), "ser_posconv_pack")
            dz = torch.empty_like(dout)
            L.check(L.lib.ser_posconv_dgrad(L.ptr(dpre), packed.data_ptr(), L.ptr(dout), B, S, H, G, K, L.ptr(dz), L.stream_ptr()), "ser_posconv_dgrad")
        # weight / bias gradients: per group dW2_g = dy_g^T . windows(slab_g) over the zero-padded slabs (see _PosConv)
        slabs = torch.zeros(G, rows + K, Cg, dtype=torch.float32, device=z.device)
        slabs[:, :rows].view(G, B, R, Cg)[:, :, K // 2:K // 2 + S] = z.view(B, S, G, Cg).permute(2, 0, 1, 3)
        dy = torch.zeros(G, rows, Cg, dtype=torch.float32, device=z.device)
        dy.view(G, B, R, Cg)[:, :, :S] = dpre.view(B, S, G, Cg).permute(2, 0, 1, 3)
        dW2 = torch.empty(H, K * Cg, dtype=torch.float32, device=z.device)
        db = torch.empty(H, dtype=torch.float32, device=z.device)
        ws = torch.empty(int(L.lib.ser_colsum_tall_workspace_bytes(H)), dtype=torch.uint8, device=z.device)
        L.check(L.lib.ser_colsum_tall(L.ptr(dpre), B * S, H, H, L.ptr(db), L.ptr(ws), L.stream_ptr()), "ser_colsum_tall")
        np_ = L.lib.ser_get_head_backward_products()
        for g in range(G):
            _gemm(dy[g].data_ptr(), 1, Cg, slabs[g].data_ptr(), Cg, 1, Cg, K * Cg, rows, dW2[g * Cg:].data_ptr(), K * Cg, products=np_)
        return dz, dW2.view(H, K, Cg).permute(0, 2, 1), db, None, None, None, None


class _Embed(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ids, pos, wemb, pemb, temb, pad_id):
        rows, D = ids.numel(), wemb.shape[1]
        e = torch.empty(rows, D, dtype=torch.float32, device=wemb.device)
        L.check(L.lib.ser_embed_fwd(L.ptr(ids), L.ptr(pos), L.ptr(wemb), L.ptr(pemb), L.ptr(temb), rows, D, wemb.shape[0], pemb.shape[0],
                                    L.ptr(e), L.stream_ptr()), "ser_embed_fwd")
        ctx.save_for_backward(ids, pos)
        ctx.shapes = (wemb.shape, pemb.shape, temb.shape, pad_id)
        return e

    @staticmethod
    def backward(ctx, de):
        ids, pos = ctx.saved_tensors
        ws, ps, ts, pad_id = ctx.shapes
        de = de.contiguous()
        dw = torch.zeros(ws, dtype=torch.float32, device=de.device)
        dp = torch.zeros(ps, dtype=torch.float32, device=de.device)
        dt = torch.zeros(ts, dtype=torch.float32, device=de.device)
        L.check(L.lib.ser_embed_bwd(L.ptr(de), L.ptr(ids), L.ptr(pos), ids.numel(), ws[1], ws[0], ps[0], pad_id, L.ptr(dw), L.ptr(dp),
                                    L.ptr(dt), L.stream_ptr()), "ser_embed_bwd")
        return None, None, dw, dp, dt, None


# ---- training-mode noise of the encoders themselves (reference: `.train()` on both encoders, src/train.py:124) ----------
# HF hidden / attention / activation dropout, LayerDrop and SpecAugment.  Off unless the owning module sets
# `encoder_train_noise` AND a dropout scope is active (SERSystem.loss in training mode): parity with the golden vectors is
# defined without it.  Dropout masks come from the head's counter-based generator (state word advanced once per step);
# LayerDrop decisions and SpecAugment spans are host draws, as in HF (`torch.rand([])` / numpy), from a generator owned by
# the encoder.  Site ids: SITE0 + 500 * encoder + 8 * layer + {0 attention probabilities, 1 hidden after attention,
# 2 activation, 3 hidden after FFN}; + 400 feature projection, + 401 encoder input, + 402 embeddings.
SITE0 = 1000


class _Dropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, state, p, site):
        ctx.key = (state, p, site)
        return O.dropout(x, (state, p), site)

    @staticmethod
    def backward(ctx, dy):
        state, p, site = ctx.key
        return O.dropout(dy, (state, p), site), None, None, None


class Noise:
    """What one forward of one encoder draws.  `plan()` fixes the host decisions (LayerDrop, SpecAugment) for a batch.

    Two ways of applying them: the eager forward calls `plan()` itself and acts on the host values (a skipped layer is not
    run).  A captured forward cannot branch on the host, so there the owner of the graph calls `plan()` + `stage()` BEFORE each
    replay (`static = True`: the forward draws nothing): the decisions travel as device words - `skip_dev[l]` non-zero = layer l
    is dropped, `spec_dev[b * S + t]` = frame masked - every layer is computed and a dropped one discarded by a select, and the
    optimizer leaves its parameters alone through the same words (FlatAdamW.set_gates)."""

    def __init__(self, cfg, enc_index, seed=0):
        import numpy as np
        self.enc = enc_index
        self.rng = np.random.default_rng(seed + 7 * enc_index)
        g = lambda n, d=0.0: float(getattr(cfg, n, d) or 0.0)
        if enc_index == 0:       # Wav2Vec2Config
            self.p_hidden, self.p_attn, self.p_act = g("hidden_dropout"), g("attention_dropout"), g("activation_dropout")
            self.p_featproj, self.layerdrop = g("feat_proj_dropout"), g("layerdrop")
            self.mask_prob, self.mask_len, self.mask_min = g("mask_time_prob"), int(getattr(cfg, "mask_time_length", 10)), int(getattr(cfg, "mask_time_min_masks", 2))
            self.spec = bool(getattr(cfg, "apply_spec_augment", True)) and self.mask_prob > 0
        else:                    # XLMRobertaConfig
            self.p_hidden, self.p_attn, self.p_act = g("hidden_dropout_prob"), g("attention_probs_dropout_prob"), 0.0
            self.p_featproj, self.layerdrop, self.spec = 0.0, 0.0, False
        self.skip, self.spec_mask = set(), None
        self.static = False
        self.skip_dev = self.spec_dev = None
        self._shape = None

    def site(self, layer, k):
        return SITE0 + 500 * self.enc + 8 * layer + k

    def plan(self, n_layers, B, S):
        """Host draws for this batch: layers to skip (hf wav2vec2 :700-703) and SpecAugment rows (one batch-1 call per clip,
        as the reference's per-utterance loop makes them: hf :1293-1302 with _compute_mask_indices :101-218)."""
        self._shape = (n_layers, B, S)
        self.skip = {l for l in range(n_layers) if self.layerdrop > 0 and self.rng.random() < self.layerdrop}
        self.spec_mask = None
        if self.spec and S >= self.mask_len:
            import numpy as np
            m = np.zeros((B, S), dtype=bool)
            for b in range(B):
                eps = self.rng.random()
                n = max(int(self.mask_prob * S / self.mask_len + eps), self.mask_min)
                if n * self.mask_len > S:
                    n = S // self.mask_len
                n = min(n, max(S - (self.mask_len - 1), 0))
                if n == 0:
                    continue
                starts = self.rng.choice(np.arange(S - (self.mask_len - 1)), n, replace=False)
                for st in starts:
                    m[b, st:st + self.mask_len] = True
            self.spec_mask = m
        return self

    def stage(self, device):
        """The planned decisions as device words (call outside a capture, before the replay that reads them)."""
        import numpy as np
        n_layers, B, S = self._shape
        if self.skip_dev is None or self.skip_dev.numel() != n_layers or self.skip_dev.device != torch.device(device):
            self.skip_dev = torch.zeros(n_layers, dtype=torch.int32, device=device)
        # one persistent buffer per batch shape: a captured graph keeps reading the buffer it was captured with, so a shape's
        # buffer must never be replaced (steps of several shapes alternate)
        bufs = self.__dict__.setdefault("_spec_devs", {})
        key = (B * S, str(torch.device(device)))
        if key not in bufs:
            bufs[key] = torch.zeros(B * S, dtype=torch.bool, device=device)
        self.spec_dev = bufs[key]
        sk = np.zeros(n_layers, dtype=np.int32)
        sk[list(self.skip)] = 1
        self.skip_dev.copy_(torch.from_numpy(sk), non_blocking=False)
        m = self.spec_mask if self.spec_mask is not None else np.zeros((B, S), dtype=bool)
        self.spec_dev.copy_(torch.from_numpy(np.ascontiguousarray(m.reshape(B * S))), non_blocking=False)
        return self

    def state(self):
        return self.rng.bit_generator.state

    def set_state(self, st):
        self.rng.bit_generator.state = st


def wav2vec2_frames(cfg, T):
    """Frames the conv stack makes of T samples (hf :1113-1130)."""
    n = int(T)
    for k, s_ in zip(cfg.conv_kernel, cfg.conv_stride):
        n = (n - k) // s_ + 1
    return n


def _drop(x, noise, p, site):
    """Dropout site of an encoder: identity unless the encoder's noise is on and a dropout scope is active."""
    if noise is None or p <= 0.0:
        return x
    d = O.dropout_ctx(p)
    if d is None:
        return x
    return _Dropout.apply(x.contiguous(), d[0], d[1], site)


def normalise_waves(wave):
    """[B, T] raw clips -> zero mean / unit variance per clip (hf feature_extraction_wav2vec2.py:78-96)."""
    wave = wave.contiguous()
    B, T = wave.shape
    out = torch.empty_like(wave)
    stats = torch.empty(B, 2, dtype=torch.float32, device=wave.device)
    L.check(L.lib.ser_wave_normalize(L.ptr(wave), B, T, L.ptr(out), L.ptr(stats), L.stream_ptr()), "ser_wave_normalize")
    return out


def _layer_entries(p, prefixes, names):
    """WeightPlanes entries of transformer layers: q | k | v stacked into one operand, the other three Linears on their own."""
    g = lambda pre, n: (p[pre + names[n] + ".weight"], p[pre + names[n] + ".bias"])
    out = {}
    for pre in prefixes:
        out[pre + "qkv"] = [g(pre, "q"), g(pre, "k"), g(pre, "v")]
        for n in ("o", "f1", "f2"):
            out[pre + n] = [g(pre, n)]
    return out


def _transformer_layer(h, p, prefix, names, B, S, heads, eps, key_mask, noise=None, layer=0, wp=None):
    """Post-LN block (hf wav2vec2 :591-608 / xlm_roberta :421-463), h [B*S, H].  With `noise`: attention-probability,
    hidden (after the attention output and after the FFN output) and activation dropout, as the HF modules place them.
    wp: the step's WeightPlanes (operand planes of every Linear weight of the encoder, refreshed by one launch)."""
    g = lambda n: p[prefix + n]
    pre = (lambda n: wp.get(prefix + n)) if wp is not None else (lambda n: None)
    Hd = h.shape[1]
    if wp is not None:
        qkv = _LinearQKV.apply(h.contiguous(), g(names["q"] + ".weight"), g(names["k"] + ".weight"), g(names["v"] + ".weight"),
                               g(names["q"] + ".bias"), g(names["k"] + ".bias"), g(names["v"] + ".bias"), pre("qkv"))
    else:
        # q, k, v as ONE product over the concatenated weights (a copy of 3 H^2 values per layer and step; autograd slices the
        # weight gradient back): one split of h, one launch, N = 3 H
        qkv = linear(h, torch.cat([g(names["q"] + ".weight"), g(names["k"] + ".weight"), g(names["v"] + ".weight")], dim=0),
                     torch.cat([g(names["q"] + ".bias"), g(names["k"] + ".bias"), g(names["v"] + ".bias")], dim=0))
    adrop = O.dropout_ctx(noise.p_attn) if noise is not None else None
    ctx = _AttentionQKV.apply(qkv, key_mask, B, S, heads, adrop, noise.site(layer, 0) if noise is not None else 0)
    a = linear(ctx, g(names["o"] + ".weight"), g(names["o"] + ".bias"), pre("o"))
    ph = noise.p_hidden if noise is not None else 0.0
    st = (lambda k: noise.site(layer, k)) if noise is not None else (lambda k: 0)
    h = layer_norm_drop(a, g(names["ln1"] + ".weight"), g(names["ln1"] + ".bias"), eps, h, noise, ph, st(1))
    f = gelu_drop(linear(h, g(names["f1"] + ".weight"), g(names["f1"] + ".bias"), pre("f1")), noise, noise.p_act if noise is not None else 0.0, st(2))
    f = linear(f, g(names["f2"] + ".weight"), g(names["f2"] + ".bias"), pre("f2"))
    return layer_norm_drop(f, g(names["ln2"] + ".weight"), g(names["ln2"] + ".bias"), eps, h, noise, ph, st(3))


W2V = dict(q="attention.q_proj", k="attention.k_proj", v="attention.v_proj", o="attention.out_proj", ln1="layer_norm",
           f1="feed_forward.intermediate_dense", f2="feed_forward.output_dense", ln2="final_layer_norm")
XLMR = dict(q="attention.self.query", k="attention.self.key", v="attention.self.value", o="attention.output.dense",
            ln1="attention.output.LayerNorm", f1="intermediate.dense", f2="output.dense", ln2="output.LayerNorm")


def wav2vec2_forward(model, wave, noise=None):
    """Wav2Vec2Model with gradients: wave [B, T] raw clips of equal length -> last_hidden_state [B, S, H].  noise=None: eval
    semantics; a planned `Noise`: HF training-mode dropout sites, LayerDrop and SpecAugment."""
    c = model.config
    assert c.feat_extract_norm == "group" and not c.do_stable_layer_norm and not c.conv_bias, \
        "only the wav2vec2-base family (group-norm front end, post-LN encoder) is implemented"
    p = dict(model.named_parameters())
    eps = c.layer_norm_eps
    x = normalise_waves(wave.to(torch.float32))
    B = x.shape[0]
    Lv, Lp = padded_lengths(x.shape[1], c.conv_kernel, c.conv_stride)
    # conv0 (1 -> C channels, k 10, s 5): window rows of ALL clips as one strided view of the padded sample buffer; with one
    # input channel the view is materialised (12 samples per row: the kernel's taps + 2 that meet zero weights) so that the
    # product and its weight gradient take the head's aligned Linear kernels (split-K over the 100 k rows)
    k0, s0 = c.conv_kernel[0], c.conv_stride[0]
    kpad = (k0 + 3) // 4 * 4
    xp = torch.zeros(B * Lp[0] + SLACK * s0 + kpad, dtype=torch.float32, device=x.device)
    n0 = min(x.shape[1], Lp[0])
    xp[:B * Lp[0]].view(B, Lp[0])[:, :n0] = x[:, :n0]
    X0 = xp.as_strided((B * Lp[1], kpad), (s0, 1)).contiguous()
    w0 = p["feature_extractor.conv_layers.0.conv.weight"]                 # [C, 1, k]
    W0 = torch.nn.functional.pad(w0.reshape(w0.shape[0], k0), (0, kpad - k0))
    h = _Linear.apply(X0, W0.contiguous(), None, None)
    h = torch.cat([h, h.new_zeros(SLACK, h.shape[1])], dim=0)
    h = _ColNorm.apply(h, p["feature_extractor.conv_layers.0.layer_norm.weight"], p["feature_extractor.conv_layers.0.layer_norm.bias"],
                       1e-5, B, Lv[1], Lp[1])
    h = gelu(h)
    for i in range(1, len(c.conv_kernel)):
        k, s_ = c.conv_kernel[i], c.conv_stride[i]
        w = p[f"feature_extractor.conv_layers.{i}.conv.weight"]           # [C_out, C_in, k]
        W2 = w.permute(0, 2, 1).reshape(w.shape[0], -1).contiguous()      # taps in (tap, channel) order (a view op under autograd)
        h = gelu(_ConvPad.apply(h, W2, k, s_, B * Lp[i + 1]))                # layer i: Lp[i] rows per clip in, Lp[i + 1] out
    n = len(c.conv_kernel)
    h = h[:B * Lp[n]].view(B, Lp[n], -1)[:, :Lv[n], :]
    S = h.shape[1]
    e = layer_norm(h.reshape(B * S, -1), p["feature_projection.layer_norm.weight"], p["feature_projection.layer_norm.bias"], eps)
    z = linear(e, p["feature_projection.projection.weight"], p["feature_projection.projection.bias"])       # [B*S, H]
    H = z.shape[1]
    if noise is not None:
        if not noise.static:
            noise.plan(c.num_hidden_layers, B, S)
        else:
            assert noise._shape == (c.num_hidden_layers, B, S), "Noise.plan() / stage() were not called for this batch shape"
        z = _drop(z, noise, noise.p_featproj, SITE0 + 400)
        if noise.static:                              # device words staged by the graph's owner (all False: nothing masked)
            z = torch.where(noise.spec_dev[:, None], p["masked_spec_embed"][None, :].to(z.dtype), z)
        elif noise.spec_mask is not None:             # SpecAugment (hf :1293-1302): masked frames <- masked_spec_embed (a select, no arithmetic)
            mk = torch.from_numpy(noise.spec_mask.reshape(B * S)).to(z.device)
            z = torch.where(mk[:, None], p["masked_spec_embed"][None, :].to(z.dtype), z)
    # positional conv: weight_norm(dim=2) W = g * v / ||v||_(0,1); grouped conv, padding K/2, last frame dropped when K is even
    g0 = p["encoder.pos_conv_embed.conv.parametrizations.weight.original0"]
    v0 = p["encoder.pos_conv_embed.conv.parametrizations.weight.original1"]
    Wp = g0 * v0 / torch.sqrt((v0 * v0).sum(dim=(0, 1), keepdim=True))                                      # [H, Cg, K]
    K, G = c.num_conv_pos_embeddings, c.num_conv_pos_embedding_groups
    Cg, R = H // G, S + K - 1
    if L.lib.ser_posconv_direct_supported(S, H, G, K) and z.is_cuda:
        # GELU(conv) + z from the resident-slab kernel, then the LayerNorm alone
        h = layer_norm(_PosConvDirect.apply(z.contiguous(), Wp, p["encoder.pos_conv_embed.conv.bias"], B, S, K, G),
                       p["encoder.layer_norm.weight"], p["encoder.layer_norm.bias"], eps)
    else:
        pc = _PosConv.apply(z.contiguous(), Wp, p["encoder.pos_conv_embed.conv.bias"], B, S, K, G)
        h = layer_norm(gelu(pc), p["encoder.layer_norm.weight"], p["encoder.layer_norm.bias"], eps, residual=z)   # LN(z + GELU(conv))
    if noise is not None:
        h = _drop(h, noise, noise.p_hidden, SITE0 + 401)
    wp = weight_planes(model, lambda: _layer_entries(p, [f"encoder.layers.{i}." for i in range(c.num_hidden_layers)], W2V), B * S)
    for i in range(c.num_hidden_layers):
        if noise is not None and noise.static and noise.layerdrop > 0:
            # captured step: the layer always runs; a dropped layer's output is discarded by a select on a device word (its
            # backward then sees zeros, and the optimizer leaves its parameters untouched through the same word)
            hn = _transformer_layer(h, p, f"encoder.layers.{i}.", W2V, B, S, c.num_attention_heads, eps, None, noise, i, wp)
            h = torch.where(noise.skip_dev[i] != 0, h, hn)
            continue
        if noise is not None and i in noise.skip:     # LayerDrop (hf :700-703)
            continue
        h = _transformer_layer(h, p, f"encoder.layers.{i}.", W2V, B, S, c.num_attention_heads, eps, None, noise, i, wp)
    return h.reshape(B, S, H)


def xlmr_forward(model, ids, attn_mask, noise=None):
    """XLMRobertaModel with gradients: ids [B, S] int64, attn_mask [B, S] 1/0 -> last_hidden_state (noise: as above)."""
    c = model.config
    p = dict(model.named_parameters())
    B, S = ids.shape
    pad = c.pad_token_id
    m = (ids != pad).to(torch.int64)
    pos = torch.cumsum(m, dim=1) * m + pad                               # hf :142-155 (index arithmetic)
    e = _Embed.apply(ids.contiguous(), pos.contiguous(), p["embeddings.word_embeddings.weight"], p["embeddings.position_embeddings.weight"],
                     p["embeddings.token_type_embeddings.weight"], pad)
    h = layer_norm(e, p["embeddings.LayerNorm.weight"], p["embeddings.LayerNorm.bias"], c.layer_norm_eps)
    if noise is not None:
        if not noise.static:
            noise.plan(c.num_hidden_layers, B, S)
        h = _drop(h, noise, noise.p_hidden, SITE0 + 500 + 402)
    mask = attn_mask.to(torch.float32).contiguous()
    wp = weight_planes(model, lambda: _layer_entries(p, [f"encoder.layer.{i}." for i in range(c.num_hidden_layers)], XLMR), B * S)
    for i in range(c.num_hidden_layers):
        h = _transformer_layer(h, p, f"encoder.layer.{i}.", XLMR, B, S, c.num_attention_heads, c.layer_norm_eps, mask, noise, i, wp)
    return h.reshape(B, S, -1)
