"""GPU parity tests for the op-level C-ABI entry points (run with -m gpu on an MI355X).
Every comparison is against a plain fp32/fp64 torch-CPU restatement of the same op."""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    import ser_amd  # noqa: F401
    import ser_amd._lib as lib
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return lib


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def test_split_bf16_reconstructs(L):
    x = _rand(1000, 37, seed=1).cuda()
    hi, lo = L.split_bf16(x)
    rec = hi.float() + lo.float()
    assert float((rec - x).abs().max() / x.abs().max()) < 2 ** -15
    assert torch.equal(hi, x.to(torch.bfloat16))      # round-to-nearest-even, same as torch


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (199, 768, 768), (3184, 2304, 768), (37, 48, 512), (1, 64, 64),
                                   (300, 3072, 768), (257, 130, 192)])
@pytest.mark.parametrize("x3", [True, False])
def test_gemm_bf16_nt(L, M, N, K, x3):
    a, w = _rand(M, K, seed=2), _rand(N, K, seed=3) / np.sqrt(K)
    bias, res = _rand(N, seed=4), _rand(M, N, seed=5)
    ah, al = L.split_bf16(a.cuda(), x3)
    wh, wl = L.split_bf16(w.cuda(), x3)
    c, ch, cl = L.gemm_bf16_nt(ah, al, wh, wl, bias.cuda(), L.ACT_GELU, res.cuda(), out_f32=True, out_split=True)
    torch.cuda.synchronize()
    if x3:
        ref = torch.nn.functional.gelu(a.double() @ w.double().t() + bias.double()) + res.double()
        tol = 3e-5
    else:   # fast mode: compare with the bf16-rounded operands multiplied exactly
        ref = torch.nn.functional.gelu(ah.cpu().double() @ wh.cpu().double().t() + bias.double()) + res.double()
        tol = 2e-5
    err = (c.cpu().double() - ref).abs().max().item()
    assert err < tol, f"max abs err {err}"
    rec = ch.float() + (cl.float() if cl is not None else 0)
    assert (rec.cpu() - c.cpu()).abs().max().item() < (1e-4 if x3 else 4e-2)


@pytest.mark.parametrize("stages", [3, 4])
@pytest.mark.parametrize("M,N,K", [(199, 768, 768), (3184, 2304, 768), (300, 768, 3072), (257, 130, 192), (37, 200, 64),
                                   (512, 768, 128)])
def test_gemm_bf16_multi_stage_pipeline_is_bit_identical(L, M, N, K, stages):
    """3- and 4-buffer LDS-DMA pipelines (counted vmcnt, raw barrier) against the double-buffered kernel: the
    arithmetic is the same, only the staging schedule differs, so the results must be identical (a stale or early
    read of a staged tile would show up here); K = 64 / 128 exercise the short-loop prologue and tail."""
    a, w = _rand(M, K, seed=12), _rand(N, K, seed=13) / np.sqrt(K)
    bias = _rand(N, seed=14)
    ah, _ = L.split_bf16(a.cuda(), False)
    wh, _ = L.split_bf16(w.cuda(), False)
    outs = []
    try:
        for st in (2, stages):
            L.lib.ser_debug_set_gemm_stages(st, st, st, st)
            for _ in range(3):
                c, _, _ = L.gemm_bf16_nt(ah, None, wh, None, bias.cuda(), L.ACT_NONE, None, out_f32=True, out_split=False)
            torch.cuda.synchronize()
            outs.append(c.clone())
    finally:
        L.lib.ser_debug_set_gemm_stages(2, 3, 2, 2)       # library default
    assert torch.equal(outs[0], outs[1]), f"differs by {(outs[0] - outs[1]).abs().max().item()}"
    ref = ah.cpu().double() @ wh.cpu().double().t() + bias.double()
    assert (outs[1].cpu().double() - ref).abs().max().item() < 2e-5


@pytest.mark.parametrize("bm", [64, 96, 160, 192])
@pytest.mark.parametrize("M,N,K", [(3696, 2304, 768), (300, 768, 3072), (257, 130, 192), (97, 200, 64), (1000, 512, 1536)])
def test_gemm_bf16_tile_heights_are_bit_identical(L, M, N, K, bm):
    """Every tile height (chosen per shape by the tuning pass / cost model) computes each output element with the same
    k order, so the results must equal those of the 128-row tiles exactly, including ragged last tiles and the
    two-pass epilogue with bias + GELU + residual + bf16 planes."""
    a, w = _rand(M, K, seed=22), _rand(N, K, seed=23) / np.sqrt(K)
    bias, res = _rand(N, seed=24), _rand(M, N, seed=25)
    ah, _ = L.split_bf16(a.cuda(), False)
    wh, _ = L.split_bf16(w.cuda(), False)
    outs = []
    try:
        for h in (128, bm):
            L.lib.ser_debug_set_gemm_bm(h)
            c, ch, _ = L.gemm_bf16_nt(ah, None, wh, None, bias.cuda(), L.ACT_GELU, res.cuda(), out_f32=True, out_split=True)
            torch.cuda.synchronize()
            outs.append((c.clone(), ch.clone()))
    finally:
        L.lib.ser_debug_set_gemm_bm(0)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    ref = torch.nn.functional.gelu(ah.cpu().double() @ wh.cpu().double().t() + bias.double()) + res.double()
    assert (outs[1][0].cpu().double() - ref).abs().max().item() < 2e-5


@pytest.mark.parametrize("shapes", [((24, 384, 128), (140, 384, 128)), ((512, 768, 768), (3184, 768, 768)), ((65, 200, 128), (140, 200, 128)),
                                    ((100, 384, 64), (140, 384, 64)), ((512, 2304, 768), (3184, 2304, 768))])
def test_gemm_bf16_paired_launch_equals_single_launches(L, shapes):
    """Two independent GEMMs in one launch (the layer-l products of the two encoders) against two single launches:
    identical results, for tile shapes picked by the larger problem (incl. 64x64, where an earlier two-call-site
    version of the kernel produced NaNs in the first problem)."""
    import ctypes as C
    f = L.lib.ser_debug_gemm_pair
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p] * 2 + [C.c_void_p]
    ops = []
    for k, (M, N, K) in enumerate(shapes):
        a, w = _rand(M, K, seed=30 + k), _rand(N, K, seed=40 + k) / np.sqrt(K)
        ah, _ = L.split_bf16(a.cuda(), False)
        wh, _ = L.split_bf16(w.cuda(), False)
        ops.append((ah, wh, torch.full((M, N), float("nan"), device="cuda"), M, N, K))
    (a0, w0, c0, M0, N0, K0), (a1, w1, c1, M1, N1, K1) = ops
    L.check(f(a0.data_ptr(), w0.data_ptr(), M0, N0, K0, c0.data_ptr(), a1.data_ptr(), w1.data_ptr(), M1, N1, K1, c1.data_ptr(),
              L.stream_ptr()), "ser_debug_gemm_pair")
    torch.cuda.synchronize()
    for ah, wh, c, M, N, K in ops:
        single, _, _ = L.gemm_bf16_nt(ah, None, wh, None, None, L.ACT_NONE, None, out_f32=True, out_split=False)
        assert torch.equal(c, single), f"{(M, N, K)}: differs by {(c - single).abs().max().item()}"


def test_gemm_strided_rows_is_conv(L):
    """Conv1d(k=3, stride=2) over channels-last activations as an NT GEMM with lda = stride*C."""
    C_, Lin, Cout, k, s = 64, 41, 64, 3, 2
    x = _rand(Lin, C_, seed=6)
    w = _rand(Cout, C_, k, seed=7) / np.sqrt(C_ * k)
    Lout = (Lin - k) // s + 1
    xh, xl = L.split_bf16(x.cuda())
    wp = w.permute(0, 2, 1).reshape(Cout, k * C_).contiguous()
    wh, wl = L.split_bf16(wp.cuda())
    c = torch.empty(Lout, Cout, device="cuda")
    L.check(L.lib.ser_gemm_bf16_nt(L.ptr(xh), L.ptr(xl), s * C_, L.ptr(wh), L.ptr(wl), k * C_, Lout, Cout, k * C_, None,
                                   L.ACT_NONE, None, 0, L.ptr(c), None, None, Cout, L.stream_ptr()))
    ref = torch.nn.functional.conv1d(x.t()[None].double(), w.double(), stride=s)[0].t()
    assert (c.cpu().double() - ref).abs().max().item() < 2e-5


@pytest.mark.parametrize("D", [64, 128, 512, 768, 1024])
def test_layernorm(L, D):
    x, x2 = _rand(77, D, seed=8, scale=3.0), _rand(77, D, seed=9)
    g, b = _rand(D, seed=10), _rand(D, seed=11)
    y, yh, yl = L.layernorm(x.cuda(), g.cuda(), b.cuda(), 1e-5, x2=x2.cuda(), out_split=True)
    ref = torch.nn.functional.layer_norm((x + x2).double(), (D,), g.double(), b.double(), 1e-5)
    assert (y.cpu().double() - ref).abs().max().item() < 5e-6
    assert ((yh.float() + yl.float()).cpu() - y.cpu()).abs().max().item() < 1e-4


@pytest.mark.parametrize("B,S,heads,masked", [(2, 199, 12, False), (3, 32, 12, True), (1, 7, 2, True), (2, 130, 2, False)])
@pytest.mark.parametrize("x3", [True, False])
def test_self_attention(L, B, S, heads, masked, x3):
    H = heads * 64
    qkv = _rand(B * S, 3 * H, seed=12)
    mask = torch.ones(B, S)
    if masked:
        for b in range(B):
            mask[b, max(1, S - 1 - 2 * b):] = 0
    qh, ql = L.split_bf16(qkv.cuda(), x3)
    ch, cl = L.self_attention(qh, ql, mask.cuda() if masked else None, B, S, heads)
    got = ch.float().cpu() + (cl.float().cpu() if cl is not None else 0)
    src = qkv.double() if x3 else qh.float().cpu().double()
    q, k, v = (src[:, i * H:(i + 1) * H].reshape(B, S, heads, 64).transpose(1, 2) for i in range(3))
    s = q @ k.transpose(2, 3) / 8.0
    s = s.masked_fill(mask[:, None, None, :] == 0, float("-inf"))
    ref = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B * S, H)
    err = (got.double() - ref).abs().max().item()
    assert err < (5e-5 if x3 else 2e-2), f"max abs err {err}"


def test_conv0_groupnorm_gelu_entry_against_torch(L):
    """K2 on its own through the C ABI: clip normalisation + Conv1d(1->C, k10, s5) + GroupNorm(C groups) + erf-GELU,
    with a DC offset and a ragged last chunk, against torch in float64."""
    import ctypes as C
    B, T, C0, KW, ST = 3, 3001, 64, 10, 5
    g = torch.Generator().manual_seed(5)
    wave = 0.1 * torch.randn(B, T, generator=g) + 0.25
    w = torch.randn(C0, KW, generator=g) * 0.3
    gn_g, gn_b = 1 + 0.1 * torch.randn(C0, generator=g), 0.1 * torch.randn(C0, generator=g)
    L0 = (T - KW) // ST + 1
    lib = L.lib
    lib.ser_conv0_workspace_bytes.restype = C.c_size_t
    lib.ser_conv0_workspace_bytes.argtypes = [C.c_int] * 3
    lib.ser_conv0_gn_gelu.restype = C.c_int
    lib.ser_conv0_gn_gelu.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    nb = lib.ser_conv0_workspace_bytes(B, L0, C0)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    hi = torch.empty(B, L0, C0, dtype=torch.bfloat16, device="cuda")
    lo = torch.empty_like(hi)
    d = [t_.cuda() for t_ in (wave, w, gn_g, gn_b)]
    L.check(lib.ser_conv0_gn_gelu(d[0].data_ptr(), B, T, d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), C0, KW, ST,
                                  hi.data_ptr(), lo.data_ptr(), ws.data_ptr(), nb, L.stream_ptr()), "ser_conv0_gn_gelu")
    torch.cuda.synchronize()
    x = wave.double()
    x = (x - x.mean(1, keepdim=True)) / torch.sqrt(x.var(1, unbiased=False, keepdim=True) + 1e-7)
    y = torch.nn.functional.conv1d(x[:, None], w.double()[:, None], stride=ST)
    y = torch.nn.functional.group_norm(y, C0, gn_g.double(), gn_b.double(), eps=1e-5)
    ref = torch.nn.functional.gelu(y).transpose(1, 2)
    got = hi.float().cpu().double() + lo.float().cpu().double()
    assert (got - ref).abs().max().item() < 3e-5


def test_xlmr_embed_entry_against_torch(L):
    """K7 on its own: position ids from the non-pad cumsum, word + type + position gather, LayerNorm."""
    import ctypes as C
    B, S, D, V, P, pad = 3, 9, 128, 50, 40, 1
    g = torch.Generator().manual_seed(6)
    ids = torch.randint(2, V, (B, S), generator=g)
    ids[0, -3:] = pad
    ids[2, -1] = pad
    we, pe, te = torch.randn(V, D, generator=g), torch.randn(P, D, generator=g), torch.randn(1, D, generator=g)
    gam, bet = 1 + 0.1 * torch.randn(D, generator=g), 0.1 * torch.randn(D, generator=g)
    lib = L.lib
    lib.ser_xlmr_embed.restype = C.c_int
    lib.ser_xlmr_embed.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_float, C.c_int, C.c_int, C.c_int, C.c_int] + \
                                  [C.c_void_p] * 5
    d = [t_.cuda() for t_ in (ids, we, pe, te, gam, bet)]
    pos = torch.empty(B * S, dtype=torch.int32, device="cuda")
    y = torch.empty(B * S, D, device="cuda")
    L.check(lib.ser_xlmr_embed(d[0].data_ptr(), B, S, d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(),
                               d[5].data_ptr(), 1e-5, D, V, P, pad, pos.data_ptr(), y.data_ptr(), None, None, L.stream_ptr()),
            "ser_xlmr_embed")
    torch.cuda.synchronize()
    m = (ids != pad).long()
    pid = torch.cumsum(m, 1) * m + pad
    ref = torch.nn.functional.layer_norm((we[ids] + pe[pid] + te[0]).double(), (D,), gam.double(), bet.double(), 1e-5)
    assert torch.equal(pos.cpu().view(B, S).long(), pid)
    assert (y.cpu().double().view(B, S, D) - ref).abs().max().item() < 1e-5
