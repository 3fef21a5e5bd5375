"""GPU parity of the interleaved split-plane layout (csrc/ser_common.h: [hi 0..31 | lo 0..31 | hi 32..63 | ...]) that the
three-product encoder kernels read and write: every producer / consumer against its planar form (bit for bit: the
layout changes where bytes live, not the arithmetic) and the GEMM against a float64 product."""
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


def test_split_interleaved_equals_planar(L):
    x = _rand(77, 96, seed=1).cuda()
    hi, lo = L.split_bf16(x)
    il = L.split_bf16_il(x)
    assert il.shape == (77, 192)
    h2, l2 = L.il_planes(il)
    assert torch.equal(h2, hi) and torch.equal(l2, lo)
    # layout spelled out: 32 hi values, then their 32 lo values
    assert torch.equal(il[:, :32], hi[:, :32]) and torch.equal(il[:, 32:64], lo[:, :32]) and torch.equal(il[:, 64:96], hi[:, 32:64])


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (199, 768, 768), (3184, 2304, 768), (37, 64, 512), (1, 64, 32),
                                   (300, 3072, 768), (257, 160, 96), (3696, 768, 3072), (512, 128, 32)])
def test_gemm_three_products_interleaved(L, M, N, K):
    a, w = _rand(M, K, seed=2), _rand(N, K, seed=3) / np.sqrt(K)
    bias, res = _rand(N, seed=4), _rand(M, N, seed=5)
    a_il, w_il = L.split_bf16_il(a.cuda()), L.split_bf16_il(w.cuda())
    c, c_il = L.gemm_bf16x3_il(a_il, w_il, bias.cuda(), L.ACT_GELU, res.cuda(), out_f32=True, out_split=True)
    torch.cuda.synchronize()
    ref = torch.nn.functional.gelu(a.double() @ w.double().t() + bias.double()) + res.double()
    err = (c.cpu().double() - ref).abs().max().item()
    assert err < 5e-5, f"max abs err {err}"
    ch, cl = L.il_planes(c_il)
    assert torch.equal(ch, c.to(torch.bfloat16)), "hi plane of the output is the rounded result"
    assert ((ch.float() + cl.float()).cpu() - c.cpu()).abs().max().item() < 1e-4
    if K % 64 == 0:      # the planar kernel multiplies the same products in the same order: identical bits
        ah, al = L.split_bf16(a.cuda())
        wh, wl = L.split_bf16(w.cuda())
        c2, _, _ = L.gemm_bf16_nt(ah, al, wh, wl, bias.cuda(), L.ACT_GELU, res.cuda(), out_f32=True, out_split=False)
        assert torch.equal(c, c2), f"interleaved vs planar differ by {(c - c2).abs().max().item()}"


@pytest.mark.parametrize("bm", [64, 96, 128, 160, 192])
def test_gemm_interleaved_tile_heights_agree(L, bm):
    M, N, K = 3696, 2304, 768
    a, w = _rand(M, K, seed=6), _rand(N, K, seed=7) / np.sqrt(K)
    a_il, w_il = L.split_bf16_il(a.cuda()), L.split_bf16_il(w.cuda())
    try:
        L.lib.ser_debug_set_gemm_bm(0)
        c0, _ = L.gemm_bf16x3_il(a_il, w_il)
        L.lib.ser_debug_set_gemm_bm(bm)
        c1, _ = L.gemm_bf16x3_il(a_il, w_il)
        torch.cuda.synchronize()
    finally:
        L.lib.ser_debug_set_gemm_bm(0)
    assert torch.equal(c0, c1)
    assert (c1.cpu().double() - a.double() @ w.double().t()).abs().max().item() < 3e-5


WIDE = {"128x256": 1128, "192x256": 1192, "256x256": 1256, "256x128": 2256, "single-buffer 64": 3064, "single-buffer 96": 3096,
        "single-buffer 128": 3128, "128x256 3 buffers": 5128, "256x128 3 buffers": 6256,
        "64x64": 7064, "96x64": 7096, "128x64": 7128, "192x64": 7192}


def _gemm_cfg(L, a_il, w_il, M, N, K, cfg, ksplit=1):
    import ctypes as C
    fn = L.lib.ser_debug_gemm_il_cfg
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    out = torch.full((max(1, ksplit), M, N), float("nan"), dtype=torch.float32, device="cuda")
    L.check(fn(a_il.data_ptr(), w_il.data_ptr(), M, N, K, cfg, ksplit, out.data_ptr(), L.stream_ptr()), "ser_debug_gemm_il_cfg")
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("name", sorted(WIDE))
@pytest.mark.parametrize("M,N,K", [(3696, 2304, 768), (700, 768, 3072), (257, 130, 96), (6399, 512, 1536)])
def test_gemm_512_thread_tiles_equal_the_classic_tiles(L, name, M, N, K):
    """2 x 4 / 4 x 2 wave grids over 256-wide / 256-tall tiles: the products and their order per accumulator are those
    of the 2 x 2 kernel, so the results are identical; ragged edges included."""
    a, w = _rand(M, K, seed=21), _rand(N, K, seed=22) / np.sqrt(K)
    a_il, w_il = L.split_bf16_il(a.cuda()), L.split_bf16_il(w.cuda())
    ref = _gemm_cfg(L, a_il, w_il, M, N, K, 128)[0]
    got = _gemm_cfg(L, a_il, w_il, M, N, K, WIDE[name])[0]
    assert torch.equal(ref, got), f"{name}: differs by {(ref - got).abs().max().item()}"
    assert (got.cpu().double() - a.double() @ w.double().t()).abs().max().item() < 5e-5


@pytest.mark.parametrize("cfg", [64, 192, 1192, 2256])
@pytest.mark.parametrize("ksplit", [2, 3, 4])
def test_gemm_split_k_slabs_sum_to_the_product(L, cfg, ksplit):
    M, N, K = 3696, 768, 3072
    a, w = _rand(M, K, seed=23), _rand(N, K, seed=24) / np.sqrt(K)
    a_il, w_il = L.split_bf16_il(a.cuda()), L.split_bf16_il(w.cuda())
    slabs = _gemm_cfg(L, a_il, w_il, M, N, K, cfg, ksplit)
    assert torch.isfinite(slabs).all(), "a slab was not written completely"
    ref = a.double() @ w.double().t()
    assert (slabs.double().sum(0).cpu() - ref).abs().max().item() < 5e-5
    # every slab is the product over its own k range
    per = -(-(K // 32) // ksplit) * 32
    for s_ in range(ksplit):
        part = a[:, s_ * per:(s_ + 1) * per].double() @ w[:, s_ * per:(s_ + 1) * per].double().t()
        assert (slabs[s_].double().cpu() - part).abs().max().item() < 5e-5


def test_layernorm_interleaved_output(L):
    x, g, b = _rand(203, 768, seed=8).cuda(), _rand(768, seed=9).cuda(), _rand(768, seed=10).cuda()
    y, yh, yl = L.layernorm(x, g, b, 1e-5, out_split=True)
    y2 = torch.empty_like(x)
    il = torch.empty(203, 1536, dtype=torch.bfloat16, device="cuda")
    hi, lo = L.il_ptrs(il)
    L.check(L.lib.ser_layernorm(x.data_ptr(), None, g.data_ptr(), b.data_ptr(), 1e-5, 203, 768, y2.data_ptr(), hi, lo, L.stream_ptr()))
    h2, l2 = L.il_planes(il)
    assert torch.equal(y, y2) and torch.equal(h2, yh) and torch.equal(l2, yl)


def _attention_reference(qkv, mask, B, S, heads):
    H = heads * 64
    q, k, v = (qkv.double()[:, i * H:(i + 1) * H].reshape(B, S, heads, 64).transpose(1, 2) for i in range(3))
    s = q @ k.transpose(2, 3) / 8.0
    if mask is not None:
        s = s.masked_fill(mask[:, None, None, :] == 0, float("-inf"))
    return (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B * S, H)


@pytest.mark.parametrize("S,masked", [(199, False), (32, True), (70, True), (149, True), (224, False), (17, True), (225, False), (499, True)])
def test_self_attention_interleaved(L, S, masked):
    """Interleaved planes in and out.  S <= 224 takes the resident-K/V kernel (swapped QK^T, transposed V reads), longer
    sequences the chunked online-softmax kernel; both against float64, and the resident kernel against the chunked one."""
    B, heads = 3, 4
    H = heads * 64
    qkv = _rand(B * S, 3 * H, seed=11)
    mask = None
    if masked:
        mask = torch.ones(B, S)
        mask[1, S - 5:] = 0
        mask[2, 3:9] = 0
    ref = _attention_reference(qkv, mask, B, S, heads)
    q_il = L.split_bf16_il(qkv.cuda())
    mk = mask.cuda() if masked else None
    outs = []
    for generic in (0, 1):
        L.lib.ser_debug_set_attention_generic(generic)
        try:
            c_il = torch.empty(B * S, 2 * H, dtype=torch.bfloat16, device="cuda")
            L.check(L.lib.ser_self_attention(*L.il_ptrs(q_il), L.ptr(mk), B, S, heads, *L.il_ptrs(c_il), L.stream_ptr()))
            torch.cuda.synchronize()
        finally:
            L.lib.ser_debug_set_attention_generic(0)
        h2, l2 = L.il_planes(c_il)
        got = h2.float().cpu().double() + l2.float().cpu().double()
        err = (got - ref).abs().max().item()
        assert err < 5e-5, f"generic={generic}: max abs err {err}"
        outs.append(got)
    assert (outs[0] - outs[1]).abs().max().item() < 2e-5


@pytest.mark.parametrize("S,masked", [(199, False), (32, True), (100, True)])
def test_self_attention_resident_one_plane(L, S, masked):
    """The resident-K/V kernel in the one-product mode (hi plane only) against the chunked kernel and float64."""
    B, heads = 2, 3
    H = heads * 64
    qkv = _rand(B * S, 3 * H, seed=13)
    mask = None
    if masked:
        mask = torch.ones(B, S)
        mask[1, S - 7:] = 0
    qh, _ = L.split_bf16(qkv.cuda(), False)
    ref = _attention_reference(qh.float().cpu(), mask, B, S, heads)
    outs = []
    for generic in (0, 1):
        L.lib.ser_debug_set_attention_generic(generic)
        try:
            ch, _ = L.self_attention(qh, None, mask.cuda() if masked else None, B, S, heads)
            torch.cuda.synchronize()
        finally:
            L.lib.ser_debug_set_attention_generic(0)
        assert (ch.float().cpu().double() - ref).abs().max().item() < 2e-2
        outs.append(ch.float().cpu())
    assert (outs[0] - outs[1]).abs().max().item() < 2e-2


@pytest.mark.parametrize("S,masked", [(199, False), (199, True), (32, True), (149, True), (224, False), (17, True), (1, False)])
@pytest.mark.parametrize("mode", ["interleaved", "one-plane", "planar-out"])
def test_self_attention_half_footprint_kernel_is_bit_identical(L, S, masked, mode):
    """The default resident kernel lets K and V take turns in one LDS region (scores and softmax of all of a wave's query
    blocks first, then V replaces K): same products in the same order as the form that keeps both resident, so the planes
    it writes are identical bit for bit, in every plane layout."""
    L.lib.ser_debug_set_attention_small_variant.argtypes = [L.i32]
    B, heads = 3, 4
    H = heads * 64
    qkv = _rand(B * S, 3 * H, seed=31 + S)
    mk = None
    if masked:
        mask = torch.ones(B, S)
        mask[1, max(0, S - 5):] = 0
        if S > 9:
            mask[2, 3:9] = 0
        mk = mask.cuda()
    outs = []
    for variant in (1, 2):
        L.lib.ser_debug_set_attention_small_variant(variant)
        try:
            if mode == "interleaved":
                q_il = L.split_bf16_il(qkv.cuda())
                c = torch.full((B * S, 2 * H), -1.0, dtype=torch.bfloat16, device="cuda")
                L.check(L.lib.ser_self_attention(*L.il_ptrs(q_il), L.ptr(mk), B, S, heads, *L.il_ptrs(c), L.stream_ptr()))
                res = (c,)
            elif mode == "one-plane":
                qh, _ = L.split_bf16(qkv.cuda(), False)
                res = (L.self_attention(qh, None, mk, B, S, heads)[0],)
            else:                      # interleaved input, planar output planes
                q_il = L.split_bf16_il(qkv.cuda())
                ch = torch.full((B * S, H), -1.0, dtype=torch.bfloat16, device="cuda")
                cl = torch.full((B * S, H), -1.0, dtype=torch.bfloat16, device="cuda")
                L.check(L.lib.ser_self_attention(*L.il_ptrs(q_il), L.ptr(mk), B, S, heads, ch.data_ptr(), cl.data_ptr(), L.stream_ptr()))
                res = (ch, cl)
            torch.cuda.synchronize()
        finally:
            L.lib.ser_debug_set_attention_small_variant(2)
        outs.append([r.clone() for r in res])
    for a, b_ in zip(*outs):
        assert torch.equal(a.view(torch.int16), b_.view(torch.int16)), f"differs by {(a.float() - b_.float()).abs().max().item()}"
