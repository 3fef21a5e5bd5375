"""GPU parity of the frozen-encoder forward (through the C ABI) against the golden vectors captured
from the reference and against the CPU oracle on fresh seeded inputs."""
import types

import numpy as np
import pytest
import torch

from oracle import ser_oracle as O
from tests.helpers import cfg_of, load_npz, split_fixture, t

pytestmark = pytest.mark.gpu


def _w2v_hf_cfg(c):
    return types.SimpleNamespace(hidden_size=c["hidden"], num_hidden_layers=c["layers"], num_attention_heads=c["heads"],
                                 intermediate_size=c["ffn"], conv_dim=list(c["conv_dim"]), conv_kernel=list(c["conv_kernel"]),
                                 conv_stride=list(c["conv_stride"]), num_conv_pos_embeddings=c["pos_kernel"],
                                 num_conv_pos_embedding_groups=c["pos_groups"], layer_norm_eps=c["eps"],
                                 feat_extract_norm="group", do_stable_layer_norm=False, conv_bias=False)


def _xlmr_hf_cfg(c):
    return types.SimpleNamespace(hidden_size=c["hidden"], num_hidden_layers=c["layers"], num_attention_heads=c["heads"],
                                 intermediate_size=c["ffn"], vocab_size=c["vocab"], max_position_embeddings=c["max_pos"],
                                 pad_token_id=c["pad_id"], layer_norm_eps=c["eps"])


@pytest.fixture(scope="module")
def eng():
    import ser_amd  # noqa: F401
    from ser_amd import _engines, _lib
    assert torch.cuda.is_available()
    return _engines, _lib


@pytest.mark.parametrize("prec,tol", [("x3", 2e-4), ("bf16", 1.5e-1)])
def test_wav2vec2_forward_golden(eng, prec, tol):
    E, L = eng
    sd, _, r = split_fixture(load_npz("audio_encoder.npz"))
    cfg = cfg_of(r)
    enc_sd = O.sub(sd, "encoder.")
    e = E.Wav2Vec2Engine(_w2v_hf_cfg(cfg), enc_sd, "cuda", L.PREC_BF16X3 if prec == "x3" else L.PREC_BF16)
    for i, name in enumerate(("wave0", "wave1")):
        w = t(r[name])
        got = e.forward(w[None].cuda())[0].cpu()
        want = O.wav2vec2_forward(enc_sd, O.normalise_waveform(w)[None], cfg)[0]
        err = (got - want).abs().max().item()
        assert err < tol, f"{name}: max abs err vs oracle {err}"
        # golden a_seq = encoder output + adapter; check the adapter-free part through the oracle's adapter
        full = O.adapter(got, O.sub(sd, "adapter."))
        gerr = (full - t(r["a_seq"])[i, : full.shape[0]]).abs().max().item()
        assert gerr < tol * 2, f"{name}: max abs err vs golden {gerr}"


def test_wav2vec2_batched_equals_single(eng):
    E, L = eng
    sd, _, r = split_fixture(load_npz("audio_encoder.npz"))
    cfg = cfg_of(r)
    e = E.Wav2Vec2Engine(_w2v_hf_cfg(cfg), O.sub(sd, "encoder."), "cuda", L.PREC_BF16X3)
    g = torch.Generator().manual_seed(5)
    waves = 0.1 * torch.randn(5, 2400, generator=g)
    batched = e.forward(waves.cuda()).cpu()
    for i in range(5):
        single = e.forward(waves[i:i + 1].cuda())[0].cpu()
        assert torch.equal(single, batched[i]), "per-clip results must not depend on batching"
    want = O.wav2vec2_forward(O.sub(sd, "encoder."), torch.stack([O.normalise_waveform(w) for w in waves]), cfg)
    assert (batched - want).abs().max().item() < 2e-4


@pytest.mark.parametrize("prec,tol", [("x3", 2e-4), ("bf16", 1.5e-1)])
def test_xlmr_forward_golden(eng, prec, tol):
    E, L = eng
    sd, _, r = split_fixture(load_npz("text_encoder.npz"))
    cfg = cfg_of(r)
    enc_sd = O.sub(sd, "encoder.")
    e = E.XlmrEngine(_xlmr_hf_cfg(cfg), enc_sd, "cuda", L.PREC_BF16X3 if prec == "x3" else L.PREC_BF16)
    ids, am = t(r["input_ids"]), t(r["attention_mask"])
    got = e.forward(ids.cuda(), am.cuda()).cpu()
    want = O.xlmr_forward(enc_sd, ids, am, cfg)
    valid = am.bool()
    assert (got - want)[valid].abs().max().item() < tol
    full = O.adapter(got, O.sub(sd, "adapter."))
    assert (full - t(r["t_seq"]))[valid].abs().max().item() < tol * 2


def test_wav2vec2_generic_conv0_geometry(eng):
    """A first conv layer that is not (kernel 10, stride 5) takes the generic conv0 kernels; DC offset and a
    different clip length exercise the clip normalisation and the ragged last chunk."""
    from transformers import Wav2Vec2Config, Wav2Vec2Model
    E, L = eng
    torch.manual_seed(3)
    for kernel, stride in (([8, 3, 2], [4, 2, 2]), ([10, 3, 2], [5, 2, 2])):
        wc = Wav2Vec2Config(hidden_size=128, num_hidden_layers=1, num_attention_heads=2, intermediate_size=128,
                            conv_dim=[64] * 3, conv_kernel=kernel, conv_stride=stride, num_conv_pos_embeddings=16,
                            num_conv_pos_embedding_groups=4)
        sd = {k: v.detach() for k, v in Wav2Vec2Model(wc).state_dict().items()}
        cfg = O.wav2vec2_config(hidden=128, layers=1, heads=2, ffn=128, conv_dim=wc.conv_dim, conv_kernel=kernel,
                                conv_stride=stride, pos_kernel=16, pos_groups=4, eps=wc.layer_norm_eps)
        e = E.Wav2Vec2Engine(wc, sd, "cuda", L.PREC_BF16X3)
        waves = 0.1 * torch.randn(3, 3001) + 0.3
        got = e.forward(waves.cuda()).cpu()
        want = O.wav2vec2_forward(sd, torch.stack([O.normalise_waveform(w) for w in waves]), cfg)
        assert (got - want).abs().max().item() < 2e-4, (kernel, stride)


@pytest.mark.parametrize("prec", ["x3", "bf16"])
def test_paired_encoders_equal_separate_calls(eng, prec):
    """ser_encoders_forward (layers of both models in lock-step, grouped launches) against the two separate
    forwards: same per-element arithmetic, so the outputs must be identical."""
    E, L = eng
    sda, _, ra = split_fixture(load_npz("audio_encoder.npz"))
    sdt, _, rt = split_fixture(load_npz("text_encoder.npz"))
    ca, ct = cfg_of(ra), cfg_of(rt)
    p = L.PREC_BF16X3 if prec == "x3" else L.PREC_BF16
    ea = E.Wav2Vec2Engine(_w2v_hf_cfg(ca), O.sub(sda, "encoder."), "cuda", p)
    et = E.XlmrEngine(_xlmr_hf_cfg(ct), O.sub(sdt, "encoder."), "cuda", p)
    assert ca["layers"] == ct["layers"], "fixtures are expected to have equal depth (paired path)"
    g = torch.Generator().manual_seed(11)
    for B, T, St in ((3, 2400, 7), (2, 4000, 70), (5, 1700, 3)):
        wave = (0.1 * torch.randn(B, T, generator=g)).cuda()
        ids = torch.randint(4, ct["vocab"], (B, St), generator=g)
        mask = torch.ones(B, St)
        if St > 4:
            ids[0, -2:] = ct["pad_id"]
            mask[0, -2:] = 0
        ids, mask = ids.cuda(), mask.cuda()
        a1, t1 = ea.forward(wave), et.forward(ids, mask)
        a2, t2 = E.forward_pair(ea, et, wave, ids, mask)
        torch.cuda.synchronize()
        assert torch.equal(a1, a2), f"audio differs by {(a1 - a2).abs().max().item()}"
        assert torch.equal(t1, t2), f"text differs by {(t1 - t2).abs().max().item()}"


@pytest.mark.parametrize("B,S,H,G", [(2, 199, 768, 16), (3, 149, 768, 16), (1, 17, 768, 16), (2, 224, 768, 16), (2, 225, 768, 16),
                                     (1, 352, 768, 16), (2, 124, 1024, 16), (1, 224, 1024, 16)])
def test_positional_conv_resident_slab_kernel(eng, B, S, H, G):
    """The one-kernel positional conv (csrc/posconv.hip: the (clip, group) slab resident in LDS, weights streamed) against the
    sliding-window GEMM it replaces and against a float64 evaluation of the same hi / lo split operands.  Same products
    in the same order per accumulator: most outputs are identical, the rest one ulp apart, and both paths sit equally close
    to the float64 value.  48- and 64-channel groups (Base / Large widths), frame counts on both sides of the 224-frame
    variant and at its limits, odd frame counts."""
    import ctypes as C
    E, L = eng
    fn = L.lib.ser_debug_posconv
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p] * 4 + [C.c_int] * 6 + [C.c_void_p, C.c_void_p]
    K, Cg = 128, H // G
    g = torch.Generator().manual_seed(S + H)
    z = torch.randn(B, S, H, generator=g).cuda()
    wl = torch.randn(G * Cg, K, Cg, generator=g) * 0.02
    w_il = L.split_bf16_il(torch.nn.functional.pad(wl, (0, 64 - Cg)).reshape(G * Cg, K * 64).cuda())
    bias = torch.randn(H, generator=g).cuda()
    slab = torch.zeros(B * G * (S + K - 1) * 128 + K * 128 + 64, dtype=torch.bfloat16, device="cuda")
    outs = []
    for direct in (0, 1):
        out = torch.full((B, S, H), float("nan"), device="cuda")
        L.check(fn(z.data_ptr(), w_il.data_ptr(), bias.data_ptr(), out.data_ptr(), B, S, H, G, K, direct, slab.data_ptr(), L.stream_ptr()),
                "ser_debug_posconv")
        torch.cuda.synchronize()
        outs.append(out)
    gemm, direct = outs
    assert torch.isfinite(direct).all()
    ulp = torch.finfo(torch.float32).eps * gemm.abs().clamp_min(1.0)
    assert ((gemm - direct).abs() <= 2 * ulp).all(), f"differs from the GEMM path by {(gemm - direct).abs().max().item()}"
    assert (gemm == direct).float().mean().item() > 0.6
    # float64 value of the same split operands
    hi, lo = L.il_planes(w_il)
    W = (hi.double() + lo.double()).reshape(G, Cg, K, 64)[..., :Cg]
    zh, zl = L.il_planes(L.split_bf16_il(z.reshape(B * S, H).contiguous()))
    pad = torch.zeros(B, S + K - 1, H, dtype=torch.float64, device="cuda")
    pad[:, K // 2:K // 2 + S] = (zh.double() + zl.double()).reshape(B, S, H)
    ref = torch.empty(B, S, H, dtype=torch.float64, device="cuda")
    for gi in range(G):
        win = pad[:, :, gi * Cg:(gi + 1) * Cg].unfold(1, K, 1)[:, :S]                  # [B, S, Cg, K]
        ref[:, :, gi * Cg:(gi + 1) * Cg] = torch.einsum("bsck,nkc->bsn", win, W[gi])
    pre = ref + bias.double()
    want = 0.5 * pre * (1 + torch.erf(pre / 2 ** 0.5)) + z.double()
    e_gemm, e_direct = (gemm.double() - want).abs().max().item(), (direct.double() - want).abs().max().item()
    assert e_direct < 5e-5 and e_direct < 1.05 * e_gemm + 1e-7, (e_gemm, e_direct)


@pytest.mark.parametrize("samples", [64000, 40000])
def test_wav2vec2_base_width_forward_with_resident_positional_conv(eng, samples):
    """Base-width encoder (768-d, 16 groups of 48 channels, kernel 128): the engine takes the resident-slab kernel; the
    output matches the oracle and the GEMM-path output to rounding."""
    from transformers import Wav2Vec2Config, Wav2Vec2Model
    E, L = eng
    L.lib.ser_debug_set_posconv_gemm.argtypes = [L.i32]
    torch.manual_seed(5)
    wc = Wav2Vec2Config(hidden_size=768, num_hidden_layers=1, num_attention_heads=12, intermediate_size=256,
                        conv_dim=[64] * 6 + [32], num_conv_pos_embeddings=128, num_conv_pos_embedding_groups=16)
    sd = {k: v.detach() for k, v in Wav2Vec2Model(wc).state_dict().items()}
    e = E.Wav2Vec2Engine(wc, sd, "cuda", L.PREC_BF16X3)
    waves = 0.1 * torch.randn(2, samples)
    try:
        L.lib.ser_debug_set_posconv_gemm(1)
        ref = e.forward(waves.cuda()).clone()
    finally:
        L.lib.ser_debug_set_posconv_gemm(0)
    got = e.forward(waves.cuda())
    assert (ref - got).abs().max().item() < 2e-5
    cfg = O.wav2vec2_config(hidden=768, layers=1, heads=12, ffn=256, conv_dim=wc.conv_dim, conv_kernel=wc.conv_kernel,
                            conv_stride=wc.conv_stride, pos_kernel=128, pos_groups=16, eps=wc.layer_norm_eps)
    want = O.wav2vec2_forward(sd, torch.stack([O.normalise_waveform(w) for w in waves]), cfg)
    assert (got.cpu() - want).abs().max().item() < 3e-4
