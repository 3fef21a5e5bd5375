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
