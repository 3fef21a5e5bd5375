"""Quality-gate / audio-conditioning kernels (csrc/frontend.hip, through the C ABI) against the vectors recorded from the
reference's AudioConditioningModule (tests/golden/dsp_frontend.npz) and against oracle/dsp_oracle.py."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "dsp_frontend.npz"))
NAMES = [str(n) for n in G["cond.names"]]
NOISE = {"unknown": 0, "low_frequency": 1, "high_frequency": 2, "mid_frequency": 3, "white_noise": 4}
DEC = {"reject": 0, "uncertain": 1, "accept": 2}


def _dev(x, dtype=torch.float32):
    return torch.as_tensor(np.asarray(x), dtype=dtype).cuda()


def accept_clip(T, rs, f0=150.0):
    """a clip the quality gates accept: 20 Hz amplitude modulation (period = 5 VAD hops, so every 5-frame median window
    holds three frames above the threshold), harmonics below 800 Hz, and a tail 26 dB down (the STFT SNR's noise frames)."""
    t = np.arange(T) / 16000.0
    v = sum(a * np.sin(2 * np.pi * f0 * (h + 1) * t + rs.uniform(0, 6)) for h, a in enumerate((0.3, 0.2, 0.1, 0.05)))
    x = (1 + 0.8 * np.sin(2 * np.pi * 20 * t + rs.uniform(0, 6))) * v * 0.4
    x[int(0.86 * T):] *= 0.05
    return (x + 1e-5 * rs.randn(T)).astype(np.float32)


def _groups():
    by_len = {}
    for n in NAMES:
        by_len.setdefault(len(G[f"cond.{n}.in"]), []).append(n)
    return list(by_len.values())


@pytest.mark.parametrize("names", _groups(), ids=lambda g: f"T{len(G[f'cond.{g[0]}.in'])}")
def test_conditioning_matches_reference_vectors(names):
    """every fixture clip of one length in ONE batched call; conditioned audio, the 12 feature inputs and the metadata."""
    from ser_amd import _ops as O
    wave = _dev(np.stack([G[f"cond.{n}.in"] for n in names]))
    out, raw, meta = O.audio_conditioning(wave)
    out, raw, meta = out.cpu().numpy(), raw.cpu().numpy(), meta.cpu().numpy()
    for i, n in enumerate(names):
        want = G[f"cond.{n}.out"]
        np.testing.assert_allclose(out[i], want, rtol=0, atol=2e-6 * max(1.0, np.abs(want).max()), err_msg=n)
        np.testing.assert_allclose(raw[i], G[f"cond.{n}.raw"], rtol=0, atol=2e-5, err_msg=n)
        m = G[f"cond.{n}.meta"]
        np.testing.assert_allclose(meta[i, :11], m, rtol=1e-5, atol=1e-4, err_msg=n)
        assert int(meta[i, 11]) == NOISE[str(G[f"cond.{n}.noise_type"])], n
        assert (meta[i, 0], bool(meta[i, 1]), bool(meta[i, 2])) == (np.float32(m[0]), bool(m[1]), bool(m[2])), n


def test_conditioning_of_rejected_clips_is_the_conditioning_of_silence():
    from ser_amd import _ops as O
    names = _groups()[0][:4]
    wave = _dev(np.stack([G[f"cond.{n}.in"] for n in names]))
    dec = torch.tensor([2, 0, 1, 2], dtype=torch.int32).cuda()
    out, raw, _ = O.audio_conditioning(wave, dec)
    ref, ref_raw, _ = O.audio_conditioning(wave)
    z, z_raw, _ = O.audio_conditioning(torch.zeros_like(wave))
    assert torch.equal(out[0], ref[0]) and torch.equal(out[3], ref[3]) and torch.equal(raw[0], ref_raw[0])
    assert not out[1].any() and not out[2].any()
    assert torch.equal(raw[1], z_raw[1]) and torch.equal(raw[2], z_raw[2])
    np.testing.assert_allclose(z_raw[0].cpu().numpy(), G["cond.zeros.raw"], atol=1e-6)


@pytest.mark.parametrize("pad_mode", ["constant", "reflect"])
def test_quality_gates_match_oracle(pad_mode):
    from oracle import dsp_oracle as D
    from ser_amd import _ops as O
    names = _groups()[0]
    clips = [G[f"cond.{n}.in"] for n in names]
    texts = [None, "hello", "hola", None, "x", None, None, "y", None][:len(names)]
    langs = [None, "en", "xx", None, None, None, None, "ja", None][:len(names)]
    rs = np.random.RandomState(11)
    for text, lang in ((None, None), ("with a transcript", "en"), (None, None)):
        names, texts, langs = names + [f"accept{len(names)}"], texts + [text], langs + [lang]
        clips.append(accept_clip(len(clips[0]), rs, rs.uniform(120, 200)))
    lid = np.array([D.language_entropy(t, l)[::2] for t, l in zip(texts, langs)], dtype=np.float32)
    raw, met, dec = O.quality_gates(_dev(np.stack(clips)), _dev(lid), pad_mode)
    raw, met, dec = raw.cpu().numpy(), met.cpu().numpy(), dec.cpu().numpy()
    seen = set()
    for i, n in enumerate(names):
        q = D.quality_metrics(clips[i], texts[i], langs[i], pad_mode=pad_mode)
        np.testing.assert_allclose(raw[i], q["features"], rtol=0, atol=2e-5, err_msg=n)
        want = [q["speech_prob"], q["snr_db"], q["clipping_percent"], q["spectral_naturalness"], q["music_prob"],
                q["laughter_prob"], q["quality_score"]]
        np.testing.assert_allclose(met[i, :7], want, rtol=1e-5, atol=1e-4, err_msg=n)
        assert dec[i] == DEC[q["decision"]] == int(met[i, 7]), n
        seen.add(q["decision"])
    assert len(seen) >= 2


def test_front_end_at_bench_size_against_oracle():
    """16 four-second clips (the BASELINE batch): speech-like mixtures with hum / rumble / noise drawn per clip."""
    from oracle import dsp_oracle as D
    from ser_amd import _ops as O
    rs = np.random.RandomState(5)
    T, B = 64000, 16
    t = np.arange(T) / 16000.0
    clips = []
    for b in range(B):
        f0 = rs.uniform(90, 220)
        env = 0.5 * (1 + np.sin(2 * np.pi * rs.uniform(2, 5) * t + rs.uniform(0, 6))) * (np.sin(2 * np.pi * rs.uniform(0.3, 0.9) * t) > -0.4)
        v = sum(rs.uniform(0.02, 0.3) / (h + 1) * np.sin(2 * np.pi * f0 * (h + 1) * t + rs.uniform(0, 6)) for h in range(12))
        x = env * v + rs.uniform(1e-4, 2e-2) * rs.randn(T)
        if b % 4 == 1:
            x = 0.3 * x + rs.uniform(0.1, 0.3) * np.sin(2 * np.pi * (50 if b % 8 == 1 else 60) * t)
        if b % 4 == 2:
            x = 0.2 * x + 0.3 * np.sin(2 * np.pi * rs.uniform(20, 70) * t)
        if b % 4 == 3:
            x = 0.3 * np.sin(2 * np.pi * rs.uniform(200, 900) * t) + 0.01 * rs.randn(T)
        if b >= 10:
            x = accept_clip(T, rs, rs.uniform(120, 200))
            if b >= 13:
                x = x + 0.2 * np.sin(2 * np.pi * (50 if b == 13 else 60) * t).astype(np.float32) * (t < 0.86 * 4)
        clips.append(x.astype(np.float32))
    wave = _dev(np.stack(clips))
    lid = np.tile(np.array([[1.0, 0.0]], dtype=np.float32), (B, 1))
    raw, met, dec = O.quality_gates(wave, _dev(lid))
    out, c_raw, c_meta = O.audio_conditioning(wave, dec)
    raw, dec, out, c_raw = raw.cpu().numpy(), dec.cpu().numpy(), out.cpu().numpy(), c_raw.cpu().numpy()
    flags = np.zeros(3)
    for b in range(B):
        fe = D.front_end(clips[b], None, None)
        np.testing.assert_allclose(raw[b], fe["quality"]["features"], atol=2e-5, err_msg=str(b))
        assert dec[b] == DEC[fe["quality"]["decision"]]
        np.testing.assert_allclose(out[b], fe["audio"], rtol=0, atol=2e-6 * max(1.0, np.abs(fe["audio"]).max()), err_msg=str(b))
        np.testing.assert_allclose(c_raw[b], fe["conditioning"]["features"], atol=2e-5, err_msg=str(b))
        flags += c_raw[b, :3]
    assert (dec == 2).sum() >= 3 and (dec != 2).sum() >= 3
    assert flags[0] >= 1 and flags[1] >= 1                      # a notch and a high-pass ran on accepted clips


def test_front_end_argument_checks():
    from ser_amd import _lib as L
    from ser_amd import _ops as O
    with pytest.raises(L.SerHipError, match="2048"):
        O.audio_conditioning(torch.zeros(2, 1000, device="cuda"))
    with pytest.raises(L.SerHipError, match="16 kHz"):
        O.audio_conditioning(torch.zeros(2, 4096, device="cuda"), sample_rate=8000)


def test_default_audio_encoder_runs_gates_and_conditioning_on_the_device():
    """AudioEncoder() with the reference's default flags (quality gates + conditioning on, ref audio_encoder.py:9-11):
    accepted clips are conditioned and encoded, the others enter as silence, the 8 + 12 features are projected and fused
    (ref :65-132) — against the oracle composition clip by clip; with transcripts every clip is silence (the reference's
    language table never lets a clip with text through)."""
    import warnings
    from transformers import Wav2Vec2Config
    from oracle import dsp_oracle as D
    from oracle import ser_oracle as R
    from ser_amd.models import AudioEncoder
    torch.manual_seed(3)
    wc = Wav2Vec2Config(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                        conv_dim=[64] * 7, num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=4)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        ae = AudioEncoder(hf_config=wc, adapter_dim=32).cuda().eval()
    assert any("webrtcvad" in str(x.message) for x in w)
    sd = {k: v.detach().cpu().float() for k, v in ae.state_dict().items()}
    cfg = R.wav2vec2_config(hidden=128, layers=2, heads=2, ffn=256, conv_dim=(64,) * 7, pos_kernel=16, pos_groups=4)
    rs = np.random.RandomState(21)
    T = 32000
    clips = [accept_clip(T, rs, 140.0), G["cond.hum50.in"], accept_clip(T, rs, 180.0), G["cond.tone_noise.in"]]
    clips[2] = clips[2] + (0.2 * np.sin(2 * np.pi * 50 * np.arange(T) / 16000.0) * (np.arange(T) < 0.86 * T)).astype(np.float32)

    def oracle(texts):
        outs, decs = [], []
        for i, c in enumerate(clips):
            fe = D.front_end(c, texts[i] if texts else None, None)
            x = R.normalise_waveform(torch.from_numpy(fe["audio"]))[None]
            s = R.adapter(R.wav2vec2_forward(R.sub(sd, "encoder."), x, cfg)[0], R.sub(sd, "adapter."))
            s = R.gate_feature_fusion(sd, s[None], torch.from_numpy(fe["quality"]["features"])[None],
                                      torch.from_numpy(fe["conditioning"]["features"])[None])[0]
            outs.append(s)
            decs.append(DEC[fe["quality"]["decision"]])
        return torch.stack(outs), decs

    with torch.no_grad():
        seq, mask = ae([torch.from_numpy(c) for c in clips])
    want, decs = oracle(None)
    assert decs.count(2) >= 2 and decs.count(2) < len(decs)
    assert ae.last_decisions.cpu().tolist() == decs
    assert mask.shape == seq.shape[:2] and bool(mask.all())
    np.testing.assert_allclose(seq.cpu().numpy(), want.numpy(), rtol=0, atol=3e-4)
    texts = ["a transcript"] * len(clips)
    with torch.no_grad():
        seq_t, _ = ae([torch.from_numpy(c) for c in clips], texts)
    want_t, decs_t = oracle(texts)
    assert 2 not in decs_t and ae.last_decisions.cpu().tolist() == decs_t
    np.testing.assert_allclose(seq_t.cpu().numpy(), want_t.numpy(), rtol=0, atol=3e-4)
    # caller-supplied raw features still bypass the front end
    q = torch.rand(len(clips), 8).cuda()
    c = torch.rand(len(clips), 12).cuda()
    with torch.no_grad():
        seq_g, _ = ae([torch.from_numpy(x) for x in clips], gate_features=(q, c))
    assert not torch.allclose(seq_g, seq)


@pytest.mark.parametrize("T", [2048, 2049, 8191, 8207, 12345, 16384 + 15, 160000])
def test_front_end_odd_lengths_against_oracle(T):
    """lengths around the kernels' internal boundaries (one analysis window, the 8192-sample filter tile with and without
    the 2 x 15 / 2 x 9 padding samples, a 10 s clip): gates + conditioning of three clips per length (hum + rumble so that
    both filters run, a tone so that the Wiener filter runs, an accepted clip) against the oracle."""
    from oracle import dsp_oracle as D
    from ser_amd import _ops as O
    rs = np.random.RandomState(T)
    t = np.arange(T) / 16000.0
    speechy = sum(0.2 / (h + 1) * np.sin(2 * np.pi * 140 * (h + 1) * t + rs.uniform(0, 6)) for h in range(6)) * (1 + 0.5 * np.sin(2 * np.pi * 3 * t))
    clips = [(0.3 * speechy + 0.25 * np.sin(2 * np.pi * 50 * t) + 0.2 * np.sin(2 * np.pi * 60 * t + 1) + 0.1 * np.sin(2 * np.pi * 30 * t)),
             0.4 * np.sin(2 * np.pi * 440 * t) + 0.01 * rs.randn(T),
             accept_clip(T, rs, 150.0)]
    clips = [c.astype(np.float32) for c in clips]
    wave = _dev(np.stack(clips))
    lid = _dev(np.tile(np.array([[1.0, 0.0]], dtype=np.float32), (3, 1)))
    raw, met, dec = O.quality_gates(wave, lid)
    out, c_raw, c_meta = O.audio_conditioning(wave)                  # conditioning of the clips themselves (no gating)
    ran = np.zeros(3)
    for b in range(3):
        q = D.quality_metrics(clips[b], None)
        c = D.condition_audio(clips[b])
        np.testing.assert_allclose(raw[b].cpu().numpy(), q["features"], atol=3e-5, err_msg=f"quality {b}")
        assert int(dec[b]) == DEC[q["decision"]]
        np.testing.assert_allclose(out[b].cpu().numpy(), c["audio"], rtol=0, atol=2e-6 * max(1.0, np.abs(c["audio"]).max()), err_msg=f"audio {b}")
        np.testing.assert_allclose(c_raw[b].cpu().numpy(), c["features"], atol=3e-5, err_msg=f"conditioning {b}")
        ran += c["features"][:3]
    assert ran[0] >= 1 and ran[1] >= 1 and (ran[2] >= 1 or T < 4000)


def test_default_audio_encoder_on_ragged_clips():
    """clips of different lengths: one front-end + encoder pass per distinct length, features scattered back in batch
    order, the padded frames of the shorter clip stay zero (ref audio_encoder.py:129-166)."""
    import warnings
    from transformers import Wav2Vec2Config
    from oracle import dsp_oracle as D
    from oracle import ser_oracle as R
    from ser_amd.models import AudioEncoder
    torch.manual_seed(5)
    wc = Wav2Vec2Config(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                        conv_dim=[64] * 7, num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=4)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ae = AudioEncoder(hf_config=wc, adapter_dim=32, vad_method="librosa").cuda().eval()
    sd = {k: v.detach().cpu().float() for k, v in ae.state_dict().items()}
    cfg = R.wav2vec2_config(hidden=128, layers=2, heads=2, ffn=256, conv_dim=(64,) * 7, pos_kernel=16, pos_groups=4)
    rs = np.random.RandomState(9)
    clips = [accept_clip(16000, rs, 150.0), accept_clip(12000, rs, 170.0), accept_clip(16000, rs, 130.0)]
    with torch.no_grad():
        seq, mask = ae([torch.from_numpy(c) for c in clips])
    S = [R.wav2vec2_forward(R.sub(sd, "encoder."), torch.zeros(1, len(c)), cfg).shape[1] for c in clips]
    assert seq.shape[1] == max(S) and bool(mask.all())
    for i, c in enumerate(clips):
        fe = D.front_end(c, None, None)
        x = R.normalise_waveform(torch.from_numpy(fe["audio"]))[None]
        s = R.adapter(R.wav2vec2_forward(R.sub(sd, "encoder."), x, cfg)[0], R.sub(sd, "adapter."))
        s = R.gate_feature_fusion(sd, s[None], torch.from_numpy(fe["quality"]["features"])[None],
                                  torch.from_numpy(fe["conditioning"]["features"])[None])[0]
        np.testing.assert_allclose(seq[i, :S[i]].cpu().numpy(), s.numpy(), rtol=0, atol=3e-4, err_msg=str(i))
        assert not seq[i, S[i]:].any()
