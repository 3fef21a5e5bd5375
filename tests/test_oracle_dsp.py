"""The DSP front-end oracle (oracle/dsp_oracle.py) against the vectors recorded from the reference's own modules
(tests/golden/make_dsp_fixtures.py -> dsp_frontend.npz), and its librosa restatements against scipy's STFT."""
import os

import numpy as np
import pytest

from oracle import dsp_oracle as D

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "dsp_frontend.npz"))
NAMES = [str(n) for n in G["cond.names"]]


@pytest.mark.parametrize("name", NAMES)
def test_conditioning_matches_reference(name):
    r = D.condition_audio(G[f"cond.{name}.in"])
    np.testing.assert_allclose(r["audio"], G[f"cond.{name}.out"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(r["features"], G[f"cond.{name}.raw"].astype(np.float32), rtol=0, atol=1e-6)
    meta = G[f"cond.{name}.meta"]
    got = [r["hpf_cutoff"], 50 in r["hum_frequencies"], 60 in r["hum_frequencies"], r["snr_before"], r["snr_after"],
           r["denoise_gain_db"], r["estimated_t60"], r["lufs_original"], r["lufs_adjustment"], r["peak_reduction_db"],
           r["compression_ratio"]]
    np.testing.assert_allclose(np.array(got, dtype=np.float64), meta, rtol=1e-9, atol=1e-9)
    assert r["noise_type"] == str(G[f"cond.{name}.noise_type"])


def test_fixture_clips_reach_every_live_branch():
    raw = np.stack([G[f"cond.{n}.raw"] for n in NAMES])
    meta = np.stack([G[f"cond.{n}.meta"] for n in NAMES])
    assert raw[:, 0].any() and raw[:, 1].any() and raw[:, 2].any()       # notch, high-pass, Wiener
    assert not raw[:, 3].any()                                          # de-reverberation cannot trigger (see estimate_t60)
    assert {80.0, 85.9375, 100.0} <= set(meta[:, 0])                    # high-pass cutoffs
    assert meta[:, 1].any() and meta[:, 2].any()                        # 50 Hz and 60 Hz
    assert (meta[:, 10] > 1.0).any()                                    # compression
    assert set(np.round(meta[:, 6], 3)) == {0.0, 0.1}


def test_clipping_language_policy_smoothing_match_reference():
    from tests.golden.make_dsp_fixtures import synth_clips
    for name, clip in synth_clips().items():
        assert D.clipping_percent(clip) == pytest.approx(float(G[f"qg.clip.{name}"]), abs=1e-9)
    rows = G["qg.lid"]
    for (avail, ent, conf), det in zip(rows, (None, "en", "ja", "hi")):
        e, lang, c = D.language_entropy("some words", det)
        assert (e, c) == (pytest.approx(ent, abs=1e-12), pytest.approx(conf, abs=1e-12))
        assert lang == str(G[f"qg.lid.{det}.lang"])
    assert D.language_entropy("   ", "en")[::2] == tuple(G["qg.lid.blank"])
    assert D.language_entropy(None, "en") == (1.0, "unknown", 0.0)
    names = {0: "reject", 1: "uncertain", 2: "accept"}
    for (snr, clipp, sp, ent, music), d, s in zip(G["qg.policy.grid"], G["qg.policy.decision"], G["qg.policy.score"]):
        assert D.abstain_decision(snr, clipp, sp, ent, music) == names[int(d)]
        assert D.quality_score(snr, sp, clipp, 0.4, ent, music) == pytest.approx(float(s), abs=1e-12)
    from scipy.ndimage import median_filter
    assert (median_filter(G["qg.median.in"], size=5) == G["qg.median.out"]).all()
    np.testing.assert_allclose(np.array(D.frames_to_segments(G["qg.median.out"], 25)), G["qg.segments"])


def test_with_text_no_clip_is_ever_accepted():
    """the reference's language table makes every clip that comes with a transcript 'uncertain' or 'reject': its three
    possible entropies are 1.5 (langdetect missing), 1.529 (a listed language) and ln 10 (any other), none below 1.5."""
    for det in (None, "en", "xx"):
        e = D.language_entropy("hello there", det)[0]
        assert e >= 1.5
        assert D.abstain_decision(40.0, 0.1, 0.99, e, 0.0) == "uncertain"
    assert D.abstain_decision(40.0, 0.1, 0.99, D.language_entropy(None, None)[0], 0.0) == "accept"


def test_stft_restatement_against_scipy():
    """librosa is absent: the STFT restatement is checked against scipy.signal.ShortTimeFFT (an independent
    implementation) with the same periodic Hann window, hop and zero centre padding."""
    from scipy.signal import ShortTimeFFT, get_window
    rs = np.random.RandomState(0)
    y = rs.randn(6000).astype(np.float32)
    mag = D.stft_mag(y, 2048, 512, "constant")
    assert mag.shape == (1025, 1 + 6000 // 512)
    st = ShortTimeFFT(get_window("hann", 2048, fftbins=True), hop=512, fs=16000, fft_mode="onesided", scale_to=None, phase_shift=None)
    S = np.abs(st.stft(y.astype(np.float64), p0=0, p1=mag.shape[1]))      # slice k is centred on sample k * hop
    np.testing.assert_allclose(mag, S, rtol=0, atol=2e-3)
    ref = D.stft_mag(y, 2048, 512, "reflect")
    assert np.abs(ref[:, 3:-3] - mag[:, 3:-3]).max() == 0 and np.abs(ref[:, 0] - mag[:, 0]).max() > 1e-2


def test_spectral_descriptors_on_known_spectra():
    f = D.fft_frequencies()
    S = np.zeros((1025, 3), dtype=np.float32)
    S[128, 0] = 1.0                                  # a single line at 1 kHz
    S[[64, 192], 1] = 1.0                            # two equal lines at 500 / 1500 Hz
    np.testing.assert_allclose(D.spectral_centroid(S), [1000.0, 1000.0, 0.0])
    np.testing.assert_allclose(D.spectral_bandwidth(S), [0.0, 500.0, 0.0])
    np.testing.assert_allclose(D.spectral_rolloff(S), [1000.0, 1500.0, 0.0])
    assert f[128] == 1000.0


def test_quality_metrics_end_to_end_shapes_and_decisions():
    from tests.golden.make_dsp_fixtures import synth_clips
    c = synth_clips()
    q = D.quality_metrics(c["speech"], None)
    assert q["features"].shape == (8,) and q["decision"] in ("accept", "uncertain", "reject")
    assert D.quality_metrics(c["zeros"], None)["decision"] == "reject"
    with pytest.raises(ValueError):
        D.quality_metrics(c["speech"], vad_method="webrtc")
    fe = D.front_end(c["speech"], "a transcript", None)
    assert fe["quality"]["decision"] != "accept" and not fe["audio"].any()
