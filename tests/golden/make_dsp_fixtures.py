#!/usr/bin/env python3
"""Golden vectors for the quality-gate / audio-conditioning front end (tests/golden/dsp_frontend.npz).

Run in the build container only (needs /root/reference, read-only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_dsp_fixtures.py

What it records
  * `AudioConditioningModule._process_single_sample` of the reference (ref audio_conditioning.py:503-584) on ten synthetic
    clips that reach every branch that can run here (hum notch 50 / 60 Hz, high-pass with each cutoff, Wiener denoise,
    compression, all-zero clip, clip shorter than one second): conditioned audio, the 12 raw feature inputs and the
    metadata scalars.  The module only needs numpy + scipy at run time; its `librosa` / `soundfile` imports are unused, so
    empty placeholder modules are registered for the import statements (as tests/golden/make_fixtures.py does).
  * the pieces of `quality_gates.py` that do not call librosa, run from the reference's own classes: clipping detector,
    language-entropy table (with the `detect` symbol set to a fixed answer, and with langdetect unavailable), abstain
    policy + quality score on a grid of metric values, median smoothing and frame -> segment conversion.
Only data is written (inputs and expected outputs).
"""
import os
import sys
import types

sys.dont_write_bytecode = True
import numpy as np
import torch

REF = "/root/reference/src"
OUT = os.path.dirname(os.path.abspath(__file__))
SR = 16000


def synth_clips():
    """name -> float32 clip.  Deterministic (legacy RandomState streams)."""
    rs = np.random.RandomState(1234)
    T = 2 * SR
    t = np.arange(T) / SR
    env = 0.5 * (1 + np.sin(2 * np.pi * 3.1 * t)) * (np.sin(2 * np.pi * 0.7 * t) > -0.3)
    voiced = sum(a * np.sin(2 * np.pi * f * t + p) for a, f, p in
                 [(0.30, 140, 0.1), (0.22, 280, 0.7), (0.15, 420, 1.3), (0.10, 840, 2.1), (0.06, 1680, 0.4), (0.03, 2520, 2.9)])
    speech = env * voiced + 0.002 * rs.randn(T)
    clips = {
        "speech": speech,
        "hum50": 0.3 * speech + 0.25 * np.sin(2 * np.pi * 50 * t),
        "hum60_rumble": 0.3 * speech + 0.2 * np.sin(2 * np.pi * 60 * t + 0.3) + 0.05 * np.sin(2 * np.pi * 23 * t),
        "tone_noise": 0.4 * np.sin(2 * np.pi * 440 * t) + 0.01 * rs.randn(T),
        "rumble": 0.2 * speech + 0.3 * np.sin(2 * np.pi * 31 * t) + 0.1 * np.sin(2 * np.pi * 97 * t + 1.0),
        "rumble_wide": 0.1 * speech + 0.2 * np.sin(2 * np.pi * 88 * t) + 0.2 * np.sin(2 * np.pi * 120 * t + 1.0),
        "clicks": 1e-4 * rs.randn(T),
        "white": 0.1 * rs.rand(T) * np.sign(rs.randn(T)),
        "zeros": np.zeros(T),
        "short": speech[: SR // 2].copy(),
    }
    for k in (3000, 14000, 25000):
        clips["clicks"][k] = 0.9
    return {k: v.astype(np.float32) for k, v in clips.items()}


def _import_reference():
    for name in ("librosa", "soundfile"):
        if name not in sys.modules:
            try:
                __import__(name)
            except ImportError:
                sys.modules[name] = types.ModuleType(name)
    sys.path.insert(0, os.path.join(REF, "models"))
    import audio_conditioning as ac
    import quality_gates as qg
    return ac, qg


def main():
    ac, qg = _import_reference()
    assert not ac.NOISEREDUCE_AVAILABLE and not ac.PYLN_AVAILABLE
    out = {}
    mod = ac.AudioConditioningModule(SR)
    names = []
    for name, clip in synth_clips().items():
        y, f = mod._process_single_sample(clip.copy())
        names.append(name)
        out[f"cond.{name}.in"] = clip
        out[f"cond.{name}.out"] = y.detach().cpu().numpy().astype(np.float32)
        # the 12 inputs of conditioning_projection exactly as ref :562-575 builds them
        raw = np.array([float(f.hum_filtered), float(f.hpf_applied), float(f.denoise_applied), float(f.dereverb_applied),
                        f.snr_before / 50.0, f.snr_after / 50.0, f.denoise_gain_db / 20.0, f.estimated_t60 / 2.0,
                        (f.lufs_original + 60) / 60, f.lufs_adjustment / 20.0, f.peak_reduction_db / 20.0,
                        f.compression_ratio / 4.0], dtype=np.float64)
        out[f"cond.{name}.raw"] = raw
        out[f"cond.{name}.meta"] = np.array([f.hpf_cutoff, 50.0 in f.hum_frequencies, 60.0 in f.hum_frequencies,
                                             f.snr_before, f.snr_after, f.denoise_gain_db, f.estimated_t60, f.lufs_original,
                                             f.lufs_adjustment, f.peak_reduction_db, f.compression_ratio], dtype=np.float64)
        out[f"cond.{name}.noise_type"] = np.array(f.noise_type_detected)
        print(f"{name:14s} hum={f.hum_frequencies} hpf={f.hpf_applied}@{f.hpf_cutoff:.2f} denoise={f.denoise_applied} "
              f"snr={f.snr_before:.2f}->{f.snr_after:.2f} t60={f.estimated_t60} lufs={f.lufs_original:.2f} "
              f"adj={f.lufs_adjustment:.2f} ratio={f.compression_ratio:.3f} noise={f.noise_type_detected}")
    out["cond.names"] = np.array(names)

    # ---- quality gates: the librosa-free pieces --------------------------------------------------------------------
    sq = qg.SignalQualityAssessor(SR)
    for name, clip in synth_clips().items():
        out[f"qg.clip.{name}"] = np.float64(sq._detect_clipping(clip))
    lid = qg.LanguageIdentifier()
    rows = []
    for avail, det in ((False, None), (True, "en"), (True, "ja"), (True, "hi")):
        lid.available = avail
        qg.detect = (lambda text, d=det: d)
        e, lang, conf = lid.identify_language("some words")
        rows.append((float(avail), e, conf))
        out[f"qg.lid.{det}.lang"] = np.array(lang)
    out["qg.lid"] = np.array(rows)
    lid.available = True
    out["qg.lid.blank"] = np.array(lid.identify_language("   ")[::2])
    pol = qg.EarlyAbstainPolicy()
    grid, dec, score = [], [], []
    for snr in (2.0, 5.0, 7.5, 10.0, 25.0):
        for clipp in (0.5, 30.0, 31.0):
            for sp in (0.2, 0.4, 0.6, 0.8, 0.95):
                for ent in (1.0, 1.5, 1.5290, 2.3026):
                    for music in (0.1, 0.2, 0.35):
                        m = qg.QualityMetrics(speech_prob=sp, speech_segments=[], snr_db=snr, clipping_percent=clipp,
                                              spectral_naturalness=0.4, lid_entropy=ent, dominant_language="en",
                                              dominant_language_conf=0.5, music_prob=music, laughter_prob=0.0,
                                              abstain_recommendation="", quality_score=0.0, quality_features=torch.zeros(8))
                        grid.append((snr, clipp, sp, ent, music))
                        dec.append({"reject": 0, "uncertain": 1, "accept": 2}[pol.make_decision(m)])
                        score.append(pol.compute_quality_score(m))
    out["qg.policy.grid"], out["qg.policy.decision"], out["qg.policy.score"] = np.array(grid), np.array(dec), np.array(score)
    vad = qg.VoiceActivityDetector(method="librosa", sample_rate=SR)
    rs = np.random.RandomState(7)
    fr = rs.rand(64) > 0.45
    out["qg.median.in"], out["qg.median.out"] = fr, vad._smooth_speech_frames(fr)
    out["qg.segments"] = np.array(vad._frames_to_segments(out["qg.median.out"]))
    path = os.path.join(OUT, "dsp_frontend.npz")
    np.savez_compressed(path, **out)
    print(f"wrote dsp_frontend.npz: {os.path.getsize(path) / 1e6:.2f} MB, {len(out)} arrays")


if __name__ == "__main__":
    main()
