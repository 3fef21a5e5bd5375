"""CPU oracle for the quality-gate / audio-conditioning front end.  TEST INFRASTRUCTURE ONLY.

numpy / scipy restatement of the two CPU DSP side-cars that the reference's default `AudioEncoder()` runs per clip
before Wav2Vec2 (ref: = /root/reference/src/models/...):

  * `quality_gates.py`      -> `quality_metrics()`  : 8 raw quality features + accept / uncertain / reject
  * `audio_conditioning.py` -> `condition_audio()`  : conditioned clip + 12 raw conditioning features

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this file.

Pinning
  * `condition_audio` is pinned: the reference module needs only numpy + scipy at run time (its `librosa` / `soundfile`
    imports are unused, `noisereduce` / `pyloudnorm` are optional and absent here exactly as in a plain
    `pip install -r requirements.txt` for pyloudnorm), so `tests/golden/make_dsp_fixtures.py` runs the reference's own
    `AudioConditioningModule._process_single_sample` on synthetic clips and `tests/test_oracle_dsp.py` compares.
    `noisereduce` (requirements.txt, unpinned ">=2.0.0") is not installed: the reference then takes its own
    `wiener_denoise` branch (ref audio_conditioning.py:250-254), which is what is restated and built.
  * of `quality_metrics`, the pieces that do not touch librosa (clipping, language-entropy table, abstain policy, quality
    score, median smoothing, frame -> segment conversion) are pinned the same way.  The librosa-defined pieces (STFT
    magnitude, rms framing, spectral centroid / rolloff / bandwidth) restate librosa's published algorithm (librosa is a
    requirements.txt dependency, "librosa>=0.9.0", not installed here): PARITY UNPINNED for those, cross-checked only
    against scipy.signal.stft in tests/test_oracle_dsp.py.  `pad_mode` selects the centre padding of librosa >= 0.10
    ("constant", the default) or of 0.9.x ("reflect").
  * webrtcvad (a compiled GMM VAD) is not restated: `vad_method="webrtc"` raises ValueError, which is what the
    reference does when webrtcvad is missing (ref quality_gates.py:61-70).  `vad_method="librosa"` is the energy VAD.
  * langdetect is not restated (a text classifier, not part of the audio path): the caller supplies the detected
    language code or None; None follows the reference's "langdetect unavailable" branch (ref :259-260).
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import numpy as np
from scipy import signal
from scipy.ndimage import median_filter
from scipy.signal import butter, filtfilt, iirnotch, wiener

SR = 16000
LID_LANGUAGES = ['en', 'es', 'fr', 'de', 'it', 'pt', 'ru', 'ja', 'ko', 'zh']     # ref quality_gates.py:269

# ---------------------------------------------------------------------------------------------------------------------
# librosa restatements (librosa/core/spectrum.py stft, librosa/feature/spectral.py rms / spectral_*; version >= 0.10)
# ---------------------------------------------------------------------------------------------------------------------

def _pad_center(y: np.ndarray, n: int, pad_mode: str) -> np.ndarray:
    if pad_mode == "constant":
        return np.pad(y, (n, n), mode="constant")
    return np.pad(y, (n, n), mode="reflect")


def frames(y: np.ndarray, frame_length: int, hop: int) -> np.ndarray:
    """[frame_length, n_frames] view like librosa.util.frame(axis=-1)."""
    n = 1 + (len(y) - frame_length) // hop
    idx = np.arange(frame_length)[:, None] + hop * np.arange(n)[None, :]
    return y[idx]


def stft_mag(y: np.ndarray, n_fft: int = 2048, hop: int = 512, pad_mode: str = "constant") -> np.ndarray:
    """|librosa.stft(y, n_fft, hop)|: periodic Hann window, centre padding n_fft/2, rfft per frame -> [1 + n_fft/2, 1 + T/hop]."""
    y = np.asarray(y, dtype=np.float32)
    win = signal.get_window("hann", n_fft, fftbins=True).astype(np.float32)
    fr = frames(_pad_center(y, n_fft // 2, pad_mode), n_fft, hop)
    return np.abs(np.fft.rfft(fr * win[:, None], axis=0)).astype(np.float32)


def rms(y: np.ndarray, frame_length: int = 2048, hop: int = 512, pad_mode: str = "constant") -> np.ndarray:
    """librosa.feature.rms(y=...)[0]: centre-padded frames, sqrt(mean(x^2))."""
    y = np.asarray(y, dtype=np.float32)
    fr = frames(_pad_center(y, frame_length // 2, pad_mode), frame_length, hop)
    return np.sqrt(np.mean(np.abs(fr) ** 2, axis=0))


def fft_frequencies(sr: int = SR, n_fft: int = 2048) -> np.ndarray:
    return np.fft.rfftfreq(n_fft, 1.0 / sr)


def _normalize_l1(S: np.ndarray) -> np.ndarray:
    """librosa.util.normalize(S, norm=1, axis=-2): columns whose sum is below `tiny` are left unscaled (i.e. zero)."""
    length = np.sum(np.abs(S), axis=0, keepdims=True)
    length = np.where(length < np.finfo(S.dtype).tiny, 1.0, length)
    return S / length


def spectral_centroid(S: np.ndarray, sr: int = SR) -> np.ndarray:
    freq = fft_frequencies(sr, 2 * (S.shape[0] - 1))
    return np.sum(freq[:, None] * _normalize_l1(S), axis=0)


def spectral_bandwidth(S: np.ndarray, sr: int = SR) -> np.ndarray:
    freq = fft_frequencies(sr, 2 * (S.shape[0] - 1))
    dev = np.abs(freq[:, None] - spectral_centroid(S, sr)[None, :])
    return np.sum(_normalize_l1(S) * dev ** 2, axis=0) ** 0.5


def spectral_rolloff(S: np.ndarray, sr: int = SR, roll_percent: float = 0.85) -> np.ndarray:
    freq = fft_frequencies(sr, 2 * (S.shape[0] - 1))
    total = np.cumsum(S, axis=0)
    thr = roll_percent * total[-1]
    ind = np.where(total < thr[None, :], np.nan, 1.0)
    return np.nanmin(ind * freq[:, None], axis=0)


# ---------------------------------------------------------------------------------------------------------------------
# quality gates (ref quality_gates.py)
# ---------------------------------------------------------------------------------------------------------------------

def energy_vad(audio: np.ndarray, sr: int = SR, pad_mode: str = "constant") -> Tuple[float, np.ndarray]:
    """ref :111-133: 25 ms / 10 ms rms frames, threshold = 30th percentile + 0.1 std, 5-tap median smoothing."""
    energy = rms(audio, int(sr * 0.025), int(sr * 0.010), pad_mode)
    thr = np.percentile(energy, 30) + 0.1 * np.std(energy)
    speech = median_filter(energy > thr, size=5)
    return float(np.mean(speech)), speech


def frames_to_segments(speech: Sequence[bool], frame_ms: float) -> List[Tuple[float, float]]:
    """ref :139-162."""
    seg, on, t0 = [], False, 0.0
    for i, s in enumerate(speech):
        if bool(s) and not on:
            t0, on = i * frame_ms / 1000.0, True
        elif not bool(s) and on:
            seg.append((t0, i * frame_ms / 1000.0))
            on = False
    if on:
        seg.append((t0, len(speech) * frame_ms / 1000.0))
    return seg


def estimate_snr_spectral(mag: np.ndarray) -> float:
    """ref :189-214: only the LAST 10 % of frames end up in the noise estimate (the first assignment is overwritten)."""
    nf = int(0.1 * mag.shape[1])
    if nf == 0:                      # mag[:, 0:-0] is empty -> nan -> python's min(50.0, nan) keeps 50.0
        return 50.0
    noise = np.mean(mag[:, -nf:], axis=1)
    sig = np.mean(mag[:, nf:-nf], axis=1)
    sp, npow = np.mean(sig ** 2), np.mean(noise ** 2)
    snr = 10 * np.log10(sp / npow) if npow > 0 else 50.0
    return max(0.0, min(50.0, float(snr)))


def clipping_percent(audio: np.ndarray) -> float:
    """ref :216-226."""
    m = np.max(np.abs(audio))
    a = audio / m if m > 0 else audio
    return float(np.sum(np.abs(a) > 0.95) / len(audio) * 100)


def spectral_naturalness(mag: np.ndarray, sr: int = SR) -> float:
    """ref :228-247 (the rolloff is in Hz, so its term is 0 for anything but a near-DC spectrum, as in the reference)."""
    c = 1.0 - np.clip(abs(np.mean(spectral_centroid(mag, sr)) - 2000) / 2000, 0, 1)
    r = 1.0 - np.clip(abs(np.mean(spectral_rolloff(mag, sr)) - 0.85) / 0.15, 0, 1)
    b = 1.0 - np.clip(abs(np.mean(spectral_bandwidth(mag, sr)) - 1000) / 1000, 0, 1)
    return float((c + r + b) / 3)


def language_entropy(text: Optional[str], detected: Optional[str], enabled: bool = True) -> Tuple[float, str, float]:
    """ref :252-301 + :514-517.  `detected` = langdetect's code for the text, or None when langdetect is unavailable."""
    if not (text and enabled):
        return 1.0, "unknown", 0.0
    if detected is None or not text.strip():
        return 1.5, "unknown", 0.0
    if detected in LID_LANGUAGES:
        probs = [0.05] * len(LID_LANGUAGES)
        probs[LID_LANGUAGES.index(detected)] = 0.7
        probs = np.array(probs) / np.sum(probs)
    else:
        probs = np.ones(len(LID_LANGUAGES)) / len(LID_LANGUAGES)
    ent = -np.sum(probs * np.log(probs + 1e-10))
    k = int(np.argmax(probs))
    return float(ent), LID_LANGUAGES[k], float(probs[k])


def abstain_decision(snr_db, clip_pct, speech_prob, lid_entropy, music_prob) -> str:
    """ref :347-380."""
    if snr_db < 5.0 or clip_pct > 30.0 or speech_prob < 0.4:
        return 'reject'
    if 5.0 <= snr_db < 10.0 or lid_entropy > 1.5 or music_prob > 0.2:
        return 'uncertain'
    if snr_db >= 10.0 and speech_prob >= 0.8 and lid_entropy < 1.5:
        return 'accept'
    return 'uncertain'


def quality_score(snr_db, speech_prob, clip_pct, naturalness, lid_entropy, music_prob) -> float:
    """ref :382-403."""
    return float(0.25 * np.clip(snr_db / 20.0, 0, 1) + 0.25 * speech_prob + 0.15 * (1.0 - np.clip(clip_pct / 100.0, 0, 1))
                 + 0.15 * naturalness + 0.10 * (1.0 - np.clip(lid_entropy / 2.0, 0, 1)) + 0.10 * (1.0 - music_prob))


def quality_metrics(audio: np.ndarray, text: Optional[str] = None, detected_lang: Optional[str] = None,
                    vad_method: str = "librosa", sr: int = SR, pad_mode: str = "constant") -> dict:
    """ref :497-560 `_process_single_sample`: raw (un-projected) 8 features, the decision and the scalar metrics."""
    if vad_method != "librosa":
        raise ValueError(f"VAD method '{vad_method}' not available")
    audio = np.asarray(audio, dtype=np.float32)
    speech_prob, speech = energy_vad(audio, sr, pad_mode)
    mag = stft_mag(audio, 2048, 512, pad_mode)
    snr = estimate_snr_spectral(mag)
    clip = clipping_percent(audio)
    nat = spectral_naturalness(mag, sr)
    ent, lang, conf = language_entropy(text, detected_lang)
    music = float(np.clip(np.mean(spectral_centroid(mag, sr)) / 4000, 0, 1))
    laughter = float(np.clip(np.var(rms(audio, 2048, 512, pad_mode)) / 0.1, 0, 1))
    decision = abstain_decision(snr, clip, speech_prob, ent, music)
    feats = np.array([speech_prob, snr / 50.0, clip / 100.0, nat, ent / 2.0, conf, music, laughter], dtype=np.float32)
    return dict(features=feats, decision=decision, speech_prob=speech_prob, snr_db=snr, clipping_percent=clip,
                spectral_naturalness=nat, lid_entropy=ent, dominant_language=lang, dominant_language_conf=conf,
                music_prob=music, laughter_prob=laughter,
                quality_score=quality_score(snr, speech_prob, clip, nat, ent, music),
                speech_segments=frames_to_segments(speech, 25))


# ---------------------------------------------------------------------------------------------------------------------
# audio conditioning (ref audio_conditioning.py), scipy calls as the reference makes them
# ---------------------------------------------------------------------------------------------------------------------

def detect_hum(audio: np.ndarray, sr: int = SR) -> List[float]:
    """ref :66-82."""
    freqs, psd = signal.welch(audio, fs=sr, nperseg=2048)
    out = []
    for f in (50, 60):
        if psd[np.argmin(np.abs(freqs - f))] > np.mean(psd) + 2 * np.std(psd):
            out.append(f)
    return out


def hpf_decision(audio: np.ndarray, sr: int = SR) -> Tuple[bool, float]:
    """ref :107-137."""
    freqs, psd = signal.welch(audio, fs=sr, nperseg=2048)
    low, tot = np.sum(psd[freqs < 200]), np.sum(psd)
    apply_ = (low / tot if tot > 0 else 0) > 0.2
    cutoff = 80
    if apply_:
        cum = np.cumsum(psd)
        idx = np.where(cum > 0.1 * cum[-1])[0]
        if len(idx) > 0:
            cutoff = max(80, min(100, freqs[idx[0]]))
    return bool(apply_), float(cutoff)


def snr_energy(audio: np.ndarray) -> float:
    """ref :161-173."""
    e, floor = np.mean(audio ** 2), np.percentile(audio ** 2, 10)
    snr = 10 * np.log10(e / floor) if floor > 0 else 50.0
    return max(0.0, min(50.0, float(snr)))


def noise_type(audio: np.ndarray, sr: int = SR) -> str:
    """ref :175-203."""
    freqs, psd = signal.welch(audio, fs=sr, nperseg=1024)
    lo, mid, hi = np.sum(psd[freqs < 500]), np.sum(psd[(freqs >= 500) & (freqs < 2000)]), np.sum(psd[freqs >= 2000])
    tot = lo + mid + hi
    if not tot > 0:
        return "unknown"
    if lo / tot > 0.5:
        return "low_frequency"
    if hi / tot > 0.4:
        return "high_frequency"
    if mid / tot > 0.6:
        return "mid_frequency"
    return "white_noise"


def wiener_denoise(audio: np.ndarray) -> Tuple[np.ndarray, float]:
    """ref :197-215: scipy.signal.wiener with a window of 2 * int(0.1 * T) samples."""
    ns = int(0.1 * len(audio))
    den = wiener(audio, mysize=2 * ns)
    e0, e1 = np.mean(audio ** 2), np.mean(den ** 2)
    return den, (10 * np.log10(e1 / e0) if e1 > 0 else 0.0)


def estimate_t60(audio: np.ndarray, sr: int = SR) -> float:
    """ref :274-301.  The cumulative energy is non-decreasing, so `where(energy < threshold)[0][0]` is either index 0 or
    absent: the value is 0.0 or 0.1, never above the 0.5 s de-reverberation threshold (ref :340-349 never runs)."""
    p = int(np.argmax(np.abs(audio)))
    dec = audio[p:]
    if len(dec) < sr:
        return 0.1
    e = np.cumsum(dec ** 2)
    if e[-1] == 0:
        return 0.1
    idx = np.where(e < e[-1] * 0.001)[0]
    return min(idx[0] / sr, 2.0) if len(idx) > 0 else 0.1


def measure_lufs(audio: np.ndarray) -> float:
    """ref :364-371 (pyloudnorm absent)."""
    r = np.sqrt(np.mean(audio ** 2))
    return 20 * np.log10(r) - 70 if r > 0 else -60


def compress(audio: np.ndarray) -> Tuple[np.ndarray, float]:
    """ref :373-400."""
    r, peak = np.sqrt(np.mean(audio ** 2)), np.max(np.abs(audio))
    dr = 20 * np.log10(peak / r) if r > 0 else 0
    if dr > 40:
        thr, ratio = r * 2, min(4.0, dr / 40)
        out = audio.copy()
        m = np.abs(audio) > thr
        out[m] = np.sign(audio[m]) * (thr + (np.abs(audio[m]) - thr) / ratio)
        return out, ratio
    return audio.copy(), 1.0


def condition_audio(audio: np.ndarray, sr: int = SR) -> dict:
    """ref :503-584 `_process_single_sample`: conditioned clip (float32) + the 12 raw (un-projected) features."""
    audio = np.asarray(audio)
    x = audio.copy()
    hum = detect_hum(x, sr)
    for f in hum:
        b, a = iirnotch(f, 30, sr)
        x = filtfilt(b, a, x)
    hpf_on, cutoff = hpf_decision(x, sr)
    if hpf_on:
        b, a = butter(4, cutoff / (sr / 2), btype='high')
        x = filtfilt(b, a, x)
    else:
        cutoff = 0.0
    snr_before = snr_energy(x)
    ntype = noise_type(x, sr)
    if snr_before < 15:
        x, gain_db = wiener_denoise(x)
    else:
        gain_db = 0.0
    denoise_on = gain_db != 0.0
    snr_after = snr_energy(x)
    t60 = estimate_t60(x, sr)
    assert t60 <= 0.5                     # see estimate_t60: simple_dereverb is unreachable
    lufs = measure_lufs(x)
    comp, ratio = compress(x)
    adj = np.clip(-23.0 - lufs, -6.0, 6.0)          # keeps the dtype of the clip (float32 unless a filter ran), as the reference does
    out = comp * 10 ** (adj / 20)
    p0, p1 = np.max(np.abs(x)), np.max(np.abs(out))
    peak_db = 20 * np.log10(p1 / p0) if p0 > 0 else 0.0
    feats = np.array([float(len(hum) > 0), float(hpf_on), float(denoise_on), 0.0, snr_before / 50.0, snr_after / 50.0,
                      gain_db / 20.0, t60 / 2.0, (lufs + 60) / 60, adj / 20.0, peak_db / 20.0, ratio / 4.0], dtype=np.float32)
    return dict(audio=out.astype(np.float32), features=feats, hum_frequencies=hum, hpf_applied=hpf_on, hpf_cutoff=cutoff,
                denoise_applied=denoise_on, snr_before=snr_before, snr_after=snr_after, denoise_gain_db=gain_db,
                estimated_t60=t60, lufs_original=lufs, lufs_adjustment=float(adj), peak_reduction_db=peak_db,
                compression_ratio=ratio, noise_type=ntype)


def front_end(audio: np.ndarray, text: Optional[str] = None, detected_lang: Optional[str] = None, vad_method="librosa",
              use_quality_gates=True, use_audio_conditioning=True, pad_mode="constant") -> dict:
    """ref audio_encoder.py:65-87: quality gates -> zero the clip unless 'accept' -> conditioning."""
    x = np.asarray(audio, dtype=np.float32)
    q = None
    if use_quality_gates:
        q = quality_metrics(x, text, detected_lang, vad_method, pad_mode=pad_mode)
        if q["decision"] != "accept":
            x = np.zeros_like(x)
    c = None
    if use_audio_conditioning:
        c = condition_audio(x)
        x = c["audio"]
    return dict(audio=x, quality=q, conditioning=c)
