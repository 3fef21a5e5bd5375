"""Quality gates and audio conditioning of the reference's default `AudioEncoder()` on the device.

`FrontEndQualityGates` (ref src/models/quality_gates.py:413-567) and `AudioConditioningModule`
(ref src/models/audio_conditioning.py:443-584) keep the reference's learnable sub-modules and state_dict keys
(`quality_projection.{0,3}.*`, `conditioning_projection.{0,3}.*`); their signal processing runs as batched HIP kernels
(`csrc/frontend.hip` through `ser_quality_gates` / `ser_audio_conditioning`) on all clips of equal length at once, with no
device -> host copy (the reference moves every clip to numpy and back, quality_gates.py:465, audio_conditioning.py:477).

What is not the reference's code path, and why:
  * `vad_method="webrtc"` (the reference default) needs the compiled webrtcvad GMM, which is not built: the energy VAD
    (`vad_method="librosa"`, ref :111-137) is used instead with a one-time warning — the same substitution the reference
    announces at import when webrtcvad is missing (ref :11-16), although its constructor then raises (ref :61-70).
    `strict_vad=True` raises that ValueError instead.
  * language identification is a text operation: `language_features()` reproduces the reference's entropy table
    (ref :252-301) from a language code.  The code comes from `langdetect.detect` when that package is importable, from
    `language_detector` (any callable text -> code) when given, else the reference's "unavailable" branch.
  * noisereduce / pyloudnorm are optional in the reference and absent here: its Wiener / RMS-loudness branches run.
"""
import math
import warnings
from typing import Callable, List, NamedTuple, Optional, Sequence

import torch
import torch.nn as nn

from .. import _ops as O

LID_LANGUAGES = ('en', 'es', 'fr', 'de', 'it', 'pt', 'ru', 'ja', 'ko', 'zh')      # ref quality_gates.py:269
DECISIONS = ('reject', 'uncertain', 'accept')
NOISE_TYPES = ('unknown', 'low_frequency', 'high_frequency', 'mid_frequency', 'white_noise')

try:                                                   # ref quality_gates.py:18-23
    from langdetect import detect as _langdetect
except ImportError:
    _langdetect = None


def _entropy(probs):
    return -sum(p * math.log(p + 1e-10) for p in probs)


_P_LISTED = [0.7 / 1.15] + [0.05 / 1.15] * 9
LID_LISTED = (_entropy(_P_LISTED), 0.7 / 1.15)         # detected language is one of the ten: (1.529, 0.609)
LID_OTHER = (_entropy([0.1] * 10), 0.1)                # any other language: uniform table (2.303, 0.1)
LID_UNAVAILABLE = (1.5, 0.0)                           # no detector, blank text or a detector error (ref :259-260, :299-301)
LID_NO_TEXT = (1.0, 0.0)                               # no transcript / detection disabled (ref :514-517)


def language_features(texts: Optional[Sequence[Optional[str]]], n: int, detector: Optional[Callable[[str], str]] = None,
                      enabled: bool = True) -> torch.Tensor:
    """[n, 2] (language entropy, dominant-language confidence) per clip, as LanguageIdentifier.identify_language
    returns them (ref quality_gates.py:252-301) behind the `if text and enable_language_detection` of ref :514-517."""
    detector = detector or _langdetect
    rows = []
    for i in range(n):
        text = texts[i] if texts is not None and i < len(texts) else None
        if not (text and enabled):
            rows.append(LID_NO_TEXT)
        elif detector is None or not text.strip():
            rows.append(LID_UNAVAILABLE)
        else:
            try:
                rows.append(LID_LISTED if detector(text) in LID_LANGUAGES else LID_OTHER)
            except Exception:                           # ref :299-301
                rows.append(LID_UNAVAILABLE)
    return torch.tensor(rows, dtype=torch.float32)


class QualityBatch(NamedTuple):
    """Device tensors for a batch of clips.  `features` are the 8 inputs of quality_projection (ref :544-553); `metrics`
    columns: speech_prob, snr_db, clipping_percent, spectral_naturalness, music_prob, laughter_prob, quality_score,
    decision; `decision` int32: 0 reject, 1 uncertain, 2 accept."""
    features: torch.Tensor
    metrics: torch.Tensor
    decision: torch.Tensor

    def recommendations(self) -> List[str]:
        return [DECISIONS[int(d)] for d in self.decision.cpu()]


class ConditioningBatch(NamedTuple):
    """`features`: the 12 inputs of conditioning_projection (ref audio_conditioning.py:562-575); `meta` columns: hpf_cutoff,
    hum50, hum60, snr_before, snr_after, denoise_gain_db, estimated_t60, lufs_original, lufs_adjustment,
    peak_reduction_db, compression_ratio, noise type index into NOISE_TYPES."""
    features: torch.Tensor
    meta: torch.Tensor


class FrontEndQualityGates(nn.Module):
    def __init__(self, sample_rate: int = 16000, vad_method: str = "webrtc", enable_language_detection: bool = True,
                 strict_vad: bool = False, pad_mode: str = "constant", language_detector=None):
        super().__init__()
        if vad_method not in ("webrtc", "librosa") or (vad_method == "webrtc" and strict_vad):
            raise ValueError(f"VAD method '{vad_method}' not available")                     # ref :69-70
        if vad_method == "webrtc":
            warnings.warn("webrtcvad not available. Using librosa-based VAD.")              # the text of ref :16
        self.sample_rate, self.vad_method = sample_rate, vad_method
        self.enable_language_detection = enable_language_detection
        self.pad_mode, self.language_detector = pad_mode, language_detector
        self.quality_projection = nn.Sequential(nn.Linear(8, 32), nn.ReLU(), nn.Dropout(0.1), nn.Linear(32, 8))

    def forward(self, audio: torch.Tensor, text=None):
        """audio [B, T] (or [T]) on the device, text: list of transcripts (or one / None) ->
        (processed audio: rejected clips zeroed (ref :556-558), QualityBatch, accept mask [B] bool)."""
        single = audio.dim() == 1
        wave = (audio[None] if single else audio).to(torch.float32)
        texts = [text] * wave.shape[0] if (text is None or isinstance(text, str)) else list(text)
        lid = language_features(texts, wave.shape[0], self.language_detector, self.enable_language_detection)
        raw, met, dec = O.quality_gates(wave, lid.to(wave.device, non_blocking=True), self.pad_mode, self.sample_rate)
        processed = wave * (dec != 0).to(wave.dtype)[:, None]
        accept = dec == 2
        q = QualityBatch(raw, met, dec)
        return (processed[0] if single else processed), q, accept


class AudioConditioningModule(nn.Module):
    def __init__(self, sample_rate: int = 16000):
        super().__init__()
        self.sample_rate = sample_rate
        self.conditioning_projection = nn.Sequential(nn.Linear(12, 32), nn.ReLU(), nn.Dropout(0.1), nn.Linear(32, 12))

    def forward(self, audio: torch.Tensor, decision: Optional[torch.Tensor] = None):
        """audio [B, T] (or [T]) -> (conditioned audio, ConditioningBatch).  `decision` (int32 [B], from the quality
        gates): clips not marked accept are conditioned as silence, which is what AudioEncoder.forward feeds the
        reference's module for them (ref audio_encoder.py:74-83)."""
        single = audio.dim() == 1
        wave = (audio[None] if single else audio).to(torch.float32)
        out, raw, meta = O.audio_conditioning(wave, decision, self.sample_rate)
        return (out[0] if single else out), ConditioningBatch(raw, meta)


def create_quality_gates(sample_rate: int = 16000, vad_method: str = "webrtc", enable_language_detection: bool = True):
    return FrontEndQualityGates(sample_rate, vad_method, enable_language_detection)        # ref :617-626


def create_audio_conditioning(sample_rate: int = 16000):
    return AudioConditioningModule(sample_rate)                                            # ref :638-640
