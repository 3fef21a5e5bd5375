"""Audio loading and the two waveform augmentations of ref src/data/preprocess.py (host side).

This is the data feed in front of the hot path ("next"): it runs on the CPU
exactly where the reference runs it.  The reference uses torchaudio, which is not installed in the
build image; `resample` below restates torchaudio.functional.resample's windowed-sinc polyphase
algorithm (defaults: lowpass_filter_width 6, rolloff 0.99, Hann window) so that `speed_perturb`
keeps the reference's semantics — a 16k -> 16k*f -> 16k round trip that leaves the clip length
unchanged (preprocess.py:50-62) — without the dependency.
"""
import math

import torch


def resample(waveform: torch.Tensor, orig_freq: int, new_freq: int, lowpass_filter_width: int = 6,
             rolloff: float = 0.99) -> torch.Tensor:
    """waveform [1,T] -> [1, ceil(T*new/orig)]."""
    if orig_freq == new_freq:
        return waveform
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
    t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None, None] / new + idx
    t = (t * base).clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    kernels = torch.where(t == 0, torch.tensor(1.0, dtype=torch.float64), t.sin() / t) * window * (base / orig)
    kernels = kernels.to(waveform.dtype)
    length = waveform.shape[-1]
    x = torch.nn.functional.pad(waveform[:, None], (width, width + orig))
    y = torch.nn.functional.conv1d(x, kernels, stride=orig)
    y = y.transpose(1, 2).reshape(waveform.shape[0], -1)
    return y[..., : math.ceil(new * length / orig)]


def load_audio(path, sr=16000, max_length=30):
    """mono float32 [T] at 16 kHz, <= 30 s, >= 0.5 s (zero-padded); 1 s of zeros when loading fails
    (ref preprocess.py:5-47)."""
    if not path.startswith('datasets/'):
        path = f"datasets/{path}"
    try:
        from scipy.io import wavfile
        orig_sr, data = wavfile.read(path)
        w = torch.as_tensor(data)
        if w.dtype in (torch.int16, torch.int32, torch.uint8):
            scale = {torch.int16: 32768.0, torch.int32: 2147483648.0, torch.uint8: 128.0}[w.dtype]
            w = (w.float() - (128.0 if w.dtype == torch.uint8 else 0.0)) / scale
        w = w.float()
        w = w.t() if w.dim() == 2 else w[None]
        if w.shape[0] > 1:
            w = w.mean(dim=0, keepdim=True)
        if orig_sr != sr:
            w = resample(w, orig_sr, sr)
        w = w[:, : sr * max_length]
        min_samples = int(sr * 0.5)
        if w.shape[1] < min_samples:
            w = torch.nn.functional.pad(w, (0, min_samples - w.shape[1]))
        return w.squeeze(0).float()
    except Exception as e:  # noqa: BLE001 - the reference swallows every loading error the same way
        print(f"Error loading {path}: {e}")
        return torch.zeros(sr, dtype=torch.float32)


def clip_length(path, sr=16000, max_length=30):
    """Number of samples `load_audio(path)` will return, from the WAV header alone (no decoding): what the length-bucketing
    sampler sorts by.  Unreadable files count as the 1 s of zeros load_audio substitutes."""
    import wave
    if not path.startswith('datasets/'):
        path = f"datasets/{path}"
    try:
        with wave.open(path, "rb") as f:
            frames, orig_sr = f.getnframes(), f.getframerate()
        n = frames if orig_sr == sr else math.ceil(frames * sr / orig_sr)
        return max(int(sr * 0.5), min(n, sr * max_length))
    except Exception:  # noqa: BLE001
        return sr


def speed_perturb(waveform: torch.Tensor, factor: float) -> torch.Tensor:
    if abs(factor - 1.0) < 1e-3:
        return waveform
    mid = int(16000 * factor)
    y = resample(waveform[None], 16000, mid)
    return resample(y, mid, 16000).squeeze(0)


def add_noise_snr(waveform: torch.Tensor, snr_db: float, generator=None) -> torch.Tensor:
    signal_power = waveform.pow(2).mean().clamp(min=1e-12)
    noise_power = (signal_power / (10 ** (snr_db / 10))).item()
    noise = torch.randn(waveform.shape, generator=generator, dtype=waveform.dtype) * math.sqrt(noise_power)
    return (waveform + noise).clamp(min=-1.0, max=1.0)
