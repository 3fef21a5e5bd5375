"""The two waveform augmentations of ref src/data/preprocess.py on the device, for whole batches
(SURVEY section 8f item 1): `--augment` no longer resamples 16 clips one by one on the host.

`resample` / `speed_perturb` reproduce torchaudio.functional.resample (the host restatement in
`preprocess.py` is the parity reference in tests/test_gpu_augment.py); `add_noise_snr` draws its
noise from the library's counter-based generator, so it matches the reference in distribution
(signal power, SNR, clamp), not sample by sample.
"""
import torch

from .. import _lib as L


def resample(wave: torch.Tensor, orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """wave [B, T] fp32 on the device -> [B, ceil(T * new / orig)]."""
    assert wave.is_cuda and wave.dtype == torch.float32 and wave.dim() == 2
    wave = wave.contiguous()
    B, T = wave.shape
    Lout = L.lib.ser_resample_out_len(T, int(orig_freq), int(new_freq))
    out = torch.empty(B, Lout, dtype=torch.float32, device=wave.device)
    L.check(L.lib.ser_resample(wave.data_ptr(), B, T, int(orig_freq), int(new_freq), int(lowpass_filter_width), float(rolloff),
                               out.data_ptr(), L.stream_ptr()), "ser_resample")
    return out


def speed_perturb(wave: torch.Tensor, factor: float) -> torch.Tensor:
    """ref preprocess.py:50-62: 16000 -> int(16000 f) -> 16000; the clip length is unchanged (band-limiting only)."""
    if abs(factor - 1.0) < 1e-3:
        return wave
    mid = int(16000 * factor)
    return resample(resample(wave, 16000, mid), mid, 16000)


def add_noise_snr(wave: torch.Tensor, snr_db, seed: int) -> torch.Tensor:
    """ref preprocess.py:65-73 for a batch: snr_db scalar or [B]."""
    assert wave.is_cuda and wave.dtype == torch.float32 and wave.dim() == 2
    wave = wave.contiguous()
    B, T = wave.shape
    snr = torch.as_tensor(snr_db, dtype=torch.float32, device=wave.device).expand(B).contiguous()
    sigma = torch.empty(B, dtype=torch.float32, device=wave.device)
    out = torch.empty_like(wave)
    L.check(L.lib.ser_add_noise_snr(wave.data_ptr(), B, T, snr.data_ptr(), int(seed) & (2 ** 64 - 1), sigma.data_ptr(),
                                    out.data_ptr(), L.stream_ptr()), "ser_add_noise_snr")
    return out
