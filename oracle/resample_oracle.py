"""Oracle for the waveform resampler of the data feed — TEST INFRASTRUCTURE ONLY.

Only tests/ may import this module; the product paths (data/preprocess.py on the host, csrc/augment.hip on the device)
never do.

What is restated: `torchaudio.functional.resample` with its defaults (sinc_interp_hann, lowpass_filter_width 6, rolloff
0.99), which the reference calls at src/data/preprocess.py:25-28 (load_audio) and :56-61 (speed_perturb: 16000 ->
int(16000 f) -> 16000).  torchaudio is a third-party dependency (requirements.txt: `torchaudio>=0.12.0`) that is NOT
installed in the build image, so this restatement follows its published algorithm and is **parity unpinned** against
torchaudio itself.  To keep the product from being compared with itself, two independent formulations are given and
tested against each other (tests/test_oracle_resample.py):

  * `resample_direct`  - the defining sum, evaluated sample by sample in float64:
        y[j] = sum_m x[m] * (base / orig) * sinc(pi * base * d) * cos^2(pi * base * d / (2 * W)),   d = m / orig - j / new,
        over the m with |base * d| < W;   orig, new reduced by their gcd, base = min(orig, new) * rolloff, W = filter width
  * `resample_table`   - torchaudio's implementation form: a [new, 1, 2 width + orig] kernel table in the waveform's dtype,
        applied as a strided convolution over the zero-padded clip (what its `_get_sinc_resample_kernel` /
        `_apply_sinc_resample_kernel` do).
"""
import math

import numpy as np


def _reduced(orig_freq, new_freq):
    g = math.gcd(int(orig_freq), int(new_freq))
    return int(orig_freq) // g, int(new_freq) // g


def out_len(T, orig_freq, new_freq):
    orig, new = _reduced(orig_freq, new_freq)
    return int(math.ceil(new * T / orig))


def resample_direct(x, orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99):
    """x [T] -> [ceil(T new / orig)], float64 throughout."""
    x = np.asarray(x, dtype=np.float64)
    if int(orig_freq) == int(new_freq):
        return x.copy()
    orig, new = _reduced(orig_freq, new_freq)
    base = min(orig, new) * rolloff
    W = float(lowpass_filter_width)
    T = x.shape[0]
    n_out = out_len(T, orig_freq, new_freq)
    y = np.zeros(n_out, dtype=np.float64)
    half = W / base                                   # support in units of input-sample time (1 / orig): |d| < W / base
    for j in range(n_out):
        c = j * orig / new                            # centre in input samples
        lo = max(0, int(math.floor(c - half * orig)) - 1)
        hi = min(T - 1, int(math.ceil(c + half * orig)) + 1)
        if hi < lo:
            continue
        m = np.arange(lo, hi + 1, dtype=np.float64)
        d = m / orig - j / new
        t = base * d
        inside = np.abs(t) < W
        win = np.cos(t * math.pi / W / 2.0) ** 2
        sinc = np.where(t == 0.0, 1.0, np.sin(math.pi * t) / np.where(t == 0.0, 1.0, math.pi * t))
        y[j] = np.sum(x[lo:hi + 1] * np.where(inside, sinc * win, 0.0)) * (base / orig)
    return y


def resample_table(x, orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99, dtype=np.float32):
    """torchaudio's form: kernel table in `dtype`, strided correlation over the padded clip.  x [T] -> [ceil(T new / orig)]."""
    x = np.asarray(x, dtype=dtype)
    if int(orig_freq) == int(new_freq):
        return x.copy()
    orig, new = _reduced(orig_freq, new_freq)
    base = min(orig, new) * rolloff
    width = int(math.ceil(lowpass_filter_width * orig / base))
    idx = np.arange(-width, width + orig, dtype=np.float64)[None, :] / orig
    t = (np.arange(0, -new, -1, dtype=np.float64)[:, None] / new + idx) * base
    t = np.clip(t, -lowpass_filter_width, lowpass_filter_width)
    window = np.cos(t * math.pi / lowpass_filter_width / 2.0) ** 2
    tp = t * math.pi
    kern = (np.where(tp == 0.0, 1.0, np.sin(tp) / np.where(tp == 0.0, 1.0, tp)) * window * (base / orig)).astype(dtype)   # [new, 2 width + orig]
    T = x.shape[0]
    xp = np.concatenate([np.zeros(width, dtype), x, np.zeros(width + orig, dtype)])
    frames = (xp.shape[0] - kern.shape[1]) // orig + 1
    win = np.lib.stride_tricks.sliding_window_view(xp, kern.shape[1])[::orig][:frames]          # [frames, taps]
    y = (win.astype(dtype) @ kern.T.astype(dtype)).reshape(-1)                                 # frame-major, phase-minor
    return y[: out_len(T, orig_freq, new_freq)]


def speed_perturb(x, factor, **kw):
    """ref src/data/preprocess.py:50-62: 16000 -> int(16000 f) -> 16000 round trip; the clip keeps its length."""
    if abs(factor - 1.0) < 1e-3:
        return np.asarray(x, dtype=np.float64).copy()
    mid = int(16000 * factor)
    return resample_direct(resample_direct(x, 16000, mid, **kw), mid, 16000, **kw)
