"""Device-side data feed (csrc/augment.hip) against the resampler ORACLE (oracle/resample_oracle.py: the defining
windowed-sinc sum evaluated in float64, independent of the package's own host code; parity with torchaudio itself is unpinned,
see there), against the package's host data feed as a second witness, and the definition of add_noise_snr
(ref src/data/preprocess.py:50-73)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def A():
    import ser_amd  # noqa: F401
    from ser_amd.data import gpu_augment, preprocess
    assert torch.cuda.is_available()
    return gpu_augment, preprocess


@pytest.mark.parametrize("orig,new", [(16000, 17123), (17123, 16000), (16000, 14400), (44100, 16000), (8000, 16000), (16000, 16000)])
def test_resample_matches_the_oracle(A, orig, new):
    G, P = A
    g = torch.Generator().manual_seed(orig + new)
    x = 0.3 * torch.randn(3, 2937, generator=g)
    from oracle import resample_oracle as R
    got = G.resample(x.cuda(), orig, new).cpu()
    for b in range(x.shape[0]):
        want = R.resample_direct(x[b].numpy(), orig, new)
        assert got[b].shape == want.shape
        np.testing.assert_allclose(got[b].numpy(), want, atol=3e-6, rtol=0)
    np.testing.assert_allclose(got.numpy(), P.resample(x, orig, new).numpy(), atol=3e-6, rtol=0)      # host data feed: same values


@pytest.mark.parametrize("factor", [0.9, 1.0, 1.07, 1.1])
def test_speed_perturb_round_trip_keeps_length(A, factor):
    G, P = A
    g = torch.Generator().manual_seed(7)
    x = 0.2 * torch.randn(2, 16000, generator=g)
    from oracle import resample_oracle as R
    got = G.speed_perturb(x.cuda(), factor).cpu()
    want = np.stack([R.speed_perturb(w.numpy(), factor) for w in x])
    assert tuple(got.shape) == want.shape == tuple(x.shape)
    np.testing.assert_allclose(got.numpy(), want, atol=5e-6, rtol=0)


def test_add_noise_snr_statistics_clamp_and_determinism(A):
    G, _ = A
    g = torch.Generator().manual_seed(3)
    x = (0.1 * torch.randn(4, 64000, generator=g)).cuda()
    x[3] *= 12.0                                     # this clip will hit the clamp
    snr = torch.tensor([20.0, 10.0, 0.0, 20.0])
    y1 = G.add_noise_snr(x, snr, seed=123)
    y2 = G.add_noise_snr(x, snr, seed=123)
    y3 = G.add_noise_snr(x, snr, seed=124)
    assert torch.equal(y1, y2) and not torch.equal(y1, y3)
    assert float(y1.abs().max()) <= 1.0
    for b in range(3):
        n = (y1[b] - x[b]).double()
        want = float(x[b].double().pow(2).mean()) / 10 ** (float(snr[b]) / 10)
        assert abs(float(n.pow(2).mean()) / want - 1.0) < 0.03, "noise power must match the requested SNR"
        assert abs(float(n.mean())) < 4 * math.sqrt(want / n.numel())
        # normality: excess kurtosis of 64000 samples
        k = float((n ** 4).mean() / (n ** 2).mean() ** 2)
        assert abs(k - 3.0) < 0.15
    # independent streams per clip
    c = torch.corrcoef(torch.stack([(y1[0] - x[0]), (y1[1] - x[1])]))[0, 1]
    assert abs(float(c)) < 0.02
