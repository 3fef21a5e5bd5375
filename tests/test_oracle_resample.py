"""CPU-only: the resampler oracle's two independent formulations agree (the defining float64 sum vs torchaudio's kernel-table
form), and the product's host data feed (data/preprocess.py, what load_audio / speed_perturb run on the CPU as the reference
does) matches them.  torchaudio itself is not installed here: parity with it is unpinned (oracle/resample_oracle.py)."""
import numpy as np
import pytest
import torch

from oracle import resample_oracle as R


# (rates whose reduced ratio is small: the table form of 16000 -> 17123 is a 17123 x 16014 matrix, which is what the
# reference's speed_perturb builds per clip; the direct sum covers those ratios in the tests below)
@pytest.mark.parametrize("orig,new", [(16000, 17600), (17600, 16000), (16000, 14400), (44100, 16000), (8000, 16000), (22050, 16000)])
def test_direct_sum_equals_kernel_table_form(orig, new):
    rng = np.random.RandomState(orig % 1000 + new % 1000)
    x = 0.3 * rng.randn(1500)
    d = R.resample_direct(x, orig, new)
    t64 = R.resample_table(x, orig, new, dtype=np.float64)
    t32 = R.resample_table(x, orig, new, dtype=np.float32)
    assert d.shape == t64.shape == t32.shape == (R.out_len(1500, orig, new),)
    np.testing.assert_allclose(t64, d, atol=1e-12, rtol=0)
    np.testing.assert_allclose(t32, d, atol=2e-6, rtol=0)


def test_identity_rate_and_length_rule():
    x = np.arange(10.0)
    assert np.array_equal(R.resample_direct(x, 16000, 16000), x)
    assert R.out_len(16000, 16000, 17123) == 17123 and R.out_len(2937, 44100, 16000) == int(np.ceil(2937 * 160 / 441))


@pytest.mark.parametrize("orig,new", [(16000, 17123), (44100, 16000), (8000, 16000), (16000, 14400)])
def test_host_data_feed_matches_the_oracle(orig, new):
    import ser_amd  # noqa: F401
    from ser_amd.data import preprocess as P
    g = torch.Generator().manual_seed(orig + new)
    x = 0.3 * torch.randn(2, 1200, generator=g)
    got = P.resample(x, orig, new).numpy()
    for b in range(2):
        np.testing.assert_allclose(got[b], R.resample_direct(x[b].numpy(), orig, new), atol=3e-6, rtol=0)


@pytest.mark.parametrize("factor", [0.9, 1.07])
def test_host_speed_perturb_matches_the_oracle(factor):
    import ser_amd  # noqa: F401
    from ser_amd.data import preprocess as P
    g = torch.Generator().manual_seed(5)
    x = 0.2 * torch.randn(4000, generator=g)
    got = P.speed_perturb(x, factor).numpy()
    want = R.speed_perturb(x.numpy(), factor)
    assert got.shape == want.shape == (4000,)
    np.testing.assert_allclose(got, want, atol=5e-6, rtol=0)
