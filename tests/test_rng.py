"""Counter-based normal generator restatement (oracle/rng_oracle.py)."""
import numpy as np
import scipy.special as ss

from oracle.rng_oracle import inv_normal_cdf, normal_fill, philox4x32_10, u01_open


def _kat(c, k):
    return [int(x) for x in philox4x32_10(np.array([c], np.uint32), np.array([k], np.uint32))[0]]


def test_philox_known_answers():
    # Random123 kat_vectors for philox4x32-10 (Salmon et al., SC'11)
    assert _kat([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert _kat([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert _kat([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_uniform_is_open_interval():
    assert 0.0 < u01_open(np.uint32(0), np.uint32(0)) < 1e-15
    assert np.isfinite(inv_normal_cdf(u01_open(np.array([0, 0xffffffff], np.uint32), np.array([0, 0xffffffff], np.uint32)))).all()
    assert 1.0 - 1e-15 < u01_open(np.uint32(0xffffffff), np.uint32(0xffffffff)) < 1.0


def test_inverse_cdf_matches_scipy():
    p = np.concatenate([np.linspace(1e-300, 1e-10, 500), np.linspace(1e-10, 1 - 1e-10, 100001),
                        10.0 ** -np.arange(1.0, 300.0)])
    ref = ss.ndtri(p)
    assert np.max(np.abs(inv_normal_cdf(p) - ref) / np.maximum(1.0, np.abs(ref))) < 5e-15


def test_moments_and_stream_independence():
    x = normal_fill(400001, seed=7, sample_id=3)
    assert abs(x.mean()) < 4 / np.sqrt(len(x)) and abs(x.var() - 1) < 0.01
    assert abs((x ** 3).mean()) < 0.03 and abs((x ** 4).mean() - 3) < 0.06
    y = normal_fill(400001, seed=7, sample_id=4)
    z = normal_fill(400001, seed=7, sample_id=3, stream=1)
    assert abs(np.corrcoef(x, y)[0, 1]) < 0.01 and abs(np.corrcoef(x, z)[0, 1]) < 0.01
    assert np.array_equal(normal_fill(11, 7, 3), x[:11])      # prefix property (odd n)
    w = normal_fill(1000, seed=7, sample_id=3, mean=2.0, sigma=3.0)
    assert np.allclose(w, 2.0 + 3.0 * x[:1000])
