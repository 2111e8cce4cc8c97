"""ORACLE (test infrastructure): counter-based normal variates, numpy restatement.

The reference draws xi sequentially from ``trng::yarn5`` + ``trng::normal_dist<double>``
(/root/reference/src/NormalDistributionSampler.cpp:31-37), i.e. uniform -> inverse normal
CDF, and splits streams by leap-frogging (:21-24).  TRNG 4.19 is not vendored and not
installed, so the yarn5 stream cannot be reproduced; the parity boundary is therefore xi
itself (the plugin API takes xi as an explicit argument, src/MLSampler.hpp:38-41).

The GPU build draws xi with Philox4x32-10 (Salmon et al., SC'11 - published algorithm,
known-answer vectors in tests/test_rng.py) keyed by (seed), countered by
(element pair, sample id, stream), followed by the same inverse-CDF construction the
reference uses; Wichura's AS241 PPND16 is the inverse CDF.  This module is the bit-level
restatement of that device code.
"""
from __future__ import annotations

import numpy as np

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = np.uint32(0x9E3779B9)
_W1 = np.uint32(0xBB67AE85)
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(ctr, key):
    """ctr: (..., 4) uint32, key: (..., 2) uint32 (broadcastable). Returns (..., 4) uint32."""
    c = [np.asarray(ctr[..., i], dtype=np.uint64) for i in range(4)]
    k0 = np.asarray(key[..., 0], dtype=np.uint64)
    k1 = np.asarray(key[..., 1], dtype=np.uint64)
    for _ in range(10):
        p0 = _M0 * c[0]
        p1 = _M1 * c[2]
        hi0, lo0 = p0 >> np.uint64(32), p0 & _MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & _MASK
        c = [(hi1 ^ c[1] ^ k0) & _MASK, lo1, (hi0 ^ c[3] ^ k1) & _MASK, lo0]
        k0 = (k0 + np.uint64(_W0)) & _MASK
        k1 = (k1 + np.uint64(_W1)) & _MASK
    return np.stack(c, axis=-1).astype(np.uint32)


def u01_open(hi, lo):
    """52-bit uniform in the OPEN interval (0,1) from two 32-bit words: (m + 1/2) / 2^52 is
    exactly representable for every 52-bit m, so neither 0 nor 1 can occur."""
    hi = np.asarray(hi, dtype=np.uint64)
    lo = np.asarray(lo, dtype=np.uint64)
    m = (hi >> np.uint64(6)) * np.uint64(1 << 26) + (lo >> np.uint64(6))     # 26 + 26 = 52 bits
    return (m.astype(np.float64) + 0.5) * (1.0 / 4503599627370496.0)


_A = [3.3871328727963666080e0, 1.3314166789178437745e+2, 1.9715909503065514427e+3, 1.3731693765509461125e+4,
      4.5921953931549871457e+4, 6.7265770927008700853e+4, 3.3430575583588128105e+4, 2.5090809287301226727e+3]
_B = [1.0, 4.2313330701600911252e+1, 6.8718700749205790830e+2, 5.3941960214247511077e+3, 2.1213794301586595867e+4,
      3.9307895800092710610e+4, 2.8729085735721942674e+4, 5.2264952788528545610e+3]
_C = [1.42343711074968357734e0, 4.63033784615654529590e0, 5.76949722146069140550e0, 3.64784832476320460504e0,
      1.27045825245236838258e0, 2.41780725177450611770e-1, 2.27238449892691845833e-2, 7.74545014278341407640e-4]
_D = [1.0, 2.05319162663775882187e0, 1.67638483018380384940e0, 6.89767334985100004550e-1, 1.48103976427480074590e-1,
      1.51986665636164571966e-2, 5.47593808499534494600e-4, 1.05075007164441684324e-9]
_E = [6.65790464350110377720e0, 5.46378491116411436990e0, 1.78482653991729133580e0, 2.96560571828504891230e-1,
      2.65321895265761230930e-2, 1.24266094738807843860e-3, 2.71155556874348757815e-5, 2.01033439929228813265e-7]
_F = [1.0, 5.99832206555887937690e-1, 1.36929880922735805310e-1, 1.48753612908506148525e-2, 7.86869131145613259100e-4,
      1.84631831751005468180e-5, 1.42151175831644588870e-7, 2.04426310338993978564e-15]


def _horner(coef, x):
    acc = np.full_like(x, coef[7])
    for c in coef[6::-1]:
        acc = acc * x + c
    return acc


def inv_normal_cdf(p):
    """Wichura AS241 PPND16."""
    p = np.asarray(p, dtype=np.float64)
    q = p - 0.5
    out = np.empty_like(p)
    central = np.abs(q) <= 0.425
    r = 0.180625 - q[central] * q[central]
    out[central] = q[central] * _horner(_A, r) / _horner(_B, r)
    t = ~central
    qt = q[t]
    r = np.where(qt < 0, p[t], 1.0 - p[t])
    r = np.sqrt(-np.log(r))
    near = r <= 5.0
    val = np.empty_like(r)
    rn = r[near] - 1.6
    val[near] = _horner(_C, rn) / _horner(_D, rn)
    rf = r[~near] - 5.0
    val[~near] = _horner(_E, rf) / _horner(_F, rf)
    out[t] = np.where(qt < 0, -val, val)
    return out


def normal_fill(n, seed, sample_id, stream=0, mean=0.0, sigma=1.0):
    """xi[0:n] for one realization: element i uses Philox counter (i//2, sample_lo, sample_hi,
    stream), key (seed_lo, seed_hi); words (0,1) -> element 2j, words (2,3) -> element 2j+1."""
    npair = (n + 1) // 2
    ctr = np.zeros((npair, 4), dtype=np.uint32)
    ctr[:, 0] = np.arange(npair, dtype=np.uint32)
    ctr[:, 1] = np.uint32(sample_id & 0xFFFFFFFF)
    ctr[:, 2] = np.uint32((sample_id >> 32) & 0xFFFFFFFF)
    ctr[:, 3] = np.uint32(stream)
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint32)
    w = philox4x32_10(ctr, key[None, :])
    u = np.stack([u01_open(w[:, 0], w[:, 1]), u01_open(w[:, 2], w[:, 3])], axis=1).reshape(-1)[:n]
    return mean + sigma * inv_normal_cdf(u)
