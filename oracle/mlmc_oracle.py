"""ORACLE (test infrastructure): MLMC / MC estimator statistics, plain-Python restatement.

Follows /root/reference/src/MLMC_Manager.cpp:103-179 (InitRun accumulators), :300-401
(computeNSamplesMSE), :181-214 (Run / grain rule) and src/Utilities.cpp:257-283
(expWRegression).  Level 0 = finest.
"""
from __future__ import annotations

import math

import numpy as np

# enum of sums, /root/reference/src/MLMC_Manager.hpp:65
Y2, Y, ABSY, Q2, Q, ABSQ, C, Y3, Y4 = range(9)
NVAR = 9


def exp_w_regression(y, x, skip_n_last):
    """Weighted (1/2^i) least-squares slope of log|y_i/y_{i+1}| on log(x_i/x_{i+1})."""
    n = len(y) - 1 - skip_n_last
    if n < 1:
        return 0.0
    num = den = 0.0
    for i in range(n):
        ldy = math.log(abs(y[i] / y[i + 1]))
        ldx = math.log(x[i] / x[i + 1])
        w = 0.5 ** i
        num += w * ldy * ldx
        den += w * ldx * ldx
    return num / den


def accumulate(sums, level, y, q, c):
    """One realization's update of the 9 running sums (MLMC_Manager.cpp:123-131,158-168)."""
    sums[level, Y3] += y * y * y
    sums[level, Y4] += y * y * y * y
    sums[level, Y2] += y * y
    sums[level, Y] += y
    sums[level, ABSY] += abs(y)
    sums[level, Q2] += q * q
    sums[level, Q] += q
    sums[level, ABSQ] += abs(q)
    sums[level, C] += c


def compute_nsamples_mse(sums, nsamples, M, eps2, ratio, cost=None):
    """Returns a dict with every quantity computeNSamplesMSE derives.  `cost` = per-level
    cost vector (wall time per sample); None = use eC (wallTime == false branch)."""
    nl = sums.shape[0]
    ns = np.asarray(nsamples, dtype=np.float64)
    ex = sums / ns[:, None]
    eY, eABSY, eQ, eABSQ, eC = ex[:, Y].copy(), ex[:, ABSY].copy(), ex[:, Q].copy(), ex[:, ABSQ].copy(), ex[:, C].copy()
    varY, varQ, kurt = ex[:, Y2].copy(), ex[:, Q2].copy(), ex[:, Y4].copy()
    for l in range(nl):
        kurt[l] /= varY[l] * varY[l]
        varY[l] -= eY[l] * eY[l]
        varY[l] *= ns[l] / (ns[l] - 1.0)
        varQ[l] -= eQ[l] * eQ[l]
        varQ[l] *= ns[l] / (ns[l] - 1.0)
    cons = np.zeros(nl)
    for l in range(nl - 1):
        cons[l] = abs(eQ[l] - eQ[l + 1] + eY[l]) / (
            3.0 * (math.sqrt(varQ[l]) + math.sqrt(varQ[l + 1]) + math.sqrt(varY[l])))
    alpha = exp_w_regression(eY, M, 1)
    alpha_abs = exp_w_regression(eABSY, M, 1)
    beta = exp_w_regression(varY, M, 1)
    if nl == 1:
        bias2 = 0.0
    else:
        m = M[0] / M[1]
        if nl > 3:
            bias2 = max(m ** (2.0 * alpha_abs) * eABSY[1] ** 2, eABSY[0] ** 2) / (m ** (-2.0 * alpha_abs) - 1.0) ** 2
        elif nl == 3:
            bias2 = eABSY[0] ** 2 / (m ** (-alpha_abs) - 1.0) ** 2
        else:
            bias2 = eABSY[0] ** 2
    if eps2 < 0:
        eps2 = bias2 / (1.0 - ratio)
    est_var = float(np.sum(varY / ns))
    costv = eC if cost is None else np.asarray(cost, dtype=np.float64)
    gamma = exp_w_regression(costv, M, 0)
    prop = float(np.sum(np.sqrt(varY * costv))) / (ratio * eps2)
    missing = np.zeros(nl, dtype=np.int64)
    VC = np.zeros(nl)
    for l in range(nl):
        miss = prop * math.sqrt(varY[l] / costv[l]) - ns[l]
        missing[l] = max(int(math.ceil(miss)), 0)
        VC[l] = varY[l] * costv[l]
    return dict(eY=eY, eABSY=eABSY, eQ=eQ, eABSQ=eABSQ, eC=eC, varY=varY, varQ=varQ, kurtosis=kurt,
                consistency=cons, alpha=alpha, alphaABS=alpha_abs, beta=beta, gamma=gamma,
                bias2=bias2, eps2=eps2, estimator_variance=est_var, actualMSE=bias2 + est_var,
                missing=missing, VC=VC, estimate=float(np.sum(eY)))
