"""ORACLE (test infrastructure): multilevel ratio estimator statistics, plain-Python restatement of
/root/reference/src/ML_BayesRatio_Manager.hpp:315-433 (InitRun accumulators) and :560-728 (computeNSamplesMSE)."""
from __future__ import annotations

import math

import numpy as np

from .mlmc_oracle import exp_w_regression

YZ2, YZ, ABS_YZ, Z2, Z, ABS_Z, YR2, YR, ABS_YR, R2, R, ABS_R = range(12)
C = 18
NVAR = 20


def accumulate(sums, level, r, y_r, z, y_z, c_tot):
    sums[level, R] += r
    sums[level, ABS_R] += abs(r)
    sums[level, R2] += r * r
    sums[level, YR] += y_r
    sums[level, ABS_YR] += abs(y_r)
    sums[level, YR2] += y_r * y_r
    sums[level, Z] += z
    sums[level, ABS_Z] += abs(z)
    sums[level, Z2] += z * z
    sums[level, YZ] += y_z
    sums[level, ABS_YZ] += abs(y_z)
    sums[level, YZ2] += y_z * y_z
    sums[level, C] += c_tot


def _bias2(nl, M, eabs, a):
    if nl == 1:
        return 0.0
    m = M[0] / M[1]
    if nl > 3:
        return max(m ** (2 * a) * eabs[1] ** 2, eabs[0] ** 2) / (m ** (-2 * a) - 1.0) ** 2
    if nl == 3:
        return eabs[0] ** 2 / (m ** (-a) - 1.0) ** 2
    return eabs[0] ** 2


def compute(sums, nsamples, M, eps2, ratio, cost=None):
    nl = sums.shape[0]
    ns = np.asarray(nsamples, float)
    ex = sums / ns[:, None]
    f = ns / (ns - 1.0)
    out = dict(eR=ex[:, R], eYR=ex[:, YR], eABS_YR=ex[:, ABS_YR], eZ=ex[:, Z], eYZ=ex[:, YZ], eABS_YZ=ex[:, ABS_YZ],
               eC=ex[:, C])
    out["varR"] = (ex[:, R2] - ex[:, R] ** 2) * f
    out["varYR"] = (ex[:, YR2] - ex[:, YR] ** 2) * f
    out["varZ"] = (ex[:, Z2] - ex[:, Z] ** 2) * f
    out["varYZ"] = (ex[:, YZ2] - ex[:, YZ] ** 2) * f
    costv = out["eC"] if cost is None else np.asarray(cost, float)
    aR = exp_w_regression(out["eABS_YR"], M, 1)
    aZ = exp_w_regression(out["eABS_YZ"], M, 1)
    out["bias2_R"], out["bias2_Z"] = _bias2(nl, M, out["eABS_YR"], aR), _bias2(nl, M, out["eABS_YZ"], aZ)
    out["bias2"] = max(out["bias2_R"], out["bias2_Z"])
    if eps2 < 0:
        eps2 = out["bias2"] / (1.0 - ratio)
    out["eps2"] = eps2
    out["var_R"] = float(np.sum(out["varYR"] / ns))
    out["var_Z"] = float(np.sum(out["varYZ"] / ns))
    out["estimator_variance"] = max(out["var_R"], out["var_Z"])
    pR = float(np.sum(np.sqrt(out["varYR"] * costv))) / (ratio * eps2)
    pZ = float(np.sum(np.sqrt(out["varYZ"] * costv))) / (ratio * eps2)
    miss = []
    for i in range(nl):
        mr = math.ceil(pR * math.sqrt(out["varYR"][i] / costv[i]) - ns[i])
        mz = math.ceil(pZ * math.sqrt(out["varYZ"][i] / costv[i]) - ns[i])
        miss.append(max(mr, mz, 0))
    out["missing"] = miss
    out["R_estimate"], out["Z_estimate"] = float(out["eYR"].sum()), float(out["eYZ"].sum())
    out["ratio_estimate"] = out["R_estimate"] / out["Z_estimate"]
    return out
