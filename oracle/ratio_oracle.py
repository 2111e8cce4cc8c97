"""ORACLE (test infrastructure): multilevel ratio estimator statistics, plain-Python restatement of
/root/reference/src/ML_BayesRatio_Manager.hpp:315-433 (InitRun accumulators) and :560-728 (computeNSamplesMSE), and of
the "divide, then subtract" variant /root/reference/src/ML_BayesRatio_Splitting_Manager.hpp:297-432, :595-737
(`accumulate_ratio`, `compute_splitting`)."""
from __future__ import annotations

import math

import numpy as np

from .mlmc_oracle import exp_w_regression

YZ2, YZ, ABS_YZ, Z2, Z, ABS_Z, YR2, YR, ABS_YR, R2, R, ABS_R = range(12)
YRATIO2, YRATIO, ABS_YRATIO, RATIO2, RATIO, ABS_RATIO = range(12, 18)
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


def accumulate_ratio(sums, level, q, y):
    """Ratio columns of the splitting manager: q = r/z, y = q - r_c/z_c (y = q on the coarsest level)."""
    sums[level, RATIO] += q
    sums[level, ABS_RATIO] += abs(q)
    sums[level, RATIO2] += q * q
    sums[level, YRATIO] += y
    sums[level, ABS_YRATIO] += abs(y)
    sums[level, YRATIO2] += y * y


def compute_splitting(sums, nsamples, M, eps2, ratio, cost=None):
    nl = sums.shape[0]
    ns = np.asarray(nsamples, float)
    ex = sums / ns[:, None]
    f = ns / (ns - 1.0)
    out = dict(eRatio=ex[:, RATIO], eYRatio=ex[:, YRATIO], eABS_YRatio=ex[:, ABS_YRATIO], eC=ex[:, C])
    out["varRatio"] = (ex[:, RATIO2] - ex[:, RATIO] ** 2) * f
    out["varYRatio"] = (ex[:, YRATIO2] - ex[:, YRATIO] ** 2) * f
    costv = out["eC"] if cost is None else np.asarray(cost, float)
    out["alpha"] = exp_w_regression(out["eYRatio"], M, 1)
    out["alpha_abs"] = exp_w_regression(out["eABS_YRatio"], M, 1)
    out["beta"] = exp_w_regression(out["varYRatio"], M, 1)
    out["bias2"] = _bias2(nl, M, out["eABS_YRatio"], out["alpha_abs"])
    if eps2 < 0:
        eps2 = out["bias2"] / (1.0 - ratio)
    out["eps2"] = eps2
    out["estimator_variance"] = float(np.sum(out["varYRatio"] / ns))
    prop = float(np.sum(np.sqrt(out["varYRatio"] * costv))) / (ratio * eps2)
    out["missing"] = [max(math.ceil(prop * math.sqrt(out["varYRatio"][i] / costv[i]) - ns[i]), 0) for i in range(nl)]
    out["ratio_estimate"] = float(out["eYRatio"].sum())
    return out
