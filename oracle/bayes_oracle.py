"""ORACLE (test infrastructure): Bayesian observation operator and likelihood, restated.

Follows /root/reference/src/BayesianInverseProblem.cpp:178-218: G_i = <g_obs_i, p> / sum(g_obs_i) with p the
pressure block of the Darcy solution, likelihood = exp(-|G - G_obs|^2 / (2 noise)), R = Q * likelihood; the
observation functionals are volume indicators around observation points on the fine level
(src/DarcySolver.cpp:321-358 style) restricted with P^T to coarser levels.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp


def observation_functionals(hierarchy, points, eps):
    """Per level a (nobs, n_p) CSR matrix; row i = g_obs_i."""
    s0 = hierarchy.spaces[0]
    cen = s0.mesh.verts[s0.mesh.elems].mean(axis=1)
    rows = []
    for x in np.atleast_2d(points):
        inside = np.linalg.norm(cen - x[None, :], axis=1) < eps
        if not inside.any():
            inside[np.argmin(np.linalg.norm(cen - x[None, :], axis=1))] = True
        rows.append(sp.csr_matrix(np.where(inside, s0.vol, 0.0)[None, :]))
    G = sp.vstack(rows).tocsr()
    out = [G]
    for P in hierarchy.P:
        G = (G @ P).tocsr()          # (P^T g)^T
        out.append(G)
    return out


def compute_G(darcy_oracle, Gobs, level, k):
    Q, C, sol = darcy_oracle.solve_fwd(level, k, return_solution=True)
    p = sol[darcy_oracle.p.levels[level].n_u:]
    g = Gobs[level]
    return (g @ p) / np.asarray(g.sum(axis=1)).ravel(), C, Q


def likelihood(G, G_obs, noise):
    return float(np.exp(-np.sum((G - G_obs) ** 2) / (2.0 * noise)))
