"""ORACLE (test infrastructure): SPDE Matérn sampler, direct-solve restatement.

Follows /root/reference/src/PDESampler.cpp:342-535 (Eval), :177-334 (operator
definition), src/EmbeddedPDESampler.cpp:552-556 (gather), src/L2ProjectionPDESampler.cpp:
738-750 (projection), src/PDESampler_Legacy.cpp:172-176,253-331 (reduced system).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla


class SamplerOracle:
    def __init__(self, problem):
        self.p = problem
        self._lu = {}

    # [M Bt; B -alpha W]   (PDESampler.cpp:279-284)
    def block_operator(self, level):
        L = self.p.levels[level]
        W = sp.diags(L.w_diag)
        return sp.bmat([[L.M, L.B.T], [L.B, -self.p.alpha * W]], format="csc")

    def _solver(self, level):
        if level not in self._lu:
            self._lu[level] = spla.splu(self.block_operator(level))
        return self._lu[level]

    def rhs_s(self, level, xi_level, xi):
        """rhs_s = -g W^{1/2} xi on xi_level, restricted with Ps^T down to `level`
        (PDESampler.cpp:423-438)."""
        assert xi_level <= level
        L = self.p.levels[xi_level]
        assert xi.shape[0] == L.n_s
        r = -self.p.matern_g * xi * np.sqrt(L.w_diag)
        while xi_level < level:
            r = self.p.levels[xi_level].P.T @ r
            xi_level += 1
        return r

    def eval_gaussian(self, level, xi_level, xi):
        """Gaussian field on the sampler's (embedded) mesh = what Eval stores in embed_s
        (PDESampler.cpp:526-527)."""
        L = self.p.levels[level]
        rhs = np.concatenate([np.zeros(L.n_u), self.rhs_s(level, xi_level, xi)])
        sol = self._solver(level).solve(rhs)
        return sol[L.n_u:]

    def eval(self, level, xi_level, xi, projection=None):
        """Returns (s, embed_s).  projection: None (PDESampler), ("gather", idx) for the
        Embedded sampler, ("l2", Gt, inv_w) for the L2-projection sampler."""
        g = self.eval_gaussian(level, xi_level, xi)
        if projection is None:
            s = g.copy()
        elif projection[0] == "gather":
            s = g[projection[1]]
        elif projection[0] == "l2":
            s = (projection[1] @ g) * projection[2]
        else:
            raise ValueError(projection[0])
        if self.p.lognormal:
            s = np.exp(s)
        return s, g

    def eval_reduced(self, level, xi_level, xi):
        """Legacy algebra: (M + a^-1 D^T W D) u = -(g/a) D^T r~,  s = a^-1 D u + (g/a) W^-1 r~,
        with r~ = W^{1/2} xi restricted (PDESampler_Legacy.cpp:172-176,262-323).  Here
        D = W^-1 B, so D^T W D = B^T W^-1 B and D^T r = B^T W^-1 r."""
        L = self.p.levels[level]
        a = self.p.alpha
        r = -self.rhs_s(level, xi_level, xi) / self.p.matern_g      # = restricted W^{1/2} xi
        winv = 1.0 / L.w_diag
        K = (L.M + (1.0 / a) * (L.B.T @ sp.diags(winv) @ L.B)).tocsc()
        u = spla.splu(K).solve(-(self.p.matern_g / a) * (L.B.T @ (winv * r)))
        return (1.0 / a) * winv * (L.B @ u) + (self.p.matern_g / a) * winv * r

    def prolongate(self, coarse_level, fine_level, s):
        """Warm-start prolongation of a Gaussian field (PDESampler.cpp:498-507)."""
        lvl = coarse_level
        while lvl > fine_level:
            s = self.p.levels[lvl - 1].P @ s
            lvl -= 1
        return s
