"""ORACLE (test infrastructure): mixed Darcy forward solve, direct-solve restatement.

Follows /root/reference/src/DarcySolver.cpp:416-437 (SolveFwd), :472-520 (assemble:
M(k), block matrix, EliminateRowCol on the essential u-dofs), :562-649 (solve) and
src/Utilities.cpp:411-420 (dot).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla


class DarcyOracle:
    def __init__(self, problem):
        self.p = problem

    def mass(self, level, k):
        """M(k) = sum_e c(k_e) M_e with c = 1/k (k_divides) or k  (DarcySolver.cpp:479)."""
        L = self.p.levels[level]
        c = (1.0 / k) if self.p.k_divides else k
        contrib = c[L.c_elem] * L.c_val
        data = np.add.reduceat(contrib, L.c_ptr[:-1])
        return sp.csr_matrix((data, L.M_pattern.indices, L.M_pattern.indptr), shape=L.M_pattern.shape)

    def assemble(self, level, k):
        """Returns (A, rhs_bc) after EliminateRowCol(ess_dofs, ess_data, rhs_bc) (:487-498)."""
        L = self.p.levels[level]
        M = self.mass(level, np.asarray(k, dtype=np.float64))
        A = sp.bmat([[M, L.B.T], [L.B, None]], format="csr")
        n = L.n_u + L.n_p
        ess = np.zeros(n, dtype=bool)
        ess[:L.n_u] = L.ess_mask.astype(bool)
        d = np.zeros(n)
        d[:L.n_u] = L.ess_data
        rhs = L.rhs - A @ (d * ess)
        rhs[ess] = d[ess]
        keep = sp.diags((~ess).astype(np.float64))
        A = (keep @ A @ keep + sp.diags(ess.astype(np.float64))).tocsc()
        return A, rhs

    def solve_fwd(self, level, k, return_solution=False):
        """Q = <obs, sol>, C = number of (global true) dofs (:427-429)."""
        L = self.p.levels[level]
        A, rhs = self.assemble(level, k)
        sol = spla.splu(A).solve(rhs)
        Q = float(L.obs @ sol)
        C = float(L.n_u + L.n_p)
        if return_solution:
            return Q, C, sol
        return Q, C
