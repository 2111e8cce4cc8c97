"""ORACLE / CPU BASELINE (test infrastructure): ctypes driver of oracle/c/pmc_ref.c - the plain-C
restatement of the reference's MINRES + block-diagonal (sym-GS x3 | V-cycle) solve, farmed over
host cores with OpenMP.  Used by tests (iterative cross-check) and by bench.py's cpu_baseline leg."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np
import scipy.sparse as sp

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "_build", "libpmc_ref.so")


class _csr(C.Structure):
    _fields_ = [("nrows", C.c_int), ("ncols", C.c_int), ("rp", C.POINTER(C.c_int)), ("ci", C.POINTER(C.c_int)),
                ("v", C.POINTER(C.c_double))]


class _level(C.Structure):
    _fields_ = [("n_u", C.c_int), ("n_s", C.c_int), ("M", _csr), ("B", _csr), ("Bt", _csr), ("S", _csr),
                ("aw", C.POINTER(C.c_double)), ("P", _csr), ("Pt", _csr)]


def build(force=False):
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(os.path.join(_HERE, "c", "pmc_ref.c")):
        subprocess.run(["make", "-C", _HERE, "_build/libpmc_ref.so"], check=True, capture_output=True)
    return LIB


class CPort:
    def __init__(self, problem):
        self.lib = C.CDLL(build())
        self.lib.pmc_ref_solve_batch.restype = C.c_int
        self.lib.pmc_ref_max_threads.restype = C.c_int
        self.p = problem
        self._keep = []
        nl = len(problem.levels)
        self.levels = (_level * nl)()
        for i, L in enumerate(problem.levels):
            dM = L.M.diagonal()
            aw = problem.alpha * L.w_diag
            S = (sp.diags(aw) + L.B @ sp.diags(1.0 / dM) @ L.B.T).tocsr()
            S.sort_indices()
            Bt = L.B.T.tocsr()
            P = L.P.tocsr() if L.P is not None else sp.csr_matrix((0, 0))
            Pt = P.T.tocsr()
            self.levels[i] = _level(L.n_u, L.n_s, self._csr(L.M), self._csr(L.B), self._csr(Bt), self._csr(S),
                                    self._f64(aw), self._csr(P), self._csr(Pt))

    def _f64(self, a):
        a = np.ascontiguousarray(a, np.float64)
        self._keep.append(a)
        return a.ctypes.data_as(C.POINTER(C.c_double))

    def _csr(self, m):
        m = m.tocsr()
        rp = np.ascontiguousarray(m.indptr, np.int32)
        ci = np.ascontiguousarray(m.indices, np.int32)
        v = np.ascontiguousarray(m.data, np.float64)
        self._keep += [rp, ci, v]
        return _csr(m.shape[0], m.shape[1], rp.ctypes.data_as(C.POINTER(C.c_int)), ci.ctypes.data_as(C.POINTER(C.c_int)),
                    v.ctypes.data_as(C.POINTER(C.c_double)))

    def max_threads(self):
        return self.lib.pmc_ref_max_threads()

    def rhs(self, level, xi_level, xi):
        """(nsamples, n_u+n_s) right-hand sides [0; -g W^{1/2} xi restricted] (PDESampler.cpp:423-442)."""
        xi = np.atleast_2d(xi)
        L = self.p.levels[xi_level]
        r = -self.p.matern_g * xi * np.sqrt(L.w_diag)[None, :]
        lvl = xi_level
        while lvl < level:
            r = (self.p.levels[lvl].P.T @ r.T).T
            lvl += 1
        out = np.zeros((xi.shape[0], self.p.levels[level].n_u + self.p.levels[level].n_s))
        out[:, self.p.levels[level].n_u:] = r
        return out

    def solve(self, level, rhs, max_iter=300, rel_tol=1e-6, abs_tol=1e-12, nthreads=0):
        rhs = np.ascontiguousarray(rhs, np.float64)
        ns, n = rhs.shape
        sol = np.empty_like(rhs)
        iters = np.zeros(ns, np.int32)
        rc = self.lib.pmc_ref_solve_batch(len(self.p.levels), self.levels, level, ns,
                                          rhs.ctypes.data_as(C.POINTER(C.c_double)),
                                          sol.ctypes.data_as(C.POINTER(C.c_double)), max_iter, C.c_double(rel_tol),
                                          C.c_double(abs_tol), nthreads, iters.ctypes.data_as(C.POINTER(C.c_int)))
        if rc != 0:
            raise RuntimeError(f"pmc_ref_solve_batch failed ({rc})")
        return sol, iters

    def eval(self, level, xi_level, xi, **kw):
        sol, iters = self.solve(level, self.rhs(level, xi_level, xi), **kw)
        s = sol[:, self.p.levels[level].n_u:]
        return (np.exp(s) if self.p.lognormal else s), iters
