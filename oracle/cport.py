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


def _cpu_flags():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return set(line.split(":", 1)[1].split())
    except OSError:
        pass
    return set()


_FLAGS = os.path.join(_HERE, "_build", "built_on_cpu_flags.txt")


def build(force=False):
    """The library is compiled -march=native where build() ran and travels to the GPU box as a built file: it is rebuilt
    there (gcc is in the image) when that host's CPU lacks a flag of the build host's, instead of dying on an illegal
    instruction."""
    here = _cpu_flags()
    if not force and os.path.exists(LIB) and os.path.exists(_FLAGS):
        with open(_FLAGS) as f:
            force = not set(f.read().split()) <= here
    elif not force and os.path.exists(LIB) and not os.path.exists(_FLAGS):
        force = True
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(os.path.join(_HERE, "c", "pmc_ref.c")):
        subprocess.run(["make", "-B", "-C", _HERE, "_build/libpmc_ref.so"], check=True, capture_output=True)
        with open(_FLAGS, "w") as f:
            f.write(" ".join(sorted(here)))
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


class _dlevel(C.Structure):
    _fields_ = [("n_u", C.c_int), ("n_p", C.c_int), ("Mpat", _csr), ("c_ptr", C.POINTER(C.c_int)),
                ("c_elem", C.POINTER(C.c_int)), ("c_val", C.POINTER(C.c_double)), ("B", _csr), ("Bt", _csr),
                ("ess", C.POINTER(C.c_ubyte)), ("ess_data", C.POINTER(C.c_double)), ("rhs", C.POINTER(C.c_double)),
                ("obs", C.POINTER(C.c_double)), ("Spat", _csr), ("parent", C.POINTER(C.c_int)), ("P", _csr), ("Pt", _csr)]


class DarcyCPort:
    """ctypes driver of pmc_ref_darcy_batch: the reference's per-sample Darcy work (assemble M(k), eliminate, REBUILD the
    preconditioner incl. the Schur hierarchy, MINRES, Q) restated in C; see oracle/c/pmc_ref.c for the line references."""

    def __init__(self, problem):
        self.lib = C.CDLL(build())
        self.lib.pmc_ref_darcy_batch.restype = C.c_int
        self.p = problem
        self._keep = []
        nl = len(problem.levels)
        self.levels = (_dlevel * nl)()
        spat = []
        for i, L in enumerate(problem.levels):
            keep = sp.diags((~L.ess_mask.astype(bool)).astype(np.float64))
            B = (L.B @ keep).tocsr()
            B.eliminate_zeros()
            B.sort_indices()
            if i == 0:
                S = (abs(B) @ abs(B).T).tocsr()
            else:                                   # Galerkin pattern of the finer level
                Pf = problem.levels[i - 1].P.tocsr()
                S = (Pf.T @ spat[i - 1] @ Pf).tocsr()
            S = (S + sp.identity(L.n_p)).tocsr()
            S.sort_indices()
            S.data[:] = 1.0
            spat.append(S)
        helper = CPort.__new__(CPort)
        helper._keep = self._keep
        for i, L in enumerate(problem.levels):
            keepm = (~L.ess_mask.astype(bool)).astype(np.float64)
            B = (L.B @ sp.diags(keepm)).tocsr()
            B.eliminate_zeros()
            B.sort_indices()
            Bt = B.T.tocsr()
            rhs = L.rhs.copy()
            rhs[L.n_u:] -= L.B @ (L.ess_data * L.ess_mask)          # k-independent part of the elimination
            P = L.P.tocsr() if L.P is not None else sp.csr_matrix((0, 0))
            parent = self._i32(P.indices) if L.P is not None else C.POINTER(C.c_int)()
            if L.P is not None:
                assert np.all(np.diff(P.indptr) == 1), "the C port handles injection-type prolongators"
            pat = L.M_pattern.tocsr()
            self.levels[i] = _dlevel(L.n_u, L.n_p, helper._csr(pat), self._i32(L.c_ptr), self._i32(L.c_elem),
                                     helper._f64(L.c_val), helper._csr(B), helper._csr(Bt), self._u8(L.ess_mask),
                                     helper._f64(L.ess_data), helper._f64(rhs), helper._f64(L.obs), helper._csr(spat[i]),
                                     parent, helper._csr(P), helper._csr(P.T.tocsr()))

    def _i32(self, a):
        a = np.ascontiguousarray(a, np.int32)
        self._keep.append(a)
        return a.ctypes.data_as(C.POINTER(C.c_int))

    def _u8(self, a):
        a = np.ascontiguousarray(a, np.uint8)
        self._keep.append(a)
        return a.ctypes.data_as(C.POINTER(C.c_ubyte))

    def solve(self, level, k, max_iter=300, rel_tol=1e-6, abs_tol=1e-12, nthreads=0, return_solution=False):
        k = np.ascontiguousarray(np.atleast_2d(k), np.float64)
        ns = k.shape[0]
        L = self.p.levels[level]
        Q = np.empty(ns)
        iters = np.zeros(ns, np.int32)
        sol = np.empty((ns, L.n_u + L.n_p)) if return_solution else None
        rc = self.lib.pmc_ref_darcy_batch(len(self.p.levels), self.levels, level, 1 if self.p.k_divides else 0, ns,
                                          k.ctypes.data_as(C.POINTER(C.c_double)), Q.ctypes.data_as(C.POINTER(C.c_double)),
                                          sol.ctypes.data_as(C.POINTER(C.c_double)) if return_solution else None, max_iter,
                                          C.c_double(rel_tol), C.c_double(abs_tol), nthreads,
                                          iters.ctypes.data_as(C.POINTER(C.c_int)))
        if rc != 0:
            raise RuntimeError(f"pmc_ref_darcy_batch failed ({rc})")
        return (Q, iters, sol) if return_solution else (Q, iters)


class _hsys(C.Structure):
    _fields_ = [("n_lambda", C.c_int), ("n_s", C.c_int), ("G", _csr), ("Gt", _csr), ("z", C.POINTER(C.c_double))]


class HybridCPort:
    """ctypes driver of pmc_ref_hybrid_batch: the reference's "Hybridization" solver restated on the CPU - PCG on
    H lambda = G f with one AMG V(1,1)-cycle (symmetric Gauss-Seidel), back-substitution s = z f - G^T lambda
    (/root/reference/examples/example_parameterlists/example_parameters.xml:200-212, src/PDESampler.cpp:383-389,451-480).
    BoomerAMG is not available: the hierarchy is smoothed aggregation built here (greedy aggregation in C, damped-Jacobi
    prolongator smoothing, Galerkin products).  problem: fe.HybridSamplerProblem (H, G, z_diag, w_diag, P per level)."""

    def __init__(self, problem, theta=0.08, min_rows=400, max_levels=12):
        self.lib = C.CDLL(build())
        self.lib.pmc_ref_hybrid_batch.restype = C.c_int
        self.lib.pmc_ref_aggregate.restype = C.c_int
        self.p = problem
        self._keep = []
        self._helper = CPort.__new__(CPort)
        self._helper._keep = self._keep
        self.theta, self.min_rows, self.max_levels = theta, min_rows, max_levels
        self._built = {}

    def _aggregate(self, A):
        A = A.tocsr()
        rp = np.ascontiguousarray(A.indptr, np.int32)
        ci = np.ascontiguousarray(A.indices, np.int32)
        v = np.ascontiguousarray(A.data, np.float64)
        agg = np.empty(A.shape[0], np.int32)
        na = self.lib.pmc_ref_aggregate(A.shape[0], rp.ctypes.data_as(C.POINTER(C.c_int)), ci.ctypes.data_as(C.POINTER(C.c_int)),
                                        v.ctypes.data_as(C.POINTER(C.c_double)), C.c_double(self.theta),
                                        agg.ctypes.data_as(C.POINTER(C.c_int)))
        if na <= 0:
            raise RuntimeError("pmc_ref_aggregate failed")
        return agg, na

    def hierarchy(self, level):
        """[(A_l, P_l)] smoothed-aggregation levels of H(level); P_l maps level l+1 -> l (None on the last)"""
        if level in self._built:
            return self._built[level]
        A = self.p.levels[level].H.tocsr().astype(np.float64)
        out = []
        while True:
            n = A.shape[0]
            if n <= self.min_rows or len(out) + 1 >= self.max_levels:
                out.append((A, None))
                break
            agg, na = self._aggregate(A)
            if na >= 0.8 * n:                 # coarsening stalled
                out.append((A, None))
                break
            T = sp.csr_matrix((np.ones(n), (np.arange(n), agg)), shape=(n, na))
            T = T @ sp.diags(1.0 / np.sqrt(np.asarray(T.power(2).sum(axis=0)).ravel()))     # orthonormal columns
            dinv = 1.0 / A.diagonal()
            # omega = 4 / (3 rho(D^-1 A)), rho by a few power iterations
            x = np.random.default_rng(0).standard_normal(n)
            rho = 1.0
            for _ in range(12):
                y = dinv * (A @ x)
                rho = float(np.linalg.norm(y) / np.linalg.norm(x))
                x = y / np.linalg.norm(y)
            P = (T - sp.diags((4.0 / 3.0) / (1.05 * rho) * dinv) @ (A @ T)).tocsr()
            P.sort_indices()
            out.append((A, P))
            A = (P.T @ A @ P).tocsr()
            A.sort_indices()
        mg = (_level * len(out))()
        h = self._helper
        empty = sp.csr_matrix((0, 0))
        for i, (Al, Pl) in enumerate(out):
            Pm = Pl if Pl is not None else empty
            mg[i] = _level(0, Al.shape[0], h._csr(empty), h._csr(empty), h._csr(empty), h._csr(Al), h._f64(np.zeros(1)),
                           h._csr(Pm), h._csr(Pm.T.tocsr()))
        L = self.p.levels[level]
        sys_ = _hsys(L.n_lambda, L.n_s, h._csr(L.G), h._csr(L.G.T.tocsr()), h._f64(L.z_diag))
        self._built[level] = (out, mg, sys_)
        return self._built[level]

    def operator_complexity(self, level):
        out = self.hierarchy(level)[0]
        return sum(A.nnz for A, _ in out) / out[0][0].nnz

    def rhs(self, level, xi_level, xi):
        """(nsamples, n_s(level)) f = -g W^{1/2} xi restricted with P^T (src/PDESampler.cpp:423-442)"""
        xi = np.atleast_2d(xi)
        r = -self.p.matern_g * xi * np.sqrt(self.p.levels[xi_level].w_diag)[None, :]
        for lvl in range(xi_level, level):
            r = (self.p.levels[lvl].P.T @ r.T).T
        return np.ascontiguousarray(r)

    def solve(self, level, f, max_iter=300, rel_tol=1e-6, abs_tol=1e-12, nthreads=0):
        out, mg, sys_ = self.hierarchy(level)
        f = np.ascontiguousarray(np.atleast_2d(f), np.float64)
        ns = f.shape[0]
        s = np.empty_like(f)
        iters = np.zeros(ns, np.int32)
        rc = self.lib.pmc_ref_hybrid_batch(len(out), mg, C.byref(sys_), ns, f.ctypes.data_as(C.POINTER(C.c_double)),
                                           s.ctypes.data_as(C.POINTER(C.c_double)), max_iter, C.c_double(rel_tol),
                                           C.c_double(abs_tol), nthreads, iters.ctypes.data_as(C.POINTER(C.c_int)))
        if rc != 0:
            raise RuntimeError(f"pmc_ref_hybrid_batch failed ({rc})")
        return s, iters

    def eval(self, level, xi_level, xi, **kw):
        s, iters = self.solve(level, self.rhs(level, xi_level, xi), **kw)
        return (np.exp(s) if self.p.lognormal else s), iters
