"""ctypes binding of libpmc_host.so (include/pmc_host.h): the MLMC / MC managers."""
from __future__ import annotations

import ctypes as C
import os
from typing import Callable, Optional, Sequence

import numpy as np

from . import capi

HOST_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libpmc_host.so")

NVAR = 9
Y2, Y, ABSY, Q2, Q, ABSQ, CC, Y3, Y4 = range(9)

REDUCE_FN = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_int, C.c_void_p)
CB_SAMPLE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_uint64, C.c_int, C.POINTER(C.c_double))
CB_EVAL = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double),
                      C.POINTER(C.c_double), C.c_int, C.c_int, C.POINTER(C.c_double))
CB_SOLVE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double),
                       C.POINTER(C.c_double))


class pmc_plugin_callbacks(C.Structure):
    _fields_ = [("user", C.c_void_p), ("sample", CB_SAMPLE), ("eval", CB_EVAL), ("solve_fwd", CB_SOLVE),
                ("xi_size", C.POINTER(C.c_int32)), ("sample_size", C.POINTER(C.c_int32)),
                ("ndofs", C.POINTER(C.c_int32))]


class pmc_mlmc_params(C.Structure):
    _fields_ = [("eps2", C.c_double), ("ratio", C.c_double), ("init_nsamples", C.c_int32),
                ("array_nsamples", C.POINTER(C.c_int32)), ("wall_time", C.c_int32), ("batch", C.c_int32),
                ("max_rounds", C.c_int32), ("log_file", C.c_char_p)]


_DPTR = C.POINTER(C.c_double)
_LPTR = C.POINTER(C.c_int64)


class pmc_mlmc_result(C.Structure):
    _fields_ = [("nlevels", C.c_int32)] + [(n, C.c_double) for n in
                ("estimate", "eps2", "actual_mse", "estimator_variance", "bias2", "alpha", "alpha_abs", "beta", "gamma")] + \
               [(n, _DPTR) for n in ("eY", "eABSY", "eQ", "eABSQ", "eC", "varY", "varQ", "consistency", "kurtosis", "VC",
                                     "cost")] + \
               [("sums", _DPTR), ("nsamples", _LPTR), ("nsamples_missing", _LPTR), ("level_seconds", _DPTR)]


_VP = C.c_void_p
HOST_SYMBOLS = {
    "pmc_mlmc_params_default": (None, [C.POINTER(pmc_mlmc_params)]),
    "pmc_mlmc_create": (C.c_int, [_VP, _VP, _VP, C.c_int, C.POINTER(pmc_mlmc_params), C.POINTER(_VP)]),
    "pmc_mlmc_create_callbacks": (C.c_int, [C.c_int, C.POINTER(pmc_plugin_callbacks), C.POINTER(pmc_mlmc_params),
                                            C.POINTER(_VP)]),
    "pmc_mlmc_add_lane": (C.c_int, [_VP, _VP, _VP, _VP]),
    "pmc_mlmc_destroy": (None, [_VP]),
    "pmc_mlmc_set_farm": (C.c_int, [_VP, C.c_int, C.c_int, REDUCE_FN, _VP]),
    "pmc_mlmc_run": (C.c_int, [_VP]),
    "pmc_mlmc_reset": (C.c_int, [_VP]),
    "pmc_mlmc_replay_log": (C.c_int, [_VP, C.c_char_p, C.POINTER(C.c_int64)]),
    "pmc_mlmc_init_run": (C.c_int, [_VP, C.POINTER(C.c_int32)]),
    "pmc_mlmc_result_get": (C.c_int, [_VP, C.POINTER(pmc_mlmc_result)]),
    "pmc_mlmc_show_me": (C.c_int, [_VP, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "pmc_mlmc_print_timers": (C.c_int, [_VP, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "pmc_mlmc_farm_times": (C.c_int, [_VP, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "pmc_mlmc_phase_times": (C.c_int, [_VP, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                       C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "pmc_bayes_likelihood": (C.c_int, [_VP, C.c_int, C.c_int, _VP, C.c_int, _DPTR, C.c_int, C.c_double, _DPTR, _DPTR, _DPTR,
                                       _DPTR]),
    "pmc_ratio_create": (C.c_int, [_VP, _VP, _VP, C.c_int, _DPTR, C.c_int, C.c_double, C.POINTER(pmc_mlmc_params),
                                   C.POINTER(_VP)]),
    "pmc_ratio_create_callbacks": (C.c_int, [C.c_int, C.POINTER(pmc_plugin_callbacks), _VP, C.POINTER(pmc_mlmc_params),
                                             C.POINTER(_VP)]),
    "pmc_ratio_destroy": (None, [_VP]),
    "pmc_ratio_set_farm": (C.c_int, [_VP, C.c_int, C.c_int, REDUCE_FN, _VP]),
    "pmc_ratio_set_splitting": (C.c_int, [_VP, C.c_int]),
    "pmc_ratio_run": (C.c_int, [_VP]),
    "pmc_ratio_init_run": (C.c_int, [_VP, C.POINTER(C.c_int32)]),
    "pmc_ratio_result_get": (C.c_int, [_VP, _VP]),
    "pmc_host_last_error": (C.c_char_p, []),
    "pmc_mortar_assemble": (C.c_int, [_VP, _VP, C.c_double, C.POINTER(_VP)]),
    "pmc_mortar_nnz": (C.c_int64, [_VP]),
    "pmc_mortar_get": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP]),
    "pmc_mortar_destroy": (None, [_VP]),
    "pmc_mortar_last_error": (C.c_char_p, []),
    "pmc_exp_w_regression": (C.c_double, [_DPTR, _DPTR, C.c_int, C.c_int]),
}

_hlib = None


def load_host_library():
    global _hlib
    if _hlib is not None:
        return _hlib
    capi.load_library()    # libpmc.so first (rpath also finds it)
    if not os.path.exists(HOST_LIB_PATH):
        raise capi.PmcError(-2, f"{HOST_LIB_PATH} not found - build it with `make`")
    lib = C.CDLL(HOST_LIB_PATH)
    for name, (res, args) in HOST_SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _hlib = lib
    return lib


def _hcheck(rc):
    if rc != 0:
        raise capi.PmcError(rc, load_host_library().pmc_host_last_error().decode("utf-8", "replace"))


def exp_w_regression(y, x, skip_n_last):
    y = np.ascontiguousarray(y, np.float64)
    x = np.ascontiguousarray(x, np.float64)
    return load_host_library().pmc_exp_w_regression(y.ctypes.data_as(_DPTR), x.ctypes.data_as(_DPTR), len(y), skip_n_last)


class pmc_mesh_view(C.Structure):
    _fields_ = [("dim", C.c_int32), ("nverts", C.c_int32), ("nelems", C.c_int32), ("verts_per_elem", C.c_int32),
                ("verts", C.c_void_p), ("elems", C.c_void_p)]


def mortar_gt(verts_a, elems_a, verts_b, elems_b, rel_tol=1e-12):
    """P0 x P0 mortar matrix G[i,j] = |A_i n B_j| between two non-matching meshes (pmc_mortar_assemble).  Returns
    (G as scipy CSR, |A_i|, |B_j|)."""
    import scipy.sparse as sp
    lib = load_host_library()
    keep = []

    def view(v, e):
        v = np.ascontiguousarray(v, np.float64)
        e = np.ascontiguousarray(e, np.int32)
        keep.extend((v, e))
        return pmc_mesh_view(v.shape[1], v.shape[0], e.shape[0], e.shape[1], v.ctypes.data, e.ctypes.data)
    va, vb = view(verts_a, elems_a), view(verts_b, elems_b)
    h = _VP()
    rc = lib.pmc_mortar_assemble(C.byref(va), C.byref(vb), float(rel_tol), C.byref(h))
    if rc != 0:
        raise capi.PmcError(rc, lib.pmc_mortar_last_error().decode("utf-8", "replace"))
    try:
        nnz = lib.pmc_mortar_nnz(h)
        rowptr = np.empty(va.nelems + 1, np.int32)
        colind = np.empty(nnz, np.int32)
        vals = np.empty(nnz)
        ma, mb = np.empty(va.nelems), np.empty(vb.nelems)
        lib.pmc_mortar_get(h, rowptr.ctypes.data, colind.ctypes.data, vals.ctypes.data, ma.ctypes.data, mb.ctypes.data)
    finally:
        lib.pmc_mortar_destroy(h)
    return sp.csr_matrix((vals, colind, rowptr), shape=(va.nelems, vb.nelems)), ma, mb


def bayes_likelihood(solver, level, k, G_obs, noise):
    """BayesianInverseProblem::ComputeLikelihoodAndQ / ComputeR for a batch: returns (likelihood, C, Q, R)."""
    lib = load_host_library()
    k = np.ascontiguousarray(np.atleast_2d(k), np.float64)
    G_obs = np.ascontiguousarray(G_obs, np.float64)
    nb = k.shape[0]
    like, Cc, Q, R = (np.empty(nb) for _ in range(4))
    _hcheck(lib.pmc_bayes_likelihood(solver.h, level, nb, k.ctypes.data, capi.PMC_MEM_HOST, G_obs.ctypes.data_as(_DPTR),
                                     len(G_obs), float(noise), like.ctypes.data_as(_DPTR), Cc.ctypes.data_as(_DPTR),
                                     Q.ctypes.data_as(_DPTR), R.ctypes.data_as(_DPTR)))
    return like, Cc, Q, R


CB_LIKE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double),
                      C.POINTER(C.c_double), C.POINTER(C.c_double))


class pmc_ratio_result(C.Structure):
    _fields_ = [("nlevels", C.c_int32)] + [(n, C.c_double) for n in
                ("R_estimate", "Z_estimate", "ratio_estimate", "eps2", "actual_mse", "estimator_variance",
                 "estimator_variance_R", "estimator_variance_Z", "bias2", "bias2_R", "bias2_Z", "alpha_R", "alpha_abs_R",
                 "beta_R", "alpha_Z", "alpha_abs_Z", "beta_Z", "gamma")] + \
               [(n, _DPTR) for n in ("eR", "varR", "eYR", "varYR", "eABS_YR", "eZ", "varZ", "eYZ", "varYZ", "eABS_YZ", "eC",
                                     "cost")] + \
               [("sums", _DPTR), ("nsamples", _LPTR), ("nsamples_missing", _LPTR)] + \
               [(n, C.c_double) for n in ("alpha", "alpha_abs", "beta")] + \
               [(n, _DPTR) for n in ("eRatio", "varRatio", "eYRatio", "varYRatio", "eABS_YRatio")]


RATIO_NVAR = 20


class RatioManager:
    """parelagmc::ML_BayesRatio_Manager (SL_BayesRatio_Manager for nlevels == 1); splitting=True gives
    ML_BayesRatio_Splitting_Manager / SL_BayesRatio_Splitting_Manager."""

    def __init__(self, nlevels, sampler=None, solver=None, G_obs=None, noise=None, callbacks=None, likelihood=None,
                 eps2=0.001, ratio=0.5, init_nsamples=10, wall_time=True, batch=256, max_rounds=1000, splitting=False):
        self.lib = load_host_library()
        self.nlevels = nlevels
        p = pmc_mlmc_params()
        self.lib.pmc_mlmc_params_default(C.byref(p))
        p.eps2, p.ratio, p.init_nsamples = eps2, ratio, init_nsamples
        p.wall_time, p.batch, p.max_rounds = (1 if wall_time else 0), batch, max_rounds
        self._keep = []
        h = _VP()
        if callbacks is not None:
            cb, keep = _make_callbacks(callbacks)
            self._keep += keep

            def _like(user, level, nbatch, k, like, R, Cp):
                try:
                    kk = np.ctypeslib.as_array(k, shape=(nbatch, int(callbacks["sample_size"][level])))
                    l, r, c = likelihood(level, kk)
                    np.ctypeslib.as_array(like, shape=(nbatch,))[...] = l
                    np.ctypeslib.as_array(R, shape=(nbatch,))[...] = r
                    np.ctypeslib.as_array(Cp, shape=(nbatch,))[...] = c
                    return 0
                except Exception:   # noqa: BLE001
                    import traceback
                    traceback.print_exc()
                    return 1
            fn = CB_LIKE(_like)
            self._keep.append(fn)
            _hcheck(self.lib.pmc_ratio_create_callbacks(nlevels, C.byref(cb), fn, C.byref(p), C.byref(h)))
        else:
            g = np.ascontiguousarray(G_obs, np.float64)
            self._keep += [sampler, solver, g]
            _hcheck(self.lib.pmc_ratio_create(sampler.ctx.h, sampler.h, solver.h, nlevels, g.ctypes.data_as(_DPTR), len(g),
                                              float(noise), C.byref(p), C.byref(h)))
        self.h = h
        if splitting:
            _hcheck(self.lib.pmc_ratio_set_splitting(self.h, 1))

    def set_farm(self, nranks, rank, reduce=None):
        if reduce is not None:
            def _red(buf, n, user):
                try:
                    reduce(np.ctypeslib.as_array(buf, shape=(n,)))
                    return 0
                except Exception:   # noqa: BLE001
                    return 1
            fn = REDUCE_FN(_red)
        else:
            fn = REDUCE_FN()
        self._keep.append(fn)
        _hcheck(self.lib.pmc_ratio_set_farm(self.h, nranks, rank, fn, None))

    def InitRun(self, nsamples):
        a = np.ascontiguousarray(nsamples, np.int32)
        _hcheck(self.lib.pmc_ratio_init_run(self.h, a.ctypes.data_as(C.POINTER(C.c_int32))))
        return self.result()

    def Run(self):
        _hcheck(self.lib.pmc_ratio_run(self.h))
        return self.result()

    def result(self):
        r = pmc_ratio_result()
        _hcheck(self.lib.pmc_ratio_result_get(self.h, C.byref(r)))
        nl = r.nlevels
        out = {n: getattr(r, n) for n, t in pmc_ratio_result._fields_ if t is C.c_double}
        for n in ("eR", "varR", "eYR", "varYR", "eABS_YR", "eZ", "varZ", "eYZ", "varYZ", "eABS_YZ", "eC", "cost", "eRatio",
                  "varRatio", "eYRatio", "varYRatio", "eABS_YRatio"):
            out[n] = np.ctypeslib.as_array(getattr(r, n), shape=(nl,)).copy()
        out["sums"] = np.ctypeslib.as_array(r.sums, shape=(nl, RATIO_NVAR)).copy()
        out["nsamples"] = np.ctypeslib.as_array(r.nsamples, shape=(nl,)).copy()
        out["missing"] = np.ctypeslib.as_array(r.nsamples_missing, shape=(nl,)).copy()
        return out

    def close(self):
        if getattr(self, "h", None):
            self.lib.pmc_ratio_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _make_callbacks(callbacks):
    """pmc_plugin_callbacks from a dict of Python callables (sample, eval, [solve]) + sizes; returns (struct, keepalive)."""
    cb = pmc_plugin_callbacks()
    xs = np.ascontiguousarray(callbacks["xi_size"], np.int32)
    ss = np.ascontiguousarray(callbacks["sample_size"], np.int32)
    nd = np.ascontiguousarray(callbacks["ndofs"], np.int32)
    f_sample, f_eval = callbacks["sample"], callbacks["eval"]

    def _sample(user, level, first_id, nbatch, xi):
        try:
            np.ctypeslib.as_array(xi, shape=(nbatch, int(xs[level])))[...] = f_sample(level, int(first_id), nbatch)
            return 0
        except Exception:   # noqa: BLE001
            import traceback
            traceback.print_exc()
            return 1

    def _eval(user, level, xi_level, nbatch, xi, s, init_s, init_level, use_init, emb):
        try:
            x = np.ctypeslib.as_array(xi, shape=(nbatch, int(xs[xi_level])))
            init = np.ctypeslib.as_array(init_s, shape=(nbatch, int(xs[init_level]))) if use_init else None
            sv, ev = f_eval(level, xi_level, x, init, init_level if use_init else None)
            np.ctypeslib.as_array(s, shape=(nbatch, int(ss[level])))[...] = sv
            if emb:
                np.ctypeslib.as_array(emb, shape=(nbatch, int(xs[level])))[...] = ev
            return 0
        except Exception:   # noqa: BLE001
            import traceback
            traceback.print_exc()
            return 1

    def _solve(user, level, nbatch, k, Qp, Cp):
        return 1     # the ratio managers never call SolveFwd directly

    cb.sample, cb.eval, cb.solve_fwd = CB_SAMPLE(_sample), CB_EVAL(_eval), CB_SOLVE(_solve)
    cb.xi_size = xs.ctypes.data_as(C.POINTER(C.c_int32))
    cb.sample_size = ss.ctypes.data_as(C.POINTER(C.c_int32))
    cb.ndofs = nd.ctypes.data_as(C.POINTER(C.c_int32))
    return cb, [xs, ss, nd, cb, cb.sample, cb.eval, cb.solve_fwd]


class MLMCManager:
    """parelagmc::MLMC_Manager (Run / InitRun / results); MC_Manager is the nlevels == 1 case.

    Either over device handles (sampler=capi.PDESampler, solver=capi.DarcySolver) or over Python
    plugin callbacks (`callbacks=dict(sample=, eval=, solve=, xi_size=, sample_size=, ndofs=)`)."""

    def __init__(self, nlevels, sampler=None, solver=None, callbacks=None, eps2=0.001, ratio=0.5, init_nsamples=10,
                 array_nsamples: Optional[Sequence[int]] = None, wall_time=True, batch=256, max_rounds=1000,
                 log_file: Optional[str] = None):
        self.lib = load_host_library()
        self.nlevels = nlevels
        p = pmc_mlmc_params()
        self.lib.pmc_mlmc_params_default(C.byref(p))
        p.eps2, p.ratio, p.init_nsamples = eps2, ratio, init_nsamples
        p.wall_time, p.batch, p.max_rounds = (1 if wall_time else 0), batch, max_rounds
        self._keep = []
        if array_nsamples is not None:
            a = np.ascontiguousarray(array_nsamples, np.int32)
            assert len(a) == nlevels
            self._keep.append(a)
            p.array_nsamples = a.ctypes.data_as(C.POINTER(C.c_int32))
        if log_file:
            p.log_file = log_file.encode()
        h = _VP()
        if callbacks is not None:
            cb = pmc_plugin_callbacks()
            xs = np.ascontiguousarray(callbacks["xi_size"], np.int32)
            ss = np.ascontiguousarray(callbacks["sample_size"], np.int32)
            nd = np.ascontiguousarray(callbacks["ndofs"], np.int32)
            self._keep += [xs, ss, nd]
            f_sample, f_eval, f_solve = callbacks["sample"], callbacks["eval"], callbacks["solve"]

            def _sample(user, level, first_id, nbatch, xi):
                try:
                    out = np.ctypeslib.as_array(xi, shape=(nbatch, int(xs[level])))
                    out[...] = f_sample(level, int(first_id), nbatch)
                    return 0
                except Exception:   # noqa: BLE001 - must not propagate through C
                    import traceback
                    traceback.print_exc()
                    return 1

            def _eval(user, level, xi_level, nbatch, xi, s, init_s, init_level, use_init, emb):
                try:
                    x = np.ctypeslib.as_array(xi, shape=(nbatch, int(xs[xi_level])))
                    init = np.ctypeslib.as_array(init_s, shape=(nbatch, int(xs[init_level]))) if use_init else None
                    sv, ev = f_eval(level, xi_level, x, init, init_level if use_init else None)
                    np.ctypeslib.as_array(s, shape=(nbatch, int(ss[level])))[...] = sv
                    if emb:
                        np.ctypeslib.as_array(emb, shape=(nbatch, int(xs[level])))[...] = ev
                    return 0
                except Exception:   # noqa: BLE001
                    import traceback
                    traceback.print_exc()
                    return 1

            def _solve(user, level, nbatch, k, Qp, Cp):
                try:
                    kk = np.ctypeslib.as_array(k, shape=(nbatch, int(ss[level])))
                    q, c = f_solve(level, kk)
                    np.ctypeslib.as_array(Qp, shape=(nbatch,))[...] = q
                    np.ctypeslib.as_array(Cp, shape=(nbatch,))[...] = c
                    return 0
                except Exception:   # noqa: BLE001
                    import traceback
                    traceback.print_exc()
                    return 1

            cb.sample, cb.eval, cb.solve_fwd = CB_SAMPLE(_sample), CB_EVAL(_eval), CB_SOLVE(_solve)
            cb.xi_size = xs.ctypes.data_as(C.POINTER(C.c_int32))
            cb.sample_size = ss.ctypes.data_as(C.POINTER(C.c_int32))
            cb.ndofs = nd.ctypes.data_as(C.POINTER(C.c_int32))
            self._keep += [cb, cb.sample, cb.eval, cb.solve_fwd]
            _hcheck(self.lib.pmc_mlmc_create_callbacks(nlevels, C.byref(cb), C.byref(p), C.byref(h)))
        else:
            assert sampler is not None and solver is not None
            self._keep += [sampler, solver]
            _hcheck(self.lib.pmc_mlmc_create(sampler.ctx.h, sampler.h, solver.h, nlevels, C.byref(p), C.byref(h)))
        self.h = h

    def add_lane(self, sampler, solver):
        """Another (context, sampler, solver) triple built from the same problem: one more HIP stream working on
        this rank's realizations concurrently."""
        self._keep += [sampler, solver]
        _hcheck(self.lib.pmc_mlmc_add_lane(self.h, sampler.ctx.h, sampler.h, solver.h))

    def set_farm(self, nranks: int, rank: int, reduce: Optional[Callable[[np.ndarray], None]] = None):
        """reduce(buf) must SUM-all-reduce the numpy buffer in place (e.g. torch.distributed);
        None with nranks > 1 uses the RCCL communicator of the device context."""
        if reduce is not None:
            def _red(buf, n, user):
                try:
                    reduce(np.ctypeslib.as_array(buf, shape=(n,)))
                    return 0
                except Exception:   # noqa: BLE001
                    import traceback
                    traceback.print_exc()
                    return 1
            fn = REDUCE_FN(_red)
        else:
            fn = REDUCE_FN()
        self._keep.append(fn)
        _hcheck(self.lib.pmc_mlmc_set_farm(self.h, nranks, rank, fn, None))

    def Run(self):
        _hcheck(self.lib.pmc_mlmc_run(self.h))
        return self.result()

    def ShowMe(self) -> str:
        """MLMC_Manager::ShowMe table as text (same labels and layout as the reference prints)."""
        need = C.c_size_t(0)
        _hcheck(self.lib.pmc_mlmc_show_me(self.h, None, 0, C.byref(need)))
        buf = C.create_string_buffer(need.value)
        _hcheck(self.lib.pmc_mlmc_show_me(self.h, buf, need.value, None))
        return buf.value.decode("utf-8", "replace")

    def PrintTimers(self) -> str:
        """TimeManager::Print of the per-realization path: "Sampler: Mult", "Darcy: Build Solver", "Darcy: Mult" per level."""
        need = C.c_size_t(0)
        _hcheck(self.lib.pmc_mlmc_print_timers(self.h, None, 0, C.byref(need)))
        buf = C.create_string_buffer(need.value)
        _hcheck(self.lib.pmc_mlmc_print_timers(self.h, buf, need.value, None))
        return buf.value.decode("utf-8", "replace")

    def phase_times(self, level):
        """dict of device milliseconds and realization counts of `level`, summed over the lanes of this rank"""
        a, b, c = C.c_double(0), C.c_double(0), C.c_double(0)
        n1, n2 = C.c_int64(0), C.c_int64(0)
        _hcheck(self.lib.pmc_mlmc_phase_times(self.h, level, C.byref(a), C.byref(b), C.byref(c), C.byref(n1), C.byref(n2)))
        return {"sampler_mult_ms": a.value, "darcy_build_ms": b.value, "darcy_mult_ms": c.value,
                "sampler_realizations": n1.value, "darcy_realizations": n2.value}

    def farm_times(self):
        """(milliseconds this rank spent in the farm's SUM all-reduce so far, number of reductions = InitRun rounds)"""
        ms, n = C.c_double(0), C.c_int64(0)
        _hcheck(self.lib.pmc_mlmc_farm_times(self.h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def Reset(self):
        _hcheck(self.lib.pmc_mlmc_reset(self.h))

    def ReplayLog(self, path: str) -> int:
        n = C.c_int64()
        _hcheck(self.lib.pmc_mlmc_replay_log(self.h, path.encode(), C.byref(n)))
        return n.value

    def InitRun(self, nsamples: Sequence[int]):
        a = np.ascontiguousarray(nsamples, np.int32)
        assert len(a) == self.nlevels
        _hcheck(self.lib.pmc_mlmc_init_run(self.h, a.ctypes.data_as(C.POINTER(C.c_int32))))
        return self.result()

    def result(self) -> dict:
        r = pmc_mlmc_result()
        _hcheck(self.lib.pmc_mlmc_result_get(self.h, C.byref(r)))
        nl = r.nlevels
        out = {k: getattr(r, k) for k in ("estimate", "eps2", "actual_mse", "estimator_variance", "bias2", "alpha",
                                          "alpha_abs", "beta", "gamma")}
        for k in ("eY", "eABSY", "eQ", "eABSQ", "eC", "varY", "varQ", "consistency", "kurtosis", "VC", "cost",
                  "level_seconds"):
            out[k] = np.ctypeslib.as_array(getattr(r, k), shape=(nl,)).copy()
        out["sums"] = np.ctypeslib.as_array(r.sums, shape=(nl, NVAR)).copy()
        out["nsamples"] = np.ctypeslib.as_array(r.nsamples, shape=(nl,)).copy()
        out["missing"] = np.ctypeslib.as_array(r.nsamples_missing, shape=(nl,)).copy()
        return out

    def close(self):
        if getattr(self, "h", None):
            self.lib.pmc_mlmc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
