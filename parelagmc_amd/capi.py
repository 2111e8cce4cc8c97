"""ctypes binding of libpmc.so (include/pmc.h) - the reference-side stub a ParELAGMC maintainer
would write in C++ is shown in INTEGRATION.md; this is the same binding for the Python test
harness and launcher.  There is NO CPU fallback: if the library cannot be loaded, or no GPU is
visible when a context is created, the call raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libpmc.so")

PMC_MEM_HOST, PMC_MEM_DEVICE = 0, 1
PMC_PROJ_NONE, PMC_PROJ_GATHER, PMC_PROJ_L2 = 0, 1, 2


class PmcError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libpmc error {code}: {msg}")
        self.code = code


class pmc_csr(C.Structure):
    _fields_ = [("nrows", C.c_int32), ("ncols", C.c_int32), ("rowptr", C.POINTER(C.c_int32)),
                ("colind", C.POINTER(C.c_int32)), ("vals", C.POINTER(C.c_double))]


class pmc_solver_opts(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("max_iter", C.c_int32), ("rel_tol", C.c_double), ("abs_tol", C.c_double),
                ("cheb_degree_M", C.c_int32), ("cheb_ratio_M", C.c_double),
                ("mg_smooth_degree", C.c_int32), ("mg_smooth_ratio", C.c_double),
                ("mg_coarse_degree", C.c_int32), ("mg_coarse_ratio", C.c_double), ("check_every", C.c_int32),
                ("use_graph", C.c_int32), ("schur_scale", C.c_double), ("mg_coarsening", C.c_int32), ("mini_max_rows", C.c_int32),
                ("two_streams", C.c_int32), ("precond_storage", C.c_int32)]


PMC_ABI_VERSION = 3          # include/pmc.h: layout of pmc_solver_opts / pmc_stats these classes restate
PMC_STORAGE_FP32, PMC_STORAGE_FP64 = 0, 1


class pmc_stats(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("converged", C.c_int32), ("initial_norm", C.c_double),
                ("final_norm", C.c_double), ("solve_ms", C.c_double), ("setup_ms", C.c_double)]


class pmc_sampler_level(C.Structure):
    _fields_ = [("n_u", C.c_int32), ("n_s", C.c_int32), ("M", pmc_csr), ("B", pmc_csr),
                ("w_diag", C.POINTER(C.c_double)), ("P", pmc_csr)]


class pmc_hybrid_level(C.Structure):
    _fields_ = [("n_lambda", C.c_int32), ("n_s", C.c_int32), ("H", pmc_csr), ("G", pmc_csr),
                ("z_diag", C.POINTER(C.c_double)), ("w_diag", C.POINTER(C.c_double)), ("P", pmc_csr)]


class pmc_hybrid_elements(C.Structure):
    _fields_ = [("n_u", C.c_int32), ("n_s", C.c_int32), ("M_pattern", pmc_csr), ("c_ptr", C.POINTER(C.c_int32)),
                ("c_elem", C.POINTER(C.c_int32)), ("c_val", C.POINTER(C.c_double)), ("B", pmc_csr),
                ("w_diag", C.POINTER(C.c_double)), ("P", pmc_csr)]


class pmc_darcy_level(C.Structure):
    _fields_ = [("n_u", C.c_int32), ("n_p", C.c_int32), ("M_pattern", pmc_csr), ("c_ptr", C.POINTER(C.c_int32)),
                ("c_elem", C.POINTER(C.c_int32)), ("c_val", C.POINTER(C.c_double)), ("B", pmc_csr),
                ("rhs", C.POINTER(C.c_double)), ("ess_mask", C.POINTER(C.c_uint8)),
                ("ess_data", C.POINTER(C.c_double)), ("obs", C.POINTER(C.c_double)), ("P", pmc_csr)]


# every symbol include/pmc.h declares: name -> (restype, argtypes)
_VP = C.c_void_p
_DP = C.c_void_p   # double* that may be a host or a device address
SYMBOLS = {
    "pmc_version": (C.c_int, []),
    "pmc_last_error": (C.c_char_p, []),
    "pmc_solver_opts_default": (None, [C.POINTER(pmc_solver_opts)]),
    "pmc_abi_version": (C.c_int, []),
    "pmc_krylov_z_bytes": (C.c_int, []),
    "pmc_sampler_krylov_z_bytes": (C.c_int, [_VP]),
    "pmc_darcy_krylov_z_bytes": (C.c_int, [_VP]),
    "pmc_kernel_launches": (C.c_uint64, []),
    "pmc_ctx_create": (C.c_int, [C.c_int, C.POINTER(_VP)]),
    "pmc_ctx_create_abi": (C.c_int, [C.c_int, C.c_int, C.POINTER(_VP)]),
    "pmc_ctx_destroy": (None, [_VP]),
    "pmc_ctx_synchronize": (C.c_int, [_VP]),
    "pmc_ctx_stream": (_VP, [_VP]),
    "pmc_timer_start": (C.c_int, [_VP]),
    "pmc_timer_stop": (C.c_int, [_VP, C.POINTER(C.c_double)]),
    "pmc_malloc": (C.c_int, [_VP, C.c_size_t, C.POINTER(_VP)]),
    "pmc_free": (C.c_int, [_VP, _VP]),
    "pmc_memcpy_h2d": (C.c_int, [_VP, _VP, _VP, C.c_size_t]),
    "pmc_memcpy_d2h": (C.c_int, [_VP, _VP, _VP, C.c_size_t]),
    "pmc_rng_seed": (C.c_int, [_VP, C.c_uint64, C.c_int, C.c_int]),
    "pmc_normal_fill": (C.c_int, [_VP, C.c_double, C.c_double, C.c_uint64, C.c_uint32, C.c_int, C.c_int, _DP, C.c_int]),
    "pmc_sampler_create": (C.c_int, [_VP, C.c_int, C.c_int, C.POINTER(pmc_sampler_level), C.c_double, C.c_double,
                                     C.c_int, C.POINTER(pmc_solver_opts), C.POINTER(_VP)]),
    "pmc_sampler_create_hybrid": (C.c_int, [_VP, C.c_int, C.POINTER(pmc_hybrid_level), C.c_double, C.c_double, C.c_int,
                                            C.POINTER(pmc_solver_opts), C.POINTER(_VP)]),
    "pmc_sampler_is_hybrid": (C.c_int, [_VP]),
    "pmc_hybrid_build": (C.c_int, [C.POINTER(pmc_hybrid_elements), C.c_double, C.POINTER(_VP)]),
    "pmc_hybrid_system_level": (C.c_int, [_VP, C.POINTER(pmc_hybrid_level)]),
    "pmc_hybrid_system_destroy": (None, [_VP]),
    "pmc_sampler_create_hybrid_from_elements": (C.c_int, [_VP, C.c_int, C.POINTER(pmc_hybrid_elements), C.c_double, C.c_double,
                                                          C.c_int, C.POINTER(pmc_solver_opts), C.POINTER(_VP)]),
    "pmc_sampler_smoother_time": (C.c_int, [_VP, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "pmc_sampler_smoother_bytes": (C.c_int, [_VP, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "pmc_sampler_vcycle_info": (C.c_int, [_VP, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    "pmc_sampler_destroy": (None, [_VP]),
    "pmc_sampler_set_projection": (C.c_int, [_VP, C.c_int, C.c_int, C.POINTER(pmc_csr), C.POINTER(C.c_int32),
                                             C.POINTER(C.c_double), C.c_int]),
    "pmc_sampler_num_levels": (C.c_int, [_VP]),
    "pmc_sampler_xi_size": (C.c_int, [_VP, C.c_int]),
    "pmc_sampler_sample_size": (C.c_int, [_VP, C.c_int]),
    "pmc_sampler_batch_width": (C.c_int, [_VP, C.c_int]),
    "pmc_darcy_batch_width": (C.c_int, [_VP, C.c_int]),
    "pmc_darcy_set_operator_timing": (C.c_int, [_VP, C.c_int]),
    "pmc_darcy_operator_time": (C.c_int, [_VP, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "pmc_darcy_operator_bytes": (C.c_int, [_VP, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "pmc_darcy_poly_time": (C.c_int, [_VP, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "pmc_darcy_poly_bytes": (C.c_int, [_VP, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "pmc_sampler_nnz": (C.c_int64, [_VP, C.c_int]),
    "pmc_sampler_true_p": (C.c_int, [_VP, C.c_int, C.POINTER(pmc_csr)]),
    "pmc_sampler_sample": (C.c_int, [_VP, C.c_int, C.c_uint64, C.c_int, _DP, C.c_int]),
    "pmc_sampler_eval": (C.c_int, [_VP, C.c_int, C.c_int, C.c_int, _DP, _DP, _DP, C.c_int, C.c_int, _DP, C.c_int,
                                   C.POINTER(pmc_stats)]),
    "pmc_sampler_mult": (C.c_int, [_VP, C.c_int, C.c_int, _DP, _DP, C.c_int, C.c_int, C.POINTER(pmc_stats)]),
    "pmc_sampler_apply_preconditioner": (C.c_int, [_VP, C.c_int, C.c_int, _DP, _DP, C.c_int]),
    "pmc_sampler_apply_operator": (C.c_int, [_VP, C.c_int, C.c_int, _DP, _DP, C.c_int, C.c_int, C.POINTER(C.c_double),
                                             C.POINTER(C.c_double)]),
    "pmc_sampler_set_operator_timing": (C.c_int, [_VP, C.c_int]),
    "pmc_sampler_operator_time": (C.c_int, [_VP, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "pmc_sampler_operator_event_overhead": (C.c_int, [_VP, C.POINTER(C.c_double)]),
    "pmc_darcy_create": (C.c_int, [_VP, C.c_int, C.c_int, C.POINTER(pmc_darcy_level), C.c_int,
                                   C.POINTER(pmc_solver_opts), C.POINTER(_VP)]),
    "pmc_darcy_create_hybrid": (C.c_int, [_VP, C.c_int, C.c_int, C.POINTER(pmc_darcy_level), C.c_int,
                                          C.POINTER(pmc_solver_opts), C.POINTER(_VP)]),
    "pmc_darcy_destroy": (None, [_VP]),
    "pmc_darcy_num_dofs": (C.c_int, [_VP, C.c_int]),
    "pmc_darcy_num_pressure_dofs": (C.c_int, [_VP, C.c_int]),
    "pmc_darcy_nnz": (C.c_int64, [_VP, C.c_int]),
    "pmc_darcy_solve_fwd": (C.c_int, [_VP, C.c_int, C.c_int, _DP, C.POINTER(C.c_double), C.POINTER(C.c_double), _DP,
                                      C.c_int, C.POINTER(pmc_stats)]),
    "pmc_darcy_solve_fwd_pressure": (C.c_int, [_VP, C.c_int, C.c_int, _DP, _DP, C.POINTER(C.c_double),
                                               C.POINTER(C.c_double), C.c_int, C.c_int, C.POINTER(pmc_stats)]),
    "pmc_darcy_set_observations": (C.c_int, [_VP, C.c_int, C.POINTER(pmc_csr)]),
    "pmc_darcy_num_observations": (C.c_int, [_VP, C.c_int]),
    "pmc_darcy_compute_G": (C.c_int, [_VP, C.c_int, C.c_int, _DP, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                      C.POINTER(C.c_double), C.c_int, C.POINTER(pmc_stats)]),
    "pmc_comm_unique_id": (C.c_int, [_VP]),
    "pmc_comm_init": (C.c_int, [_VP, _VP, C.c_int, C.c_int]),
    "pmc_comm_destroy": (C.c_int, [_VP]),
    "pmc_allreduce_sum_f64": (C.c_int, [_VP, C.POINTER(C.c_double), C.c_int]),
}

_lib = None


def load_library(path: Optional[str] = None):
    """dlopen libpmc.so and bind every declared symbol (raises if one is missing)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    # PMC_LIB (harness only - the library itself reads no PMC_* variable): another build of the library, e.g. the laboratory
    # one for a profiling pass, selected WITHOUT overwriting the product's file
    p = path or os.environ.get("PMC_LIB") or LIB_PATH
    if not os.path.exists(p):
        raise PmcError(-2, f"{p} not found - build it with `make` / __graft_entry__.build() (no CPU fallback)")
    lib = C.CDLL(p)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)     # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def _check(rc):
    if rc != 0:
        raise PmcError(rc, load_library().pmc_last_error().decode("utf-8", "replace"))


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a, ctype):
    return a.ctypes.data_as(C.POINTER(ctype))


class _Keep:
    """Holds numpy arrays alive while a C struct points into them."""

    def __init__(self):
        self.refs = []

    def csr(self, m) -> pmc_csr:
        if m is None:
            return pmc_csr(0, 0, None, None, None)
        m = m.tocsr()
        rp, ci, v = _i32(m.indptr), _i32(m.indices), _f64(m.data)
        self.refs += [rp, ci, v]
        return pmc_csr(m.shape[0], m.shape[1], _ptr(rp, C.c_int32), _ptr(ci, C.c_int32), _ptr(v, C.c_double))

    def f64(self, a):
        a = _f64(a)
        self.refs.append(a)
        return _ptr(a, C.c_double)

    def i32(self, a):
        a = _i32(a)
        self.refs.append(a)
        return _ptr(a, C.c_int32)

    def u8(self, a):
        a = np.ascontiguousarray(a, dtype=np.uint8)
        self.refs.append(a)
        return _ptr(a, C.c_uint8)


def hybrid_build(pattern, c_ptr, c_elem, c_val, B, w_diag, alpha, P=None):
    """pmc_hybrid_build (host code of libpmc.so, no GPU needed): the element-local elimination of one sampler level from the
    element decomposition of the u-mass matrix (fe.rt0.mass_contributions), B without boundary elimination and diag(W).
    Returns (H, G, z_diag) as scipy CSR / numpy COPIES of what pmc_hybrid_system_level exposes."""
    import scipy.sparse as sp
    lib = load_library()
    keep = _Keep()
    n_s, n_u = B.shape
    lv = pmc_hybrid_elements(n_u, n_s, keep.csr(pattern), keep.i32(c_ptr), keep.i32(c_elem), keep.f64(c_val), keep.csr(B),
                             keep.f64(w_diag), keep.csr(P))
    h = _VP()
    _check(lib.pmc_hybrid_build(C.byref(lv), float(alpha), C.byref(h)))
    try:
        v = pmc_hybrid_level()
        _check(lib.pmc_hybrid_system_level(h, C.byref(v)))

        def mat(c):
            rp = np.ctypeslib.as_array(c.rowptr, (c.nrows + 1,)).copy()
            nnz = int(rp[-1])
            return sp.csr_matrix((np.ctypeslib.as_array(c.vals, (nnz,)).copy(), np.ctypeslib.as_array(c.colind, (nnz,)).copy(), rp),
                                 shape=(c.nrows, c.ncols))
        return mat(v.H), mat(v.G), np.ctypeslib.as_array(v.z_diag, (n_s,)).copy()
    finally:
        lib.pmc_hybrid_system_destroy(h)


def library_hybrid_builder(space, alpha):
    """fe.hybrid's `builder` hook: the library's own elimination (pmc_hybrid_build) instead of the numpy stand-in"""
    from .fe.rt0 import mass_contributions
    pat, c_ptr, c_elem, c_val = mass_contributions(space.emass)
    return hybrid_build(pat, c_ptr, c_elem, c_val, space.B, space.vol, alpha)


def solver_opts(**kw) -> pmc_solver_opts:
    o = pmc_solver_opts()
    load_library().pmc_solver_opts_default(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise TypeError(f"unknown solver option {k!r}")
        setattr(o, k, v)
    return o


class DeviceArray:
    """fp64 array in HBM owned by a Context (pmc_malloc / pmc_free)."""

    def __init__(self, ctx: "Context", n: int):
        self.ctx, self.n = ctx, int(n)
        p = _VP()
        _check(ctx.lib.pmc_malloc(ctx.h, self.n * 8, C.byref(p)))
        self.ptr = p.value or 0
        ctx._adopt(self)

    def upload(self, a):
        a = _f64(a).ravel()
        assert a.size == self.n
        _check(self.ctx.lib.pmc_memcpy_h2d(self.ctx.h, self.ptr, a.ctypes.data, a.nbytes))
        return self

    def download(self) -> np.ndarray:
        out = np.empty(self.n, np.float64)
        _check(self.ctx.lib.pmc_memcpy_d2h(self.ctx.h, out.ctypes.data, self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr and self.ctx.h:
            self.ctx.lib.pmc_free(self.ctx.h, self.ptr)
        self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _addr(x):
    """(address, memspace) of a numpy array, a DeviceArray or a torch tensor."""
    if x is None:
        return None, None
    if isinstance(x, DeviceArray):
        return x.ptr, PMC_MEM_DEVICE
    if isinstance(x, np.ndarray):
        assert x.dtype == np.float64 and x.flags["C_CONTIGUOUS"]
        return x.ctypes.data, PMC_MEM_HOST
    if hasattr(x, "data_ptr"):   # torch tensor: plumbing only (device memory owner)
        assert str(x.dtype) == "torch.float64" and x.is_contiguous()
        return x.data_ptr(), (PMC_MEM_DEVICE if x.is_cuda else PMC_MEM_HOST)
    raise TypeError(type(x))


class Context:
    def __init__(self, device_id: int = 0, seed: int = 0):
        self.lib = load_library()
        h = _VP()
        # the versioned entry point: a library with another pmc_solver_opts / pmc_stats layout refuses this binding
        _check(self.lib.pmc_ctx_create_abi(int(device_id), PMC_ABI_VERSION, C.byref(h)))
        self.h = h
        self.device_id = device_id
        self._children = []     # weakrefs to handles that must die before the context does
        self.seed(seed)

    def _adopt(self, obj):
        import weakref
        self._children.append(weakref.ref(obj))

    def seed(self, seed: int, nparts: int = 1, mypart: int = 0):
        _check(self.lib.pmc_rng_seed(self.h, C.c_uint64(seed), nparts, mypart))

    def synchronize(self):
        _check(self.lib.pmc_ctx_synchronize(self.h))

    def stream(self) -> int:
        return self.lib.pmc_ctx_stream(self.h) or 0

    def timer_start(self):
        _check(self.lib.pmc_timer_start(self.h))

    def timer_stop(self) -> float:
        ms = C.c_double()
        _check(self.lib.pmc_timer_stop(self.h, C.byref(ms)))
        return ms.value

    def empty(self, n) -> DeviceArray:
        return DeviceArray(self, n)

    def array(self, a) -> DeviceArray:
        a = _f64(a)
        return DeviceArray(self, a.size).upload(a)

    def normal_fill(self, n, nbatch=1, first_id=0, stream=0, mean=0.0, sigma2=1.0, out=None):
        """NormalDistributionSampler::operator()(Vector&)."""
        if out is None:
            out = np.empty((nbatch, n))
        p, ms = _addr(out)
        _check(self.lib.pmc_normal_fill(self.h, mean, sigma2, C.c_uint64(first_id), stream, nbatch, n, p, ms))
        return out

    # communicator
    def comm_unique_id(self) -> bytes:
        buf = C.create_string_buffer(128)
        _check(self.lib.pmc_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, uid: bytes, nranks: int, rank: int):
        buf = C.create_string_buffer(uid, 128)
        _check(self.lib.pmc_comm_init(self.h, buf, nranks, rank))

    def allreduce_sum(self, a: np.ndarray) -> np.ndarray:
        a = _f64(a)
        _check(self.lib.pmc_allreduce_sum_f64(self.h, _ptr(a, C.c_double), a.size))
        return a

    def close(self):
        if getattr(self, "h", None):
            for ref in self._children:      # samplers / solvers / device arrays hold a reference to this ctx
                obj = ref()
                if obj is not None:
                    (obj.close if hasattr(obj, "close") else obj.free)()
            self._children = []
            self.lib.pmc_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PDESampler:
    """Device SPDE sampler; mirrors parelagmc::PDESampler / EmbeddedPDESampler /
    L2ProjectionPDESampler (Sample, Eval, SampleSize, GetNNZ)."""

    def __init__(self, ctx: Context, problem, opts: Optional[pmc_solver_opts] = None, projection: str = "none",
                 l2_ops=None):
        self.ctx, self.problem = ctx, problem
        lib = ctx.lib
        keep = _Keep()
        nl = len(problem.levels)
        h = _VP()
        o = opts if opts is not None else solver_opts()
        self.hybrid = hasattr(problem.levels[0], "n_lambda")     # fe.HybridSamplerProblem: the hybridized solver
        if self.hybrid:
            harr = (pmc_hybrid_level * nl)()
            for i, L in enumerate(problem.levels):
                harr[i] = pmc_hybrid_level(L.n_lambda, L.n_s, keep.csr(L.H), keep.csr(L.G), keep.f64(L.z_diag),
                                           keep.f64(L.w_diag), keep.csr(L.P))
            _check(lib.pmc_sampler_create_hybrid(ctx.h, nl, harr, problem.alpha, problem.matern_g,
                                                 1 if problem.lognormal else 0, C.byref(o), C.byref(h)))
        else:
            arr = (pmc_sampler_level * nl)()
            for i, L in enumerate(problem.levels):
                arr[i] = pmc_sampler_level(L.n_u, L.n_s, keep.csr(L.M), keep.csr(L.B), keep.f64(L.w_diag), keep.csr(L.P))
            _check(lib.pmc_sampler_create(ctx.h, nl, problem.n_mc_levels, arr, problem.alpha, problem.matern_g,
                                          1 if problem.lognormal else 0, C.byref(o), C.byref(h)))
        self.h = h
        ctx._adopt(self)
        self.nlevels = problem.n_mc_levels
        if projection == "gather":
            for lvl, idx in enumerate(problem.orig_index):
                k2 = _Keep()
                _check(lib.pmc_sampler_set_projection(h, lvl, PMC_PROJ_GATHER, None, k2.i32(idx), None, len(idx)))
        elif projection == "l2":
            for lvl, (Gt, inv_w) in enumerate(l2_ops):
                k2 = _Keep()
                g = k2.csr(Gt)
                _check(lib.pmc_sampler_set_projection(h, lvl, PMC_PROJ_L2, C.byref(g), None, k2.f64(inv_w), Gt.shape[0]))
        elif projection != "none":
            raise ValueError(projection)

    def xi_size(self, level):
        return self.ctx.lib.pmc_sampler_xi_size(self.h, level)

    def SampleSize(self, level):
        return self.ctx.lib.pmc_sampler_sample_size(self.h, level)

    def BatchWidth(self, level):
        """realizations of `level` one launch of the solver kernels carries (pmc_sampler_batch_width)"""
        return self.ctx.lib.pmc_sampler_batch_width(self.h, level)

    def vcycle_levels(self, level):
        """[{rows, nnz, slots, sp_nnz, sp_slots, in_tail, fused_restriction, narrow_dense, narrow_pieces}] of the V-cycle
        hierarchy of `level` (narrow_*: what launches of at most 8 realizations do on the level, see include/pmc.h)"""
        out, nv, info = [], C.c_int(0), (C.c_int64 * 7)()
        v = 0
        while True:
            _check(self.ctx.lib.pmc_sampler_vcycle_info(self.h, level, v, C.byref(nv), info))
            d = dict(zip(("rows", "nnz", "slots", "sp_nnz", "sp_slots", "in_tail", "fused_restriction"), [int(x) for x in info]))
            flags = d["in_tail"]
            d.update(in_tail=flags & 1, narrow_dense=(flags >> 1) & 1, narrow_pieces=1 << (flags >> 4))
            out.append(d)
            v += 1
            if v >= nv.value:
                return out

    def z_bytes(self):
        """bytes per entry of the preconditioned Krylov vectors of this handle (4: PMC_STORAGE_FP32, 8: PMC_STORAGE_FP64)"""
        return self.ctx.lib.pmc_sampler_krylov_z_bytes(self.h)

    def GetNNZ(self, level):
        return self.ctx.lib.pmc_sampler_nnz(self.h, level)

    def GetTrueP(self, level):
        """MLSampler::GetTrueP: the s-space prolongator from level+1 to level as a scipy CSR matrix (a copy)."""
        import scipy.sparse as sp
        c = pmc_csr()
        _check(self.ctx.lib.pmc_sampler_true_p(self.h, level, C.byref(c)))
        nnz = c.rowptr[c.nrows]
        rp = np.ctypeslib.as_array(c.rowptr, shape=(c.nrows + 1,)).copy()
        ci = np.ctypeslib.as_array(c.colind, shape=(nnz,)).copy()
        v = np.ctypeslib.as_array(c.vals, shape=(nnz,)).copy()
        return sp.csr_matrix((v, ci, rp), shape=(c.nrows, c.ncols))

    def Sample(self, level, first_id=0, nbatch=1, out=None):
        n = self.xi_size(level)
        if out is None:
            out = np.empty((nbatch, n))
        p, ms = _addr(out)
        _check(self.ctx.lib.pmc_sampler_sample(self.h, level, C.c_uint64(first_id), nbatch, p, ms))
        return out

    def Eval(self, level, xi, xi_level=None, init_s=None, init_level=None, use_init=False, s_out=None,
             embed_out=None, want_embed=False, return_stats=False):
        """xi: (nbatch, n_xi) numpy (host) or DeviceArray/torch tensor (device, with nbatch=...).
        Returns s (and embed_s, stats) as numpy arrays for host inputs."""
        lib = self.ctx.lib
        if not (0 <= level < self.nlevels):
            raise PmcError(-1, f"Eval: level {level} out of range")
        if isinstance(xi, np.ndarray):
            xi = _f64(np.atleast_2d(xi))
            nbatch = xi.shape[0]
            if xi_level is None:     # reference behaviour: infer from the length (PDESampler.cpp:419)
                sizes = [self.xi_size(l) for l in range(self.nlevels)]
                xi_level = sizes.index(xi.shape[1])
            if s_out is None:
                s_out = np.empty((nbatch, self.SampleSize(level)))
            if want_embed and embed_out is None:
                embed_out = np.empty((nbatch, self.xi_size(level)))
            if init_s is not None:
                init_s = _f64(np.atleast_2d(init_s))
        else:
            nbatch = xi.n // self.xi_size(xi_level)
        stats = (pmc_stats * nbatch)()
        pxi, ms = _addr(xi)
        ps, _ = _addr(s_out)
        pinit, _ = _addr(init_s)
        pemb, _ = _addr(embed_out)
        _check(lib.pmc_sampler_eval(self.h, level, xi_level, nbatch, pxi, ps, pinit,
                                    -1 if init_level is None else init_level, 1 if use_init else 0, pemb, ms, stats))
        # device milliseconds of this call (HIP events on the handle's stream): (right-hand sides / initial guesses, solves)
        self.last_phase_ms = (sum(s.setup_ms for s in stats), sum(s.solve_ms for s in stats))
        out = [s_out]
        if want_embed or embed_out is not None:
            out.append(embed_out)
        if return_stats:
            out.append([(s.iterations, s.converged, s.initial_norm, s.final_norm) for s in stats])
        return out[0] if len(out) == 1 else tuple(out)

    def set_operator_timing(self, on: bool):
        """Bracket every K5 launch of the MINRES loop with HIP events (in-situ kernel time for the roofline)."""
        _check(self.ctx.lib.pmc_sampler_set_operator_timing(self.h, 1 if on else 0))

    def operator_time(self):
        """(total ms, launches) of the timed K5 launches since the last call."""
        ms, n = C.c_double(0.0), C.c_int64(0)
        _check(self.ctx.lib.pmc_sampler_operator_time(self.h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def operator_event_overhead(self):
        """total ms of the empty event brackets recorded behind the timed launches since the last call."""
        ms = C.c_double(0.0)
        _check(self.ctx.lib.pmc_sampler_operator_event_overhead(self.h, C.byref(ms)))
        return ms.value

    def smoother_time(self):
        """(total ms, launches, event overhead ms) of the bracketed post-smoothing launches (hybridized samplers) since the
        last call"""
        ms, n, g = C.c_double(0.0), C.c_int64(0), C.c_double(0.0)
        _check(self.ctx.lib.pmc_sampler_smoother_time(self.h, C.byref(ms), C.byref(n), C.byref(g)))
        return ms.value, n.value, g.value

    def smoother_bytes(self, level, nbatch):
        b = C.c_double(0.0)
        _check(self.ctx.lib.pmc_sampler_smoother_bytes(self.h, level, nbatch, C.byref(b)))
        return b.value

    def Mult(self, level, x, repeat=1):
        """y = [M Bt; B -aW] x (the block operator's Mult).  x: (nbatch, n_u+n_s) numpy or a
        DeviceArray (then pass nbatch via x.n).  Returns (y, avg_kernel_ms, algorithmic_bytes)."""
        L = self.problem.levels[level]
        n = L.n_lambda if self.hybrid else L.n_u + L.n_s     # hybrid: y = H x
        if isinstance(x, np.ndarray):
            x = _f64(np.atleast_2d(x))
            nb = x.shape[0]
            y = np.empty_like(x)
        else:
            nb = x.n // n
            y = self.ctx.empty(x.n)
        px, ms = _addr(x)
        py, _ = _addr(y)
        t, b = C.c_double(), C.c_double()
        _check(self.ctx.lib.pmc_sampler_apply_operator(self.h, level, nb, px, py, ms, repeat, C.byref(t), C.byref(b)))
        return y, t.value, b.value

    def ApplyPreconditioner(self, level, r):
        """z = B^-1 r, one application of the MINRES preconditioner of `level` (pmc_sampler_apply_preconditioner); r: (nbatch,
        n_u+n_s) numpy, nbatch one of 1, 2, 4, ... up to the level's launch width"""
        r = _f64(np.atleast_2d(r))
        z = np.empty_like(r)
        pr, ms = _addr(r)
        pz, _ = _addr(z)
        _check(self.ctx.lib.pmc_sampler_apply_preconditioner(self.h, level, r.shape[0], pr, pz, ms))
        return z

    def Solve(self, level, rhs, guess=None, return_stats=False):
        """invA[level]->Mult(rhs, sol) (pmc_sampler_mult): the full solution [u; s] of A x = rhs.  rhs: (nbatch, n_u+n_s)
        numpy or a DeviceArray; guess (same kind): initial guess (iterative_mode)."""
        L = self.problem.levels[level]
        n = L.n_lambda if self.hybrid else L.n_u + L.n_s     # hybrid: H lambda = rhs on the multipliers
        if isinstance(rhs, np.ndarray):
            rhs = _f64(np.atleast_2d(rhs))
            nb = rhs.shape[0]
            sol = np.empty_like(rhs) if guess is None else _f64(np.atleast_2d(guess)).copy()
        else:
            nb = rhs.n // n
            sol = self.ctx.empty(rhs.n) if guess is None else guess
        pr, ms = _addr(rhs)
        px, _ = _addr(sol)
        st = (pmc_stats * nb)()
        _check(self.ctx.lib.pmc_sampler_mult(self.h, level, nb, pr, px, 0 if guess is None else 1, ms, st))
        if return_stats:
            return sol, [(t.iterations, t.converged, t.initial_norm, t.final_norm) for t in st]
        return sol

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.pmc_sampler_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DarcySolver:
    """Device mixed Darcy solver; mirrors parelagmc::DarcySolver (SolveFwd, GetNNZ, ...)."""

    def __init__(self, ctx: Context, problem, opts: Optional[pmc_solver_opts] = None, hybrid: bool = False):
        """hybrid: SolveFwd through the hybridized form (pmc_darcy_create_hybrid, the reference's "Hybridization" option)"""
        self.ctx, self.problem = ctx, problem
        keep = _Keep()
        nl = len(problem.levels)
        arr = (pmc_darcy_level * nl)()
        for i, L in enumerate(problem.levels):
            arr[i] = pmc_darcy_level(L.n_u, L.n_p, keep.csr(L.M_pattern), keep.i32(L.c_ptr), keep.i32(L.c_elem),
                                     keep.f64(L.c_val), keep.csr(L.B), keep.f64(L.rhs), keep.u8(L.ess_mask),
                                     keep.f64(L.ess_data), keep.f64(L.obs), keep.csr(L.P))
        h = _VP()
        o = opts if opts is not None else solver_opts()
        create = ctx.lib.pmc_darcy_create_hybrid if hybrid else ctx.lib.pmc_darcy_create
        _check(create(ctx.h, nl, problem.n_mc_levels, arr, 1 if problem.k_divides else 0, C.byref(o), C.byref(h)))
        self.h = h
        ctx._adopt(self)
        self.nlevels = problem.n_mc_levels

    def GetGlobalNumberOfDofs(self, level):
        return self.ctx.lib.pmc_darcy_num_dofs(self.h, level)

    GetNumberOfDofs = GetGlobalNumberOfDofs

    def GetNNZ(self, level):
        return self.ctx.lib.pmc_darcy_nnz(self.h, level)

    def BatchWidth(self, level):
        return self.ctx.lib.pmc_darcy_batch_width(self.h, level)

    def z_bytes(self):
        return self.ctx.lib.pmc_darcy_krylov_z_bytes(self.h)

    def set_operator_timing(self, on: bool):
        """Bracket every in-loop launch of the u-rows [M(k) | B^T] x (eg_pair_spmm) with HIP events."""
        _check(self.ctx.lib.pmc_darcy_set_operator_timing(self.h, 1 if on else 0))

    def operator_time(self):
        """(total bracket ms, launches, total empty-bracket ms) since the last call."""
        ms, n, gap = C.c_double(0.0), C.c_int64(0), C.c_double(0.0)
        _check(self.ctx.lib.pmc_darcy_operator_time(self.h, C.byref(ms), C.byref(n), C.byref(gap)))
        return ms.value, n.value, gap.value

    def operator_bytes(self, level, nbatch):
        b = C.c_double(0.0)
        _check(self.ctx.lib.pmc_darcy_operator_bytes(self.h, level, nbatch, C.byref(b)))
        return b.value

    def poly_time(self):
        """the same for the M-block polynomial of the preconditioner (eg_poly2): (bracket ms, launches, empty-bracket ms)"""
        ms, n, gap = C.c_double(0.0), C.c_int64(0), C.c_double(0.0)
        _check(self.ctx.lib.pmc_darcy_poly_time(self.h, C.byref(ms), C.byref(n), C.byref(gap)))
        return ms.value, n.value, gap.value

    def poly_bytes(self, level, nbatch):
        b = C.c_double(0.0)
        _check(self.ctx.lib.pmc_darcy_poly_bytes(self.h, level, nbatch, C.byref(b)))
        return b.value

    def SolveFwd(self, level, k, nbatch=None, want_solution=False, sol_out=None, return_stats=False):
        """Returns (Q, C) arrays of length nbatch (plus solution / stats on request)."""
        lib = self.ctx.lib
        if isinstance(k, np.ndarray):
            k = _f64(np.atleast_2d(k))
            nbatch = k.shape[0]
            if want_solution and sol_out is None:
                sol_out = np.empty((nbatch, self.GetGlobalNumberOfDofs(level)))
        assert nbatch is not None
        Q = np.empty(nbatch)
        Cc = np.empty(nbatch)
        stats = (pmc_stats * nbatch)()
        pk, ms = _addr(k)
        psol, _ = _addr(sol_out)
        _check(lib.pmc_darcy_solve_fwd(self.h, level, nbatch, pk, _ptr(Q, C.c_double), _ptr(Cc, C.c_double), psol, ms,
                                       stats))
        # device milliseconds of this call: ("Darcy: Build Solver" = M(k), elimination, Schur hierarchy refresh; "Darcy: Mult")
        self.last_phase_ms = (sum(s.setup_ms for s in stats), sum(s.solve_ms for s in stats))
        out = [Q, Cc]
        if want_solution or sol_out is not None:
            out.append(sol_out)
        if return_stats:
            out.append([(s.iterations, s.converged, s.initial_norm, s.final_norm) for s in stats])
        return tuple(out)

    def SetObservations(self, level, Gobs):
        """Gobs: scipy sparse (nobs, n_p): rows are the observation functionals g_obs_i of the level."""
        keep = _Keep()
        g = keep.csr(Gobs)
        _check(self.ctx.lib.pmc_darcy_set_observations(self.h, level, C.byref(g)))

    def ComputeG(self, level, k, nbatch=None):
        """BayesianInverseProblem::ComputeG: returns (G (nbatch, nobs), C, Q)."""
        if isinstance(k, np.ndarray):
            k = _f64(np.atleast_2d(k))
            nbatch = k.shape[0]
        nobs = self.ctx.lib.pmc_darcy_num_observations(self.h, level)
        G = np.empty((nbatch, nobs))
        Q = np.empty(nbatch)
        Cc = np.empty(nbatch)
        pk, ms = _addr(k)
        _check(self.ctx.lib.pmc_darcy_compute_G(self.h, level, nbatch, pk, _ptr(G, C.c_double), _ptr(Cc, C.c_double),
                                                _ptr(Q, C.c_double), ms, None))
        return G, Cc, Q

    def SolveFwd_RtnPressure(self, level, k, compute_Q=True):
        """Returns (P, C, Q): pressure block (nbatch, n_p), dof counts, QoI (None unless compute_Q)."""
        k = _f64(np.atleast_2d(k))
        nb = k.shape[0]
        n_p = self.problem.levels[level].n_p
        P = np.empty((nb, n_p))
        Q = np.empty(nb)
        Cc = np.empty(nb)
        stats = (pmc_stats * nb)()
        _check(self.ctx.lib.pmc_darcy_solve_fwd_pressure(self.h, level, nb, k.ctypes.data, P.ctypes.data,
                                                         _ptr(Cc, C.c_double), _ptr(Q, C.c_double),
                                                         1 if compute_Q else 0, PMC_MEM_HOST, stats))
        return P, Cc, (Q if compute_Q else None)

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.pmc_darcy_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
