"""The drop-in boundary exercised from compiled C and C++ (not ctypes): tests/c/abi_smoke.c includes include/pmc.h only;
tests/c/adapter_smoke.cpp goes through parelagmc_amd/host/mfem_adapter.hpp (compiled against a stand-in for the MFEM
containers) and the reference-named mirror classes of parelagmc.hpp.  The CPU test builds both (the headers compile as
C11 with -Werror and as C++17); the GPU test runs them on a problem file with the oracle's expected fields and QoIs."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

BIN = os.path.join(ROOT, "tests", "c", "bin")


def _build():
    r = subprocess.run(["make", "-C", ROOT, "test-abi"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def _w_csr(f, A):
    A = A.tocsr()
    A.sort_indices()
    np.array([A.shape[0], A.shape[1], A.nnz], np.int32).tofile(f)
    A.indptr.astype(np.int32).tofile(f)
    A.indices.astype(np.int32).tofile(f)
    A.data.astype(np.float64).tofile(f)


def write_problem_file(path, sp_, dp, xi, s_expect, k, q_expect):
    """layout read by tests/c/prob_io.h"""
    with open(path, "wb") as f:
        np.array([0x504d4332, len(sp_.levels)], np.int32).tofile(f)
        np.array([sp_.alpha, sp_.matern_g], np.float64).tofile(f)
        np.array([1 if sp_.lognormal else 0], np.int32).tofile(f)
        for L in sp_.levels:
            np.array([L.n_u, L.n_s], np.int32).tofile(f)
            _w_csr(f, L.M)
            _w_csr(f, L.B)
            L.w_diag.astype(np.float64).tofile(f)
            np.array([0 if L.P is None else 1], np.int32).tofile(f)
            if L.P is not None:
                _w_csr(f, L.P)
        np.array([xi.shape[0]], np.int32).tofile(f)
        xi.astype(np.float64).tofile(f)
        for s in s_expect:
            s.astype(np.float64).tofile(f)
        np.array([len(dp.levels), 1 if dp.k_divides else 0], np.int32).tofile(f)
        for L in dp.levels:
            np.array([L.n_u, L.n_p], np.int32).tofile(f)
            _w_csr(f, L.M_pattern)
            L.c_ptr.astype(np.int32).tofile(f)
            np.array([len(L.c_elem)], np.int32).tofile(f)
            L.c_elem.astype(np.int32).tofile(f)
            L.c_val.astype(np.float64).tofile(f)
            _w_csr(f, L.B)
            L.rhs.astype(np.float64).tofile(f)
            L.ess_mask.astype(np.uint8).tofile(f)
            L.ess_data.astype(np.float64).tofile(f)
            L.obs.astype(np.float64).tofile(f)
            np.array([0 if L.P is None else 1], np.int32).tofile(f)
            if L.P is not None:
                _w_csr(f, L.P)
        for kk, qq in zip(k, q_expect):
            kk.astype(np.float64).tofile(f)
            qq.astype(np.float64).tofile(f)


def test_boundary_compiles_from_c_and_cpp():
    """include/pmc.h as C11 (-Wall -Wextra -Werror), mfem_adapter.hpp + parelagmc.hpp as C++17, linked against the libraries"""
    _build()
    assert os.access(os.path.join(BIN, "abi_smoke"), os.X_OK) and os.access(os.path.join(BIN, "adapter_smoke"), os.X_OK)


@pytest.mark.gpu
def test_c_and_cpp_callers_reproduce_the_oracle(tmp_path, hex_hierarchy_small, seeded_rng):
    from oracle.darcy_oracle import DarcyOracle
    from oracle.sampler_oracle import SamplerOracle
    from parelagmc_amd.fe import build_darcy_problem, build_sampler_problem
    _build()
    sp_ = build_sampler_problem(hex_hierarchy_small, corlen=0.1, lognormal=True)
    dp = build_darcy_problem(hex_hierarchy_small, [0, 1, 1, 1, 1, 0], [1, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 1])
    so, do = SamplerOracle(sp_), DarcyOracle(dp)
    nb = 3
    xi = seeded_rng.standard_normal((nb, sp_.levels[0].n_s))
    s_expect = [np.stack([so.eval(l, 0, x)[0] for x in xi]) for l in range(2)]
    k = [np.exp(0.5 * seeded_rng.standard_normal((nb, dp.levels[l].n_p))) for l in range(2)]
    q_expect = [np.array([do.solve_fwd(l, kk)[0] for kk in k[l]]) for l in range(2)]
    path = str(tmp_path / "problem.bin")
    write_problem_file(path, sp_, dp, xi, s_expect, k, q_expect)
    for prog in ("abi_smoke", "adapter_smoke"):
        r = subprocess.run([os.path.join(BIN, prog), path], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and f"{prog} OK" in r.stdout, r.stdout + r.stderr
