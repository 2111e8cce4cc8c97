"""Round-4 GPU tests: the storage option of the preconditioner (pmc_solver_opts.precond_storage, ABI 3), proven at FULL size
by TRUE residuals - ||b - A x|| formed with the fp64 block operator (pmc_sampler_apply_operator, K5) from the solution the
solver returns (pmc_sampler_mult = invA[level]->Mult, /root/reference/src/PDESampler.cpp:397,521) - and the drop-in
single-realization path."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, golden_path

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(b)


def _tet_problem(nref):
    from parelagmc_amd.fe import build_hierarchy, build_sampler_problem, mesh_from_json
    h = build_hierarchy(mesh_from_json(golden_path("meshes", "cube_tet.json")), nref)
    return build_sampler_problem(h, corlen=0.1, n_mc_levels=1)


def _true_residuals(smp, L, g, rhs, x):
    """per realization: (||b - A x||_2 / ||b||_2, sqrt(<r, B^-1 r>)) with A x from the device's fp64 block operator"""
    Ax = smp.Mult(0, x)[0]
    r = rhs - Ax
    z = smp.ApplyPreconditioner(0, r)
    two = np.linalg.norm(r, axis=1) / np.linalg.norm(rhs, axis=1)
    pnorm = np.sqrt(np.einsum("ij,ij->i", r, z))
    return two, pnorm, Ax


@pytest.mark.parametrize("nref", [5, 6])
def test_true_residual_of_the_sampler_solve_at_full_size_both_storages(gpu_ctx, nref):
    """cube_tet r = 5 (595 968 DoF, BASELINE config 2) and r = 6 (4 743 168 DoF): for fp32 storage inside the preconditioner
    (the default) and for everything fp64, at the reference's tolerance 1e-6 and at 1e-12,
      * the TRUE preconditioned residual norm sqrt(<r, B^-1 r>), r = b - A x with the fp64 K5, agrees with the norm MINRES's
        recurrence reports (pmc_stats.final_norm) to 1e-4 relative at 1e-6 and to 1 % at 1e-12 (measured on MI355X: nine
        digits at 1e-6, five at 1e-12, for both storages - the recurrence does not drift from the truth);
      * the two storages return the same field to the solver tolerance, with the same iteration counts."""
    from parelagmc_amd import capi
    sp = _tet_problem(nref)
    L = sp.levels[0]
    n = L.n_u + L.n_s
    rng = np.random.default_rng(4 + nref)
    nb = 2
    rhs = np.zeros((nb, n))
    rhs[:, L.n_u:] = -sp.matern_g * np.sqrt(L.w_diag) * rng.standard_normal((nb, L.n_s))   # what Eval solves for
    report = {}
    fields = {}
    for storage in (capi.PMC_STORAGE_FP32, capi.PMC_STORAGE_FP64):
        for tol in (1e-6, 1e-12):
            smp = capi.PDESampler(gpu_ctx, sp, capi.solver_opts(rel_tol=tol, abs_tol=1e-300, precond_storage=storage))
            assert smp.z_bytes() == (4 if storage == capi.PMC_STORAGE_FP32 else 8)
            x, st = smp.Solve(0, rhs, return_stats=True)
            assert all(t[1] == 1 for t in st), st
            two, pnorm, Ax = _true_residuals(smp, L, sp.matern_g, rhs, x)
            if nref == 5 and storage == capi.PMC_STORAGE_FP32 and tol == 1e-6:
                # the device operator against scipy on the same vector (the residual is only as good as this product)
                from oracle.sampler_oracle import SamplerOracle
                A = SamplerOracle(sp).block_operator(0).tocsr()
                assert rel(Ax[0], A @ x[0]) < 1e-13
            eta0 = np.array([t[2] for t in st])
            eta = np.array([t[3] for t in st])
            report[(storage, tol)] = dict(its=[t[0] for t in st], two=two, true=pnorm / eta0, reported=eta / eta0)
            fields[(storage, tol)] = x[:, L.n_u:].copy()
            ratio = pnorm / eta
            assert np.all(np.abs(ratio - 1.0) < (1e-4 if tol == 1e-6 else 1e-2)), (storage, tol, ratio)
            assert np.all(pnorm / eta0 <= tol)
            smp.close()
    print("true residuals r=%d:" % nref, {k: {a: np.asarray(b).tolist() for a, b in v.items()} for k, v in report.items()})
    for tol in (1e-6, 1e-12):
        a, b = report[(0, tol)], report[(1, tol)]
        assert a["its"] == b["its"]
        assert np.all(np.abs(a["true"] / b["true"] - 1.0) < 1e-3)
        assert rel(fields[(0, tol)], fields[(1, tol)]) < (2e-6 if tol == 1e-6 else 1e-9)


def test_true_residual_of_the_darcy_solve_on_hex64_both_storages(gpu_ctx):
    """The Darcy solve of BASELINE config 3's finest level (cube_hex 64^3, 1 060 864 DoF) for a log-normal permeability:
    b - A(k) x with A(k), b assembled on the host exactly as src/DarcySolver.cpp:472-520 does (oracle/darcy_oracle.py) from
    the solution pmc_darcy_solve_fwd returns.  fp32 storage (default) against everything fp64, at 1e-6 and 1e-12."""
    from oracle.darcy_oracle import DarcyOracle
    from parelagmc_amd import capi
    from parelagmc_amd.fe import box_mesh, build_darcy_problem, build_hierarchy, build_sampler_problem
    h = build_hierarchy(box_mesh([4, 4, 4], [2, 2, 2], "hex"), 4)
    dp = build_darcy_problem(h, [0, 1, 1, 1, 1, 0], [1, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 1], n_mc_levels=1)
    sp = build_sampler_problem(h, corlen=0.1, lognormal=True, n_mc_levels=1)
    smp = capi.PDESampler(gpu_ctx, sp)
    k = smp.Eval(0, smp.Sample(0, first_id=7, nbatch=2))
    smp.close()
    assert k.shape == (2, 262144) and k.min() > 0
    do = DarcyOracle(dp)
    sys_h = [do.assemble(0, kk) for kk in k]
    out = {}
    for storage in (capi.PMC_STORAGE_FP32, capi.PMC_STORAGE_FP64):
        for tol in (1e-6, 1e-12):
            ds = capi.DarcySolver(gpu_ctx, dp, capi.solver_opts(rel_tol=tol, abs_tol=1e-300, precond_storage=storage))
            assert ds.z_bytes() == (4 if storage == capi.PMC_STORAGE_FP32 else 8)
            Q, C, sol, st = ds.SolveFwd(0, k, want_solution=True, return_stats=True)
            assert all(t[1] == 1 for t in st), st
            two = np.array([np.linalg.norm(b - A @ x) / np.linalg.norm(b) for (A, b), x in zip(sys_h, sol)])
            out[(storage, tol)] = dict(two=two, Q=np.array(Q), its=[t[0] for t in st], rep=[t[3] / t[2] for t in st])
            ds.close()
    print("darcy true residuals:", {k_: {a: np.asarray(b).tolist() for a, b in v.items()} for k_, v in out.items()})
    for tol in (1e-6, 1e-12):
        a, b = out[(0, tol)], out[(1, tol)]
        assert a["its"] == b["its"]
        assert np.all(np.abs(a["two"] / b["two"] - 1.0) < 1e-2)
        assert np.all(np.abs(a["Q"] - b["Q"]) <= (1e-5 if tol == 1e-6 else 1e-9) * np.abs(b["Q"]))
        # 2-norm of the true residual against the preconditioned norm the recurrence reports: one order at most
        assert np.all(a["two"] < 10.0 * max(tol, 1e-10)) and np.all(b["two"] < 10.0 * max(tol, 1e-10))


def test_fp64_storage_reproduces_the_oracle_on_every_preconditioner_path(gpu_ctx, hex_hierarchy, seeded_rng):
    """precond_storage = PMC_STORAGE_FP64 selects other kernel instantiations on every path (typed last kernels of both
    preconditioner blocks, operator products, w / x updates, LDS tail, persistent small-level solver): all of them against
    the oracle's direct solves, sampler on three levels with warm start, Darcy, M-block degrees 2 / 3 / 4."""
    from oracle.darcy_oracle import DarcyOracle
    from oracle.sampler_oracle import SamplerOracle
    from parelagmc_amd import capi
    from parelagmc_amd.fe import build_darcy_problem, build_sampler_problem
    sp = build_sampler_problem(hex_hierarchy, corlen=0.1, lognormal=True)
    dp = build_darcy_problem(hex_hierarchy, [0, 1, 1, 1, 1, 0], [1, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 1])
    so, do = SamplerOracle(sp), DarcyOracle(dp)
    for deg in (0, 3, 4):
        o = capi.solver_opts(rel_tol=1e-12, abs_tol=1e-300, precond_storage=capi.PMC_STORAGE_FP64, cheb_degree_M=deg)
        smp = capi.PDESampler(gpu_ctx, sp, o)
        ds = capi.DarcySolver(gpu_ctx, dp, o)
        for lvl in range(3):
            xi = seeded_rng.standard_normal((5, sp.levels[lvl].n_s))
            s, emb = smp.Eval(lvl, xi, want_embed=True)
            ref = np.stack([so.eval(lvl, lvl, x)[0] for x in xi])
            assert rel(s, ref) < 1e-9, (deg, lvl)
            if lvl > 0:
                # warm start of the finer level from this field (src/PDESampler.cpp:498-510) through the fp64 path
                xif = seeded_rng.standard_normal((5, sp.levels[lvl - 1].n_s))
                sc, embc = smp.Eval(lvl, xif, xi_level=lvl - 1, want_embed=True)
                sf = smp.Eval(lvl - 1, xif, init_s=embc, init_level=lvl, use_init=True)
                assert rel(sf, np.stack([so.eval(lvl - 1, lvl - 1, x)[0] for x in xif])) < 1e-9
            Q = ds.SolveFwd(lvl, s)[0]
            assert np.allclose(Q, [do.solve_fwd(lvl, kk)[0] for kk in s], rtol=1e-9)
        ds.close()
        smp.close()


def test_abi_handshake_refuses_another_layout(gpu_ctx):
    """pmc_ctx_create_abi (what the header's pmc_ctx_create macro calls) refuses a caller built against another
    PMC_ABI_VERSION before any handle exists - also callers that later pass opts == NULL with a pmc_stats array."""
    import ctypes as C
    lib = gpu_ctx.lib
    h = C.c_void_p()
    assert lib.pmc_ctx_create_abi(0, 2, C.byref(h)) == -1 and not h.value
    assert b"PMC_ABI_VERSION 2" in lib.pmc_last_error()
    assert lib.pmc_ctx_create_abi(0, lib.pmc_abi_version(), C.byref(h)) == 0
    lib.pmc_ctx_destroy(h)


def test_ratio_manager_cuts_its_plugin_calls_like_the_mlmc_manager(gpu_ctx, hex_hierarchy_small):
    """ML_BayesRatio_Manager with the library's default batch (256, pmc_mlmc_params_default) hands the plugins at most what
    they prefer on a level (ML_BayesRatio_Manager::level_batch, as MLMC_Manager::level_batch): same realizations, same
    sums as with 4 per call (src/ML_BayesRatio_Manager.hpp:323-430 is one realization per call)."""
    from oracle.bayes_oracle import compute_G, observation_functionals
    from oracle.darcy_oracle import DarcyOracle
    from oracle.rng_oracle import normal_fill
    from oracle.sampler_oracle import SamplerOracle
    from parelagmc_amd import capi, host_api
    from parelagmc_amd.fe import build_darcy_problem, build_sampler_problem
    h = hex_hierarchy_small
    sp = build_sampler_problem(h, corlen=0.1, lognormal=True)
    dp = build_darcy_problem(h, [0, 1, 1, 1, 1, 0], [1, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 1])
    so, do = SamplerOracle(sp), DarcyOracle(dp)
    Gobs = observation_functionals(h, np.array([[0.5, 0.5, 0.5], [1.4, 1.2, 0.6]]), eps=0.3)
    G_obs = compute_G(do, Gobs, 0, so.eval(0, 0, normal_fill(sp.levels[0].n_s, 20261003, 12345, 0))[0])[0]
    tight = dict(rel_tol=1e-12, abs_tol=1e-300)
    smp = capi.PDESampler(gpu_ctx, sp, capi.solver_opts(**tight))
    ds = capi.DarcySolver(gpu_ctx, dp, capi.solver_opts(**tight))
    for lvl in range(2):
        ds.SetObservations(lvl, Gobs[lvl])
    out = []
    for kw in (dict(), dict(batch=4)):
        mgr = host_api.RatioManager(2, sampler=smp, solver=ds, G_obs=G_obs, noise=0.05, wall_time=False, **kw)
        out.append(mgr.InitRun([300, 520]))          # more than one launch width (256) on both levels
        mgr.close()
    assert list(out[0]["nsamples"]) == list(out[1]["nsamples"]) == [300, 520]
    assert np.allclose(out[0]["sums"], out[1]["sums"], rtol=1e-9, atol=1e-12)
    ds.close()
    smp.close()


def test_sampler_mult_is_the_reference_solver_seam(gpu_ctx, hex_hierarchy, seeded_rng):
    """pmc_sampler_mult = invA[level]->Mult(rhs, sol) (/root/reference/src/PDESampler.cpp:397,521): ARBITRARY right-hand sides
    (a nonzero u-block, which Eval never produces), full [u; s] solutions against the oracle's sparse direct solve of the block
    system, on every level, host pointers; iterative_mode (:510) from a perturbed solution starts from a residual four orders smaller and converges to the
    same vector; the s-block equals what Eval returns for the right-hand side Eval builds."""
    from oracle.sampler_oracle import SamplerOracle
    from parelagmc_amd import capi
    from parelagmc_amd.fe import build_sampler_problem
    import scipy.sparse.linalg as spla
    sp = build_sampler_problem(hex_hierarchy, corlen=0.1)
    so = SamplerOracle(sp)
    smp = capi.PDESampler(gpu_ctx, sp, capi.solver_opts(rel_tol=1e-12, abs_tol=1e-300))
    for lvl in range(3):
        L = sp.levels[lvl]
        n = L.n_u + L.n_s
        rhs = seeded_rng.standard_normal((3, n))
        x, st = smp.Solve(lvl, rhs, return_stats=True)
        assert all(t[1] == 1 for t in st)
        lu = spla.splu(so.block_operator(lvl))
        ref = np.stack([lu.solve(b) for b in rhs])
        assert rel(x, ref) < 1e-9, lvl
        x2, st2 = smp.Solve(lvl, rhs, guess=ref * (1.0 + 1e-4 * seeded_rng.standard_normal(ref.shape)), return_stats=True)
        # the tolerance is relative to the solve's OWN initial residual (as MFEM's): the warm start begins four orders lower
        assert rel(x2, ref) < 1e-9 and all(t[1] == 1 for t in st2)
        assert all(t2[2] < 1e-2 * t1[2] for t1, t2 in zip(st, st2))
        xi = seeded_rng.standard_normal((2, L.n_s))
        b = np.zeros((2, n))
        b[:, L.n_u:] = -sp.matern_g * np.sqrt(L.w_diag) * xi
        assert rel(smp.Solve(lvl, b)[:, L.n_u:], smp.Eval(lvl, xi)) < 1e-9
    smp.close()


@pytest.mark.parametrize("storage", [0, 1])
def test_preconditioner_is_symmetric_positive_definite(gpu_ctx, hex_hierarchy_small, seeded_rng, storage):
    """What MINRES needs of B^-1 (block Jacobi: M-block polynomial | V-cycle; the reference's "BJ-GS" block,
    examples/example_helpers/CreateSamplerParameterList.hpp:68-113): <x, B^-1 y> == <B^-1 x, y> and <x, B^-1 x> > 0, through
    pmc_sampler_apply_preconditioner, for fp32 and fp64 storage inside the preconditioner (fp32: symmetric to rounding)."""
    from parelagmc_amd import capi
    from parelagmc_amd.fe import build_sampler_problem
    sp = build_sampler_problem(hex_hierarchy_small, corlen=0.1)
    smp = capi.PDESampler(gpu_ctx, sp, capi.solver_opts(precond_storage=storage, mini_max_rows=0))
    for lvl in range(2):
        n = sp.levels[lvl].n_u + sp.levels[lvl].n_s
        v = seeded_rng.standard_normal((4, n))
        z = smp.ApplyPreconditioner(lvl, v)
        G = v @ z.T                                   # G[i, j] = <v_i, B^-1 v_j>
        assert np.all(np.diag(G) > 0)
        assert np.abs(G - G.T).max() <= (1e-6 if storage == 0 else 1e-11) * np.abs(G).max()
        assert np.all(np.linalg.eigvalsh(0.5 * (G + G.T)) > 0)
        # linearity
        assert rel(smp.ApplyPreconditioner(lvl, (2.0 * v[0] - 3.0 * v[1])[None])[0], 2.0 * z[0] - 3.0 * z[1]) < (1e-6 if storage == 0 else 1e-12)
    smp.close()
