"""The hybridized Darcy solver on the device (pmc_darcy_create_hybrid: the reference's "Hybridization" branch of DarcySolver,
src/DarcySolver.cpp:586,619) against the oracle's saddle-point direct solve and against the default device solver.  Run with
-m gpu on an MI355X."""
import numpy as np
import pytest

from conftest import golden_path

pytestmark = pytest.mark.gpu

TIGHT = dict(rel_tol=1e-12, abs_tol=1e-30, max_iter=400)


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(b)


@pytest.mark.parametrize("k_divides", [True, False])
def test_hybrid_darcy_matches_direct_solve_all_levels(gpu_ctx, hex_hierarchy, seeded_rng, k_divides):
    """flux, pressure and QoI of every level against the direct solve: 1e-8 at rel 1e-12, QoI 1e-4 at the default 1e-6; ragged
    batch (3 = 2 + 1); all converged"""
    from oracle.darcy_oracle import DarcyOracle
    from parelagmc_amd import capi
    from parelagmc_amd.fe import build_darcy_problem
    dp = build_darcy_problem(hex_hierarchy, [0, 1, 1, 1, 1, 0], [1, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 1], k_divides=k_divides)
    do = DarcyOracle(dp)
    for opts, qtol, stol in ((capi.solver_opts(**TIGHT), 1e-9, 1e-8), (capi.solver_opts(), 1e-4, 1e-3)):
        ds = capi.DarcySolver(gpu_ctx, dp, opts, hybrid=True)
        for lvl in range(3):
            k = np.exp(seeded_rng.standard_normal((3, dp.levels[lvl].n_p)))
            Q, C, sol, st = ds.SolveFwd(lvl, k, want_solution=True, return_stats=True)
            for b in range(3):
                Qr, Cr, sr = do.solve_fwd(lvl, k[b], return_solution=True)
                assert abs(Q[b] - Qr) < qtol * abs(Qr) and C[b] == Cr
                assert rel(sol[b], sr) < stol
            assert all(t[1] == 1 for t in st)
        ds.close()


def test_hybrid_darcy_general_boundary_data_and_volume_qoi(gpu_ctx, hex_hierarchy_small, seeded_rng):
    """nonzero essential fluxes, a volume source, the pressure-integral QoI: the right-hand side and back-substitution pieces
    that the drivers' default data leave zero"""
    from oracle.darcy_oracle import DarcyOracle
    from parelagmc_amd import capi
    from parelagmc_amd.fe import build_darcy_problem
    dp = build_darcy_problem(hex_hierarchy_small, [0, 1, 1, 1, 1, 0], [1, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 1], qoi="p_int")
    for L in dp.levels:
        L.ess_data = 0.05 * seeded_rng.standard_normal(L.n_u) * L.ess_mask
        L.rhs[L.n_u:] = 0.1 * seeded_rng.standard_normal(L.n_p)
    do = DarcyOracle(dp)
    ds = capi.DarcySolver(gpu_ctx, dp, capi.solver_opts(**TIGHT), hybrid=True)
    for lvl in range(2):
        k = np.exp(seeded_rng.standard_normal((2, dp.levels[lvl].n_p)))
        Q, _, sol = ds.SolveFwd(lvl, k, want_solution=True)
        for b in range(2):
            Qr, _, sr = do.solve_fwd(lvl, k[b], return_solution=True)
            assert abs(Q[b] - Qr) < 1e-8 * max(1.0, abs(Qr)) and rel(sol[b], sr) < 1e-8
    ds.close()


def test_hybrid_darcy_on_tetrahedra_and_against_the_default_solver(gpu_ctx, seeded_rng):
    """cube_tet refined twice with the boundary relabelled by position (inflow x = min, outflow x = max, no-flux elsewhere): the
    hybridized and the saddle-point device solvers against the oracle and each other"""
    from oracle.darcy_oracle import DarcyOracle
    from parelagmc_amd import capi
    from parelagmc_amd.fe import build_darcy_problem, build_hierarchy, mesh_from_json
    m = mesh_from_json(golden_path("meshes", "cube_tet.json"))
    cen = m.verts[m.bdr].mean(axis=1)
    lo, hi = m.verts[:, 0].min(), m.verts[:, 0].max()
    m.bdr_attr = np.where(np.isclose(cen[:, 0], lo), 1, np.where(np.isclose(cen[:, 0], hi), 6, 2)).astype(m.bdr_attr.dtype)
    h = build_hierarchy(m, 2)
    dp = build_darcy_problem(h, [0, 1, 1, 1, 1, 0], [1, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 1], n_mc_levels=2)
    do = DarcyOracle(dp)
    hy = capi.DarcySolver(gpu_ctx, dp, capi.solver_opts(**TIGHT), hybrid=True)
    sa = capi.DarcySolver(gpu_ctx, dp, capi.solver_opts(**TIGHT))
    for lvl in range(2):
        k = np.exp(seeded_rng.standard_normal((5, dp.levels[lvl].n_p)))
        Qh, _, sh, st = hy.SolveFwd(lvl, k, want_solution=True, return_stats=True)
        Qs, _, ss = sa.SolveFwd(lvl, k, want_solution=True)
        assert all(t[1] == 1 for t in st)
        for b in range(5):
            Qr, _, sr = do.solve_fwd(lvl, k[b], return_solution=True)
            assert abs(Qh[b] - Qr) < 1e-8 * abs(Qr) and rel(sh[b], sr) < 1e-8
            assert rel(sh[b], ss[b]) < 1e-7
    hy.close()
    sa.close()


def test_hybrid_darcy_on_stretched_cells(gpu_ctx, seeded_rng):
    """SPE10-shaped cells (1200 x 2200 x 170 box, 7 x 27 x 10 coarse cells refined once: aspect ratio ~10): the aggregation
    hierarchy is built from the coupling magnitudes of H(1), so the anisotropy is followed; converged, QoI against the oracle"""
    from oracle.darcy_oracle import DarcyOracle
    from parelagmc_amd import capi
    from parelagmc_amd.fe import box_mesh, build_darcy_problem, build_hierarchy
    h = build_hierarchy(box_mesh([7, 27, 10], [1200.0, 2200.0, 170.0], "hex"), 1)
    dp = build_darcy_problem(h, [1, 0, 1, 0, 1, 1], [0, 1, 0, 0, 0, 0], [0, 0, 0, 1, 0, 0], n_mc_levels=1)
    do = DarcyOracle(dp)
    k = np.exp(1.2 * seeded_rng.standard_normal((4, dp.levels[0].n_p)))
    for opts, qtol in ((capi.solver_opts(**TIGHT), 1e-8), (capi.solver_opts(), 1e-4)):
        ds = capi.DarcySolver(gpu_ctx, dp, opts, hybrid=True)
        Q, _, st = ds.SolveFwd(0, k, return_stats=True)
        assert all(t[1] == 1 for t in st), st
        for b in range(4):
            assert abs(Q[b] - do.solve_fwd(0, k[b])[0]) < qtol * abs(Q[b])
        ds.close()


def test_mlmc_manager_runs_unchanged_on_a_hybridized_darcy_solver(gpu_ctx, hex_hierarchy_small):
    """the managers see a pmc_darcy: the same InitRun (same generator, same realizations, level pairs with warm starts) through
    the hybridized handle gives the accumulators of the default handle to the solver tolerance; ragged plugin batches included"""
    from parelagmc_amd import capi, host_api
    from parelagmc_amd.fe import build_darcy_problem, build_sampler_problem
    sp = build_sampler_problem(hex_hierarchy_small, corlen=0.1, lognormal=True)
    dp = build_darcy_problem(hex_hierarchy_small, [0, 1, 1, 1, 1, 0], [1, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 1])
    smp = capi.PDESampler(gpu_ctx, sp, capi.solver_opts(**TIGHT))
    out = []
    for hybrid in (False, True):
        ds = capi.DarcySolver(gpu_ctx, dp, capi.solver_opts(**TIGHT), hybrid=hybrid)
        mgr = host_api.MLMCManager(2, sampler=smp, solver=ds, wall_time=False, batch=4)
        out.append(mgr.InitRun([5, 9]))
        mgr.close()
        ds.close()
    smp.close()
    assert np.allclose(out[0]["sums"], out[1]["sums"], rtol=1e-7, atol=1e-9)
    assert out[0]["estimate"] == pytest.approx(out[1]["estimate"], rel=1e-7)


def test_hybrid_darcy_hipgraph_replay_gives_the_eager_result(gpu_ctx, hex_hierarchy, seeded_rng):
    """pairs of MINRES iterations replayed as one hipGraph (pmc_solver_opts.use_graph): the element-grouped cycle of the
    hybridized solver is captured like any other preconditioner - same QoI, same iteration counts"""
    from parelagmc_amd import capi
    from parelagmc_amd.fe import build_darcy_problem
    dp = build_darcy_problem(hex_hierarchy, [0, 1, 1, 1, 1, 0], [1, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 1], n_mc_levels=1)
    k = np.exp(seeded_rng.standard_normal((6, dp.levels[0].n_p)))
    out = []
    for g in (0, 1):
        ds = capi.DarcySolver(gpu_ctx, dp, capi.solver_opts(use_graph=g), hybrid=True)
        Q, _, st = ds.SolveFwd(0, k, return_stats=True)
        Q2, _ = ds.SolveFwd(0, k)                     # second call: the captured graph is replayed from the first pair on
        out.append((Q, [t[0] for t in st], Q2))
        ds.close()
    assert np.array_equal(out[0][0], out[1][0]) and out[0][1] == out[1][1]
    assert np.array_equal(out[1][0], out[1][2])
