"""Row f3 of SURVEY.md 8: the device path on ALGEBRAICALLY AGGLOMERATED hierarchies - what ParELAG hands over when it
coarsens with METIS (/root/reference/src/Utilities.cpp:125-155, src/PDESampler.cpp:189-193) instead of nested refinement:
agglomerates of 3 to ~20 elements, coarse flux dofs that collect any number of fine faces, coarse rows of 10-19 entries,
P_s with variable children per parent (unit and non-unit weights), coarse mass matrices whose dofs touch MORE than two
coefficient entries (the element-grouped fast path of M(k) does not apply there).  The operator sets come from
parelagmc_amd/fe/agglomerate.py (greedy agglomeration + Galerkin products); the oracle solves the same matrices directly.
Both V-cycle modes: the caller's (agglomerated) levels and the internally built smoothed-aggregation hierarchy."""
import numpy as np
import pytest

from conftest import golden_path

pytestmark = pytest.mark.gpu

TIGHT = dict(rel_tol=1e-12, abs_tol=1e-30, max_iter=600)


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(b)


@pytest.fixture(scope="module")
def tet_spaces():
    from parelagmc_amd.fe import build_spaces, mesh_from_json, refine_uniform
    m = mesh_from_json(golden_path("meshes", "cube_tet.json"))
    for _ in range(3):
        m = refine_uniform(m)[0]
    # six boundary attributes as on MFEM's cube: 1 z=0, 2 y=0, 3 x=1, 4 y=1, 5 x=0, 6 z=1
    c = m.verts[m.bdr].mean(axis=1)
    attr = np.zeros(len(m.bdr), np.int32)
    for a, (ax, val) in enumerate([(2, 0.0), (1, 0.0), (0, 1.0), (1, 1.0), (0, 0.0), (2, 1.0)], start=1):
        attr[np.abs(c[:, ax] - val) < 1e-12] = a
    assert attr.min() == 1
    m.bdr_attr = attr
    return build_spaces(m)


@pytest.mark.parametrize("ps_weights", ["unit", "nonunit"])
@pytest.mark.parametrize("mg", [0, 1])
def test_sampler_on_agglomerated_levels(gpu_ctx, tet_spaces, seeded_rng, ps_weights, mg):
    from oracle.sampler_oracle import SamplerOracle
    from parelagmc_amd import capi
    from parelagmc_amd.fe.agglomerate import build_agglomerated_sampler_problem
    sp_ = build_agglomerated_sampler_problem(tet_spaces, 3, corlen=0.3, ps_weights=ps_weights, lognormal=True)
    sizes = np.bincount(np.asarray(sp_.levels[0].P.tocsr().indices))
    assert sizes.min() >= 3 and sizes.max() >= 3 * sizes.min()                 # non-uniform agglomerates
    assert np.diff(sp_.levels[1].M.indptr).max() > 9 and np.diff(sp_.levels[1].B.indptr).max() > 8
    so = SamplerOracle(sp_)
    smp = capi.PDESampler(gpu_ctx, sp_, capi.solver_opts(mg_coarsening=mg, **TIGHT))
    xi = seeded_rng.standard_normal((5, sp_.levels[0].n_s))
    emb = None
    for lvl in (2, 1, 0):                      # the managers' order: coarse first, its field warm-starts the finer solve
        if emb is None:
            s, emb, st = smp.Eval(lvl, xi, xi_level=0, want_embed=True, return_stats=True)
        else:
            s, emb, st = smp.Eval(lvl, xi, xi_level=0, init_s=emb, init_level=lvl + 1, use_init=True, want_embed=True,
                                  return_stats=True)
        ref = np.stack([so.eval(lvl, 0, x)[0] for x in xi])
        assert rel(s, ref) < 1e-8, (lvl, rel(s, ref))
        assert all(t[1] == 1 for t in st)
    for lvl in (1, 2):                         # xi drawn on the level itself
        x = seeded_rng.standard_normal((2, sp_.levels[lvl].n_s))
        assert rel(smp.Eval(lvl, x), np.stack([so.eval(lvl, lvl, v)[0] for v in x])) < 1e-8
    P = smp.GetTrueP(0)
    assert (abs(P - sp_.levels[0].P)).max() == 0.0
    smp.close()


@pytest.mark.parametrize("smooth", [0.0, 0.5])
@pytest.mark.parametrize("mg", [0, 1])
def test_darcy_and_mlmc_on_agglomerated_levels(gpu_ctx, tet_spaces, seeded_rng, smooth, mg):
    """smooth = 0.5: one damped Jacobi step on the coarsest prolongator - its flux dofs then belong to up to ~11
    agglomerates, so M(k) on that level runs through the general per-realization-values path instead of the element-grouped
    one."""
    from oracle import mlmc_oracle as mo
    from oracle.darcy_oracle import DarcyOracle
    from oracle.rng_oracle import normal_fill
    from oracle.sampler_oracle import SamplerOracle
    from parelagmc_amd import capi, host_api
    from parelagmc_amd.fe.agglomerate import build_agglomerated_darcy_problem, build_agglomerated_sampler_problem
    bc = ([0, 1, 1, 1, 1, 0], [1, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 1])
    dp = build_agglomerated_darcy_problem(tet_spaces, 3, *bc, smooth_pu=smooth)
    sp_ = build_agglomerated_sampler_problem(tet_spaces, 3, corlen=0.3, lognormal=True, smooth_pu=smooth)
    assert [L.n_s for L in sp_.levels] == [L.n_p for L in dp.levels]
    do, so = DarcyOracle(dp), SamplerOracle(sp_)
    o = capi.solver_opts(mg_coarsening=mg, **TIGHT)
    ds = capi.DarcySolver(gpu_ctx, dp, o)
    for lvl in range(3):
        k = np.exp(0.6 * seeded_rng.standard_normal((4, dp.levels[lvl].n_p)))
        Q, C, st = ds.SolveFwd(lvl, k, return_stats=True)
        Qr = np.array([do.solve_fwd(lvl, kk)[0] for kk in k])
        assert np.allclose(Q, Qr, rtol=1e-7), (lvl, Q, Qr)
        assert all(t[1] == 1 for t in st) and np.all(C == dp.levels[lvl].ndofs)
    assert abs(ds.SolveFwd(0, np.ones((1, dp.levels[0].n_p)))[0][0] - 1.0) < 1e-8      # unit cube, k == 1: unit flux
    # MLMC level pairs through the manager against the same loop with the oracle
    smp = capi.PDESampler(gpu_ctx, sp_, o)
    mgr = host_api.MLMCManager(3, sampler=smp, solver=ds, wall_time=False, batch=4)
    ns = [3, 5, 6]
    r = mgr.InitRun(ns)
    sums = np.zeros((3, mo.NVAR))
    for lvl in (2, 1, 0):
        for i in range(ns[lvl]):
            xi = normal_fill(sp_.levels[lvl].n_s, 20261003, i, lvl)
            q, c = do.solve_fwd(lvl, so.eval(lvl, lvl, xi)[0])
            if lvl == 2:
                mo.accumulate(sums, lvl, q, q, c)
            else:
                qc, cc = do.solve_fwd(lvl + 1, so.eval(lvl + 1, lvl, xi)[0])
                mo.accumulate(sums, lvl, q - qc, q, c + cc)
    assert np.allclose(r["sums"], sums, rtol=1e-6, atol=1e-8)
    mgr.close()
    smp.close()
    ds.close()
