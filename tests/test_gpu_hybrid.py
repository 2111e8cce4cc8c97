"""GPU tests of the HYBRIDIZED sampler (pmc_sampler_create_hybrid): the reference's alternative solver of PDESampler::Eval
("Hybridization" in the parameter lists, /root/reference/src/PDESampler.cpp:291,307-311,451-480).  The field it returns is the
field of the saddle-point system - the oracle (oracle/sampler_oracle.py, sparse direct solves of [M B^T; B -alpha W]) is the
checker throughout; the hybrid operators come from parelagmc_amd/fe/hybrid.py."""
import numpy as np
import pytest

from conftest import golden_path

pytestmark = pytest.mark.gpu

TIGHT = dict(rel_tol=1e-12, abs_tol=1e-300, max_iter=300)


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(b)


def _pair(h, **kw):
    from parelagmc_amd.fe import build_hybrid_sampler_problem, build_sampler_problem
    return build_sampler_problem(h, **kw), build_hybrid_sampler_problem(h, **kw)


@pytest.mark.parametrize("tol,bound", [(TIGHT, 1e-9), (dict(), 1e-5)])
def test_hybrid_sampler_matches_direct_solve_all_levels(gpu_ctx, hex_hierarchy, seeded_rng, tol, bound):
    """every level, xi drawn on the finest level (restricted with P^T, the last step landing on z f directly) and on the
    level itself - against the oracle's direct solves of the saddle-point system"""
    from oracle.sampler_oracle import SamplerOracle
    from parelagmc_amd import capi
    sp, hp = _pair(hex_hierarchy, corlen=0.1)
    so = SamplerOracle(sp)
    smp = capi.PDESampler(gpu_ctx, hp, capi.solver_opts(**tol))
    assert smp.hybrid and gpu_ctx.lib.pmc_sampler_is_hybrid(smp.h) == 1
    assert smp.GetNNZ(0) == hp.levels[0].H.nnz
    xi = seeded_rng.standard_normal((3, sp.levels[0].n_s))
    for lvl in range(3):
        assert smp.xi_size(lvl) == sp.levels[lvl].n_s == smp.SampleSize(lvl)
        s, st = smp.Eval(lvl, xi, xi_level=0, return_stats=True)
        ref = np.stack([so.eval(lvl, 0, x)[0] for x in xi])
        assert rel(s, ref) < bound
        assert all(t[1] == 1 for t in st) and all(0 < t[0] <= 300 for t in st)
    for lvl in (1, 2):
        x = seeded_rng.standard_normal((2, sp.levels[lvl].n_s))
        assert rel(smp.Eval(lvl, x), np.stack([so.eval(lvl, lvl, v)[0] for v in x])) < bound
    x = seeded_rng.standard_normal((2, sp.levels[1].n_s))      # xi of level 1 evaluated on level 2
    assert rel(smp.Eval(2, x, xi_level=1), np.stack([so.eval(2, 1, v)[0] for v in x])) < bound
    # GetTrueP hands back what was given
    assert abs(smp.GetTrueP(0) - hp.levels[0].P).max() == 0.0
    smp.close()


@pytest.mark.parametrize("mesh,nref", [("cube_tet", 2), ("inline_quad", 3)])
def test_hybrid_sampler_on_simplices_and_in_2d(gpu_ctx, seeded_rng, mesh, nref):
    """tetrahedra (H has positive couplings across obtuse dihedral angles: the aggregation matches by magnitude) and the 2D
    quadrilateral mesh of config 1"""
    from oracle.sampler_oracle import SamplerOracle
    from parelagmc_amd import capi
    from parelagmc_amd.fe import build_hierarchy, mesh_from_json
    h = build_hierarchy(mesh_from_json(golden_path("meshes", mesh + ".json")), nref)
    sp, hp = _pair(h, corlen=0.1, n_mc_levels=2)
    so = SamplerOracle(sp)
    smp = capi.PDESampler(gpu_ctx, hp, capi.solver_opts(**TIGHT))
    xi = seeded_rng.standard_normal((2, sp.levels[0].n_s))
    for lvl in range(2):
        s, st = smp.Eval(lvl, xi, xi_level=0, return_stats=True)
        assert rel(s, np.stack([so.eval(lvl, 0, x)[0] for x in xi])) < 1e-9
        assert all(t[1] == 1 for t in st)
    smp.close()


def test_hybrid_embedded_gather_l2_projection_lognormal_and_embed_output(gpu_ctx, seeded_rng):
    from oracle.sampler_oracle import SamplerOracle
    from parelagmc_amd import capi
    from parelagmc_amd.fe import box_mesh, build_hierarchy, l2_projection_ops
    m = box_mesh([6, 6, 6], [3.0, 3.0, 3.0], "hex", origin=[-0.5, -0.5, -0.5])
    cen = m.verts[m.elems].mean(1)
    m.elem_attr[:] = np.where(np.all((cen > 0) & (cen < 2), axis=1), 1, 2)
    h = build_hierarchy(m, 1)
    sp, hp = _pair(h, corlen=0.1, embedded=True, lognormal=True)
    assert all(np.array_equal(a, b) for a, b in zip(sp.orig_index, hp.orig_index))
    so = SamplerOracle(sp)
    l2 = l2_projection_ops(h, sp.orig_index)
    ga = capi.PDESampler(gpu_ctx, hp, capi.solver_opts(**TIGHT), projection="gather")
    pr = capi.PDESampler(gpu_ctx, hp, capi.solver_opts(**TIGHT), projection="l2", l2_ops=l2)
    xi = seeded_rng.standard_normal((3, sp.levels[0].n_s))
    for lvl in range(2):
        assert ga.SampleSize(lvl) == len(sp.orig_index[lvl])
        a, emb = ga.Eval(lvl, xi, xi_level=0, want_embed=True)
        b = pr.Eval(lvl, xi, xi_level=0)
        ref = np.stack([so.eval(lvl, 0, x, projection=("gather", sp.orig_index[lvl]))[0] for x in xi])
        assert rel(a, ref) < 1e-9 and rel(b, ref) < 1e-9
        gauss = np.stack([so.eval_gaussian(lvl, 0, x) for x in xi])       # embed_s: the Gaussian field on the sampler mesh
        assert rel(emb, gauss) < 1e-9
    # use_init is accepted and ignored (the reference: "The HybridizationSolver cannot be used in iterative mode")
    a0 = ga.Eval(0, xi, xi_level=0)
    a1 = ga.Eval(0, xi, xi_level=0, init_s=np.zeros((3, sp.levels[1].n_s)), init_level=1, use_init=True)
    assert np.array_equal(a0, a1)
    ga.close()
    pr.close()


@pytest.mark.parametrize("nbatch", [1, 3, 32, 37, 70])
def test_hybrid_ragged_batches_equal_single_evaluations(gpu_ctx, hex_hierarchy_small, nbatch):
    from parelagmc_amd import capi
    from parelagmc_amd.fe import build_hybrid_sampler_problem
    hp = build_hybrid_sampler_problem(hex_hierarchy_small, corlen=0.1)
    smp = capi.PDESampler(gpu_ctx, hp, capi.solver_opts(**TIGHT))
    xi = smp.Sample(0, first_id=11, nbatch=nbatch)
    s = smp.Eval(0, xi)
    for k in sorted({0, nbatch // 2, nbatch - 1}):
        assert rel(s[k], smp.Eval(0, xi[k:k + 1])[0]) < 1e-10
    # device-resident operands give the same bits as host operands
    d = gpu_ctx.empty(nbatch * smp.SampleSize(0))
    smp.Eval(0, gpu_ctx.array(xi), xi_level=0, s_out=d)
    assert np.array_equal(d.download().reshape(nbatch, -1), s)
    smp.close()


@pytest.mark.parametrize("storage", ["fp32", "fp64"])
def test_hybrid_true_residual_and_field_at_full_size(gpu_ctx, storage):
    """cube_tet r = 5 (399 360 multipliers, 196 608 elements; BASELINE config 2): lambda = H^-1 G f from the solver seam
    (pmc_sampler_mult), TRUE residual formed with the fp64 operator kernel, and the field against the saddle-point sampler
    of the same handle family on the same xi, both at 1e-10."""
    from parelagmc_amd import capi
    from parelagmc_amd.fe import build_hierarchy, mesh_from_json
    h = build_hierarchy(mesh_from_json(golden_path("meshes", "cube_tet.json")), 5)
    sp, hp = _pair(h, corlen=0.1, n_mc_levels=1)
    L = hp.levels[0]
    st_ = capi.PMC_STORAGE_FP32 if storage == "fp32" else capi.PMC_STORAGE_FP64
    o = capi.solver_opts(rel_tol=1e-10, abs_tol=1e-300, precond_storage=st_)
    hy = capi.PDESampler(gpu_ctx, hp, o)
    sa = capi.PDESampler(gpu_ctx, sp, o)
    assert hy.z_bytes() == (4 if storage == "fp32" else 8)
    xi = hy.Sample(0, first_id=3, nbatch=2)
    f = -hp.matern_g * np.sqrt(L.w_diag) * xi
    rhs = (L.G @ f.T).T.copy()
    lam, st = hy.Solve(0, rhs, return_stats=True)
    assert all(t[1] == 1 for t in st) and all(t[0] < 60 for t in st), st
    Hl = hy.Mult(0, lam)[0]
    assert rel(Hl[0], L.H @ lam[0]) < 1e-13                       # the device operator is H
    r = rhs - Hl
    z = hy.ApplyPreconditioner(0, r)
    pnorm = np.sqrt(np.einsum("ij,ij->i", r, z))
    eta0, eta = np.array([t[2] for t in st]), np.array([t[3] for t in st])
    assert np.all(pnorm / eta0 <= 1e-10) and np.all(np.abs(pnorm / eta - 1.0) < 1e-2), (pnorm, eta, eta0)
    # element-local back-substitution of the solver's lambda = what Eval returns
    s_h, sth = hy.Eval(0, xi, return_stats=True)
    assert rel(s_h, L.z_diag * f - (L.G.T @ lam.T).T) < 1e-12
    s_s, sts = sa.Eval(0, xi, return_stats=True)
    assert rel(s_h, s_s) < 1e-8
    print("hybrid r=5 %s: iterations %s against %s of the saddle-point solve at 1e-10" %
          (storage, [t[0] for t in sth], [t[0] for t in sts]))
    assert max(t[0] for t in sth) * 1.7 < min(t[0] for t in sts)   # measured 31-35 against 72
    hy.close()
    sa.close()


def test_hybrid_preconditioner_is_symmetric_positive_definite(gpu_ctx, hex_hierarchy_small, seeded_rng):
    from parelagmc_amd import capi
    from parelagmc_amd.fe import build_hybrid_sampler_problem
    hp = build_hybrid_sampler_problem(hex_hierarchy_small, corlen=0.1)
    smp = capi.PDESampler(gpu_ctx, hp, capi.solver_opts(precond_storage=capi.PMC_STORAGE_FP64))
    n = hp.levels[0].n_lambda
    a, b = seeded_rng.standard_normal((2, n))
    za, zb = smp.ApplyPreconditioner(0, np.stack([a, b]))
    assert abs(a @ zb - b @ za) < 1e-10 * abs(a @ zb)
    assert a @ za > 0 and b @ zb > 0
    smp.close()


def test_mlmc_manager_runs_unchanged_on_a_hybrid_sampler(gpu_ctx, hex_hierarchy_small):
    """the managers see a pmc_sampler: the same InitRun with the hybridized handle gives the sums of the saddle-point handle
    (same generator, same realizations) to the solver tolerance"""
    from parelagmc_amd import capi, host_api
    from parelagmc_amd.fe import build_darcy_problem
    sp, hp = _pair(hex_hierarchy_small, corlen=0.1, lognormal=True)
    dp = build_darcy_problem(hex_hierarchy_small, [0, 1, 1, 1, 1, 0], [1, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 1])
    ds = capi.DarcySolver(gpu_ctx, dp, capi.solver_opts(**TIGHT))
    out = []
    for prob in (sp, hp):
        smp = capi.PDESampler(gpu_ctx, prob, capi.solver_opts(**TIGHT))
        mgr = host_api.MLMCManager(2, sampler=smp, solver=ds, wall_time=False, batch=4)
        out.append(mgr.InitRun([5, 9]))
        mgr.close()
        smp.close()
    ds.close()
    assert np.allclose(out[0]["sums"], out[1]["sums"], rtol=1e-7, atol=1e-9)
    assert out[0]["estimate"] == pytest.approx(out[1]["estimate"], rel=1e-7)


def test_hybrid_error_paths(gpu_ctx, hex_hierarchy_small):
    import copy
    from parelagmc_amd import capi
    from parelagmc_amd.fe import build_hybrid_sampler_problem
    hp = build_hybrid_sampler_problem(hex_hierarchy_small, corlen=0.1)
    bad = copy.copy(hp)
    bad.levels = [copy.copy(hp.levels[0]), hp.levels[1]]
    bad.levels[0].G = hp.levels[0].G[:-1]
    with pytest.raises(capi.PmcError):
        capi.PDESampler(gpu_ctx, bad)
    bad.levels[0] = copy.copy(hp.levels[0])
    bad.levels[0].z_diag = np.zeros_like(hp.levels[0].z_diag)
    with pytest.raises(capi.PmcError):
        capi.PDESampler(gpu_ctx, bad)
    smp = capi.PDESampler(gpu_ctx, hp)
    with pytest.raises(capi.PmcError):
        smp.Eval(2, np.zeros((1, hp.levels[0].n_s)), xi_level=0)
    smp.close()


def test_hybrid_hipgraph_replay_and_wide_column_groups_give_the_eager_field(gpu_ctx, hex_hierarchy_small):
    """opts.use_graph captures the MINRES iterations of the multiplier solve as it does the saddle-point one; a small level
    runs 64 ... 256 realizations per launch as column groups of 32: both must return the eager single-group bits / field"""
    from parelagmc_amd import capi
    from parelagmc_amd.fe import build_hybrid_sampler_problem
    hp = build_hybrid_sampler_problem(hex_hierarchy_small, corlen=0.1)
    eager = capi.PDESampler(gpu_ctx, hp, capi.solver_opts(**TIGHT))
    graph = capi.PDESampler(gpu_ctx, hp, capi.solver_opts(use_graph=1, **TIGHT))
    w = eager.BatchWidth(1)
    assert w >= 64                                   # 8^3 ... 4^3 hexes: far below the width-32 threshold
    xi = eager.Sample(1, first_id=5, nbatch=w)
    a, st = eager.Eval(1, xi, return_stats=True)
    assert all(t[1] == 1 for t in st)
    for _ in range(2):                               # capture, then replay
        b = graph.Eval(1, xi)
        assert rel(b, a) < 1e-10
    for k in (0, w // 2 + 1, w - 1):                 # a column of the wide launch against its own launch of one
        assert rel(a[k], eager.Eval(1, xi[k:k + 1])[0]) < 1e-10
    eager.close()
    graph.close()


def test_hybrid_config4_full_size_field_equals_the_saddle_point_field(gpu_ctx):
    """BASELINE config 4 at full size (cube_tet_embed refined 4 x: 2 502 400 / 313 792 DoF on the two finest levels, embedded
    gather, lognormal): the hybridized handle bench.py reports under extra.c4 returns the field of the default solver on the
    same xi (both at 1e-9), in a third of the iterations on these badly shaped cells"""
    from parelagmc_amd import capi
    from parelagmc_amd.fe import build_hierarchy, mesh_from_json
    h = build_hierarchy(mesh_from_json(golden_path("meshes", "cube_tet_embed.json")), 4)
    sp, hp = _pair(h, corlen=0.1, embedded=True, lognormal=True, n_mc_levels=2)
    o = capi.solver_opts(rel_tol=1e-9, abs_tol=1e-300)
    hy = capi.PDESampler(gpu_ctx, hp, o, projection="gather")
    sa = capi.PDESampler(gpu_ctx, sp, o, projection="gather")
    xi = hy.Sample(0, first_id=21, nbatch=2)
    for lvl in (0, 1):
        a, sta = hy.Eval(lvl, xi, xi_level=0, return_stats=True)
        b, stb = sa.Eval(lvl, xi, xi_level=0, return_stats=True)
        assert a.shape == (2, len(sp.orig_index[lvl]))
        assert all(t[1] == 1 for t in sta) and all(t[1] == 1 for t in stb)
        assert rel(np.log(a), np.log(b)) < 1e-6
        assert max(t[0] for t in sta) * 2 < min(t[0] for t in stb), (sta, stb)
    hy.close()
    sa.close()


def test_hybrid_ragged_remainder_takes_the_late_tail_and_gives_the_same_field(gpu_ctx):
    """cube_tet r = 4 (50 688 multipliers): the first coarse level of the aggregation hierarchy has more than 4 096 rows, so a
    launch of at most 8 realizations - the remainder chunk of a ragged call, or the drop-in path's single realization - runs
    that level as kernels and starts the LDS tail one level further down (Multigrid::tail_later_nb, set for this hierarchy
    only), while a full launch starts the tail there.  Different kernels, same preconditioner up to rounding: every realization
    of a ragged call must equal the same realization evaluated alone and inside a full launch (1e-9 at rel 1e-12; within the
    solver tolerance at the default 1e-6)."""
    from parelagmc_amd import capi
    from parelagmc_amd.fe import build_hierarchy, build_hybrid_sampler_problem, mesh_from_json
    h = build_hierarchy(mesh_from_json(golden_path("meshes", "cube_tet.json")), 4)
    hp = build_hybrid_sampler_problem(h, corlen=0.1, n_mc_levels=1, builder=capi.library_hybrid_builder)
    for opts, tol in ((capi.solver_opts(rel_tol=1e-12, abs_tol=1e-300), 1e-9), (None, 2e-5)):
        smp = capi.PDESampler(gpu_ctx, hp, opts)
        lv = smp.vcycle_levels(0)
        assert lv[1]["rows"] > 4096 and lv[1]["in_tail"] == 1 and lv[0]["fused_restriction"] == 1, lv
        # ... and narrow launches run that level with its rows cut into pieces and end on the exact solve of the next one
        assert lv[1]["narrow_pieces"] >= 2 and lv[2]["narrow_dense"] == 1 and lv[2]["rows"] <= 768, lv
        w = smp.BatchWidth(0)
        n = w + 3                                           # chunks of w, 2 and 1
        xi = smp.Sample(0, first_id=77, nbatch=n)
        s, st = smp.Eval(0, xi, return_stats=True)
        assert all(t[1] == 1 for t in st)
        full = smp.Eval(0, xi[:w])
        for k in (0, w - 1):                                # members of the full launch
            assert rel(s[k], full[k]) == 0.0
        for k in (w, w + 1, w + 2, 0, w // 2):              # remainder chunks (late tail) and launch members, each alone
            assert rel(s[k], smp.Eval(0, xi[k:k + 1])[0]) < tol
        # a member of the full launch against the same realization in a narrow chunk of 2 (late tail)
        assert rel(full[3], smp.Eval(0, xi[2:4])[1]) < tol
        for m in (4, 8):                                    # the 4- and 8-wide forms of the row-split kernels
            assert rel(smp.Eval(0, xi[:m]), full[:m]) < tol
        smp.close()


@pytest.mark.parametrize("solver", ["hybridization", "saddle-point"])
def test_both_device_solvers_reproduce_the_tetrahedral_golden_vectors(gpu_ctx, solver):
    """tests/golden/gold_sampler_tet.npz (cube_tet refined 3 x / 2 x; fields from oracle/fe_ref.py's independent tetrahedral
    operators): the HIP path against COMMITTED vectors on the headline's mesh family - fine level, coarse level from a fine xi
    (the P^T coupling) and coarse level from its own xi; 1e-9 at rel 1e-12, 1e-5 at the default 1e-6.  The hybridized handle is
    built from the element arrays by the library itself (pmc_hybrid_build)."""
    from parelagmc_amd import capi
    from parelagmc_amd.fe import build_hierarchy, build_hybrid_sampler_problem, build_sampler_problem, mesh_from_json
    g = np.load(golden_path("gold_sampler_tet.npz"))
    h = build_hierarchy(mesh_from_json(golden_path("meshes", "cube_tet.json")), 3)
    kw = dict(corlen=0.1, n_mc_levels=2)
    prob = (build_hybrid_sampler_problem(h, builder=capi.library_hybrid_builder, **kw) if solver == "hybridization"
            else build_sampler_problem(h, **kw))
    for opts, tol in ((capi.solver_opts(**TIGHT), 1e-9), (None, 1e-5)):
        smp = capi.PDESampler(gpu_ctx, prob, opts)
        for key, lvl, xl, xi in (("s00", 0, 0, g["xi0"]), ("s10", 1, 0, g["xi0"]), ("s11", 1, 1, g["xi1"])):
            s, st = smp.Eval(lvl, xi, xi_level=xl, return_stats=True)
            assert all(t[1] == 1 for t in st)
            assert rel(s, g[key]) < tol, (solver, key, rel(s, g[key]))
        smp.close()


def test_vcycle_info_reports_the_hierarchy_the_sampler_runs(gpu_ctx, hex_hierarchy_small):
    """pmc_sampler_vcycle_info (what scripts/roofline_table.py prices the per-kernel table with): levels shrink, the finest level
    of a hybridized handle is H itself and has its restriction fused (fp32 storage) or not (fp64 storage: generic V-cycle), the
    saddle-point handle reports its Schur-complement hierarchy; out-of-range queries are errors, not garbage."""
    import ctypes as C
    from parelagmc_amd import capi
    from parelagmc_amd.fe import build_hybrid_sampler_problem, build_sampler_problem
    hp = build_hybrid_sampler_problem(hex_hierarchy_small, corlen=0.1, builder=capi.library_hybrid_builder)
    for storage, fused in ((capi.PMC_STORAGE_FP32, 1), (capi.PMC_STORAGE_FP64, 0)):
        smp = capi.PDESampler(gpu_ctx, hp, capi.solver_opts(precond_storage=storage))
        lv = smp.vcycle_levels(0)
        assert lv[0]["rows"] == hp.levels[0].n_lambda and lv[0]["nnz"] == hp.levels[0].H.nnz
        assert all(a["rows"] > b["rows"] for a, b in zip(lv, lv[1:])) and lv[0]["slots"] >= lv[0]["nnz"]
        assert lv[0]["fused_restriction"] == fused and lv[0]["sp_nnz"] > 0
        nv, info = C.c_int(0), (C.c_int64 * 7)()
        assert gpu_ctx.lib.pmc_sampler_vcycle_info(smp.h, 0, len(lv), C.byref(nv), info) != 0      # vlevel out of range
        assert gpu_ctx.lib.pmc_sampler_vcycle_info(smp.h, 9, 0, C.byref(nv), info) != 0            # level out of range
        smp.close()
    sa = capi.PDESampler(gpu_ctx, build_sampler_problem(hex_hierarchy_small, corlen=0.1))
    lv = sa.vcycle_levels(0)
    assert lv[0]["rows"] == hex_hierarchy_small.spaces[0].n_s and len(lv) == 2
    sa.close()
