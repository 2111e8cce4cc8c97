"""oracle/fe_ref.py (closed-form Cartesian-hex RT0/P0 operators, no import of parelagmc_amd.fe) against the product's
builders, entry by entry, and against the golden fixtures.

The two builders number cells and faces differently and orient faces differently; both are matched GEOMETRICALLY (cell and
face centroids), the orientation signs are read off the divergence matrices and must then explain the mass matrices as
well.  Spec: /root/reference/src/PDESampler.cpp:232-258 (M, W, D, boundary elimination), SURVEY.md Appendix A.5."""
import json
import os

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import fe_ref
from oracle.darcy_oracle import DarcyOracle
from oracle.sampler_oracle import SamplerOracle
from parelagmc_amd.fe import (box_mesh, build_darcy_problem, build_hierarchy, build_hybrid_sampler_problem,
                              build_sampler_problem)
from parelagmc_amd.fe.mesh import element_centroids
from parelagmc_amd.fe.rt0 import mass_matrix

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _match(a, b, tol=1e-9):
    """perm with a[i] == b[perm[i]] for two point clouds holding the same points"""
    def key(x):
        return np.lexsort(np.round(x / tol).astype(np.int64).T[::-1])
    ia, ib = key(a), key(b)
    assert np.allclose(a[ia], b[ib], atol=10 * tol)
    perm = np.empty(len(a), np.int64)
    perm[ia] = ib
    return perm


def _maps(space, ref):
    """(cell permutation, face permutation, face signs): product index -> reference index, s_f = +-1 so that a product
    u-dof equals s_f times the reference u-dof of the same face"""
    cperm = _match(element_centroids(space.mesh), ref.cell_centroids())
    fc = space.mesh.verts[space.faces.face_verts].mean(axis=1)
    fperm = _match(fc, ref.face_centroids())
    Bp = space.B.tocsr()
    Br = ref.divergence()[cperm][:, fperm].tocsr()
    assert (Bp != 0).multiply(Br != 0).nnz == Bp.nnz == Br.nnz          # same pattern
    ratio = sp.csr_matrix(Bp.multiply(Br))                                # entries +-1: product of the two signs
    s = np.zeros(space.n_u)
    coo = ratio.tocoo()
    s[coo.col] = coo.data
    # both cells of an interior face must agree on the sign
    assert abs(ratio - sp.csr_matrix(abs(Br) @ sp.diags(s))).max() == 0.0
    assert set(np.unique(s)) <= {-1.0, 1.0}
    return cperm, fperm, s


@pytest.mark.parametrize("n0,nref", [(4, 0), (4, 1), (4, 2)])
def test_hex_operators_equal_the_closed_forms_entry_by_entry(n0, nref):
    h = build_hierarchy(box_mesh([n0] * 3, [2.0, 2.0, 2.0], "hex"), nref)
    levels = fe_ref.hex_hierarchy([n0] * 3, [2.0, 2.0, 2.0], nref)
    maps = []
    for space, ref in zip(h.spaces, levels):
        cperm, fperm, s = _maps(space, ref)
        maps.append(cperm)
        S = sp.diags(s)
        assert space.n_u == ref.n_u and space.n_s == ref.n_s
        np.testing.assert_allclose(space.vol, ref.w_diag()[cperm], rtol=1e-14)
        # unconstrained mass matrix and M(c) for a random coefficient (the Darcy M(k))
        rng = np.random.default_rng(5)
        c = np.exp(rng.standard_normal(space.n_s))
        for coeff_p, coeff_r in ((None, None), (c, None)):
            Mp = mass_matrix(space.emass, coeff_p)
            cr = None
            if coeff_p is not None:
                cr = np.empty(ref.n_s)
                cr[cperm] = c
            Mr = (S @ ref.mass(cr)[fperm][:, fperm] @ S).tocsr()
            d = abs(Mp - Mr)
            assert d.max() <= 1e-14 * abs(Mr).max()
            assert (Mp != 0).nnz == (Mr != 0).nnz                         # no extra stored couplings either way
        # boundary attributes (the Darcy BC arrays index them)
        np.testing.assert_array_equal(space.faces.face_bdr_attr, ref.boundary_attribute()[fperm])
    # P0 prolongators
    for l in range(nref):
        Pr = levels[l].prolongation(levels[l + 1])[maps[l]][:, maps[l + 1]]
        assert abs(h.P[l] - Pr).max() == 0.0


def test_sampler_operators_and_fields_from_the_independent_builder():
    """the arrays the HIP path receives (build_sampler_problem) == the closed forms after boundary elimination, the two
    direct solves give the same field for the same white noise (ids matched geometrically), and the golden fixture holds
    for the independent builder"""
    h = build_hierarchy(box_mesh([4, 4, 4], [2.0, 2.0, 2.0], "hex"), 1)
    spb = build_sampler_problem(h, corlen=0.1)
    levels = fe_ref.hex_hierarchy([4, 4, 4], [2.0, 2.0, 2.0], 1)
    ref = fe_ref.RefSampler(levels, 0.1)
    assert abs(spb.matern_g - ref.g) <= 1e-13 * ref.g and spb.alpha == ref.alpha
    kat = json.load(open(os.path.join(GOLD, "kat.json")))["matern_g"]["cases"]
    for case in kat:
        assert abs(fe_ref.matern_g(case["corlen"], case["dim"]) - case["g"]) <= 1e-12 * case["g"]
    cperms = []
    for l, (space, L) in enumerate(zip(h.spaces, spb.levels)):
        cperm, fperm, s = _maps(space, levels[l])
        cperms.append(cperm)
        S = sp.diags(s)
        M, B, w = ref.operators(l)
        assert abs(L.M - S @ M[fperm][:, fperm] @ S).max() <= 1e-14 * abs(M).max()
        assert abs(L.B - B[cperm][:, fperm] @ S).max() == 0.0
        np.testing.assert_allclose(L.w_diag, w[cperm], rtol=1e-14)
    gold = np.load(os.path.join(GOLD, "gold_sampler_hex.npz"))
    so = SamplerOracle(spb)

    def to_ref(v, l):                 # product cell order -> reference cell order
        out = np.empty_like(v)
        out[cperms[l]] = v
        return out

    for xi, s00, s10 in zip(gold["xi0"], gold["s00"], gold["s10"]):
        f0 = ref.eval(0, 0, to_ref(xi, 0))
        f1 = ref.eval(1, 0, to_ref(xi, 0))
        np.testing.assert_allclose(f0[cperms[0]], s00, rtol=0, atol=1e-10 * np.abs(s00).max())
        np.testing.assert_allclose(f1[cperms[1]], s10, rtol=0, atol=1e-10 * np.abs(s10).max())
        np.testing.assert_allclose(f0[cperms[0]], so.eval(0, 0, xi)[0], rtol=0, atol=1e-10 * np.abs(s00).max())
    for xi, s11 in zip(gold["xi1"], gold["s11"]):
        np.testing.assert_allclose(ref.eval(1, 1, to_ref(xi, 1))[cperms[1]], s11, rtol=0, atol=1e-10 * np.abs(s11).max())


@pytest.mark.parametrize("builder", ["numpy", "library"])
def test_hybridized_system_from_the_closed_forms(builder):
    """oracle/fe_ref.RefHybrid (one 7 x 7 inverse per level, this module's own multiplier signs) (i) reproduces the field of
    the independent saddle-point direct solve - the hybridized system IS the sampler's system - and (ii) equals, entry by entry
    after geometric matching and the per-face multiplier sign, the H, G, z the HIP path receives (fe/hybrid.py)"""
    h = build_hierarchy(box_mesh([4, 4, 4], [2.0, 2.0, 2.0], "hex"), 1)
    # "library": pmc_hybrid_build, the C++ elimination a caller of libpmc.so uses (host code, no GPU)
    from parelagmc_amd import capi
    hp = build_hybrid_sampler_problem(h, corlen=0.1, builder=capi.library_hybrid_builder if builder == "library" else None)
    levels = fe_ref.hex_hierarchy([4, 4, 4], [2.0, 2.0, 2.0], 1)
    ref, hyb = fe_ref.RefSampler(levels, 0.1), fe_ref.RefHybrid(levels, 0.1)
    rng = np.random.default_rng(11)
    for lvl, xl in ((0, 0), (1, 0), (1, 1)):
        xi = rng.standard_normal(levels[xl].n_s)
        a, b = ref.eval(lvl, xl, xi), hyb.eval(lvl, xl, xi)
        assert np.linalg.norm(a - b) <= 1e-12 * np.linalg.norm(a)
    for l, (space, L) in enumerate(zip(h.spaces, hp.levels)):
        cperm, fperm, _ = _maps(space, levels[l])
        H, G, z = hyb.operators(l)
        Hr, Gr = H[fperm][:, fperm].tocsr(), G[fperm][:, cperm].tocsr()
        np.testing.assert_allclose(L.z_diag, z[cperm], rtol=1e-13)
        # multiplier sign per face: read off G (one sign per row), must then explain H as well
        d = np.zeros(space.n_u)
        coo = sp.csr_matrix(L.G.multiply(Gr)).tocoo()
        d[coo.row] = np.sign(coo.data)
        assert set(np.unique(d)) <= {-1.0, 1.0}
        D = sp.diags(d)
        assert abs(L.G - D @ Gr).max() <= 1e-13 * abs(Gr).max()
        assert abs(L.H - D @ Hr @ D).max() <= 1e-13 * abs(Hr).max()
        assert (L.H != 0).nnz == (Hr != 0).nnz


def test_darcy_known_answer_and_goldens_from_the_independent_builder():
    """DarcyDeterministicTest (/root/reference/examples/CMakeLists.txt:62-66): Q = 2 and 17152 / 2240 / 304 dofs from the
    closed-form operators alone; for log-normal k the independent solve reproduces the golden QoIs and the oracle's"""
    ess, obs, inflow = [0, 1, 1, 1, 1, 0], [1, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 1]
    levels = fe_ref.hex_hierarchy([4, 4, 4], [2.0, 2.0, 2.0], 2)
    for L, dofs in zip(levels, (17152, 2240, 304)):
        Q, C, _ = fe_ref.RefDarcy(L, ess, obs, inflow).solve_fwd(np.ones(L.n_s))
        assert abs(Q - 2.0) < 1e-11 and int(C) == dofs
    h = build_hierarchy(box_mesh([4, 4, 4], [2.0, 2.0, 2.0], "hex"), 1)
    gold = np.load(os.path.join(GOLD, "gold_darcy_hex.npz"))
    lv = fe_ref.hex_hierarchy([4, 4, 4], [2.0, 2.0, 2.0], 1)
    for kd, tag in ((True, "div"), (False, "mul")):
        dp = build_darcy_problem(h, ess, obs, inflow, k_divides=kd)
        do = DarcyOracle(dp)
        for l in range(2):
            cperm, fperm, s = _maps(h.spaces[l], lv[l])
            rd = fe_ref.RefDarcy(lv[l], ess, obs, inflow, k_divides=kd)
            # rhs / observation functional / essential mask of the product == the closed forms (up to the face signs)
            np.testing.assert_array_equal(dp.levels[l].ess_mask.astype(bool), rd.ess[fperm])
            np.testing.assert_allclose(dp.levels[l].rhs[:lv[l].n_u], s * rd.rhs_u[fperm], atol=0)
            np.testing.assert_allclose(dp.levels[l].obs[:lv[l].n_u], s * rd.obs_u[fperm], atol=0)
            for k, Qg in zip(gold[f"k_L{l}"], gold[f"Q_L{l}_{tag}"]):
                kr = np.empty_like(k)
                kr[cperm] = k
                Q, _, sol = rd.solve_fwd(kr)
                assert abs(Q - Qg) <= 1e-10 * abs(Qg)
                Qo, _, solo = do.solve_fwd(l, k, return_solution=True)
                assert abs(Q - Qo) <= 1e-10 * abs(Qo)
                np.testing.assert_allclose(solo[:lv[l].n_u], s * sol[:lv[l].n_u][fperm], atol=1e-10 * np.abs(sol).max())


# ---------------------------------------------------------------------------------------------------------------------
# tetrahedra (the headline mesh family): SURVEY Appendix A.5's RT0 element matrices by quadrature, own faces / orientations


def _tet_hierarchy(name, nref):
    from conftest import ROOT
    from parelagmc_amd.fe import mesh_from_json
    return build_hierarchy(mesh_from_json(os.path.join(ROOT, "tests", "golden", "meshes", name + ".json")), nref)


def _tet_maps(space, ref):
    """as _maps for a TetLevel built from the same vertex / element DATA: cells are in the same order by construction
    (checked through their centroids), faces are matched by centroid, signs read off the divergence matrices"""
    np.testing.assert_allclose(element_centroids(space.mesh), ref.cell_centroids(), atol=1e-14)
    fc = space.mesh.verts[space.faces.face_verts].mean(axis=1)
    fperm = _match(fc, ref.face_centroids())
    Bp = space.B.tocsr()
    Br = ref.divergence()[:, fperm].tocsr()
    assert (Bp != 0).multiply(Br != 0).nnz == Bp.nnz == Br.nnz
    coo = sp.csr_matrix(Bp.multiply(Br)).tocoo()
    s = np.zeros(space.n_u)
    s[coo.col] = coo.data
    assert abs(sp.csr_matrix(Bp.multiply(Br)) - sp.csr_matrix(abs(Br) @ sp.diags(s))).max() == 0.0
    assert set(np.unique(s)) <= {-1.0, 1.0}
    return fperm, s


@pytest.mark.parametrize("name,nref", [("cube_tet", 0), ("cube_tet", 1), ("cube_tet", 2), ("cube_tet_embed", 1)])
def test_tet_operators_equal_the_quadrature_restatement_entry_by_entry(name, nref):
    """M_e, M, M(c), B, W of fe/rt0.py on tetrahedra (the barycentric closed form) == SURVEY A.5's
    M_e[i, j] = +-(1 / (9 |T|^2)) int (x - v_i).(x - v_j) evaluated by quadrature with this oracle's own faces and signs"""
    h = _tet_hierarchy(name, nref)
    rng = np.random.default_rng(17)
    for space in h.spaces:
        ref = fe_ref.TetLevel(space.mesh.verts, space.mesh.elems)
        assert ref.n_u == space.n_u and ref.n_s == space.n_s
        # the refinement handed over is conforming and fills the domain: own volume sum, every face has 1 or 2 elements
        box = np.prod(space.mesh.verts.max(axis=0) - space.mesh.verts.min(axis=0))
        assert abs(ref.vol.sum() - box) <= 1e-12 * box
        np.testing.assert_allclose(space.vol, ref.w_diag(), rtol=1e-13)
        fperm, s = _tet_maps(space, ref)
        S = sp.diags(s)
        c = np.exp(rng.standard_normal(space.n_s))
        for coeff in (None, c):
            Mp = mass_matrix(space.emass, coeff)
            Mr = (S @ ref.mass(coeff)[fperm][:, fperm] @ S).tocsr()
            assert abs(Mp - Mr).max() <= 1e-13 * abs(Mr).max()
        # element matrices themselves: the product's COO entries against Me[e, i, j] of the oracle
        Me = ref.element_mass()
        inv = np.empty(space.n_u, np.int64)
        inv[fperm] = np.arange(space.n_u)                 # reference face -> product face
        pf = inv[ref.elem_face]                           # (ne, 4) product face ids in the oracle's local order
        dense = {}
        em = space.emass
        for r_, c_, e_, v_ in zip(em.rows, em.cols, em.elem, em.vals):
            dense[(int(e_), int(r_), int(c_))] = v_
        worst = 0.0
        for e in range(0, space.n_s, max(1, space.n_s // 97)):          # a sample of elements, every entry of each
            for i in range(4):
                for j in range(4):
                    want = Me[e, i, j] * s[pf[e, i]] * s[pf[e, j]]
                    got = dense.get((e, int(pf[e, i]), int(pf[e, j])), 0.0)      # exact zeros are not stored by the product
                    worst = max(worst, abs(got - want) / abs(Me[e]).max())
        assert worst <= 1e-13
        # boundary faces
        np.testing.assert_array_equal(space.faces.face_elem[:, 1] < 0, ref.boundary_faces()[fperm])


@pytest.mark.parametrize("builder", ["numpy", "library"])
def test_tet_hybridized_system_and_fields_from_the_quadrature_restatement(builder):
    """H, G, z the HIP path receives on tetrahedra (fe/hybrid.py, and pmc_hybrid_build = what a C++ caller uses) == the
    oracle's own element-local elimination entry by entry (after the per-face multiplier sign), and the fields of the
    independent saddle-point direct solve, of the independent hybridized solve and of the product-side oracle agree"""
    from parelagmc_amd import capi
    h = _tet_hierarchy("cube_tet", 2)
    kw = dict(corlen=0.1, n_mc_levels=2)
    hp = build_hybrid_sampler_problem(h, builder=capi.library_hybrid_builder if builder == "library" else None, **kw)
    spb = build_sampler_problem(h, **kw)
    so = SamplerOracle(spb)
    levels = [fe_ref.TetLevel(sp_.mesh.verts, sp_.mesh.elems) for sp_ in h.spaces]
    ref = fe_ref.RefTetSampler(levels, 0.1)
    rng = np.random.default_rng(23)
    # P0 prolongator by the oracle's own parent search
    assert abs(h.P[0] - levels[0].prolongation(levels[1])).max() == 0.0
    for l in range(2):
        space, L = h.spaces[l], hp.levels[l]
        fperm, s = _tet_maps(space, levels[l])
        hy = fe_ref.RefTetHybrid(levels[l], 0.1)
        np.testing.assert_allclose(L.z_diag, hy.z, rtol=1e-12)
        Hr, Gr = hy.H[fperm][:, fperm].tocsr(), hy.G[fperm].tocsr()
        d = np.zeros(space.n_u)
        coo = sp.csr_matrix(L.G.multiply(Gr)).tocoo()
        d[coo.row] = np.sign(coo.data)
        assert set(np.unique(d)) <= {-1.0, 1.0}
        D = sp.diags(d)
        assert abs(L.G - D @ Gr).max() <= 1e-12 * abs(Gr).max()
        assert abs(L.H - D @ Hr @ D).max() <= 1e-12 * abs(Hr).max()
        # sampler operators after boundary elimination
        M, B, w = ref.operators(l)
        S = sp.diags(s)
        assert abs(spb.levels[l].M - S @ M[fperm][:, fperm] @ S).max() <= 1e-13 * abs(M).max()
        assert abs(spb.levels[l].B - B[:, fperm] @ S).max() == 0.0
        xi = rng.standard_normal(levels[l].n_s)
        a, b, c = ref.eval(l, l, xi), hy.eval(xi), so.eval(l, l, xi)[0]
        assert np.linalg.norm(a - b) <= 1e-11 * np.linalg.norm(a)
        assert np.linalg.norm(a - c) <= 1e-11 * np.linalg.norm(a)
    xi = rng.standard_normal(levels[0].n_s)
    assert np.linalg.norm(ref.eval(1, 0, xi) - so.eval(1, 0, xi)[0]) <= 1e-11 * np.linalg.norm(ref.eval(1, 0, xi))
