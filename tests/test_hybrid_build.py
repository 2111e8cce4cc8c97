"""pmc_hybrid_build - the library's element-local elimination (the setup step behind the reference's "Hybridization" solver,
/root/reference/src/PDESampler.cpp:302-318) - against the numpy stand-in fe/hybrid.py and against the independent closed-form
restatement oracle/fe_ref.py.  Host code of libpmc.so: runs without a GPU."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import ROOT
from parelagmc_amd import capi
from parelagmc_amd.fe import box_mesh, build_hierarchy, mesh_from_json
from parelagmc_amd.fe.hybrid import hybrid_level_ops
from parelagmc_amd.fe.rt0 import mass_contributions


def _mesh(name):
    return mesh_from_json(os.path.join(ROOT, "tests", "golden", "meshes", name + ".json"))


CASES = [("hex", lambda: build_hierarchy(box_mesh([4, 4, 4], [2, 2, 2], "hex"), 1)),
         ("stretched_hex", lambda: build_hierarchy(box_mesh([3, 5, 2], [1200.0, 2200.0, 170.0], "hex"), 1)),
         ("cube_tet", lambda: build_hierarchy(_mesh("cube_tet"), 2)),
         ("cube_tet_embed", lambda: build_hierarchy(_mesh("cube_tet_embed"), 1)),
         ("quad", lambda: build_hierarchy(box_mesh([5, 3], [1.0, 1.0], "quad"), 1))]


@pytest.mark.parametrize("name,make", CASES, ids=[c[0] for c in CASES])
def test_library_elimination_equals_the_numpy_one_entry_by_entry(name, make):
    h = make()
    for corlen in (0.1, 100.0):
        alpha = 1.0 / corlen ** 2
        for space in h.spaces:
            ref = hybrid_level_ops(space, alpha, None)
            H, G, z = capi.library_hybrid_builder(space, alpha)
            assert H.shape == ref.H.shape and G.shape == ref.G.shape
            # same pattern up to entries the numpy version stores as explicit zeros / drops
            dH = abs(H - ref.H)
            tol = 1e-10 if name == "stretched_hex" else 1e-12   # two elimination orders of a local matrix with condition ~1e6
            assert dH.max() <= tol * abs(ref.H).max(), (name, dH.max())
            assert abs(G - ref.G).max() <= tol * abs(ref.G).max()
            assert np.max(np.abs(z - ref.z_diag)) <= tol * np.max(np.abs(ref.z_diag))
            assert abs(H - H.T).max() == 0.0                      # symmetrised exactly
            assert H.has_sorted_indices or (np.diff(H.indices)[np.diff(H.indices) < 0].size <= H.shape[0])


def test_library_elimination_reproduces_the_saddle_point_field():
    """H lambda = G f, s = z f - G^T lambda built by the LIBRARY equals the direct solve of [M B^T; B -aW][u; s] = [0; f]"""
    import scipy.sparse.linalg as spla
    from parelagmc_amd.fe import build_sampler_problem
    h = build_hierarchy(_mesh("cube_tet"), 2)
    sp_ = build_sampler_problem(h, corlen=0.1)
    L = sp_.levels[0]
    H, G, z = capi.library_hybrid_builder(h.spaces[0], sp_.alpha)
    f = np.random.default_rng(3).standard_normal(L.n_s)
    A = sp.bmat([[L.M, L.B.T], [L.B, -sp_.alpha * sp.diags(L.w_diag)]]).tocsc()
    s_ref = spla.splu(A).solve(np.concatenate([np.zeros(L.n_u), f]))[L.n_u:]
    lam = spla.splu(H.tocsc()).solve(G @ f)
    s = z * f - G.T @ lam
    assert np.linalg.norm(s - s_ref) <= 1e-11 * np.linalg.norm(s_ref)


def test_library_elimination_refuses_bad_input():
    h = build_hierarchy(box_mesh([2, 2, 2], [1, 1, 1], "hex"), 0)
    space = h.spaces[0]
    pat, c_ptr, c_elem, c_val = mass_contributions(space.emass)
    # B with eliminated (explicitly zeroed) boundary columns: refused - the elimination needs every face of an element
    Bz = space.B.copy()
    Bz.data[0] = 0.0
    with pytest.raises(capi.PmcError) as e:
        capi.hybrid_build(pat, c_ptr, c_elem, c_val, Bz, space.vol, 1.0)
    assert e.value.code == -1 and "explicit zero" in str(e.value)
    with pytest.raises(capi.PmcError):
        capi.hybrid_build(pat, c_ptr, c_elem, c_val, space.B, -space.vol, 1.0)
    with pytest.raises(capi.PmcError):
        capi.hybrid_build(pat, c_ptr, c_elem, c_val, space.B, space.vol, 0.0)
    bad = c_elem.copy()
    bad[0] = (bad[0] + 5) % space.n_s          # an element that does not own the face
    with pytest.raises(capi.PmcError):
        capi.hybrid_build(pat, c_ptr, bad, c_val, space.B, space.vol, 1.0)
    # a singular local matrix: all mass contributions zero
    with pytest.raises(capi.PmcError) as e:
        capi.hybrid_build(pat, c_ptr, c_elem, 0.0 * c_val, space.B[:, :], space.vol, 1.0)
    assert "singular" in str(e.value) or e.value.code == -1
