"""Regenerates the fixtures under tests/golden/ (run in the build container, where
/root/reference exists).  Fixtures are DATA only:
  * meshes/*.json        - mesh data files the reference's tests/drivers read
                           (/root/reference/meshes/*.mesh), converted to JSON arrays;
  * meshes/*.mesh        - six of those data files in their on-disk format (MFEM mesh v1.0 /
                           INLINE, 68 B ... 4.6 KB each): fixtures of the mesh READER;
  * kat.json             - the reference's RNG-free known answers
                           (examples/CMakeLists.txt:62-66 DarcyDeterministicTest) and the
                           closed-form Matern coefficients of src/Utilities.hpp:188-200;
  * gold_sampler_hex.npz, gold_quad.npz, gold_darcy_hex.npz
                         - SELF-GENERATED oracle outputs (sparse direct solves) for seeded
                           inputs; they guard the oracle against drift, they are not
                           reference outputs (the reference cannot be built here).
                           The two hex fixtures are computed by oracle/fe_ref.py - closed-form
                           RT0/P0 operators that import nothing of parelagmc_amd.fe - and stored in
                           the product's cell order (cells matched by centroid); the oracle on the
                           product's builders must reproduce them to 1e-10 or this script stops.
  * gold_sampler_tet.npz - the same for tetrahedra (cube_tet refined 3 x / 2 x): fields from oracle/fe_ref.py's TetLevel
                           (quadrature element matrices, own faces / signs / parent search), `--only-tet` writes this file alone.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from parelagmc_amd.fe import (box_mesh, build_darcy_problem, build_hierarchy, build_sampler_problem,  # noqa: E402
                              read_mfem_mesh)
from oracle import fe_ref  # noqa: E402
from oracle.darcy_oracle import DarcyOracle  # noqa: E402
from oracle.sampler_oracle import SamplerOracle  # noqa: E402
from parelagmc_amd.fe.mesh import element_centroids  # noqa: E402

REF_MESHES = "/root/reference/meshes"


def mesh_to_json(name):
    m = read_mfem_mesh(os.path.join(REF_MESHES, name + ".mesh"))
    d = dict(etype=m.etype, verts=m.verts.tolist(), elems=m.elems.tolist(), elem_attr=m.elem_attr.tolist(),
             bdr=m.bdr.tolist(), bdr_attr=m.bdr_attr.tolist(), source=f"meshes/{name}.mesh")
    with open(os.path.join(HERE, "meshes", name + ".json"), "w") as f:
        json.dump(d, f)


def cell_perm(space, ref):
    """perm with: product cell i == reference cell perm[i] (matched by centroid)"""
    def order(x):
        return np.lexsort(np.round(x * 1e9).astype(np.int64).T[::-1])
    a, b = element_centroids(space.mesh), ref.cell_centroids()
    ia, ib = order(a), order(b)
    assert np.allclose(a[ia], b[ib])
    perm = np.empty(len(a), np.int64)
    perm[ia] = ib
    return perm


def agree(a, b, what):
    if not np.allclose(a, b, rtol=0, atol=1e-10 * np.abs(b).max()):
        raise SystemExit(f"{what}: oracle on the product's builders and oracle/fe_ref.py disagree")


def tet_golden():
    """GOLD-4 (round 5): cube_tet refined 3 x / 2 x (3 072 / 384 tetrahedra, the headline's mesh family), Gaussian, correlation
    length 0.1: fields for fixed xi from oracle/fe_ref.py's TetLevel / RefTetSampler - own faces, orientations, quadrature
    element matrices and parent search; only the vertex coordinates and vertex quadruples of the refined meshes are taken from
    the product as data - cross-checked against the independent hybridized solve (RefTetHybrid) and against the oracle on the
    product's builders before anything is written.  Cells are in the product's order (TetLevel keeps the order it is given)."""
    from parelagmc_amd.fe import mesh_from_json
    h = build_hierarchy(mesh_from_json(os.path.join(HERE, "meshes", "cube_tet.json")), 3)
    sp = build_sampler_problem(h, corlen=0.1, n_mc_levels=2)
    so = SamplerOracle(sp)
    levels = [fe_ref.TetLevel(x.mesh.verts, x.mesh.elems) for x in h.spaces[:2]]
    rs = fe_ref.RefTetSampler(levels, 0.1)
    rng = np.random.Generator(np.random.PCG64(20261005))
    xi = rng.standard_normal((2, levels[0].n_s))
    xi1 = rng.standard_normal((2, levels[1].n_s))
    s00 = np.stack([rs.eval(0, 0, x) for x in xi])
    s10 = np.stack([rs.eval(1, 0, x) for x in xi])
    s11 = np.stack([rs.eval(1, 1, x) for x in xi1])
    hy0, hy1 = fe_ref.RefTetHybrid(levels[0], 0.1), fe_ref.RefTetHybrid(levels[1], 0.1)
    agree(np.stack([hy0.eval(x) for x in xi]), s00, "tet s00 (hybridized restatement)")
    agree(np.stack([hy1.eval(x) for x in xi1]), s11, "tet s11 (hybridized restatement)")
    agree(np.stack([so.eval(0, 0, x)[0] for x in xi]), s00, "tet s00")
    agree(np.stack([so.eval(1, 0, x)[0] for x in xi]), s10, "tet s10")
    agree(np.stack([so.eval(1, 1, x)[0] for x in xi1]), s11, "tet s11")
    np.savez_compressed(os.path.join(HERE, "gold_sampler_tet.npz"), xi0=xi, s00=s00.astype(np.float64), s10=s10, xi1=xi1, s11=s11)
    print("gold_sampler_tet.npz written")


def main():
    if "--only-tet" in sys.argv:
        tet_golden()
        return
    if os.path.isdir(REF_MESHES):
        # the mesh DATA files of the BASELINE configurations in their on-disk format (MFEM mesh v1.0 / INLINE), as fixtures of
        # the reader: tests/test_fe.py reads them with fe/mesh.py::read_mfem_mesh and compares with the JSON arrays below
        import shutil
        for name in ("inline_quad", "inline_hex", "inline_tri", "cube_hex", "cube_tet", "cube_tet_embed"):
            shutil.copy(os.path.join(REF_MESHES, name + ".mesh"), os.path.join(HERE, "meshes", name + ".mesh"))
            os.chmod(os.path.join(HERE, "meshes", name + ".mesh"), 0o644)
        for name in ("inline_quad", "cube_hex", "cube_tet", "cube_tet_embed", "cube_hex_enlarge", "cube_tet_enlarge", "square",
                     "square_enlarge"):
            mesh_to_json(name)
    kat = {
        "darcy_deterministic": {"source": "examples/CMakeLists.txt:62-66", "Q": [2.0, 2.0, 2.0],
                                "dofs": [17152, 2240, 304],
                                "mesh": "Build3DHexMesh 4x4x4 on [0,2]^3, 2 parallel refinements",
                                "ess": [0, 1, 1, 1, 1, 0], "obs": [1, 0, 0, 0, 0, 0], "inflow": [0, 0, 0, 0, 0, 1]},
        "matern_g": {"source": "src/Utilities.hpp:188-200", "cases": [
            {"corlen": 0.1, "dim": 3, "g": 28.90067818451249},
            {"corlen": 0.1, "dim": 2, "g": 50.13256549262001},
            {"corlen": 100.0, "dim": 3, "g": 0.9139196898659948}]},
    }
    with open(os.path.join(HERE, "kat.json"), "w") as f:
        json.dump(kat, f, indent=1)

    rng = np.random.Generator(np.random.PCG64(20261003))
    # GOLD-2: hex 4^3 -> 8^3 (2 levels), Gaussian
    h = build_hierarchy(box_mesh([4, 4, 4], [2, 2, 2], "hex"), 1)
    sp = build_sampler_problem(h, corlen=0.1)
    so = SamplerOracle(sp)
    ref_levels = fe_ref.hex_hierarchy([4, 4, 4], [2.0, 2.0, 2.0], 1)
    rs = fe_ref.RefSampler(ref_levels, 0.1)
    perm = [cell_perm(h.spaces[l], ref_levels[l]) for l in range(2)]

    def to_ref(v, l):
        out = np.empty_like(v)
        out[perm[l]] = v
        return out

    xi = rng.standard_normal((2, sp.levels[0].n_s))
    s00 = np.stack([rs.eval(0, 0, to_ref(x, 0))[perm[0]] for x in xi])
    s10 = np.stack([rs.eval(1, 0, to_ref(x, 0))[perm[1]] for x in xi])      # coarse field from fine xi (Ps^T coupling)
    xi1 = rng.standard_normal((2, sp.levels[1].n_s))
    s11 = np.stack([rs.eval(1, 1, to_ref(x, 1))[perm[1]] for x in xi1])
    agree(np.stack([so.eval(0, 0, x)[0] for x in xi]), s00, "s00")
    agree(np.stack([so.eval(1, 0, x)[0] for x in xi]), s10, "s10")
    agree(np.stack([so.eval(1, 1, x)[0] for x in xi1]), s11, "s11")
    np.savez_compressed(os.path.join(HERE, "gold_sampler_hex.npz"), xi0=xi, s00=s00, s10=s10, xi1=xi1, s11=s11)
    # GOLD-1: inline_quad, 1 level, d=2
    hq = build_hierarchy(box_mesh([2, 2], [1.0, 1.0], "quad"), 0)
    spq = build_sampler_problem(hq, corlen=0.1)
    xq = rng.standard_normal((16, spq.levels[0].n_s))
    sq = np.stack([SamplerOracle(spq).eval(0, 0, x)[0] for x in xq])
    np.savez_compressed(os.path.join(HERE, "gold_quad.npz"), xi=xq, s=sq)
    # GOLD-3: Darcy Q for lognormal k (both k_divides settings)
    spl = build_sampler_problem(h, corlen=0.1, lognormal=True)
    sol = SamplerOracle(spl)
    out = {}
    for kd in (True, False):
        dp = build_darcy_problem(h, [0, 1, 1, 1, 1, 0], [1, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 1], k_divides=kd)
        do = DarcyOracle(dp)
        for lvl in range(2):
            k = np.stack([sol.eval(lvl, 0, x)[0] for x in xi])
            out[f"k_L{lvl}"] = k
            rd = fe_ref.RefDarcy(ref_levels[lvl], [0, 1, 1, 1, 1, 0], [1, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 1], k_divides=kd)
            Q = np.array([rd.solve_fwd(to_ref(kk, lvl))[0] for kk in k])
            agree(np.array([do.solve_fwd(lvl, kk)[0] for kk in k]), Q, f"Q level {lvl}")
            out[f"Q_L{lvl}_{'div' if kd else 'mul'}"] = Q
    np.savez_compressed(os.path.join(HERE, "gold_darcy_hex.npz"), **out)
    print("fixtures written")
    tet_golden()


if __name__ == "__main__":
    main()
