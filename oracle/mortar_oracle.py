"""ORACLE (test infrastructure): P0 x P0 mortar matrix G[i,j] = |A_i ∩ B_j| between two non-matching meshes.

Restates what the reference's mortar assembly computes for piecewise constants
(/root/reference/src/transfer/MortarAssembler.cpp:35-125: candidate pairs, polytope intersection, its measure;
/root/reference/src/L2ProjectionPDESampler.cpp:488-505 assembles Gt from it) by a route that shares nothing with the
product code: every element is taken as the intersection of its face half-spaces, the intersection of two elements
as the union of the two half-space sets, its vertices by scipy.spatial.HalfspaceIntersection around the Chebyshev
centre (scipy.optimize.linprog) and its measure by scipy.spatial.ConvexHull.  Convex elements with planar faces only
(simplices, parallelepipeds); brute force over bounding-box candidates - small meshes only.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
from scipy.optimize import linprog
from scipy.spatial import ConvexHull, HalfspaceIntersection

_FACES = {
    "tri": [(0, 1), (1, 2), (2, 0)],
    "quad": [(0, 1), (1, 2), (2, 3), (3, 0)],
    "tet": [(1, 2, 3), (0, 3, 2), (0, 1, 3), (0, 2, 1)],
    "hex": [(0, 3, 2, 1), (4, 5, 6, 7), (0, 1, 5, 4), (1, 2, 6, 5), (2, 3, 7, 6), (3, 0, 4, 7)],
}


def element_halfspaces(pts: np.ndarray, etype: str) -> np.ndarray:
    """Rows [n, -n.p0] with n the outward unit normal of each face: inside <=> row . [x, 1] <= 0."""
    c = pts.mean(axis=0)
    rows = []
    for f in _FACES[etype]:
        p = pts[list(f)]
        if pts.shape[1] == 2:
            t = p[1] - p[0]
            n = np.array([t[1], -t[0]])
        else:
            n = np.cross(p[1] - p[0], p[2] - p[0])
        n = n / np.linalg.norm(n)
        if n @ (c - p[0]) > 0:
            n = -n
        rows.append(np.concatenate([n, [-n @ p[0]]]))
    return np.array(rows)


def convex_intersection_measure(hs: np.ndarray) -> float:
    d = hs.shape[1] - 1
    A, b = hs[:, :d], hs[:, d]
    # Chebyshev centre: max r  s.t.  A x + r |a| <= -b
    res = linprog(np.concatenate([np.zeros(d), [-1.0]]), A_ub=np.hstack([A, np.ones((len(A), 1))]), b_ub=-b,
                  bounds=[(None, None)] * d + [(0, None)], method="highs")
    if res.status != 0 or res.x[-1] < 1e-10:
        return 0.0
    try:
        pts = HalfspaceIntersection(hs, res.x[:d]).intersections
        return float(ConvexHull(pts).volume)
    except Exception:
        return 0.0


def mortar_gt(verts_a, elems_a, etype_a, verts_b, elems_b, etype_b, tol=1e-12) -> sp.csr_matrix:
    pa, pb = verts_a[elems_a], verts_b[elems_b]
    lo_a, hi_a, lo_b, hi_b = pa.min(1), pa.max(1), pb.min(1), pb.max(1)
    hs_b = [element_halfspaces(p, etype_b) for p in pb]
    rows, cols, vals = [], [], []
    for i, p in enumerate(pa):
        hs_i = element_halfspaces(p, etype_a)
        cand = np.nonzero(np.all(lo_a[i] < hi_b - 1e-14, axis=1) & np.all(lo_b < hi_a[i] - 1e-14, axis=1))[0]
        for j in cand:
            v = convex_intersection_measure(np.vstack([hs_i, hs_b[j]]))
            if v > tol:
                rows.append(i); cols.append(int(j)); vals.append(v)
    return sp.csr_matrix((vals, (rows, cols)), shape=(len(pa), len(pb)))
