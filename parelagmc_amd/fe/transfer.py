"""L2 (mortar) projector between NON-MATCHING meshes (setup side).

The reference assembles ``Gt[i,j] = |e_orig,i  ∩  e_embed,j|`` for P0 x P0 with a general polytope-clipping
mortar assembler (/root/reference/src/transfer/ParMortarAssembler.cpp:1127-1144, used by
src/L2ProjectionPDESampler.cpp:488-505) and obtains the coarse levels by ``RAP(orig_Ps, Gt, Ps)`` (:512-513).
For quadrilateral / hexahedral meshes whose elements are axis-aligned boxes (meshes/cube_hex.mesh inside
meshes/cube_hex_enlarge.mesh: 4^3 cells of size 0.5 on [0,2]^3 inside 5^3 cells of size 0.6 on [-0.5,2.5]^3) the
intersection volume is the product of the per-axis interval overlaps, which is what this module computes - exactly,
without clipping.  Every other pair (triangles, tetrahedra, general quadrilaterals / hexahedra, mixed) goes through the
simplex-clipping assembler of libpmc_host.so (``pmc_mortar_assemble``, host/mortar.cpp).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

from .mesh import Mesh


def _boxes(m: Mesh):
    if m.etype not in ("quad", "hex"):
        raise ValueError("box intersection needs quad / hex meshes")
    p = m.verts[m.elems]
    lo, hi = p.min(axis=1), p.max(axis=1)
    if np.any(np.abs(np.prod(hi - lo, axis=1) - _vol_from_corners(p)) > 1e-10 * np.prod(hi - lo, axis=1)):
        raise ValueError("elements are not axis-aligned boxes")
    return lo, hi


def _vol_from_corners(p):
    # volume of the bounding box spanned by the distinct coordinates actually used (equal to the element volume
    # iff the element is an axis-aligned box whose corners use exactly two values per axis)
    d = p.shape[2]
    ext = []
    for a in range(d):
        c = p[:, :, a]
        two = (np.abs(c - c.min(axis=1, keepdims=True)) < 1e-12) | (np.abs(c - c.max(axis=1, keepdims=True)) < 1e-12)
        ext.append(np.where(two.all(axis=1), c.max(axis=1) - c.min(axis=1), np.nan))
    return np.prod(np.stack(ext, axis=1), axis=1)


def box_intersection_gt(orig: Mesh, embed: Mesh, tol: float = 1e-12) -> sp.csr_matrix:
    """Gt (n_orig x n_embed) with Gt[i,j] = |e_orig,i ∩ e_embed,j|, computed by a sort-and-sweep over the first axis
    and exact interval products; no structure of the numbering is assumed."""
    lo_o, hi_o = _boxes(orig)
    lo_e, hi_e = _boxes(embed)
    d = orig.dim
    # candidate pairs: overlap along axis 0 via searchsorted on the embed intervals sorted by lower end
    order = np.argsort(lo_e[:, 0], kind="stable")
    los = lo_e[order, 0]
    rows, cols = [], []
    # every embed cell whose lower end is < hi_o and upper end is > lo_o overlaps along axis 0
    upto = np.searchsorted(los, hi_o[:, 0] - tol, side="left")
    max_len = float((hi_e[:, 0] - lo_e[:, 0]).max())
    frm = np.searchsorted(los, lo_o[:, 0] - max_len + tol, side="left")
    cnt = upto - frm
    ii = np.repeat(np.arange(orig.ne), cnt)
    off = np.concatenate([[0], np.cumsum(cnt)])
    jj = order[np.arange(off[-1]) - np.repeat(off[:-1], cnt) + np.repeat(frm, cnt)]
    ov = np.ones(len(ii))
    for a in range(d):
        w = np.minimum(hi_o[ii, a], hi_e[jj, a]) - np.maximum(lo_o[ii, a], lo_e[jj, a])
        ov *= np.clip(w, 0.0, None)
    vol_scale = float(np.prod(hi_o - lo_o, axis=1).max())
    keep = ov > tol * vol_scale
    return sp.csr_matrix((ov[keep], (ii[keep], jj[keep])), shape=(orig.ne, embed.ne))


def clipped_intersection_gt(orig: Mesh, embed: Mesh, rel_tol: float = 1e-12) -> sp.csr_matrix:
    """Gt for arbitrary (convex-element) mesh pairs through the C ABI of libpmc_host.so."""
    from ..host_api import mortar_gt
    if orig.dim != embed.dim:
        raise ValueError("meshes of different dimension")
    G, _, _ = mortar_gt(orig.verts, orig.elems, embed.verts, embed.elems, rel_tol)
    return G


def intersection_gt(orig: Mesh, embed: Mesh, method: str = "auto") -> sp.csr_matrix:
    """method: 'box' (axis-aligned boxes, interval products), 'clip' (general), 'auto' (box when both meshes qualify)."""
    if method not in ("auto", "box", "clip"):
        raise ValueError("method must be auto, box or clip")
    if method == "box":
        return box_intersection_gt(orig, embed)
    # the C++ assembler has its own fast path for pairs of axis-aligned boxes and a bucket-grid candidate search, so it
    # also is the scalable route for large box meshes (box_intersection_gt enumerates every overlap along one axis)
    return clipped_intersection_gt(orig, embed)


def l2_projection_hierarchy(h_orig, h_embed, method: str = "auto"):
    """Per level (Gt, 1/|e_orig|): the finest level by geometry, coarser ones by RAP with the P0 prolongators as in
    L2ProjectionPDESampler.cpp:512-513.  Both hierarchies must have the same number of levels."""
    if h_orig.nlevels != h_embed.nlevels:
        raise ValueError("original and embedded hierarchies need the same number of levels")
    Gt = intersection_gt(h_orig.spaces[0].mesh, h_embed.spaces[0].mesh, method)
    out = [(Gt, 1.0 / h_orig.spaces[0].vol)]
    for lvl in range(h_orig.nlevels - 1):
        Gt = (h_orig.P[lvl].T @ Gt @ h_embed.P[lvl]).tocsr()
        Gt.sort_indices()
        out.append((Gt, 1.0 / h_orig.spaces[lvl + 1].vol))
    return out
