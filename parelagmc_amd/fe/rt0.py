"""Lowest-order Raviart-Thomas / piecewise-constant (RT0/P0) building blocks.

These are the *inputs* of the hot path - what the reference obtains from ParELAG's
``DeRhamSequence::ComputeMassOperator`` / ``GetDerivativeOperator``
(/root/reference/src/PDESampler.cpp:232-234, src/DarcySolver.cpp:194-208,479).
Normalisation: a u-dof is the total flux through a face in
the direction of the face's global normal, a p/s-dof is the cell value.  Then

    W = diag(|e|),   B[e,f] = +-1 (outward = +1)  (= W*D of the reference),
    M_e[i,j] = int_e phi_i . phi_j,   M(c) = sum_e c_e M_e.

The parity quantities (s per element, Q, C, sizes) do not depend on this choice.
"""
from __future__ import annotations

import dataclasses

import numpy as np
import scipy.sparse as sp

from .mesh import FaceTable, Mesh, build_faces, element_volumes


@dataclasses.dataclass
class ElementMass:
    """Element mass matrices in global-orientation sign convention, COO form."""
    rows: np.ndarray   # (ne*nfe*nfe,) global face index i
    cols: np.ndarray   # (ne*nfe*nfe,) global face index j
    elem: np.ndarray   # (ne*nfe*nfe,) element
    vals: np.ndarray   # (ne*nfe*nfe,) value for coefficient 1
    n_u: int
    n_e: int


@dataclasses.dataclass
class LevelSpaces:
    mesh: Mesh
    faces: FaceTable
    vol: np.ndarray            # (ne,)
    n_u: int
    n_s: int
    B: sp.csr_matrix           # (n_s, n_u), entries +-1, no BC elimination
    emass: ElementMass


def _local_mass(m: Mesh, ft: FaceTable, vol: np.ndarray) -> np.ndarray:
    """(ne, nfe, nfe) element mass matrices for outward-oriented local basis functions."""
    ne = m.ne
    if m.etype in ("tri", "tet"):
        d = m.dim
        nv = d + 1
        X = m.verts[m.elems]                                  # (ne, nv, d)
        # phi_i(x) = (x - v_i) / (d |K|);   x - v_i = sum_a lambda_a (v_a - v_i),
        # int lambda_a lambda_b = |K| (1 + delta_ab) / ((d+1)(d+2))   =>
        # (1/|K|) int (x-v_i).(x-v_j) = q - (v_i+v_j).c + v_i.v_j
        #   with c = centroid, q = (|sum_a v_a|^2 + sum_a |v_a|^2) / ((d+1)(d+2)).
        X = X - X.mean(axis=1, keepdims=True)                 # shift: c = 0 (better conditioning)
        ssum = X.sum(axis=1)
        q = ((ssum ** 2).sum(axis=1) + (X ** 2).sum(axis=(1, 2))) / ((d + 1.0) * (d + 2.0))
        I = q[:, None, None] + np.einsum("eix,ejx->eij", X, X)
        return I / (d * d * vol[:, None, None])
    # axis-aligned boxes
    P = m.verts[m.elems]
    ext = P.max(axis=1) - P.min(axis=1)                        # (ne, dim)
    if m.etype == "quad":
        # local faces: y-, x+, y+, x-   -> axis of each face and its partner
        axis = np.array([1, 0, 1, 0])
        partner = np.array([2, 3, 0, 1])
    else:
        # z-, y-, x+, y+, x-, z+
        axis = np.array([2, 1, 0, 1, 0, 2])
        partner = np.array([5, 3, 4, 1, 2, 0])
    # check axis alignment: each element's vertices must span exactly a box
    chk = np.abs(np.prod(ext, axis=1) - vol)
    if (chk > 1e-10 * vol).any():
        raise ValueError("quad/hex elements must be axis-aligned boxes")
    nfe = len(axis)
    Ml = np.zeros((ne, nfe, nfe))
    # int phi_f.phi_f = h_a / (3 A_f),   int phi_f.phi_partner = -h_a/(6 A_f)  (both outward)
    for f in range(nfe):
        h = ext[:, axis[f]]
        area = vol / h
        Ml[:, f, f] = h / (3.0 * area)
        Ml[:, f, partner[f]] = -h / (6.0 * area)
    return Ml


def build_spaces(m: Mesh) -> LevelSpaces:
    ft = build_faces(m)
    vol = element_volumes(m)
    ne, nfe = ft.elem_face.shape
    nf = ft.face_verts.shape[0]
    sign = ft.elem_sign.astype(np.float64)
    rows = np.repeat(np.arange(ne), nfe)
    B = sp.csr_matrix((sign.ravel(), (rows, ft.elem_face.ravel())), shape=(ne, nf))
    Ml = _local_mass(m, ft, vol) * sign[:, :, None] * sign[:, None, :]
    gi = np.broadcast_to(ft.elem_face[:, :, None], Ml.shape)
    gj = np.broadcast_to(ft.elem_face[:, None, :], Ml.shape)
    ge = np.broadcast_to(np.arange(ne)[:, None, None], Ml.shape)
    keep = Ml != 0.0          # hexes: cross-direction entries are exact zeros, not stored
    em = ElementMass(gi[keep].astype(np.int64), gj[keep].astype(np.int64), ge[keep].astype(np.int64),
                     Ml[keep], nf, ne)
    return LevelSpaces(m, ft, vol, nf, ne, B, em)


def mass_matrix(em: ElementMass, coeff=None) -> sp.csr_matrix:
    """M(c) = sum_e c_e M_e as CSR with sorted indices."""
    v = em.vals if coeff is None else em.vals * np.asarray(coeff)[em.elem]
    M = sp.coo_matrix((v, (em.rows, em.cols)), shape=(em.n_u, em.n_u)).tocsr()
    M.sum_duplicates()
    M.sort_indices()
    return M


def mass_contributions(em: ElementMass):
    """Symbolic structure for the per-sample refresh M(k): a CSR pattern plus, for every
    stored nonzero p, the list of (element, unit value) pairs that sum into it.

    Returns (pattern_csr, c_ptr[nnz+1], c_elem[ncontrib], c_val[ncontrib]) with
    M(c).data[p] = sum_{t in c_ptr[p]:c_ptr[p+1]} c[c_elem[t]] * c_val[t]."""
    n = em.n_u
    key = em.rows * n + em.cols
    order = np.argsort(key, kind="stable")
    skey = key[order]
    ukey, start = np.unique(skey, return_index=True)
    c_ptr = np.concatenate([start, [len(skey)]]).astype(np.int32)
    r = (ukey // n).astype(np.int64)
    c = (ukey % n).astype(np.int32)
    rowptr = np.zeros(n + 1, np.int64)
    np.add.at(rowptr, r + 1, 1)
    rowptr = np.cumsum(rowptr).astype(np.int32)
    pat = sp.csr_matrix((np.ones(len(ukey)), c, rowptr), shape=(n, n))
    return pat, c_ptr, em.elem[order].astype(np.int32), em.vals[order].copy()


def prolongation_p0(parent: np.ndarray, n_coarse: int) -> sp.csr_matrix:
    """Piecewise-constant prolongator (children <- parent), the ``ComputeTrueP(sform)``
    of /root/reference/src/PDESampler.cpp:189-193 for nested order-0 spaces."""
    nfine = len(parent)
    return sp.csr_matrix((np.ones(nfine), (np.arange(nfine), parent)), shape=(nfine, n_coarse))
