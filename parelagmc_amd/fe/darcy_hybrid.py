"""Hybridized form of the mixed DARCY system (setup side, numpy) - the algebra of the reference's "Hybridization" branch of
`DarcySolver` (/root/reference/src/DarcySolver.cpp:586,619: the solver factory is handed the block operator, ParELAG's
HybridHdivL2 eliminates (u, p) element by element).  The numpy statement of the reduction the library performs in
`pmc_darcy_create_hybrid` (csrc/darcy.hip::build_hybrid; LAB_NOTES 10.15 and 10.18), pinned against the oracle.

With the flux continuity across interior faces (and the essential value on no-flux boundary faces) imposed by one Lagrange
multiplier per such face, every element keeps its own copy u_e of the fluxes through its faces (GLOBAL face orientation):

    (1 / kappa_e) M_e u_e + b_e p_e + C_e^T lambda_e = f_e          b_e^T u_e = g_e          sum_e C_e u_e = d~

kappa_e = 1 / c(k_e) is the element's coefficient in the form that MULTIPLIES the inverse (kappa = k when M(k) divides by k,
DarcySolver.cpp:479; 1 / k otherwise), C_e = +1 / -1 for the first / second element of a face, d~ = 0 on interior faces and the
essential datum on essential boundary faces.  Pressure-boundary (natural) faces carry no multiplier - the local copy IS the
global unknown there and f_e holds the boundary term.  The local inverse scales with the coefficient,

    [[M_e / kappa, b], [b^T, 0]]^-1 = [[kappa X_e, Y_e], [Y_e^T, -Z_e / kappa]],
    X_e = M_e^-1 - M_e^-1 b (b^T M_e^-1 b)^-1 b^T M_e^-1,   Y_e = M_e^-1 b / (b^T M_e^-1 b),   Z_e = 1 / (b^T M_e^-1 b),

so the multiplier system is LINEAR in the realization's coefficients,

    H(kappa) lambda = rhs(kappa),     H(kappa) = sum_e kappa_e C_e X_e C_e^T,     rhs(kappa) = R kappa + b_0,

and so is the back-substitution  u = kappa_owner (U_0 - U_L lambda) + u_g,   p = P_0 - P_L lambda - z_g / kappa.  H is given as
contribution lists over the elements in exactly the format `DarcyLevel` uses for M(k) (pattern + c_ptr / c_elem / c_val): the
device's element-grouped layout and the refresh lists of its per-realization hierarchies take that format as it is.
"""
from __future__ import annotations

import dataclasses

import numpy as np
import scipy.sparse as sp

from .problems import DarcyLevel
from .rt0 import LevelSpaces


@dataclasses.dataclass
class DarcyHybridLevel:
    n_lambda: int
    faces: np.ndarray            # (n_lambda,) global face (u-dof) of every multiplier
    H_pattern: sp.csr_matrix     # (n_lambda, n_lambda), values = H(kappa = 1)
    h_ptr: np.ndarray            # (nnz + 1,) contributions per stored nonzero, as DarcyLevel.c_ptr
    h_elem: np.ndarray
    h_val: np.ndarray
    R: sp.csr_matrix             # (n_lambda, n_p)
    b0: np.ndarray               # (n_lambda,)
    owner: np.ndarray            # (n_u,) the element whose copy of the flux is reported as the global one
    U0: np.ndarray               # (n_u,)
    UL: sp.csr_matrix            # (n_u, n_lambda)
    ug: np.ndarray               # (n_u,)
    P0: np.ndarray               # (n_p,)
    PL: sp.csr_matrix            # (n_p, n_lambda)
    zg: np.ndarray               # (n_p,)

    def operator(self, kappa: np.ndarray) -> sp.csr_matrix:
        data = np.add.reduceat(kappa[self.h_elem] * self.h_val, self.h_ptr[:-1])
        return sp.csr_matrix((data, self.H_pattern.indices, self.H_pattern.indptr), shape=self.H_pattern.shape)

    def rhs(self, kappa: np.ndarray) -> np.ndarray:
        return self.R @ kappa + self.b0

    def back_substitute(self, kappa: np.ndarray, lam: np.ndarray):
        u = kappa[self.owner] * (self.U0 - self.UL @ lam) + self.ug
        p = self.P0 - self.PL @ lam - self.zg / kappa
        return u, p


def darcy_hybrid_level(space: LevelSpaces, L: DarcyLevel) -> DarcyHybridLevel:
    ft = space.faces
    ef = ft.elem_face
    ne, nfe = ef.shape
    nf = space.n_u
    em = space.emass
    la = (ef[em.elem] == em.rows[:, None]).argmax(axis=1)
    lb = (ef[em.elem] == em.cols[:, None]).argmax(axis=1)
    Me = np.zeros((ne, nfe, nfe))
    Me[em.elem, la, lb] = em.vals
    b = ft.elem_sign.astype(np.float64)                    # B[e, f] = +-1
    Mi = np.linalg.inv(Me)
    Mib = np.einsum("eab,eb->ea", Mi, b)
    sig = np.einsum("ea,ea->e", b, Mib)
    X = Mi - Mib[:, :, None] * Mib[:, None, :] / sig[:, None, None]
    X = 0.5 * (X + X.transpose(0, 2, 1))
    Y = Mib / sig[:, None]
    Z = 1.0 / sig
    first = ft.face_elem[:, 0]
    c = np.where(first[ef] == np.arange(ne)[:, None], 1.0, -1.0)
    isb = ft.face_bdr_attr > 0
    ess = L.ess_mask.astype(bool)
    active = ~isb | ess
    faces = np.nonzero(active)[0]
    new = -np.ones(nf, np.int64)
    new[faces] = np.arange(len(faces))
    nl = len(faces)
    # element shares of the right-hand side: rhs_u of a face goes to its first element (boundary faces have only that one)
    rhs_u, rhs_p = L.rhs[:nf], L.rhs[nf:]
    fe_ = np.where(first[ef] == np.arange(ne)[:, None], rhs_u[ef], 0.0)           # (ne, nfe)
    # H as contribution lists on its own sparsity pattern
    rows = np.repeat(ef, nfe, axis=1).ravel()
    cols = np.tile(ef, (1, nfe)).ravel()
    elem = np.repeat(np.arange(ne), nfe * nfe)
    vals = (c[:, :, None] * X * c[:, None, :]).ravel()
    keep = active[rows] & active[cols]
    r, cc, el, v = new[rows[keep]], new[cols[keep]], elem[keep], vals[keep]
    order = np.lexsort((el, cc, r))
    r, cc, el, v = r[order], cc[order], el[order], v[order]
    key = r * nl + cc
    start = np.concatenate([[True], key[1:] != key[:-1]])
    h_ptr = np.concatenate([np.nonzero(start)[0], [len(key)]]).astype(np.int64)
    pat = sp.csr_matrix((np.add.reduceat(v, h_ptr[:-1]), (r[start], cc[start])), shape=(nl, nl))
    pat.sort_indices()
    # (lexsort by (row, col) is the CSR order of the pattern, so h_ptr indexes its stored entries directly)
    Xf = np.einsum("eab,eb->ea", X, fe_)                   # X_e f_e
    g = rhs_p
    act_e = active[ef]
    R = sp.csr_matrix(((c * Xf)[act_e], (new[ef][act_e], np.repeat(np.arange(ne), nfe).reshape(ne, nfe)[act_e])), shape=(nl, ne))
    b0 = np.zeros(nl)
    np.add.at(b0, new[ef][act_e], (c * Y * g[:, None])[act_e])
    b0 -= np.where(ess[faces], L.ess_data[faces], 0.0)
    # back-substitution, the flux of a face taken from its first element
    owner = first.astype(np.int64)
    lo = (ef[owner] == np.arange(nf)[:, None]).argmax(axis=1)           # local index of the face in its owner
    U0 = Xf[owner, lo]
    ug = (Y * g[:, None])[owner, lo]
    xr = X[owner, lo, :] * c[owner]                                     # row of X_e C_e^T, (nf, nfe)
    colf = ef[owner]                                                    # global faces of the owner, (nf, nfe)
    ok = active[colf]
    UL = sp.csr_matrix((xr[ok], (np.repeat(np.arange(nf), nfe).reshape(nf, nfe)[ok], new[colf][ok])), shape=(nf, nl))
    P0 = np.einsum("ea,ea->e", Y, fe_)
    PL = sp.csr_matrix(((Y * c)[act_e], (np.repeat(np.arange(ne), nfe).reshape(ne, nfe)[act_e], new[ef][act_e])), shape=(ne, nl))
    zg = Z * g
    return DarcyHybridLevel(nl, faces, pat, h_ptr, el.astype(np.int64), v, R, b0, owner, U0, UL, ug, P0, PL, zg)
