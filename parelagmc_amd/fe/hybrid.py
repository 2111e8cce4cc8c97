"""Hybridized form of the SPDE sampler's mixed system (setup side, numpy).

The reference offers hybridization as an alternative solver of `PDESampler::Eval`
(/root/reference/src/PDESampler.cpp:291,307-311,383-389,451-475: "Hybridization" in the parameter list,
examples/example_parameterlists/example_parameters.xml:205-213; ParELAG's HybridHdivL2 does the element-local
elimination).  This module is the stand-in for that setup step: from the element matrices of the RT0 / P0 pair it builds,
per level,

    H      = sum_e C_e X_e C_e^T          SPD, one unknown (Lagrange multiplier) per face, the pattern of M
    G      = sum_e C_e y_e                n_lambda x n_s:  rhs_lambda = G f
    z_diag                                n_s:             s = z f - G^T lambda

where, element by element, [[X, y], [y^T, z]] is the inverse of the local saddle-point matrix [[M_e, b_e^T], [b_e, -alpha w_e]]
(broken flux dofs in the GLOBAL face orientation) and C_e = +1 / -1 for the first / second element of a face, +1 alone on a
boundary face - every boundary face is essential for the sampler (u.n = 0, src/PDESampler.cpp:210-214), which in hybrid form
is the same continuity constraint with nothing on the other side.  H lambda = G f followed by s = z f - G^T lambda is an EXACT
algebraic reformulation of [M B^T; B -alpha W][u; s] = [0; f]: the field s is the same to rounding (tests/test_fe.py).
"""
from __future__ import annotations

import dataclasses
from typing import List, Optional

import numpy as np
import scipy.sparse as sp

from .problems import Hierarchy, matern_coefficient
from .rt0 import LevelSpaces


@dataclasses.dataclass
class HybridLevel:
    n_lambda: int
    n_s: int
    H: sp.csr_matrix          # (n_lambda, n_lambda) SPD
    G: sp.csr_matrix          # (n_lambda, n_s)
    z_diag: np.ndarray        # (n_s,) negative: the (s, s) entry of the local inverses
    w_diag: np.ndarray        # (n_s,)
    P: Optional[sp.csr_matrix]  # s-space prolongator to the next coarser level (couples xi between levels), None on the last

    @property
    def n_u(self) -> int:
        """flux unknowns of the saddle-point form of this level: one per face, as the multipliers"""
        return self.n_lambda

    @property
    def nnz(self) -> int:
        return int(self.H.nnz)


@dataclasses.dataclass
class HybridSamplerProblem:
    levels: List[HybridLevel]
    n_mc_levels: int
    corlen: float
    alpha: float
    matern_g: float
    dim: int
    lognormal: bool
    # embedded variants: per level, indices of the original-mesh elements (attr == 1), as SamplerProblem.orig_index
    orig_index: Optional[List[np.ndarray]] = None


def hybrid_level_ops(space: LevelSpaces, alpha: float, P, builder=None) -> HybridLevel:
    """builder: None = the numpy elimination below; a callable (space, alpha) -> (H, G, z_diag) = somebody else's - the
    library's own pmc_hybrid_build through capi.library_hybrid_builder (what a C++ caller of libpmc.so uses)"""
    if builder is not None:
        H, G, z = builder(space, alpha)
        return HybridLevel(space.n_u, space.n_s, H, G, z, space.vol.copy(), P)
    ft = space.faces
    ef = ft.elem_face
    ne, nfe = ef.shape
    nf = space.n_u
    em = space.emass
    # element mass matrices Me[e, a, b] (global face orientation); hexes store no cross-direction zeros
    la = (ef[em.elem] == em.rows[:, None]).argmax(axis=1)
    lb = (ef[em.elem] == em.cols[:, None]).argmax(axis=1)
    Me = np.zeros((ne, nfe, nfe))
    Me[em.elem, la, lb] = em.vals
    sign = ft.elem_sign.astype(np.float64)                 # b_e: B[e, f] = +-1
    A = np.zeros((ne, nfe + 1, nfe + 1))
    A[:, :nfe, :nfe] = Me
    A[:, :nfe, nfe] = sign
    A[:, nfe, :nfe] = sign
    A[:, nfe, nfe] = -alpha * space.vol
    Ainv = np.linalg.inv(A)
    X, y, z = Ainv[:, :nfe, :nfe], Ainv[:, :nfe, nfe], Ainv[:, nfe, nfe]
    first = ft.face_elem[:, 0]
    c = np.where(first[ef] == np.arange(ne)[:, None], 1.0, -1.0)
    rows = np.repeat(ef, nfe, axis=1).ravel()
    cols = np.tile(ef, (1, nfe)).ravel()
    H = sp.coo_matrix(((c[:, :, None] * X * c[:, None, :]).ravel(), (rows, cols)), shape=(nf, nf)).tocsr()
    H.sum_duplicates()
    H.sort_indices()
    G = sp.coo_matrix(((c * y).ravel(), (ef.ravel(), np.repeat(np.arange(ne), nfe))), shape=(nf, ne)).tocsr()
    G.sort_indices()
    return HybridLevel(nf, ne, H, G, z.copy(), space.vol.copy(), P)


def build_hybrid_sampler_problem(h: Hierarchy, corlen=0.1, lognormal=False, n_mc_levels=None,
                                 embedded=False, builder=None) -> HybridSamplerProblem:
    """The hybridized twin of build_sampler_problem (same arguments).  Only the Monte Carlo levels are built: the
    multiplier system brings its own algebraic hierarchy, coarser mesh levels have no role in it."""
    dim = h.spaces[0].mesh.dim
    alpha = 1.0 / (corlen * corlen)
    nmc = h.nlevels if n_mc_levels is None else n_mc_levels
    levels = [hybrid_level_ops(h.spaces[i], alpha, h.P[i] if i < nmc - 1 else None, builder) for i in range(nmc)]
    orig = None
    if embedded:
        orig = [np.nonzero(h.spaces[i].mesh.elem_attr == 1)[0].astype(np.int32) for i in range(nmc)]
    return HybridSamplerProblem(levels, nmc, corlen, alpha, matern_coefficient(corlen, dim), dim, lognormal, orig)
