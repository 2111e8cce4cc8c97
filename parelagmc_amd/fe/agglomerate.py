"""Algebraically agglomerated coarse levels (setup side, host only).

ParELAG coarsens an unstructured mesh by graph partitioning (METIS) of the element graph
(/root/reference/src/Utilities.cpp:125-155): coarse "elements" are agglomerates of very different sizes, coarse "faces"
are the sets of fine faces two agglomerates share (any number of them), and the coarse operators are Galerkin products
with the prolongators of the coarse de Rham sequence (src/PDESampler.cpp:189-193 ComputeTrueP; M_c = Pu^T M Pu,
B_c = Ps^T B Pu, W_c = Ps^T W Ps).  METIS and ParELAG are not available here; this module produces operator sets with the
SAME algebraic structure from a greedy graph agglomeration, so that the device path is exercised on what such a
hierarchy hands over - not on nested refinement, where P_s is a uniform 8-children injection and every dof touches at
most two equal cells:

  * agglomerates of non-uniform size (breadth-first growth to prescribed, varying target sizes),
  * one coarse flux dof per agglomerated face = all fine faces between the same two agglomerates (or on the same
    boundary attribute of one agglomerate); fine extension by area weights, optionally smoothed with one damped Jacobi
    step of the fine mass matrix (the support of a coarse flux dof then reaches into third agglomerates, as the
    energy-minimising extensions of an AMGe hierarchy do near agglomerate corners),
  * P_s = agglomerate indicator with one entry per fine element, unit or non-unit weights.

Any full-column-rank pair (Pu, Ps) with disjoint Ps columns gives a valid coarse saddle-point system of the form the C ABI
accepts (M_c SPD, W_c diagonal, B_c onto); parity is checked against the oracle's direct solves on the SAME matrices.
"""
from __future__ import annotations

import dataclasses
from typing import List, Optional, Sequence

import numpy as np
import scipy.sparse as sp

from .problems import DarcyLevel, DarcyProblem, SamplerLevel, SamplerProblem, matern_coefficient
from .rt0 import LevelSpaces, mass_matrix


def greedy_agglomerates(adj: sp.csr_matrix, sizes: Sequence[int], seed: int = 0) -> np.ndarray:
    """Partition the vertices of the graph `adj` into connected agglomerates grown breadth-first to the target sizes
    `sizes` (cycled); leftovers smaller than 3 join a neighbouring agglomerate.  Returns agg[v] in 0..nagg-1."""
    n = adj.shape[0]
    adj = adj.tocsr()
    agg = np.full(n, -1, np.int64)
    rng = np.random.Generator(np.random.PCG64(seed))
    order = rng.permutation(n)
    nagg = 0
    for start in order:
        if agg[start] >= 0:
            continue
        target = int(sizes[nagg % len(sizes)])
        members = [start]
        agg[start] = nagg
        head = 0
        while head < len(members) and len(members) < target:
            v = members[head]
            head += 1
            for w in adj.indices[adj.indptr[v]:adj.indptr[v + 1]]:
                if agg[w] < 0 and len(members) < target:
                    agg[w] = nagg
                    members.append(w)
        nagg += 1
    # merge tiny agglomerates (what the breadth-first growth leaves between finished ones) into a neighbour, whole
    for _ in range(4):
        counts = np.bincount(agg, minlength=nagg)
        small = np.nonzero((counts > 0) & (counts < 3))[0]
        if len(small) == 0:
            break
        for a in small:
            mem = np.nonzero(agg == a)[0]
            nb = [agg[w] for v in mem for w in adj.indices[adj.indptr[v]:adj.indptr[v + 1]] if agg[w] != a]
            if nb:
                agg[mem] = nb[0]
    _, agg = np.unique(agg, return_inverse=True)
    return agg


@dataclasses.dataclass
class AggLevel:
    """Coarse level produced from a finer one: operators WITHOUT boundary elimination plus the transfers."""
    n_u: int
    n_s: int
    M: sp.csr_matrix              # Pu^T M_f Pu (unit coefficient)
    B: sp.csr_matrix              # Ps^T B_f Pu
    w_diag: np.ndarray            # diag(Ps^T W_f Ps)
    Ps: sp.csr_matrix             # n_s(fine) x n_s
    Pu: sp.csr_matrix             # n_u(fine) x n_u
    bdr_attr: np.ndarray          # (n_u,) boundary attribute of a coarse flux dof, 0 if interior
    agg: np.ndarray               # fine element -> agglomerate
    elem_adj: sp.csr_matrix       # agglomerate adjacency (for the next coarsening)
    Me: Optional[List[sp.csr_matrix]] = None   # per agglomerate: its contribution to M (for M(k) on this level)


def _finest_as_agg(s: LevelSpaces) -> AggLevel:
    ne = s.n_s
    fe = s.faces.face_elem
    inter = fe[:, 1] >= 0
    adj = sp.coo_matrix((np.ones(inter.sum()), (fe[inter, 0], fe[inter, 1])), shape=(ne, ne))
    adj = (adj + adj.T).tocsr()
    return AggLevel(s.n_u, s.n_s, mass_matrix(s.emass), s.B.tocsr(), s.vol.copy(), sp.identity(ne, format="csr"),
                    sp.identity(s.n_u, format="csr"), s.faces.face_bdr_attr.astype(np.int64), np.arange(ne), adj)


def coarsen(fine: AggLevel, face_weight: np.ndarray, sizes: Sequence[int], seed: int = 0, ps_weights: str = "unit",
            smooth_pu: float = 0.0, fine_elem_M: Optional[List[sp.csr_matrix]] = None):
    """One agglomerated level below `fine`; returns (level, weights of its flux dofs).  face_weight: positive weight of every fine flux dof (face areas on the finest
    level, sums of them below).  fine_elem_M: per fine element its contribution to fine.M (gives AggLevel.Me)."""
    agg = greedy_agglomerates(fine.elem_adj, sizes, seed)
    na = int(agg.max()) + 1
    nf_e = fine.n_s
    # ---- Ps
    if ps_weights == "unit":
        pw = np.ones(nf_e)
    else:                          # non-unit weights: W_c stays diagonal, transfers are no longer 0/1 injections
        pw = 0.5 + (np.arange(nf_e) % 3) * 0.25
    Ps = sp.csr_matrix((pw, (np.arange(nf_e), agg)), shape=(nf_e, na))
    # ---- agglomerated faces: fine flux dofs grouped by the unordered pair of agglomerates they separate
    Bf = fine.B.tocsc()
    nfu = fine.n_u
    a_lo = np.full(nfu, -1, np.int64)
    a_hi = np.full(nfu, -1, np.int64)
    sgn = np.zeros(nfu)
    for f in range(nfu):
        rows = Bf.indices[Bf.indptr[f]:Bf.indptr[f + 1]]
        vals = Bf.data[Bf.indptr[f]:Bf.indptr[f + 1]]
        ags = agg[rows]
        if len(rows) == 1:
            a_lo[f] = ags[0]
            sgn[f] = np.sign(vals[0])                  # outward from the only agglomerate
        elif len(rows) == 2 and ags[0] != ags[1]:
            i = int(np.argmin(ags))
            a_lo[f], a_hi[f] = ags[i], ags[1 - i]
            sgn[f] = np.sign(vals[i])                  # positive = from the lower to the higher agglomerate
        # faces interior to an agglomerate (or flux dofs touching > 2 elements) carry no coarse dof of their own
    has = a_lo >= 0
    key = np.where(a_hi >= 0, a_lo * (na + 1) + a_hi + 1, -(a_lo * 64 + np.maximum(fine.bdr_attr, 0)) - 1)
    ukey, inv = np.unique(key[has], return_inverse=True)
    nF = len(ukey)
    fidx = np.nonzero(has)[0]
    wsum = np.bincount(inv, weights=face_weight[fidx], minlength=nF)
    Pu = sp.csr_matrix((sgn[fidx] * face_weight[fidx] / wsum[inv], (fidx, inv)), shape=(nfu, nF))
    if smooth_pu > 0.0:
        dM = fine.M.diagonal()
        Pu = (Pu - smooth_pu * sp.diags(1.0 / dM) @ (fine.M @ Pu)).tocsr()
        Pu.data[np.abs(Pu.data) < 1e-14] = 0.0
        Pu.eliminate_zeros()
    battr = np.zeros(nF, np.int64)
    bsel = a_hi[fidx] < 0
    battr[inv[bsel]] = fine.bdr_attr[fidx[bsel]]
    M = (Pu.T @ fine.M @ Pu).tocsr()
    B = (Ps.T @ fine.B @ Pu).tocsr()
    B.data[np.abs(B.data) < 1e-13] = 0.0
    B.eliminate_zeros()
    w = np.asarray((Ps.multiply(Ps)).T @ fine.w_diag).ravel()
    adj = (Ps.T @ fine.elem_adj @ Ps).tocsr()
    adj.setdiag(0)
    adj.eliminate_zeros()
    adj.data[:] = 1.0
    Me = None
    if fine_elem_M is not None:
        Me = []
        for a in range(na):
            acc = None
            for e in np.nonzero(agg == a)[0]:
                acc = fine_elem_M[e] if acc is None else acc + fine_elem_M[e]
            Me.append((Pu.T @ acc @ Pu).tocsr())
    return AggLevel(nF, na, M, B, w, Ps, Pu, battr, agg, adj, Me), wsum


def _eliminate(M, B, ess):
    keep = sp.diags((~ess).astype(np.float64))
    Me = (keep @ M @ keep + sp.diags(ess.astype(np.float64))).tocsr()
    Me.eliminate_zeros()
    Me.sort_indices()
    Be = (B @ keep).tocsr()
    Be.eliminate_zeros()
    Be.sort_indices()
    return Me, Be


def _element_mass_matrices(s: LevelSpaces) -> List[sp.csr_matrix]:
    em = s.emass
    out = []
    order = np.argsort(em.elem, kind="stable")
    bounds = np.searchsorted(em.elem[order], np.arange(s.n_s + 1))
    for e in range(s.n_s):
        sel = order[bounds[e]:bounds[e + 1]]
        out.append(sp.csr_matrix((em.vals[sel], (em.rows[sel], em.cols[sel])), shape=(s.n_u, s.n_u)))
    return out


def build_agglomerated_levels(s0: LevelSpaces, nlevels: int, sizes=(5, 9, 14, 7, 11), seed=3, ps_weights="unit",
                              smooth_pu=0.0, with_element_matrices=False) -> List[AggLevel]:
    """[finest (as AggLevel), agglomerated level 1, ...]: each level coarsened from the previous one."""
    # face areas of the finest level from the mass matrix are not available; use |B| column sums * a geometric weight:
    # the area of a face equals the jump of the element volumes' derivative - simpler: take the RT0 normalisation, where a
    # flux dof is the TOTAL flux through the face, so unit weights per unit area are obtained from the diagonal of M
    # (M_ff ~ h / area): any positive weights give a valid (full-rank) prolongator, areas only make it a sensible one
    levels = [_finest_as_agg(s0)]
    X = s0.mesh.verts[s0.faces.face_verts]
    if s0.mesh.etype == "tet":
        area = 0.5 * np.linalg.norm(np.cross(X[:, 1] - X[:, 0], X[:, 2] - X[:, 0]), axis=1)
    elif s0.mesh.etype == "tri":
        area = np.linalg.norm(X[:, 1] - X[:, 0], axis=1)
    else:
        ext = X.max(axis=1) - X.min(axis=1)
        area = np.prod(np.where(ext > 0, ext, 1.0), axis=1)
    wts = area
    elemM = _element_mass_matrices(s0) if with_element_matrices else None
    for l in range(1, nlevels):
        # only the coarsest level is smoothed: the next coarsening reads the agglomerated faces off the sparsity of B
        lvl, wts = coarsen(levels[-1], wts, sizes, seed + l, ps_weights, smooth_pu if l == nlevels - 1 else 0.0, elemM)
        elemM = lvl.Me
        levels.append(lvl)
    return levels


def build_agglomerated_sampler_problem(s0: LevelSpaces, nlevels: int, corlen=0.1, lognormal=False, **kw) -> SamplerProblem:
    lv = build_agglomerated_levels(s0, nlevels, **kw)
    out = []
    for i, L in enumerate(lv):
        ess = L.bdr_attr > 0                              # every boundary flux dof is essential (PDESampler.cpp:210-214)
        M, B = _eliminate(L.M, L.B, ess)
        P = lv[i + 1].Ps if i + 1 < len(lv) else None
        out.append(SamplerLevel(L.n_u, L.n_s, M, B, L.w_diag.copy(), P))
    dim = s0.mesh.dim
    return SamplerProblem(out, nlevels, corlen, 1.0 / (corlen * corlen), matern_coefficient(corlen, dim), dim, lognormal)


def build_agglomerated_darcy_problem(s0: LevelSpaces, nlevels: int, ess_attr, obs_attr, inflow_attr, p_inflow=-1.0,
                                     k_divides=True, **kw) -> DarcyProblem:
    """Darcy hierarchy on the agglomerated levels: M(k) = sum_A c(k_A) Me_A with one coefficient per agglomerate, right-hand
    side / observation functional restricted with Pu^T (src/DarcySolver.cpp:313-314,411-412)."""
    lv = build_agglomerated_levels(s0, nlevels, with_element_matrices=True, **kw)
    ess_attr, obs_attr, inflow_attr = (np.asarray(a, dtype=bool) for a in (ess_attr, obs_attr, inflow_attr))
    out = []
    rhs_u = obs_u = None
    for i, L in enumerate(lv):
        fattr = L.bdr_attr
        isb = fattr > 0
        a = np.where(isb, fattr - 1, 0)
        ess = isb & ess_attr[a]
        if i == 0:
            rhs_u = np.where(isb & inflow_attr[a], p_inflow, 0.0)
            obs_u = np.where(isb & obs_attr[a], 1.0, 0.0)
        else:
            rhs_u = L.Pu.T @ rhs_u
            obs_u = L.Pu.T @ obs_u
        # contributions per stored nonzero of the pattern
        if i == 0:
            from .rt0 import mass_contributions
            pat, c_ptr, c_elem, c_val = mass_contributions(s0.emass)
            pat = mass_matrix(s0.emass)
        else:
            pat = L.M.copy()
            pat.sort_indices()
            rows = np.repeat(np.arange(pat.shape[0]), np.diff(pat.indptr))
            pos = {}
            for p_, (r_, c_) in enumerate(zip(rows, pat.indices)):
                pos[(int(r_), int(c_))] = p_
            lists = [[] for _ in range(pat.nnz)]
            for A, MA in enumerate(L.Me):
                MA = MA.tocoo()
                for r_, c_, v_ in zip(MA.row, MA.col, MA.data):
                    if v_ != 0.0 and (int(r_), int(c_)) in pos:
                        lists[pos[(int(r_), int(c_))]].append((A, v_))
            c_ptr = np.concatenate([[0], np.cumsum([len(x) for x in lists])]).astype(np.int32)
            c_elem = np.array([a_ for x in lists for a_, _ in x], np.int32)
            c_val = np.array([v_ for x in lists for _, v_ in x], np.float64)
        rhs = np.concatenate([rhs_u, np.zeros(L.n_s)])
        obs = np.concatenate([obs_u, np.zeros(L.n_s)])
        B = L.B.tocsr().copy()
        B.sort_indices()
        out.append(DarcyLevel(L.n_u, L.n_s, pat, c_ptr, c_elem, c_val, B, rhs, ess.astype(np.uint8), np.zeros(L.n_u), obs,
                              lv[i + 1].Ps if i + 1 < len(lv) else None))
    return DarcyProblem(out, nlevels, k_divides)
