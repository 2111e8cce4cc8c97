"""Level hierarchies and the per-level operator data the hot path consumes.

Restates the *setup* the reference performs once before sampling:
  * PDESampler::BuildHierarchy        /root/reference/src/PDESampler.cpp:177-334
  * EmbeddedPDESampler material ids   /root/reference/src/EmbeddedPDESampler.cpp:63-89
  * L2ProjectionPDESampler Gt / RAP   /root/reference/src/L2ProjectionPDESampler.cpp:488-514
  * DarcySolver hierarchy, forcing, BCs, observation functional
                                      /root/reference/src/DarcySolver.cpp:194-227,297-319,360-414
Level 0 is the finest level, as in ParELAG.
"""
from __future__ import annotations

import dataclasses
import math
from typing import List, Optional, Sequence

import numpy as np
import scipy.sparse as sp

from .mesh import Mesh, refine_uniform
from .rt0 import LevelSpaces, build_spaces, mass_contributions, mass_matrix, prolongation_p0


def matern_coefficient(corlen: float, dim: int) -> float:
    """g of /root/reference/src/Utilities.hpp:188-200 (follows the code: Gamma(nu+d))."""
    d = float(dim)
    nu = 2.0 - d / 2.0
    c = (16.0 * math.atan(1.0)) ** (0.5 * d)
    k = (1.0 / corlen) ** (2.0 * nu)
    return math.sqrt(c * math.gamma(nu + d) * k / math.gamma(nu))


@dataclasses.dataclass
class Hierarchy:
    spaces: List[LevelSpaces]          # [0] finest
    P: List[sp.csr_matrix]             # P[i]: n_s(i) x n_s(i+1)

    @property
    def nlevels(self) -> int:
        return len(self.spaces)


def build_hierarchy(coarse: Mesh, n_refine: int) -> Hierarchy:
    """Refine `coarse` n_refine times; returns n_refine+1 nested levels, finest first."""
    meshes = [coarse]
    parents = []
    for _ in range(n_refine):
        fine, parent = refine_uniform(meshes[-1])
        meshes.append(fine)
        parents.append(parent)
    meshes.reverse()
    parents.reverse()
    spaces = [build_spaces(m) for m in meshes]
    P = [prolongation_p0(parents[i], spaces[i + 1].n_s) for i in range(n_refine)]
    return Hierarchy(spaces, P)


# --------------------------------------------------------------------------- SPDE sampler
@dataclasses.dataclass
class SamplerLevel:
    n_u: int
    n_s: int
    M: sp.csr_matrix          # boundary rows/cols eliminated -> identity (PDESampler.cpp:236-241)
    B: sp.csr_matrix          # boundary columns removed          (PDESampler.cpp:243-246)
    w_diag: np.ndarray        # diag(W) = |e|
    P: Optional[sp.csr_matrix]  # to the next coarser level, None on the last

    @property
    def w_sqrt(self) -> np.ndarray:
        return np.sqrt(self.w_diag)

    @property
    def nnz(self) -> int:
        """nnz of the block operator [M Bt; B -aW] (PDESampler.cpp:265)."""
        return int(self.M.nnz + 2 * self.B.nnz + self.n_s)


@dataclasses.dataclass
class SamplerProblem:
    levels: List[SamplerLevel]
    n_mc_levels: int          # levels the MLMC hierarchy uses; the rest only feed the V-cycle
    corlen: float
    alpha: float
    matern_g: float
    dim: int
    lognormal: bool
    # embedded variants: per MC level, indices of the original-mesh elements (attr==1)
    orig_index: Optional[List[np.ndarray]] = None


def sampler_level_ops(sp_: LevelSpaces, P) -> SamplerLevel:
    ess = sp_.faces.face_elem[:, 1] < 0            # every boundary face is essential (:210-214)
    M = mass_matrix(sp_.emass)
    keep = sp.diags((~ess).astype(np.float64))
    M = (keep @ M @ keep + sp.diags(ess.astype(np.float64))).tocsr()
    M.eliminate_zeros()
    M.sort_indices()
    B = (sp_.B @ keep).tocsr()
    B.eliminate_zeros()
    B.sort_indices()
    return SamplerLevel(sp_.n_u, sp_.n_s, M, B, sp_.vol.copy(), P)


def build_sampler_problem(h: Hierarchy, corlen=0.1, lognormal=False, n_mc_levels=None,
                          embedded=False) -> SamplerProblem:
    dim = h.spaces[0].mesh.dim
    levels = [sampler_level_ops(h.spaces[i], h.P[i] if i < h.nlevels - 1 else None)
              for i in range(h.nlevels)]
    nmc = h.nlevels if n_mc_levels is None else n_mc_levels
    orig = None
    if embedded:
        orig = [np.nonzero(h.spaces[i].mesh.elem_attr == 1)[0].astype(np.int32) for i in range(nmc)]
    return SamplerProblem(levels, nmc, corlen, 1.0 / (corlen * corlen), matern_coefficient(corlen, dim),
                          dim, lognormal, orig)


def l2_projection_ops(h_embed: Hierarchy, orig_index: Sequence[np.ndarray]):
    """Gt and 1/|e_orig| for an *element-aligned* original mesh (the original elements are the
    attr-1 elements of the embedded mesh).  Gt[i,j] = |e_orig,i ∩ e_emb,j|
    (L2ProjectionPDESampler.cpp:489-505); coarse levels by RAP with the P0 prolongators
    (:512-513), which for aligned nested meshes is again the volume-weighted selection."""
    out = []
    for lvl, idx in enumerate(orig_index):
        vol = h_embed.spaces[lvl].vol
        n_emb = h_embed.spaces[lvl].n_s
        Gt = sp.csr_matrix((vol[idx], (np.arange(len(idx)), idx)), shape=(len(idx), n_emb))
        out.append((Gt, 1.0 / vol[idx]))
    return out


# --------------------------------------------------------------------------- Darcy
@dataclasses.dataclass
class DarcyLevel:
    n_u: int
    n_p: int
    M_pattern: sp.csr_matrix     # pattern (values = M(1))
    c_ptr: np.ndarray            # (nnz+1,) contributions per stored nonzero
    c_elem: np.ndarray
    c_val: np.ndarray
    B: sp.csr_matrix             # (n_p, n_u), no elimination (DarcySolver.cpp:203-207)
    rhs: np.ndarray              # (n_u+n_p,)
    ess_mask: np.ndarray         # (n_u,) uint8
    ess_data: np.ndarray         # (n_u,)
    obs: np.ndarray              # (n_u+n_p,)
    P: Optional[sp.csr_matrix]   # P0 prolongator to next coarser level (V-cycle on the Schur block)

    @property
    def ndofs(self) -> int:
        return self.n_u + self.n_p


@dataclasses.dataclass
class DarcyProblem:
    levels: List[DarcyLevel]
    n_mc_levels: int
    k_divides: bool = True


def elements_near_points(mesh: Mesh, points, eps: float) -> np.ndarray:
    """(npoints, nelem) bool: element e is marked for point j when the point lies in the element's bounding box enlarged
    by eps, lower bounds inclusive and upper bounds exclusive - ChangeMeshAttributes of the reference
    (/root/reference/src/MeshUtilities.cpp:268-334), which labels the elements a local pressure average is taken over."""
    pts = np.atleast_2d(np.asarray(points, dtype=np.float64))
    xv = mesh.verts[mesh.elems]                      # (nelem, nverts, dim)
    lo, hi = xv.min(axis=1) - eps, xv.max(axis=1) + eps
    return np.stack([np.all((lo <= x[None, :]) & (x[None, :] < hi), axis=1) for x in pts])


def build_darcy_problem(h: Hierarchy, ess_attr, obs_attr, inflow_attr, n_mc_levels=None,
                        p_inflow=-1.0, qoi="eff_perm", k_divides=True, qoi_point=(0.5, 0.5, 0.5),
                        qoi_eps=0.1) -> DarcyProblem:
    """Default test problem of the reference drivers (examples/DarcyTest.cpp:186-215,
    example_helpers/CreateDarcyParameterList.hpp:32-36): u.n = 0 on `ess_attr`, boundary
    pressure coefficient `p_inflow` on `inflow_attr`, QoI = boundary flux through `obs_attr`.
    Volume forcing f = 0 and q = 0, so rhs / obs live on boundary faces only and their
    restriction with P^T (DarcySolver.cpp:313-314,411-412) equals direct evaluation on each
    (nested) level.  qoi: "eff_perm" (boundary flux through obs_attr, default), "p_int" (integral of the pressure,
    DarcySolver.cpp:267-295) or "local_avg_p" (integral of the pressure over the fine elements around qoi_point,
    BuildPWObservationFunctional_p, DarcySolver.cpp:321-358; defaults of examples/MLMC.cpp:105-109), the three the
    reference's drivers select (examples/MLMC.cpp:228-236)."""
    ess_attr = np.asarray(ess_attr, dtype=bool)
    obs_attr = np.asarray(obs_attr, dtype=bool)
    inflow_attr = np.asarray(inflow_attr, dtype=bool)
    levels = []
    obs_p = None
    if qoi == "local_avg_p":
        # fine-level indicator weighted by the element volumes (DomainLFIntegrator of a restricted unit coefficient),
        # restricted to the coarser levels with P^T (:353-354)
        s0 = h.spaces[0]
        mark = elements_near_points(s0.mesh, np.asarray(qoi_point, dtype=np.float64)[None, :s0.mesh.dim], qoi_eps)[0]
        obs_p = [np.where(mark, s0.vol, 0.0)]
        for P in h.P:
            obs_p.append(P.T @ obs_p[-1])
    for i, s in enumerate(h.spaces):
        fattr = s.faces.face_bdr_attr
        isb = fattr > 0
        a = np.where(isb, fattr - 1, 0)
        if isb.any() and a.max() >= len(ess_attr):
            raise ValueError("boundary attribute exceeds the attribute arrays")
        ess = isb & ess_attr[a]
        obs_f = isb & obs_attr[a]
        inflow = isb & inflow_attr[a]
        n_u, n_p = s.n_u, s.n_s
        rhs = np.zeros(n_u + n_p)
        rhs[:n_u][inflow] = p_inflow          # int_F p_bdr phi_F.n = p_bdr * (unit outward flux)
        obs = np.zeros(n_u + n_p)
        if qoi == "eff_perm":
            obs[:n_u][obs_f] = 1.0
        elif qoi == "p_int":
            obs[n_u:] = s.vol
        elif qoi == "local_avg_p":
            obs[n_u:] = obs_p[i]
        else:
            raise ValueError(qoi)
        pat, c_ptr, c_elem, c_val = mass_contributions(s.emass)
        pat = mass_matrix(s.emass)
        levels.append(DarcyLevel(n_u, n_p, pat, c_ptr, c_elem, c_val, s.B.tocsr().copy(), rhs,
                                 ess.astype(np.uint8), np.zeros(n_u), obs,
                                 h.P[i] if i < h.nlevels - 1 else None))
        levels[-1].B.sort_indices()
    return DarcyProblem(levels, h.nlevels if n_mc_levels is None else n_mc_levels, k_divides)
