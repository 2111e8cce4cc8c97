"""ORACLE (test infrastructure): RT0/P0 operators on Cartesian hexahedra from closed forms.

Independent of the product's finite-element builders: nothing of ``parelagmc_amd`` is imported
here.  ``oracle/sampler_oracle.py`` and ``oracle/darcy_oracle.py`` consume the arrays the HIP path
receives (``parelagmc_amd/fe``); this module builds the SAME operators a second time, from the
formulas alone, on the one mesh family where they have a closed form - axis-aligned boxes cut into
nx x ny x nz equal cells - so that a wrong mass entry, orientation or boundary elimination in
``fe/rt0.py`` / ``fe/problems.py`` shows up as an entry-wise difference (tests/test_fe_ref.py) and
the golden fixtures can be regenerated without the product's builders (tests/golden/make_golden.py).

What is restated (reference lines, all under /root/reference):
  * operator definition of the sampler   src/PDESampler.cpp:232-258 (M, W, D; all boundary faces
                                         essential :210-214; B = W D; w_sqrt; -alpha W)
  * level coupling                       src/PDESampler.cpp:189-193 (P = ComputeTrueP(sform)),
                                         :423-438 (rhs = -g W^1/2 xi, restricted with P^T)
  * scaling coefficient                  src/Utilities.hpp:188-200 (the code's Gamma(nu + d))
  * Darcy system, BCs, QoI               src/DarcySolver.cpp:386-414 (rhs), :360-384 (ess_data),
                                         :297-319 (observation functional), :472-520 (assemble),
                                         :416-437 (Q, C)
  * hybridization branch of Eval         src/PDESampler.cpp:291,307-311,451-480 (RefHybrid: element-local elimination)
  * element matrices                     SURVEY.md Appendix A.5: cell (hx, hy, hz); a u-dof is the
                                         total flux through a face along +axis; the two faces of one
                                         direction couple with  h_a / A_a * [[1/3, 1/6], [1/6, 1/3]],
                                         A_a = |cell| / h_a; W = diag(|cell|); B[e, f] = +1 on the
                                         high face, -1 on the low face (outward = +).

Numbering (this module's own, deliberately not the product's): cell (i, j, k) -> i + nx (j + ny k);
x-faces (i, j, k), i = 0..nx, first, then y-faces, then z-faces.  Boundary attributes as MFEM's
Cartesian hex meshes number them: 1 z-low, 2 y-low, 3 x-high, 4 y-high, 5 x-low, 6 z-high.
"""
from __future__ import annotations

import math

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla


class HexLevel:
    """One uniform box grid with its RT0/P0 operators (global +axis face orientation)."""

    def __init__(self, n, size, origin=(0.0, 0.0, 0.0)):
        self.n = tuple(int(v) for v in n)
        self.size = tuple(float(v) for v in size)
        self.origin = tuple(float(v) for v in origin)
        nx, ny, nz = self.n
        self.h = tuple(s / m for s, m in zip(self.size, self.n))
        self.vol = self.h[0] * self.h[1] * self.h[2]
        self.n_s = nx * ny * nz
        self.nface = ((nx + 1) * ny * nz, nx * (ny + 1) * nz, nx * ny * (nz + 1))
        self.foff = (0, self.nface[0], self.nface[0] + self.nface[1])
        self.n_u = sum(self.nface)

    # ------------------------------------------------------------------ numbering
    def cell_index(self, i, j, k):
        nx, ny, _ = self.n
        return i + nx * (j + ny * k)

    def face_index(self, axis, i, j, k):
        """face of direction `axis` at grid position (i, j, k); the index along `axis` runs to n[axis] inclusive"""
        nx, ny, nz = self.n
        dims = [nx, ny, nz]
        dims[axis] += 1
        return self.foff[axis] + i + dims[0] * (j + dims[1] * k)

    def _cells(self):
        nx, ny, nz = self.n
        k, j, i = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
        return i.ravel(), j.ravel(), k.ravel()

    def cell_faces(self, axis):
        """(low face, high face) of every cell along `axis`, cells in index order"""
        i, j, k = self._cells()
        lo = self.face_index(axis, i, j, k)
        d = [0, 0, 0]
        d[axis] = 1
        hi = self.face_index(axis, i + d[0], j + d[1], k + d[2])
        return lo, hi

    def cell_centroids(self):
        i, j, k = self._cells()
        return np.stack([self.origin[0] + (i + 0.5) * self.h[0], self.origin[1] + (j + 0.5) * self.h[1],
                         self.origin[2] + (k + 0.5) * self.h[2]], axis=1)

    def face_centroids(self):
        out = np.zeros((self.n_u, 3))
        for axis in range(3):
            dims = list(self.n)
            dims[axis] += 1
            k, j, i = np.meshgrid(np.arange(dims[2]), np.arange(dims[1]), np.arange(dims[0]), indexing="ij")
            idx = [i.ravel().astype(float), j.ravel().astype(float), k.ravel().astype(float)]
            for a in range(3):
                if a != axis:
                    idx[a] = idx[a] + 0.5
            f = self.foff[axis] + np.arange(self.nface[axis])
            for a in range(3):
                out[f, a] = self.origin[a] + idx[a] * self.h[a]
        return out

    def boundary_attribute(self):
        """(n_u,) 0 for interior faces, 1..6 for boundary faces (MFEM Cartesian-hex numbering, see module header)"""
        attr = np.zeros(self.n_u, dtype=np.int64)
        low = (5, 2, 1)
        high = (3, 4, 6)
        for axis in range(3):
            dims = list(self.n)
            dims[axis] += 1
            k, j, i = np.meshgrid(np.arange(dims[2]), np.arange(dims[1]), np.arange(dims[0]), indexing="ij")
            pos = (i, j, k)[axis].ravel()
            f = self.foff[axis] + np.arange(self.nface[axis])
            attr[f[pos == 0]] = low[axis]
            attr[f[pos == self.n[axis]]] = high[axis]
        return attr

    # ------------------------------------------------------------------ operators
    def w_diag(self):
        return np.full(self.n_s, self.vol)

    def divergence(self):
        """B = W D, entries +-1: +1 on the high face of a cell (outward along +axis), -1 on the low face"""
        rows, cols, vals = [], [], []
        e = np.arange(self.n_s)
        for axis in range(3):
            lo, hi = self.cell_faces(axis)
            rows += [e, e]
            cols += [lo, hi]
            vals += [-np.ones(self.n_s), np.ones(self.n_s)]
        B = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(self.n_s, self.n_u))
        return B.tocsr()

    def mass(self, coeff=None):
        """M(c) = sum_e c_e M_e; per cell and direction the 2 x 2 block h_a / A_a [[1/3, 1/6], [1/6, 1/3]]"""
        c = np.ones(self.n_s) if coeff is None else np.asarray(coeff, dtype=np.float64)
        rows, cols, vals = [], [], []
        for axis in range(3):
            lo, hi = self.cell_faces(axis)
            scale = self.h[axis] / (self.vol / self.h[axis])
            rows += [lo, hi, lo, hi]
            cols += [lo, hi, hi, lo]
            vals += [c * scale / 3.0, c * scale / 3.0, c * scale / 6.0, c * scale / 6.0]
        M = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(self.n_u, self.n_u))
        return M.tocsr()

    def prolongation(self, coarse: "HexLevel"):
        """P0 prolongator children <- parent for a grid with twice the cells per direction"""
        assert all(f == 2 * c for f, c in zip(self.n, coarse.n))
        i, j, k = self._cells()
        parent = coarse.cell_index(i // 2, j // 2, k // 2)
        return sp.csr_matrix((np.ones(self.n_s), (np.arange(self.n_s), parent)), shape=(self.n_s, coarse.n_s))


def matern_g(corlen, dim):
    """src/Utilities.hpp:188-200, as coded: sqrt((4 pi)^(d/2) Gamma(nu + d) kappa^(2 nu) / Gamma(nu)), nu = 2 - d/2"""
    nu = 2.0 - dim / 2.0
    return math.sqrt((4.0 * math.pi) ** (dim / 2.0) * math.gamma(nu + dim) * (1.0 / corlen) ** (2.0 * nu) / math.gamma(nu))


def hex_hierarchy(n_coarse, size, n_refine, origin=(0.0, 0.0, 0.0)):
    """levels [0] finest .. [n_refine] coarsest of a box cut n_coarse * 2^r times"""
    return [HexLevel([m * 2 ** r for m in n_coarse], size, origin) for r in range(n_refine, -1, -1)]


class RefSampler:
    """PDESampler::Eval on a HexLevel hierarchy, sparse direct solve (src/PDESampler.cpp:342-409)."""

    def __init__(self, levels, corlen):
        self.levels = levels
        self.alpha = 1.0 / (corlen * corlen)
        self.g = matern_g(corlen, 3)
        self._lu = {}

    def operators(self, l):
        """(M, B, w): boundary rows / columns of M replaced by the identity, boundary columns of B removed"""
        L = self.levels[l]
        ess = L.boundary_attribute() > 0
        keep = sp.diags((~ess).astype(np.float64))
        M = (keep @ L.mass() @ keep + sp.diags(ess.astype(np.float64))).tocsr()
        B = (L.divergence() @ keep).tocsr()
        return M, B, L.w_diag()

    def _solver(self, l):
        if l not in self._lu:
            M, B, w = self.operators(l)
            self._lu[l] = spla.splu(sp.bmat([[M, B.T], [B, -self.alpha * sp.diags(w)]], format="csc"))
        return self._lu[l]

    def eval(self, level, xi_level, xi):
        """Gaussian field on `level` from white noise drawn on xi_level <= level (finer)"""
        r = -self.g * np.sqrt(self.levels[xi_level].w_diag()) * xi
        for l in range(xi_level, level):
            r = self.levels[l].prolongation(self.levels[l + 1]).T @ r
        L = self.levels[level]
        return self._solver(level).solve(np.concatenate([np.zeros(L.n_u), r]))[L.n_u:]


class RefHybrid:
    """The sampler's hybridization solver restated from the closed forms (src/PDESampler.cpp:291,307-311,451-480: the
    "Hybridization" branch of Eval; ParELAG's HybridHdivL2 does the elimination).  Fluxes are broken across faces, one Lagrange
    multiplier per face re-imposes continuity (on a boundary face: u.n = 0, every boundary face being essential for the sampler,
    :210-214).  All cells of a HexLevel are equal, so ONE 7 x 7 inverse gives every element's blocks:

        [[X, y], [y^T, z]] = [[M_e, b_e^T], [b_e, -alpha |e|]]^-1,   H = sum_e C_e X C_e^T,  G = sum_e C_e y,
        H lambda = G f,    s = z f - G^T lambda.

    Multiplier sign convention of THIS module: C_e = +1 for the cell on the low side of a face (the face is that cell's high
    face), -1 for the cell on its high side; a boundary face has its one cell with the sign that rule gives it."""

    def __init__(self, levels, corlen):
        self.levels = levels
        self.alpha = 1.0 / (corlen * corlen)
        self.g = matern_g(corlen, 3)
        self._ops = {}

    def local_inverse(self, l):
        L = self.levels[l]
        A = np.zeros((7, 7))
        b = np.zeros(6)
        for axis in range(3):
            scale = L.h[axis] / (L.vol / L.h[axis])
            lo, hi = 2 * axis, 2 * axis + 1
            A[lo, lo] = A[hi, hi] = scale / 3.0
            A[lo, hi] = A[hi, lo] = scale / 6.0
            b[lo], b[hi] = -1.0, 1.0                       # global +axis orientation: outward on the high face
        A[:6, 6] = A[6, :6] = b
        A[6, 6] = -self.alpha * L.vol
        Ai = np.linalg.inv(A)
        return Ai[:6, :6], Ai[:6, 6], Ai[6, 6]

    def operators(self, l):
        """(H, G, z) of level l in this module's face / cell numbering"""
        if l in self._ops:
            return self._ops[l]
        L = self.levels[l]
        X, y, z = self.local_inverse(l)
        faces = np.empty((L.n_s, 6), dtype=np.int64)       # local order: x-low, x-high, y-low, y-high, z-low, z-high
        c = np.empty(6)
        for axis in range(3):
            lo, hi = L.cell_faces(axis)
            faces[:, 2 * axis], faces[:, 2 * axis + 1] = lo, hi
            c[2 * axis], c[2 * axis + 1] = -1.0, 1.0       # the cell sits on the HIGH side of its low face
        Xc = c[:, None] * X * c[None, :]
        rows = np.repeat(faces, 6, axis=1).ravel()
        cols = np.tile(faces, (1, 6)).ravel()
        H = sp.coo_matrix((np.tile(Xc.ravel(), L.n_s), (rows, cols)), shape=(L.n_u, L.n_u)).tocsr()
        G = sp.coo_matrix((np.tile(c * y, L.n_s), (faces.ravel(), np.repeat(np.arange(L.n_s), 6))), shape=(L.n_u, L.n_s)).tocsr()
        self._ops[l] = (H, G, np.full(L.n_s, z))
        return self._ops[l]

    def eval(self, level, xi_level, xi):
        """Gaussian field on `level` from white noise of xi_level <= level: the hybridized solve, sparse direct"""
        f = -self.g * np.sqrt(self.levels[xi_level].w_diag()) * xi
        for l in range(xi_level, level):
            f = self.levels[l].prolongation(self.levels[l + 1]).T @ f
        H, G, z = self.operators(level)
        lam = spla.spsolve(H.tocsc(), G @ f)
        return z * f - G.T @ lam


class RefDarcy:
    """DarcySolver::SolveFwd on one HexLevel (src/DarcySolver.cpp:416-437,472-520): u.n = 0 on the attributes flagged in
    `ess`, pressure coefficient p_inflow on `inflow`, Q = flux through `obs` (outward)."""

    def __init__(self, level, ess, obs, inflow, p_inflow=-1.0, k_divides=True):
        self.L = level
        attr = level.boundary_attribute()
        isb = attr > 0
        a = np.where(isb, attr - 1, 0)
        # outward = +axis on the high faces, -axis on the low ones
        outward = np.where(np.isin(attr, (3, 4, 6)), 1.0, -1.0)
        self.ess = isb & np.asarray(ess, dtype=bool)[a]
        self.rhs_u = np.where(isb & np.asarray(inflow, dtype=bool)[a], p_inflow * outward, 0.0)
        self.obs_u = np.where(isb & np.asarray(obs, dtype=bool)[a], outward, 0.0)
        self.k_divides = k_divides
        self.B = level.divergence()

    def solve_fwd(self, k):
        L = self.L
        k = np.asarray(k, dtype=np.float64)
        M = L.mass(1.0 / k if self.k_divides else k)
        A = sp.bmat([[M, self.B.T], [self.B, None]], format="csr")
        n = L.n_u + L.n_s
        ess = np.concatenate([self.ess, np.zeros(L.n_s, dtype=bool)])
        rhs = np.concatenate([self.rhs_u, np.zeros(L.n_s)])
        rhs[ess] = 0.0                                   # homogeneous essential data (ess_data == 0)
        keep = sp.diags((~ess).astype(np.float64))
        A = (keep @ A @ keep + sp.diags(ess.astype(np.float64))).tocsc()
        sol = spla.splu(A).solve(rhs)
        return float(self.obs_u @ sol[:L.n_u]), float(n), sol
