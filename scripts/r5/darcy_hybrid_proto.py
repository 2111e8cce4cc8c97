"""CPU prototype (round 5): is the hybridized Darcy system worth building on the device?

The Darcy saddle-point system [M(k) B^T; B 0] with M(k) = sum_e (1 / k_e) M_e reduces, after breaking the flux continuity with
one multiplier per interior / no-flux face, to the SPD system

    H(k) lambda = sum_e k_e C_e X_e C_e^T lambda = rhs,    X_e = M_e^-1 - M_e^-1 b (b^T M_e^-1 b)^-1 b^T M_e^-1,

LINEAR in the realization's element coefficients (the element-grouped layout the device already has for M(k)).  The question
is the iteration count of MINRES + one plain-aggregation V(1,1) cycle when the AGGREGATES are fixed (built once from H(1), the
only thing a batched device solver can afford) and the coarse operators are the per-realization Galerkin products, against
the 40-65 iterations of the block-diagonal preconditioner the product runs on the saddle-point system.

usage: darcy_hybrid_proto.py [nref (2 = 16^3, 3 = 32^3)] [nsamples]      Development aid, nothing here is product code."""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amg_proto import cheb, my_minres  # noqa: E402
from parelagmc_amd.fe import box_mesh, build_darcy_problem, build_hierarchy, build_sampler_problem  # noqa: E402
from oracle.sampler_oracle import SamplerOracle  # noqa: E402
from oracle.cport import build  # noqa: E402

nref = int(sys.argv[1]) if len(sys.argv) > 1 else 2
nsamp = int(sys.argv[2]) if len(sys.argv) > 2 else 3
h = build_hierarchy(box_mesh([4, 4, 4], [2, 2, 2], "hex"), nref)
klevel = 1 if nref >= 4 else 0          # 64^3: the coefficient is drawn on 32^3 and injected (a direct solve of the 1 M-DoF sampler system is out of reach here)
spb = build_sampler_problem(h, corlen=0.1, lognormal=True, n_mc_levels=klevel + 1)
dp = build_darcy_problem(h, [0, 1, 1, 1, 1, 0], [1, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 1], n_mc_levels=1)
space = h.spaces[0]
L = dp.levels[0]
ft = space.faces
ef = ft.elem_face
ne, nfe = ef.shape
nf = space.n_u
em = space.emass
la = (ef[em.elem] == em.rows[:, None]).argmax(axis=1)
lb = (ef[em.elem] == em.cols[:, None]).argmax(axis=1)
Me = np.zeros((ne, nfe, nfe))
Me[em.elem, la, lb] = em.vals
b = ft.elem_sign.astype(np.float64)
Mi = np.linalg.inv(Me)
Mib = np.einsum("eab,eb->ea", Mi, b)
s = np.einsum("ea,ea->e", b, Mib)
X = Mi - Mib[:, :, None] * Mib[:, None, :] / s[:, None, None]
first = ft.face_elem[:, 0]
c = np.where(first[ef] == np.arange(ne)[:, None], 1.0, -1.0)
CXC = c[:, :, None] * X * c[:, None, :]
rows = np.repeat(ef, nfe, axis=1).ravel()
cols = np.tile(ef, (1, nfe)).ravel()
# multipliers: interior faces and no-flux (essential) boundary faces; pressure-boundary faces carry lambda = data
isb = ft.face_bdr_attr > 0
active = ~isb | L.ess_mask.astype(bool)
new = -np.ones(nf, int)
new[active] = np.arange(active.sum())
keep = active[rows] & active[cols]
nl = int(active.sum())


def hybrid_matrix(k):
    v = (k[:, None, None] * CXC).ravel()
    Hm = sp.coo_matrix((v[keep], (new[rows[keep]], new[cols[keep]])), shape=(nl, nl)).tocsr()
    Hm.sum_duplicates()
    return Hm


import ctypes as C  # noqa: E402
lib = C.CDLL(build())
lib.pmc_ref_aggregate.restype = C.c_int


def aggregates(K, theta=0.08):
    K = K.tocsr()
    K.sort_indices()
    n = K.shape[0]
    agg = np.zeros(n, dtype=np.int32)
    ip = K.indptr.astype(np.int32)
    ix = K.indices.astype(np.int32)
    dv = K.data.astype(np.float64)
    nc = lib.pmc_ref_aggregate(C.c_int(n), ip.ctypes.data_as(C.c_void_p), ix.ctypes.data_as(C.c_void_p),
                               dv.ctypes.data_as(C.c_void_p), C.c_double(theta), agg.ctypes.data_as(C.c_void_p))
    return agg, nc


def hierarchy_P(H1):
    Ps = []
    Kc = H1
    while Kc.shape[0] > 300 and len(Ps) < 12:
        agg, nc = aggregates(Kc)
        P = sp.csr_matrix((np.ones(len(agg)), (np.arange(len(agg)), agg)), shape=(len(agg), nc))
        Ps.append(P)
        Kc = (P.T @ Kc @ P).tocsr()
    return Ps


def levels_for(Hk, Ps, l1=True, scale=1.0):
    """scale < 1: the coarse operators are scale x the Galerkin products (over-correction: a piecewise-constant prolongator
    makes the Galerkin operator too stiff by about the linear coarsening ratio - Braess 1995; the P0 Schur hierarchy of the
    product does the same with 1/2)"""
    lv = []
    Kc = Hk
    for P in Ps + [None]:
        if l1:
            dinv, lmax = 1.0 / (abs(Kc) @ np.ones(Kc.shape[0])), 1.0
        else:
            d = Kc.diagonal()
            dinv, lmax = 1.0 / d, (abs(Kc) @ np.ones(Kc.shape[0]) / d).max() * 1.0001
        lv.append((Kc.tocsr(), dinv, lmax, P))
        if P is not None:
            Kc = (scale * (P.T @ Kc @ P)).tocsr()
    return lv


def vcycle(lv, ratio=16.0, deg=2):
    def v(l, r):
        S, dinv, lmax, P = lv[l]
        if P is None:
            return cheb(S, dinv, lmax, 100.0, 12, r)
        x = cheb(S, dinv, lmax, ratio, deg, r)
        x = x + P @ v(l + 1, P.T @ (r - S @ x))
        return cheb(S, dinv, lmax, ratio, deg, r, x)
    return lambda r: v(0, r)


t0 = time.time()
H1 = hybrid_matrix(np.ones(ne))
Ps = hierarchy_P(H1)
print(f"hex {round(ne ** (1 / 3))}^3: {nl} multipliers of {nf} faces, nnz/row {H1.nnz / nl:.1f}, fixed hierarchy "
      f"{[nl] + [P.shape[1] for P in Ps]} ({time.time() - t0:.1f} s)", flush=True)
orc = SamplerOracle(spb)
rng = np.random.default_rng(5)
f_u = L.rhs[:nf]
for smp in range(nsamp):
    xi = rng.standard_normal(spb.levels[klevel].n_s)
    k, _ = orc.eval(klevel, klevel, xi)
    if klevel:
        k = h.P[0] @ k                      # piecewise-constant injection (P0 prolongator, unit entries)
    Hk = hybrid_matrix(k)
    # rhs: C A^-1 [f; 0], u-part of the local inverse = k X f_e (f lives on pressure-boundary faces: one element each)
    fe = f_u[ef] * np.where(isb[ef], 1.0, 0.0)
    rl = np.zeros(nf)
    np.add.at(rl, ef.ravel(), (c * k[:, None] * np.einsum("eab,eb->ea", X, fe)).ravel())
    rhs = rl[active]
    if not np.any(rhs):
        rhs = rng.standard_normal(nl)
    out = [f"sample {smp}: log10 contrast {np.log10(k.max() / k.min()):.1f}"]
    for name, lv in (("fixed aggregates (from H(1)), Galerkin per k, l1 scaling", levels_for(Hk, Ps, True)),
                     ("  ... coarse operators 0.7 x Galerkin", levels_for(Hk, Ps, True, 0.7)),
                     ("  ... coarse operators 0.55 x Galerkin", levels_for(Hk, Ps, True, 0.55)),
                     ("  ... coarse operators 0.45 x Galerkin", levels_for(Hk, Ps, True, 0.45)),
                     ("fixed aggregates, diagonal scaling + per-k lmax", levels_for(Hk, Ps, False)),
                     ("aggregates from H(k) itself (per-realization setup: bound)", None))[:4 if klevel else 6]:
        if lv is None:
            lv = levels_for(Hk, hierarchy_P(Hk), False)
        x, it = my_minres(Hk, vcycle(lv), rhs, 1e-6, 300)
        res = np.linalg.norm(rhs - Hk @ x) / np.linalg.norm(rhs)
        out.append(f"   {name:62s} MINRES iterations {it:3d}  (true residual {res:.1e})")
    print("\n".join(out), flush=True)
