"""CPU prototype (round 5): over-corrected coarse operators in the plain-aggregation V-cycle of the SAMPLER's multiplier system.
A piecewise-constant prolongator makes the Galerkin operator P^T H P too stiff by about the linear coarsening ratio; scaling it by
s < 1 (equivalently: the coarse correction by 1 / s) is the classical remedy (Braess 1995).  How many MINRES iterations does the
cube_tet multiplier system need with s x Galerkin on every coarse level?  usage: overcorrect_proto.py [nref] [scales ...]
Development aid, nothing here is product code."""
import ctypes as C
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amg_proto import cheb, my_minres, pairwise  # noqa: E402
from parelagmc_amd.fe import build_hierarchy, build_hybrid_sampler_problem, mesh_from_json  # noqa: E402
from oracle.cport import build  # noqa: E402

nref = int(sys.argv[1]) if len(sys.argv) > 1 else 4
scales = [float(x) for x in sys.argv[2:]] or [1.0, 0.8, 0.7, 0.6, 0.5]
mesh = os.environ.get("MESH", "cube_tet")
h = build_hierarchy(mesh_from_json(os.path.join(ROOT, "tests", "golden", "meshes", mesh + ".json")), nref)
hp = build_hybrid_sampler_problem(h, corlen=0.1, n_mc_levels=1)
L = hp.levels[0]
H, G = L.H.tocsr(), L.G
lib = C.CDLL(build())
lib.pmc_ref_aggregate.restype = C.c_int


def aggregates(K, theta=0.08):
    K = K.tocsr()
    K.sort_indices()
    n = K.shape[0]
    agg = np.zeros(n, dtype=np.int32)
    ip, ix, dv = K.indptr.astype(np.int32), K.indices.astype(np.int32), K.data.astype(np.float64)
    nc = lib.pmc_ref_aggregate(C.c_int(n), ip.ctypes.data_as(C.c_void_p), ix.ctypes.data_as(C.c_void_p),
                               dv.ctypes.data_as(C.c_void_p), C.c_double(theta), agg.ctypes.data_as(C.c_void_p))
    return agg, nc


def triple_pairwise(K):
    """the product's coarsening: three passes of magnitude-based pairwise matching (aggregates of ~8)"""
    def one(Kx):
        Ka = Kx.copy().tocsr()
        dg = Ka.diagonal()
        Ka.data = -np.abs(Ka.data)
        Ka.setdiag(dg)
        return pairwise(Ka, 0.25)
    agg, nc = one(K)
    cur, Kc = agg, K
    for _ in range(2):
        P = sp.csr_matrix((np.ones(len(cur)), (np.arange(len(cur)), cur)), shape=(len(cur), nc))
        Kc = (P.T @ Kc @ P).tocsr()
        nxt, nc2 = one(Kc)
        agg = nxt[agg]
        cur, nc = nxt, nc2
    return agg, nc


t0 = time.time()
Ps = []
Kc = H
coarsen = triple_pairwise if os.environ.get("PAIRWISE", "0") == "1" else aggregates
while Kc.shape[0] > 300 and len(Ps) < 12:
    agg, nc = coarsen(Kc)
    if nc > 0.8 * Kc.shape[0]:
        break
    P = sp.csr_matrix((np.ones(len(agg)), (np.arange(len(agg)), agg)), shape=(len(agg), nc))
    Ps.append(P)
    Kc = (P.T @ Kc @ P).tocsr()
print(f"{mesh} r={nref}: {H.shape[0]} multipliers, hierarchy {[H.shape[0]] + [P.shape[1] for P in Ps]} ({time.time() - t0:.1f} s)", flush=True)
rng = np.random.default_rng(0)
b = G @ (-hp.matern_g * np.sqrt(L.w_diag) * rng.standard_normal(L.n_s))
for s in scales:
    lv = []
    Kc = H
    for P in Ps + [None]:
        d = Kc.diagonal()
        lv.append((Kc.tocsr(), 1.0 / d, (abs(Kc) @ np.ones(Kc.shape[0]) / d).max() * 1.0001, P))
        if P is not None:
            first_only = os.environ.get("FIRST_ONLY", "0") == "1"     # scale the first coarse operator only
            Kc = ((s if (not first_only or len(lv) == 1) else 1.0) * (P.T @ Kc @ P)).tocsr()

    def v(l, r):
        S, dinv, lmax, P = lv[l]
        if P is None:
            return cheb(S, dinv, lmax, 100.0, 12, r)
        x = cheb(S, dinv, lmax, 16.0, 2, r)
        x = x + P @ v(l + 1, P.T @ (r - S @ x))
        return cheb(S, dinv, lmax, 16.0, 2, r, x)
    x, it = my_minres(H, lambda r: v(0, r), b, 1e-6, 300)
    print(f"coarse operators {s:4.2f} x Galerkin: MINRES iterations {it:3d}  (true residual "
          f"{np.linalg.norm(b - H @ x) / np.linalg.norm(b):.1e})", flush=True)
