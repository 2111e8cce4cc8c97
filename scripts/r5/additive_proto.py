"""CPU prototype (round 5): the finest level of the multiplier V-cycle made ADDITIVE - z = p2(H) r + P B_1 P^T r, no residual,
no post-smoothing: two passes over H per MINRES iteration (operator + smoother) instead of four.  How many iterations does
MINRES need then?  (multiplicative V(1,1) below the finest level in both cases)  usage: additive_proto.py [nref]
Development aid, nothing here is product code."""
import ctypes as C
import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amg_proto import cheb, my_minres  # noqa: E402
from parelagmc_amd.fe import build_hierarchy, build_hybrid_sampler_problem, mesh_from_json  # noqa: E402
from oracle.cport import build  # noqa: E402

nref = int(sys.argv[1]) if len(sys.argv) > 1 else 4
h = build_hierarchy(mesh_from_json(os.path.join(ROOT, "tests", "golden", "meshes", "cube_tet.json")), nref)
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


Ps, Kc = [], H
while Kc.shape[0] > 300 and len(Ps) < 12:
    agg, nc = aggregates(Kc)
    P = sp.csr_matrix((np.ones(len(agg)), (np.arange(len(agg)), agg)), shape=(len(agg), nc))
    Ps.append(P)
    Kc = (P.T @ Kc @ P).tocsr()
lv, Kc = [], H
for P in Ps + [None]:
    d = Kc.diagonal()
    lv.append((Kc.tocsr(), 1.0 / d, (abs(Kc) @ np.ones(Kc.shape[0]) / d).max() * 1.0001, P))
    if P is not None:
        Kc = (P.T @ Kc @ P).tocsr()
print(f"cube_tet r={nref}: hierarchy {[t[0].shape[0] for t in lv]}", flush=True)


def v(l, r):
    S, dinv, lmax, P = lv[l]
    if P is None:
        return cheb(S, dinv, lmax, 100.0, 12, r)
    x = cheb(S, dinv, lmax, 16.0, 2, r)
    x = x + P @ v(l + 1, P.T @ (r - S @ x))
    return cheb(S, dinv, lmax, 16.0, 2, r, x)


def additive(r, wc=1.0, deg=2):
    S, dinv, lmax, P = lv[0]
    return cheb(S, dinv, lmax, 16.0, deg, r) + wc * (P @ v(1, P.T @ r))


rng = np.random.default_rng(0)
b = G @ (-hp.matern_g * np.sqrt(L.w_diag) * rng.standard_normal(L.n_s))
for name, prec in (("multiplicative V(1,1) (product)", lambda r: v(0, r)),
                   ("additive finest level", lambda r: additive(r)),
                   ("additive, coarse part x 0.7", lambda r: additive(r, 0.7)),
                   ("additive, coarse part x 1.5", lambda r: additive(r, 1.5)),
                   ("additive, degree-1 smoother", lambda r: additive(r, 1.0, 1))):
    x, it = my_minres(H, prec, b, 1e-6, 300)
    print(f"{name:36s} MINRES iterations {it:3d}  (true residual {np.linalg.norm(b - H @ x) / np.linalg.norm(b):.1e})", flush=True)
