"""CPU prototype (round 5): how many MINRES iterations does the multiplier system need when the COARSE part of the aggregation
cycle is made stronger - W-cycle from level `wfrom` on, higher smoothing degree on the coarse levels, two smoothing passes -
while the finest level (where the bytes are) keeps its V(1,1) one-pass degree-2 smoothing?  With four lanes the coarse levels
are latency that hides behind the other lanes' bandwidth-bound kernels, so coarse work is nearly free there and every
iteration saved is 5 % of the run.  Development aid, nothing here is product code."""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amg_proto import cheb, level_tuple, my_minres  # noqa: E402
from parelagmc_amd.fe import build_hierarchy, build_hybrid_sampler_problem, mesh_from_json  # noqa: E402
from oracle.cport import HybridCPort  # noqa: E402  (greedy aggregation in C: fast enough for r = 5)

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
nref = int(sys.argv[1]) if len(sys.argv) > 1 else 4
h = build_hierarchy(mesh_from_json(os.path.join(ROOT, "tests", "golden", "meshes", "cube_tet.json")), nref)
hp = build_hybrid_sampler_problem(h, corlen=0.1, n_mc_levels=1)
L = hp.levels[0]
H, G = L.H.tocsr(), L.G


def pairwise_pass(K, theta=0.25):
    """one pass of magnitude-based pairwise matching with the allow-weak rule, vectorised poorly but in numpy loops only over
    rows (python): fine up to ~50 k rows; larger levels use the C greedy aggregation of the oracle instead"""
    import amg_proto
    Ka = K.copy()
    dg = Ka.diagonal()
    Ka.data = -np.abs(Ka.data)
    Ka.setdiag(dg)
    return amg_proto.pairwise(Ka, theta)


def aggregates(K, target=9.0):
    n = K.shape[0]
    if n > 60000:
        hc = HybridCPort.__new__(HybridCPort)
        import ctypes as C
        from oracle.cport import build
        hc.lib = C.CDLL(build())
        hc.lib.pmc_ref_aggregate.restype = C.c_int
        hc.theta = 0.08
        return hc._aggregate(K)
    agg, nc = pairwise_pass(K)
    cur, Kc = agg, K
    for _ in range(2):
        P = sp.csr_matrix((np.ones(len(cur)), (np.arange(len(cur)), cur)), shape=(len(cur), nc))
        Kc = (P.T @ Kc @ P).tocsr()
        nxt, nc2 = pairwise_pass(Kc)
        agg = nxt[agg]
        cur, nc = nxt, nc2
    return agg, nc


t0 = time.time()
lv = []
Kc = H
while True:
    if Kc.shape[0] <= 300 or len(lv) >= 12:
        lv.append(level_tuple(Kc, None))
        break
    agg, nc = aggregates(Kc)
    P = sp.csr_matrix((np.ones(len(agg)), (np.arange(len(agg)), agg)), shape=(len(agg), nc))
    lv.append(level_tuple(Kc, P))
    Kc = (P.T @ Kc @ P).tocsr()
print(f"r={nref}: levels {[t[0].shape[0] for t in lv]} ({time.time() - t0:.1f} s)", flush=True)


class Cycle:
    def __init__(self, lv, ratio0=16.0, deg0=2, ratio_c=16.0, deg_c=2, gamma=1, wfrom=1, nsm_c=1, cdeg=12, cratio=100.0):
        self.__dict__.update(locals())

    def __call__(self, r):
        return self.v(0, r)

    def smooth(self, l, r, x=None):
        S, dinv, lmax, P = self.lv[l]
        ratio, deg, n = (self.ratio0, self.deg0, 1) if l == 0 else (self.ratio_c, self.deg_c, self.nsm_c)
        for _ in range(n):
            x = cheb(S, dinv, lmax, ratio, deg, r, x)
        return x

    def v(self, l, r):
        S, dinv, lmax, P = self.lv[l]
        if l == len(self.lv) - 1:
            return cheb(S, dinv, lmax, self.cratio, self.cdeg, r)
        x = self.smooth(l, r)
        visits = self.gamma if l + 1 >= self.wfrom and l + 2 < len(self.lv) else 1
        for _ in range(visits):
            x = x + P @ self.v(l + 1, P.T @ (r - S @ x))
        return self.smooth(l, r, x)


rng = np.random.default_rng(0)
b = G @ (-hp.matern_g * np.sqrt(L.w_diag) * rng.standard_normal(L.n_s))
cases = [("product: V(1,1), degree 2 everywhere", {}),
         ("W-cycle from level 1", dict(gamma=2, wfrom=1)),
         ("W-cycle from level 2", dict(gamma=2, wfrom=2)),
         ("degree 4 on the coarse levels", dict(deg_c=4)),
         ("two smoothing passes on the coarse levels", dict(nsm_c=2)),
         ("W from level 1 + degree 4 coarse", dict(gamma=2, wfrom=1, deg_c=4)),
         ("degree 3 on the finest level too", dict(deg0=3, deg_c=3)),
         ("degree 1 (damped Jacobi) on the finest level only", dict(deg0=1)),
         ("degree 1 on the finest level, ratio 4", dict(deg0=1, ratio0=4.0)),
         ("degree 1 on the finest level, degree 4 coarse", dict(deg0=1, deg_c=4)),
         ("exact coarse solve from level 1 (bound)", None)]
for name, kw in cases:
    if kw is None:
        import scipy.sparse.linalg as spla
        S1 = lv[1][0].tocsc()
        lu = spla.splu(S1) if S1.shape[0] < 80000 else None
        if lu is None:
            continue
        cyc = Cycle(lv)
        cyc.v = (lambda self_v: None)
        S, dinv, lmax, P = lv[0]

        def prec(r, S=S, dinv=dinv, lmax=lmax, P=P):
            x = cheb(S, dinv, lmax, 16.0, 2, r)
            x = x + P @ lu.solve(P.T @ (r - S @ x))
            return cheb(S, dinv, lmax, 16.0, 2, r, x)
    else:
        prec = Cycle(lv, **kw)
    t0 = time.time()
    x, it = my_minres(H, prec, b, 1e-6, 300)
    print(f"{name:52s} MINRES iterations {it:3d}   ({time.time() - t0:.1f} s)", flush=True)
