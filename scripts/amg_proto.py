"""CPU prototype: geometric vs pairwise-aggregation V-cycle as the Schur block of the MINRES preconditioner on an
anisotropic (SPE10-shaped) box.  Development aid for the algebraic coarsening option."""
import sys
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from parelagmc_amd.fe import box_mesh, build_hierarchy, build_sampler_problem  # noqa: E402


def cheb(A, dinv, lmax, ratio, deg, r, x=None):
    lmin = lmax / ratio
    th, de = 0.5 * (lmax + lmin), 0.5 * (lmax - lmin)
    sg = th / de
    rho_old = 1.0 / sg
    if x is None:
        d = dinv * r / th
        x = d.copy()
        k0 = 1
    else:
        d = dinv * (r - A @ x) / th
        x = x + d
        k0 = 1
    for _ in range(k0, deg):
        rho = 1.0 / (2 * sg - rho_old)
        d = rho * rho_old * d + 2 * rho / de * dinv * (r - A @ x)
        x = x + d
        rho_old = rho
    return x


def pairwise(K, theta=0.25):
    K = K.tocsr()
    n = K.shape[0]
    agg = -np.ones(n, int)
    nc = 0
    ip, ix, dv = K.indptr, K.indices, K.data
    for i in range(n):
        if agg[i] >= 0:
            continue
        cols = ix[ip[i]:ip[i + 1]]
        vals = -dv[ip[i]:ip[i + 1]]
        off = cols != i
        smax = vals[off].max() if off.any() else 0.0
        best, bval = -1, 0.0
        for c, v in zip(cols[off], vals[off]):
            if agg[c] < 0 and v >= theta * smax and v > bval:
                best, bval = c, v
        agg[i] = nc
        if best >= 0:
            agg[best] = nc
        nc += 1
    return agg, nc


def aggregate(K, passes):
    agg, nc = pairwise(K)
    cur = agg
    Kc = K
    for _ in range(1, passes):
        P = sp.csr_matrix((np.ones(len(cur)), (np.arange(len(cur)), cur)), shape=(len(cur), nc))
        Kc = (P.T @ Kc @ P).tocsr()
        nxt, nc2 = pairwise(Kc)
        if nc2 == nc:
            break
        agg = nxt[agg]
        cur, nc = nxt, nc2
    return agg, nc


def my_minres(A, prec, b, rel, maxit):
    x = np.zeros_like(b)
    v0 = np.zeros_like(b); w0 = np.zeros_like(b); w1 = np.zeros_like(b)
    v1 = b.copy(); u1 = prec(v1)
    beta = np.sqrt(v1 @ u1); eta = beta; g0 = g1 = 1.0; s0 = s1 = 0.0
    goal = rel * eta
    for it in range(1, maxit + 1):
        v1 = v1 / beta; u1 = u1 / beta
        q = A @ u1
        alpha = u1 @ q
        v0 = q - alpha * v1 - beta * v0
        delta = g1 * alpha - g0 * s1 * beta
        rho3 = s0 * beta; rho2 = s1 * alpha + g0 * g1 * beta
        q2 = prec(v0)
        beta_new = np.sqrt(max(v0 @ q2, 0.0))
        rho1 = np.hypot(delta, beta_new)
        w0 = (u1 - rho3 * w0 - rho2 * w1) / rho1
        g0, g1 = g1, delta / rho1
        x = x + g1 * eta * w0
        s0, s1 = s1, beta_new / rho1
        eta = -s1 * eta
        u1 = q2; v0, v1 = v1, v0; w0, w1 = w1, w0; beta = beta_new
        if abs(eta) <= goal:
            return x, it
    return x, maxit


class MG:
    def __init__(self, levels, deg=2, ratio=8.0, cdeg=12, cratio=100.0):
        self.L, self.deg, self.ratio, self.cdeg, self.cratio = levels, deg, ratio, cdeg, cratio

    def v(self, l, r):
        S, dinv, lmax, P = self.L[l]
        if l == len(self.L) - 1:
            return cheb(S, dinv, lmax, self.cratio, self.cdeg, r)
        x = cheb(S, dinv, lmax, self.ratio, self.deg, r)
        rc = P.T @ (r - S @ x)
        x = x + P @ self.v(l + 1, rc)
        return cheb(S, dinv, lmax, self.ratio, self.deg, r, x)


def level_tuple(S, P):
    d = S.diagonal()
    lmax = (abs(S) @ np.ones(S.shape[0]) / d).max() * 1.0001
    return (S.tocsr(), 1.0 / d, lmax, P)


def main():
    import os
    nref = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    omega = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
    passes = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    if os.environ.get("ISO", "0") == "1":
        h = build_hierarchy(box_mesh([4, 4, 4], [2.0, 2.0, 2.0], "hex"), nref)
        sp_ = build_sampler_problem(h, corlen=0.1)
    elif os.environ.get("ISO", "0") == "2":
        from parelagmc_amd.fe import kuhn_cube_tet
        h = build_hierarchy(kuhn_cube_tet(), nref)
        sp_ = build_sampler_problem(h, corlen=0.1)
    else:
        h = build_hierarchy(box_mesh([7, 27, 10], [1200.0, 2200.0, 170.0], "hex"), nref)
        sp_ = build_sampler_problem(h, corlen=100.0)
    L = sp_.levels[0]
    a = sp_.alpha
    M, B = L.M, L.B
    A = sp.bmat([[M, B.T], [B, -a * sp.diags(L.w_diag)]], format="csr")
    n_u = L.n_u
    l1 = 1.0 / (abs(M) @ np.ones(n_u))
    dM = M.diagonal()
    K0 = (B @ sp.diags(1.0 / dM) @ B.T).tocsr()
    W0 = sp.diags(a * L.w_diag)
    # geometric hierarchy
    geo = []
    for i, lv in enumerate(sp_.levels):
        S = (sp.diags(a * lv.w_diag) + lv.B @ sp.diags(1.0 / lv.M.diagonal()) @ lv.B.T).tocsr()
        geo.append(level_tuple(S, lv.P))
    # algebraic hierarchy
    alg = []
    Kc, Wc = K0, W0
    while True:
        S = (Wc + Kc).tocsr()
        if S.shape[0] <= 200 or len(alg) >= 12:
            alg.append(level_tuple(S, None))
            break
        agg, nc = aggregate(Kc, passes)
        P = sp.csr_matrix((np.ones(len(agg)), (np.arange(len(agg)), agg)), shape=(len(agg), nc))
        if os.environ.get("SA", "0") == "1":
            # smoothed aggregation: P = (I - w D^-1 S_f) P_tent with the strength-filtered operator
            Sf = S.tocsr().copy()
            d = Sf.diagonal()
            # filter weak connections (lump into diagonal)
            Sc = Sf.tocoo()
            rowmax = np.zeros(S.shape[0]); np.maximum.at(rowmax, Sc.row[Sc.row != Sc.col], -Sc.data[Sc.row != Sc.col])
            weak = (Sc.row != Sc.col) & (-Sc.data < 0.25 * rowmax[Sc.row])
            lump = np.zeros(S.shape[0]); np.add.at(lump, Sc.row[weak], Sc.data[weak])
            data = Sc.data.copy(); data[weak] = 0.0
            Sf = sp.csr_matrix((data, (Sc.row, Sc.col)), shape=S.shape); Sf.eliminate_zeros()
            Sf = Sf + sp.diags(lump)
            df = Sf.diagonal()
            lam = (abs(Sf) @ np.ones(S.shape[0]) / df).max()
            P = (P - (4.0 / (3.0 * lam)) * (sp.diags(1.0 / df) @ (Sf @ P))).tocsr()
            alg.append(level_tuple(S, P))
            Kc = (P.T @ Kc @ P).tocsr()
            Wc = (P.T @ Wc @ P).tocsr()
            continue
        alg.append(level_tuple(S, P))
        Kc = (P.T @ Kc @ P).tocsr() / omega
        Wc = (P.T @ Wc @ P).tocsr()
    print("geo sizes", [t[0].shape[0] for t in geo], "alg sizes", [t[0].shape[0] for t in alg])
    rng = np.random.default_rng(0)
    rhs = np.concatenate([np.zeros(n_u), -sp_.matern_g * np.sqrt(L.w_diag) * rng.standard_normal(L.n_s)])
    lu = spla.splu((W0 + K0).tocsc())
    luM = spla.splu(M.tocsc())

    class Exact:
        def v(self, l, r):
            return lu.solve(r)
    import os
    sdeg = int(os.environ.get("SDEG", "2")); srat = float(os.environ.get("SRAT", "8"))
    for name, lv in (("geometric", geo), ("algebraic", alg)):
        mg = Exact() if lv is None else MG(lv, deg=sdeg, ratio=srat)
        exactM = name.endswith("exactM")

        def prec(r):
            z = np.empty_like(r)
            z[:n_u] = luM.solve(r[:n_u]) if exactM else cheb(M, l1, 1.0, 8.0, 2, r[:n_u])
            z[n_u:] = mg.v(0, r[n_u:])
            return z
        x, nit = my_minres(A, prec, rhs, 1e-6, 400)
        print(f"{name}: minres iterations {nit} resid {np.linalg.norm(A @ x - rhs) / np.linalg.norm(rhs):.2e}", flush=True)


if __name__ == "__main__":
    main()
