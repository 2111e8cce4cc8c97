"""CPU prototype (scipy): MINRES iteration counts of the sampler's saddle-point system for variants of the S-block of the
block-diagonal preconditioner - the product's multiplicative V(1,1)-cycle with one-pass degree-2 Chebyshev smoothing against
ADDITIVE (BPX-type) multilevel variants that need ONE gather pass per level instead of four.  Development aid, round 4."""
import sys
import os
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

nref = int(sys.argv[1]) if len(sys.argv) > 1 else 4
p = bench.build_problem(nref)
L = p.levels
alpha, g = p.alpha, p.matern_g


def cheb2(lmax, ratio):
    lmin = lmax / ratio
    theta, delta = 0.5 * (lmax + lmin), 0.5 * (lmax - lmin)
    sigma = theta / delta
    rho_old = 1.0 / sigma
    rho1 = 1.0 / (2.0 * sigma - rho_old)
    return (1.0 + rho1 * rho_old) / theta + 2.0 * rho1 / delta, 2.0 * rho1 / (delta * theta)


class Level:
    def __init__(self, lv):
        dM = lv.M.diagonal()
        self.S = (alpha * sp.diags(lv.w_diag) + lv.B @ sp.diags(1.0 / dM) @ lv.B.T).tocsr()
        self.d = self.S.diagonal()
        self.lmax = (abs(self.S) @ np.ones(self.S.shape[0]) / self.d).max() * 1.0001
        self.c0, self.c1 = cheb2(self.lmax, 8.0)
        self.P = lv.P

    def p2(self, r):   # one-pass degree-2 polynomial from a zero guess: x = D^-1 (c0 r - c1 S D^-1 r)
        t = r / self.d
        return (self.c0 * r - self.c1 * (self.S @ t)) / self.d


lv = [Level(x) for x in L]
nl = len(lv)
coarse = spla.splu(lv[-1].S.tocsc())


def vcycle(l, r):
    if l == nl - 1:
        return coarse.solve(r)
    x = lv[l].p2(r)
    res = r - lv[l].S @ x
    x = x + lv[l].P @ vcycle(l + 1, lv[l].P.T @ res)
    return x + lv[l].p2(r - lv[l].S @ x)


def additive(l, r, scale):
    if l == nl - 1:
        return coarse.solve(r)
    return scale * lv[l].p2(r) + lv[l].P @ additive(l + 1, lv[l].P.T @ r, scale)


def additive_sym2(l, r):
    """two-level-wise symmetric multiplicative without pre-smoothing is not symmetric; this is additive with the smoother
    applied to the part of r the coarse space does not see: z = p2(r) + P C^-1 P^T (r - S p2(r))  [one more S pass]"""
    if l == nl - 1:
        return coarse.solve(r)
    x = lv[l].p2(r)
    return x + lv[l].P @ additive_sym2(l + 1, lv[l].P.T @ (r - lv[l].S @ x))


M, B = L[0].M.tocsr(), L[0].B.tocsr()
n_u, n_s = L[0].n_u, L[0].n_s
A = sp.bmat([[M, B.T], [B, -alpha * sp.diags(L[0].w_diag)]], format="csr")
l1 = abs(M) @ np.ones(n_u)
# Chebyshev interval of D_l1^-1 M
lam = spla.eigsh(sp.diags(1.0 / np.sqrt(l1)) @ M @ sp.diags(1.0 / np.sqrt(l1)), k=1, which="SA", return_eigenvectors=False)[0]
ratioM = 1.0 / lam
mc0, mc1 = cheb2(1.0, ratioM)


def mblock(r):
    t = r / l1
    return (mc0 * r - mc1 * (M @ t)) / l1


def minres(prec_s, b, rel=1e-6, maxit=300):
    """preconditioned MINRES, stopping on the preconditioned residual norm estimate as the product does"""
    def prec(v):
        return np.concatenate([mblock(v[:n_u]), prec_s(v[n_u:])])
    x = np.zeros_like(b)
    v0 = np.zeros_like(b)
    v1 = b.copy()
    z1 = prec(v1)
    beta = np.sqrt(v1 @ z1)
    eta = eta0 = beta
    g0 = g1 = 1.0
    s0 = s1 = 0.0
    w0 = np.zeros_like(b)
    w1 = np.zeros_like(b)
    beta_old = 1.0
    for it in range(1, maxit + 1):
        zq = z1 / beta
        q = A @ zq
        alpha_ = zq @ q
        vn = q - (alpha_ / beta) * v1 - (beta / beta_old) * v0
        zn = prec(vn)
        d2 = vn @ zn
        if d2 < 0:
            return -it
        beta_new = np.sqrt(d2)
        delta = g1 * alpha_ - g0 * s1 * beta
        rho1 = np.hypot(delta, beta_new)
        rho2 = s1 * alpha_ + g0 * g1 * beta
        rho3 = s0 * beta
        g0, g1 = g1, delta / rho1
        s0, s1 = s1, beta_new / rho1
        wn = (zq - rho3 * w0 - rho2 * w1) / rho1
        x = x + g1 * eta * wn
        w0, w1 = w1, wn
        eta = -s1 * eta
        v0, v1, z1 = v1, vn, zn
        beta_old, beta = beta, beta_new
        if abs(eta) <= rel * eta0:
            return it
    return maxit


rng = np.random.default_rng(0)
for name, fn in (("V(1,1) multiplicative (product)", lambda r: vcycle(0, r)),
                 ("additive, scale 1", lambda r: additive(0, r, 1.0)),
                 ("additive, scale 0.5", lambda r: additive(0, r, 0.5)),
                 ("smoother + coarse correction of its residual (2 passes)", lambda r: additive_sym2(0, r))):
    its = []
    for _ in range(3):
        b = np.concatenate([np.zeros(n_u), -g * np.sqrt(L[0].w_diag) * rng.standard_normal(n_s)])
        its.append(minres(fn, b))
    print(f"r={nref} n={n_u + n_s}  {name:60s} iterations {its}", flush=True)
