"""CPU prototype (round 4): the HYBRIDIZED form of the sampler's mixed system (the reference's alternative solver branch,
/root/reference/src/PDESampler.cpp:291,307-311,383-389: "hybridization + PCG-AMG") - element-local elimination of (u, s)
leaves an SPD system H lambda = g for one multiplier per face; s follows element by element.  Question: how many Krylov
iterations does H need with an aggregation multigrid of the kind the library already builds (sa_hierarchy), against the 33-42
of block-diagonally preconditioned MINRES on the saddle-point system?  Development aid, nothing here is product code."""
import os
import sys
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import amg_proto  # noqa: E402
from amg_proto import cheb, level_tuple, MG  # noqa: E402


def pairwise_weak(K, theta=0.25):
    """amg_proto.pairwise with the library's allow_weak rule (csrc/sparse.hip::pairwise_match): a row whose strong
    neighbours are all taken pairs with its strongest FREE neighbour instead of staying a singleton"""
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
        best, bval, weak, wval = -1, 0.0, -1, 0.0
        for c, v in zip(cols[off], vals[off]):
            if agg[c] >= 0:
                continue
            if v >= theta * smax and v > bval:
                best, bval = c, v
            if v > wval:
                weak, wval = c, v
        if best < 0:
            best = weak
        agg[i] = nc
        if best >= 0:
            agg[best] = nc
        nc += 1
    if os.environ.get("JOIN", "1") == "1":
        # rows left alone (every neighbour taken) join the aggregate of their strongest neighbour: without this the
        # coarsening stalls at the ~2-6 % of rows that end up isolated in every pass
        size = np.bincount(agg, minlength=nc)
        for i in range(n):
            if size[agg[i]] != 1:
                continue
            cols = ix[ip[i]:ip[i + 1]]
            vals = -dv[ip[i]:ip[i + 1]]
            off = (cols != i)
            if not off.any():
                continue
            cand = cols[off][np.argsort(-vals[off])]
            for c in cand:
                if size[agg[c]] >= 2 and size[agg[c]] < 3:
                    size[agg[i]] -= 1
                    agg[i] = agg[c]
                    size[agg[c]] += 1
                    break
        used = np.unique(agg)
        remap = -np.ones(nc, int)
        remap[used] = np.arange(len(used))
        agg = remap[agg]
        nc = len(used)
    return agg, nc


if os.environ.get("WEAK", "1") == "1":
    amg_proto.pairwise = pairwise_weak
aggregate = amg_proto.aggregate
from parelagmc_amd.fe import build_hierarchy, mesh_from_json  # noqa: E402

nref = int(sys.argv[1]) if len(sys.argv) > 1 else 3
h = build_hierarchy(mesh_from_json(os.path.join(bench.ROOT, "tests", "golden", "meshes", "cube_tet.json")), nref)
p = bench.build_problem(nref)
L = p.levels[0]
alpha, g = p.alpha, p.matern_g
space = h.spaces[0]
ne, nfe = space.faces.elem_face.shape
nf = space.n_u
em = space.emass
# element matrices in the global face orientation: Me[e, a, b]
Me = np.zeros((ne, nfe, nfe))
lf = {}
ef = space.faces.elem_face
pos = np.zeros((ne, nf if nf < 1 else 1), int)  # placeholder
loc = {}
order = np.argsort(em.elem, kind="stable")
# local index of a global face inside its element
inv = np.full((ne, nfe), -1)
for a in range(nfe):
    inv[:, a] = ef[:, a]
def local_of(e, f):
    return (ef[e] == f[:, None]).argmax(axis=1)
la = local_of(em.elem, em.rows)
lb = local_of(em.elem, em.cols)
Me[em.elem, la, lb] = em.vals
sign = space.faces.elem_sign.astype(float)          # B[e, f] = sign
vol = space.vol
# local saddle matrix [[Me, b^T], [b, -alpha w]] and its inverse, batched
A5 = np.zeros((ne, nfe + 1, nfe + 1))
A5[:, :nfe, :nfe] = Me
A5[:, :nfe, nfe] = sign
A5[:, nfe, :nfe] = sign
A5[:, nfe, nfe] = -alpha * vol
Ainv = np.linalg.inv(A5)
X = Ainv[:, :nfe, :nfe]
y = Ainv[:, :nfe, nfe]
z = Ainv[:, nfe, nfe]
# continuity constraint: broken flux dofs in the global orientation; C_e = +1 for the face's first element, -1 for the second,
# +1 alone on boundary faces (u.n = 0 there: all boundary faces are essential, src/PDESampler.cpp:210-214)
first = space.faces.face_elem[:, 0]
c = np.where(first[ef] == np.arange(ne)[:, None], 1.0, -1.0)
rows = np.repeat(ef, nfe, axis=1).ravel()
cols = np.tile(ef, (1, nfe)).ravel()
vals = (c[:, :, None] * X * c[:, None, :]).ravel()
H = sp.coo_matrix((vals, (rows, cols)), shape=(nf, nf)).tocsr()
G = sp.coo_matrix(((c * y).ravel(), (ef.ravel(), np.repeat(np.arange(ne), nfe))), shape=(nf, ne)).tocsr()
print(f"r={nref}: faces {nf}, elements {ne}, nnz(H) {H.nnz} ({H.nnz / nf:.1f} per row), saddle system {L.n_u + L.n_s} dofs nnz {L.nnz}")

rng = np.random.default_rng(0)
xi = rng.standard_normal(ne)
f = -g * np.sqrt(L.w_diag) * xi
# reference: saddle-point direct solve
DIRECT = nf < 100000          # sparse direct solves of 3D systems beyond that take too long for a prototype
if DIRECT:
    A = sp.bmat([[L.M, L.B.T], [L.B, -alpha * sp.diags(L.w_diag)]], format="csc")
    s_ref = spla.splu(A).solve(np.concatenate([np.zeros(L.n_u), f]))[L.n_u:]
    lam = spla.splu(H.tocsc()).solve(G @ f)
    s_h = z * f - G.T @ lam
    print("hybridized s vs saddle-point s:", np.linalg.norm(s_h - s_ref) / np.linalg.norm(s_ref))
else:
    s_ref = None

# aggregation hierarchy on H (plain / smoothed), V(1,1) with the product's one-pass degree-2 smoothing; PCG and MINRES counts
def hierarchy(K, passes, smoothed):
    lv = []
    Kc = K
    while True:
        if Kc.shape[0] <= 300 or len(lv) >= 12:
            lv.append(level_tuple(Kc, None))
            break
        Ka = Kc
        if os.environ.get("ABS", "1") == "1":     # strength by magnitude: H has positive off-diagonals on right / obtuse tetrahedra
            Ka = Kc.copy()
            dg = Ka.diagonal()
            Ka.data = -np.abs(Ka.data)
            Ka.setdiag(dg)
        agg, nc = aggregate(Ka, passes)
        P = sp.csr_matrix((np.ones(len(agg)), (np.arange(len(agg)), agg)), shape=(len(agg), nc))
        if smoothed:
            d = Kc.diagonal()
            lmax = (abs(Kc) @ np.ones(Kc.shape[0]) / d).max()
            P = (P - (4.0 / (3.0 * lmax)) * (sp.diags(1.0 / d) @ (Kc @ P))).tocsr()
        lv.append(level_tuple(Kc, P))
        Kc = ((P.T @ Kc @ P) * (1.0 / float(os.environ.get("OMEGA", "1")))).tocsr()   # over-correction of plain aggregation
    return lv

def pcg(Aop, prec, b, rel=1e-6, maxit=300):
    x = np.zeros_like(b)
    r = b.copy()
    zz = prec(r)
    pp = zz.copy()
    rz = r @ zz
    r0 = np.sqrt(rz)
    for it in range(1, maxit + 1):
        Ap = Aop @ pp
        a = rz / (pp @ Ap)
        x += a * pp
        r -= a * Ap
        zz = prec(r)
        rz_new = r @ zz
        if np.sqrt(rz_new) <= rel * r0:
            return x, it
        pp = zz + (rz_new / rz) * pp
        rz = rz_new
    return x, maxit

b = G @ f
for passes in (2, 3):
    for smoothed in ((False,) if not DIRECT else (False, True)):
        lv = hierarchy(H, passes, smoothed)
        mg = MG(lv, deg=int(os.environ.get("DEG", "2")), ratio=float(os.environ.get("RATIO", "8.0")))
        x, it = pcg(H, lambda r: mg.v(0, r), b)
        s_it = z * f - G.T @ x
        opc = sum(t[0].nnz for t in lv) / lv[0][0].nnz
        err = np.linalg.norm(s_it - s_ref) / np.linalg.norm(s_ref) if s_ref is not None else float("nan")
        print(f"aggregation passes {passes} smoothed {smoothed}: levels {[t[0].shape[0] for t in lv]} operator complexity {opc:.2f} "
              f"PCG iterations {it}  field error {err:.1e}", flush=True)
