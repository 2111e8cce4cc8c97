"""Conforming meshes with nested uniform refinement (setup side, host only).

The reference gets its meshes from MFEM (``mfem::Mesh(nx,ny,nz,HEXAHEDRON,...)``,
``mfem::Mesh(imesh,1,1)`` and ``UniformRefinement()``; /root/reference/examples/
DarcyTest.cpp:139-172, examples/example_helpers/Build3DMesh.hpp:24-28).  MFEM is not
available here, so this module provides the small subset the hot path's *inputs*
need: box meshes with MFEM's boundary-attribute convention, a reader for the
"MFEM mesh v1.0" / "MFEM INLINE mesh v1.0" text formats, and nested uniform
refinement that remembers each child's parent (the P0 prolongator of the level
hierarchy, /root/reference/src/PDESampler.cpp:189-193).

Element types: 'quad' and 'hex' (axis-aligned boxes only), 'tri' and 'tet'.
Everything is vectorised numpy; no per-element Python loops.
"""
from __future__ import annotations

import dataclasses
import io
import re

import numpy as np

# local faces, MFEM vertex numbering (mesh/geom.hpp in MFEM; restated from its
# public documentation, not copied).  Face i of a simplex is opposite vertex i.
_LOCAL_FACES = {
    "tri": np.array([[1, 2], [2, 0], [0, 1]]),
    "tet": np.array([[1, 2, 3], [0, 3, 2], [0, 1, 3], [0, 2, 1]]),
    "quad": np.array([[0, 1], [1, 2], [2, 3], [3, 0]]),
    # z-, y-, x+, y+, x-, z+
    "hex": np.array([[3, 2, 1, 0], [0, 1, 5, 4], [1, 2, 6, 5],
                     [2, 3, 7, 6], [3, 0, 4, 7], [4, 5, 6, 7]]),
}
_DIM = {"tri": 2, "quad": 2, "tet": 3, "hex": 3}
_GEOM_TO_TYPE = {2: "tri", 3: "quad", 4: "tet", 5: "hex"}
_FACE_NV = {"tri": 2, "quad": 2, "tet": 3, "hex": 4}


@dataclasses.dataclass
class Mesh:
    etype: str
    verts: np.ndarray      # (nv, dim) float64
    elems: np.ndarray      # (ne, nv_per_elem) int64
    elem_attr: np.ndarray  # (ne,) int32
    bdr: np.ndarray        # (nb, nv_per_face) int64
    bdr_attr: np.ndarray   # (nb,) int32

    @property
    def dim(self) -> int:
        return _DIM[self.etype]

    @property
    def ne(self) -> int:
        return self.elems.shape[0]

    def local_faces(self) -> np.ndarray:
        return _LOCAL_FACES[self.etype]


# --------------------------------------------------------------------------- box meshes
def box_mesh(n, sizes, etype="hex", origin=None) -> Mesh:
    """Cartesian box mesh with MFEM's INLINE conventions.

    Boundary attributes follow mfem::Mesh::Make3D / Make2D: hex 1=bottom(z=0),
    2=front(y=0), 3=right(x=sx), 4=back(y=sy), 5=left(x=0), 6=top(z=sz);
    quad 1=bottom(y=0), 2=right(x=sx), 3=top(y=sy), 4=left(x=0).  Elements are
    numbered x-fastest.
    """
    n = np.asarray(n, dtype=np.int64)
    sizes = np.asarray(sizes, dtype=np.float64)
    dim = len(n)
    org = np.zeros(dim) if origin is None else np.asarray(origin, dtype=np.float64)
    if etype == "tri":
        # mfem::Mesh::Make2D with triangles: every cell of the quad grid split along the diagonal (i, j) - (i + 1, j + 1)
        # [the order of the two triangles inside a cell is from memory of MFEM, not verified; nothing here depends on it]
        q = box_mesh(n, sizes, "quad", origin)
        e = q.elems
        tri = np.stack([np.stack([e[:, 0], e[:, 1], e[:, 2]], axis=1), np.stack([e[:, 0], e[:, 2], e[:, 3]], axis=1)], axis=1)
        m = Mesh("tri", q.verts, tri.reshape(-1, 3), np.ones(2 * len(e), np.int32), q.bdr, q.bdr_attr)
        _orient_simplices(m)
        return m
    if etype == "quad":
        nx, ny = n
        xs = org[0] + np.linspace(0.0, sizes[0], nx + 1)
        ys = org[1] + np.linspace(0.0, sizes[1], ny + 1)
        Y, X = np.meshgrid(ys, xs, indexing="ij")
        verts = np.stack([X.ravel(), Y.ravel()], axis=1)
        vid = lambda i, j: j * (nx + 1) + i
        J, I = np.meshgrid(np.arange(ny), np.arange(nx), indexing="ij")
        I, J = I.ravel(), J.ravel()
        elems = np.stack([vid(I, J), vid(I + 1, J), vid(I + 1, J + 1), vid(I, J + 1)], axis=1)
        i = np.arange(nx)
        j = np.arange(ny)
        b1 = np.stack([vid(i, 0), vid(i + 1, 0)], axis=1)
        b2 = np.stack([vid(nx, j), vid(nx, j + 1)], axis=1)
        b3 = np.stack([vid(i + 1, ny), vid(i, ny)], axis=1)
        b4 = np.stack([vid(0, j + 1), vid(0, j)], axis=1)
        bdr = np.concatenate([b1, b2, b3, b4])
        battr = np.concatenate([np.full(len(b), a) for b, a in ((b1, 1), (b2, 2), (b3, 3), (b4, 4))])
    elif etype == "hex":
        nx, ny, nz = n
        xs = org[0] + np.linspace(0.0, sizes[0], nx + 1)
        ys = org[1] + np.linspace(0.0, sizes[1], ny + 1)
        zs = org[2] + np.linspace(0.0, sizes[2], nz + 1)
        Z, Y, X = np.meshgrid(zs, ys, xs, indexing="ij")
        verts = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
        vid = lambda i, j, k: (k * (ny + 1) + j) * (nx + 1) + i
        K, J, I = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
        I, J, K = I.ravel(), J.ravel(), K.ravel()
        elems = np.stack([vid(I, J, K), vid(I + 1, J, K), vid(I + 1, J + 1, K), vid(I, J + 1, K),
                          vid(I, J, K + 1), vid(I + 1, J, K + 1), vid(I + 1, J + 1, K + 1),
                          vid(I, J + 1, K + 1)], axis=1)

        def grid2(na, nb):
            B, A = np.meshgrid(np.arange(nb), np.arange(na), indexing="ij")
            return A.ravel(), B.ravel()
        i, j = grid2(nx, ny)
        b1 = np.stack([vid(i, j, 0), vid(i, j + 1, 0), vid(i + 1, j + 1, 0), vid(i + 1, j, 0)], axis=1)
        b6 = np.stack([vid(i, j, nz), vid(i + 1, j, nz), vid(i + 1, j + 1, nz), vid(i, j + 1, nz)], axis=1)
        i, k = grid2(nx, nz)
        b2 = np.stack([vid(i, 0, k), vid(i + 1, 0, k), vid(i + 1, 0, k + 1), vid(i, 0, k + 1)], axis=1)
        b4 = np.stack([vid(i + 1, ny, k), vid(i, ny, k), vid(i, ny, k + 1), vid(i + 1, ny, k + 1)], axis=1)
        j, k = grid2(ny, nz)
        b3 = np.stack([vid(nx, j, k), vid(nx, j + 1, k), vid(nx, j + 1, k + 1), vid(nx, j, k + 1)], axis=1)
        b5 = np.stack([vid(0, j + 1, k), vid(0, j, k), vid(0, j, k + 1), vid(0, j + 1, k + 1)], axis=1)
        bl = ((b1, 1), (b2, 2), (b3, 3), (b4, 4), (b5, 5), (b6, 6))
        bdr = np.concatenate([b for b, _ in bl])
        battr = np.concatenate([np.full(len(b), a) for b, a in bl])
    else:
        raise ValueError(f"box_mesh: unsupported element type {etype!r}")
    return Mesh(etype, verts, elems.astype(np.int64), np.ones(len(elems), np.int32),
                bdr.astype(np.int64), battr.astype(np.int32))


def kuhn_cube_tet(size=1.0) -> Mesh:
    """[0,size]^3 split into 6 tetrahedra around the main diagonal (Kuhn), 12 boundary
    triangles with attribute 1 (same shape as the reference's meshes/cube_tet.mesh:
    6 tets, 8 vertices, 12 boundary triangles)."""
    import itertools
    verts = np.array([[i, j, k] for k in (0, 1) for j in (0, 1) for i in (0, 1)], dtype=np.float64) * size
    vid = lambda p: int(p[0] + 2 * p[1] + 4 * p[2])
    elems = []
    for perm in itertools.permutations(range(3)):
        p = np.zeros(3, dtype=int)
        tet = [vid(p)]
        for ax in perm:
            p = p.copy()
            p[ax] = 1
            tet.append(vid(p))
        elems.append(tet)
    elems = np.array(elems, dtype=np.int64)
    m = Mesh("tet", verts, elems, np.ones(6, np.int32), np.zeros((0, 3), np.int64), np.zeros(0, np.int32))
    _orient_simplices(m)
    ft = build_faces(m)
    bf = np.nonzero(ft.face_elem[:, 1] < 0)[0]
    m.bdr = ft.face_verts[bf]
    m.bdr_attr = np.ones(len(bf), np.int32)
    return m


# --------------------------------------------------------------------------- MFEM text formats
def read_mfem_mesh(text_or_path) -> Mesh:
    """Parse "MFEM mesh v1.0" (conforming, straight-sided) or "MFEM INLINE mesh v1.0"."""
    if "\n" not in str(text_or_path):
        with open(text_or_path, "r") as f:
            text = f.read()
    else:
        text = str(text_or_path)
    lines = [ln.split("#", 1)[0].strip() for ln in text.splitlines()]
    lines = [ln for ln in lines if ln]
    header = lines[0]
    if header.startswith("MFEM INLINE mesh"):
        kv = {}
        for ln in lines[1:]:
            m = re.match(r"(\w+)\s*=\s*(\S+)", ln)
            if m:
                kv[m.group(1)] = m.group(2)
        et = kv["type"]
        if et == "quad":
            return box_mesh([int(kv["nx"]), int(kv["ny"])], [float(kv["sx"]), float(kv["sy"])], "quad")
        if et == "hex":
            return box_mesh([int(kv["nx"]), int(kv["ny"]), int(kv["nz"])],
                            [float(kv["sx"]), float(kv["sy"]), float(kv["sz"])], "hex")
        if et == "tri":
            return box_mesh([int(kv["nx"]), int(kv["ny"])], [float(kv["sx"]), float(kv["sy"])], "tri")
        raise ValueError(f"INLINE mesh type {et!r} not supported")
    if not header.startswith("MFEM mesh v1.0"):
        raise ValueError(f"unsupported mesh header {header!r}")
    it = iter(lines[1:])
    dim = None
    elems = attrs = bdr = battr = verts = None
    etype = None
    for ln in it:
        if ln == "dimension":
            dim = int(next(it))
        elif ln == "elements":
            ne = int(next(it))
            rows = [list(map(int, next(it).split())) for _ in range(ne)]
            geoms = {r[1] for r in rows}
            if len(geoms) != 1:
                raise ValueError("mixed meshes are not supported")
            etype = _GEOM_TO_TYPE[geoms.pop()]
            attrs = np.array([r[0] for r in rows], np.int32)
            elems = np.array([r[2:] for r in rows], np.int64)
        elif ln == "boundary":
            nb = int(next(it))
            rows = [list(map(int, next(it).split())) for _ in range(nb)]
            battr = np.array([r[0] for r in rows], np.int32)
            bdr = np.array([r[2:] for r in rows], np.int64).reshape(nb, -1)
        elif ln == "vertices":
            nv = int(next(it))
            vdim = int(next(it))
            verts = np.array([list(map(float, next(it).split())) for _ in range(nv)], np.float64)
            verts = verts.reshape(nv, vdim)
    if dim is None or elems is None or verts is None:
        raise ValueError("incomplete mesh file")
    m = Mesh(etype, verts, elems, attrs, bdr, battr)
    if etype in ("tri", "tet"):
        _orient_simplices(m)
    return m


def _simplex_volume(verts, elems):
    d = verts.shape[1]
    e = verts[elems[:, 1:]] - verts[elems[:, :1]]
    return np.linalg.det(e) / (2.0 if d == 2 else 6.0)


def _orient_simplices(m: Mesh) -> None:
    """Make every simplex positively oriented (swap two vertices where needed)."""
    vol = _simplex_volume(m.verts, m.elems)
    neg = vol < 0
    if neg.any():
        a = m.elems[neg, 0].copy()
        m.elems[neg, 0] = m.elems[neg, 1]
        m.elems[neg, 1] = a


# --------------------------------------------------------------------------- face table
@dataclasses.dataclass
class FaceTable:
    face_verts: np.ndarray   # (nf, nvf) sorted vertex ids (key)
    elem_face: np.ndarray    # (ne, nfe) global face index of each local face
    elem_sign: np.ndarray    # (ne, nfe) +1 if the element's outward normal is the face's global normal
    face_elem: np.ndarray    # (nf, 2) adjacent elements (second = -1 on the boundary)
    face_bdr_attr: np.ndarray  # (nf,) boundary attribute, 0 for interior faces


def _row_keys(rows: np.ndarray, nv: int) -> np.ndarray:
    """Injective int64 key of sorted small-width rows."""
    rows = np.sort(rows, axis=1)
    if rows.shape[1] == 4:
        # quadrilateral faces of a conforming mesh share at most an edge, so the three smallest vertex
        # ids identify a face uniquely (keeps the packed key inside 62 bits for large meshes)
        rows = rows[:, :3]
    base = np.int64(nv + 1)
    w = rows.shape[1]
    if float(nv + 1) ** w >= 2.0 ** 62:
        raise ValueError("mesh too large for packed face keys")
    key = np.zeros(rows.shape[0], np.int64)
    for c in range(w):
        key = key * base + rows[:, c]
    return key


def build_faces(m: Mesh) -> FaceTable:
    lf = m.local_faces()
    nfe, nvf = lf.shape
    fv = m.elems[:, lf]                       # (ne, nfe, nvf)
    flat = fv.reshape(-1, nvf)
    nv = m.verts.shape[0]
    key = _row_keys(flat, nv)
    ukey, first, inv = np.unique(key, return_index=True, return_inverse=True)
    nf = len(ukey)
    # number the faces by their owning (first adjacent) element, so that the u-dofs an element touches
    # are close in memory when the elements are (children of a parent are contiguous): this is what makes
    # the gathers of the SpMV kernels hit in L2 instead of scattering over the whole vector
    owner_u = first // nfe
    order = np.argsort(owner_u * nfe + (first % nfe), kind="stable")
    rank = np.empty(nf, np.int64)
    rank[order] = np.arange(nf)
    inv = rank[inv]
    first = first[order]
    ukey_sorted_pos = order          # new face id j corresponds to ukey[order[j]]
    elem_face = inv.reshape(m.ne, nfe)
    face_verts = np.sort(flat[first], axis=1)
    # adjacency: the element that owns the first occurrence defines the global normal
    owner = first // nfe
    eidx = np.repeat(np.arange(m.ne), nfe)
    face_elem = np.full((nf, 2), -1, np.int64)
    face_elem[:, 0] = owner
    other = eidx != owner[inv]
    face_elem[inv[other], 1] = eidx[other]
    elem_sign = np.where(other, -1, 1).reshape(m.ne, nfe).astype(np.int8)
    # boundary attributes
    fattr = np.zeros(nf, np.int32)
    if m.bdr is not None and len(m.bdr):
        bkey = _row_keys(m.bdr, nv)
        pos = np.searchsorted(ukey, bkey)
        ok = (pos < nf) & (ukey[np.minimum(pos, nf - 1)] == bkey)
        if not ok.all():
            raise ValueError("boundary element does not match any mesh face")
        fattr[rank[pos]] = m.bdr_attr
    return FaceTable(face_verts, elem_face, elem_sign, face_elem, fattr)


# --------------------------------------------------------------------------- uniform refinement
def _edge_midpoints(verts, edges):
    """edges: (m,2) vertex pairs (any order, repeats allowed). Returns (new_verts, idx[m])
    where idx indexes into concat(verts, new_verts)."""
    nv = verts.shape[0]
    key = _row_keys(edges, nv)
    ukey, first, inv = np.unique(key, return_index=True, return_inverse=True)
    mid = 0.5 * (verts[edges[first, 0]] + verts[edges[first, 1]])
    return mid, nv + inv, ukey


def refine_uniform(m: Mesh):
    """One level of nested uniform refinement.

    Returns (fine_mesh, parent) with parent[child] = coarse element index.  Children of
    coarse element p are stored contiguously at 2^dim*p ... 2^dim*p + 2^dim - 1 (a layout
    choice of this build; MFEM appends children differently, which only permutes
    element numbers)."""
    v = m.verts
    nv = v.shape[0]
    E = m.elems
    et = m.etype
    if et == "tri":
        pairs = np.array([[0, 1], [1, 2], [2, 0]])
        edges = E[:, pairs].reshape(-1, 2)
        mid, midx, ukey = _edge_midpoints(v, edges)
        midx = midx.reshape(-1, 3)
        a, b, c = E[:, 0], E[:, 1], E[:, 2]
        ab, bc, ca = midx[:, 0], midx[:, 1], midx[:, 2]
        ch = np.stack([np.stack([a, ab, ca], 1), np.stack([ab, b, bc], 1),
                       np.stack([ca, bc, c], 1), np.stack([ab, bc, ca], 1)], 1)
        verts = np.concatenate([v, mid])
        bmid = _lookup_mid(m.bdr, ukey, nv)
        nb = np.stack([np.stack([m.bdr[:, 0], bmid], 1), np.stack([bmid, m.bdr[:, 1]], 1)], 1).reshape(-1, 2)
        nbattr = np.repeat(m.bdr_attr, 2)
    elif et == "quad":
        pairs = np.array([[0, 1], [1, 2], [2, 3], [3, 0]])
        edges = E[:, pairs].reshape(-1, 2)
        mid, midx, ukey = _edge_midpoints(v, edges)
        midx = midx.reshape(-1, 4)
        cen = v[E].mean(axis=1)
        cidx = nv + len(mid) + np.arange(m.ne)
        a, b, c, d = E.T
        ab, bc, cd, da = midx.T
        ch = np.stack([np.stack([a, ab, cidx, da], 1), np.stack([ab, b, bc, cidx], 1),
                       np.stack([da, cidx, cd, d], 1), np.stack([cidx, bc, c, cd], 1)], 1)
        verts = np.concatenate([v, mid, cen])
        bmid = _lookup_mid(m.bdr, ukey, nv)
        nb = np.stack([np.stack([m.bdr[:, 0], bmid], 1), np.stack([bmid, m.bdr[:, 1]], 1)], 1).reshape(-1, 2)
        nbattr = np.repeat(m.bdr_attr, 2)
    elif et == "tet":
        pairs = np.array([[0, 1], [0, 2], [0, 3], [1, 2], [1, 3], [2, 3]])
        edges = E[:, pairs].reshape(-1, 2)
        mid, midx, ukey = _edge_midpoints(v, edges)
        midx = midx.reshape(-1, 6)
        verts = np.concatenate([v, mid])
        v0, v1, v2, v3 = E.T
        m01, m02, m03, m12, m13, m23 = midx.T
        # inner octahedron split along its shortest diagonal (keeps shape regularity)
        d0 = np.linalg.norm(verts[m01] - verts[m23], axis=1)
        d1 = np.linalg.norm(verts[m02] - verts[m13], axis=1)
        d2 = np.linalg.norm(verts[m03] - verts[m12], axis=1)
        choice = np.argmin(np.stack([d0, d1 * (1 + 1e-12), d2 * (1 + 2e-12)], 1), axis=1)
        corner = [np.stack([v0, m01, m02, m03], 1), np.stack([m01, v1, m12, m13], 1),
                  np.stack([m02, m12, v2, m23], 1), np.stack([m03, m13, m23, v3], 1)]

        def octa(p, q, ring):
            # diagonal p-q, ring = 4 vertices in cyclic order around it
            r0, r1, r2, r3 = ring
            return [np.stack([p, q, r0, r1], 1), np.stack([p, q, r1, r2], 1),
                    np.stack([p, q, r2, r3], 1), np.stack([p, q, r3, r0], 1)]
        o0 = octa(m01, m23, (m02, m03, m13, m12))
        o1 = octa(m02, m13, (m01, m03, m23, m12))
        o2 = octa(m03, m12, (m01, m02, m23, m13))
        inner = []
        for k in range(4):
            sel = np.where((choice == 0)[:, None], o0[k], np.where((choice == 1)[:, None], o1[k], o2[k]))
            inner.append(sel)
        ch = np.stack(corner + inner, 1)           # (ne, 8, 4)
        # boundary triangles -> 4
        bp = np.array([[0, 1], [1, 2], [2, 0]])
        bm = _lookup_mid(m.bdr[:, bp].reshape(-1, 2), ukey, nv).reshape(-1, 3)
        a, b, c = m.bdr.T
        ab, bc, ca = bm.T
        nb = np.stack([np.stack([a, ab, ca], 1), np.stack([ab, b, bc], 1),
                       np.stack([ca, bc, c], 1), np.stack([ab, bc, ca], 1)], 1).reshape(-1, 3)
        nbattr = np.repeat(m.bdr_attr, 4)
    elif et == "hex":
        epairs = np.array([[0, 1], [1, 2], [2, 3], [3, 0], [4, 5], [5, 6], [6, 7], [7, 4],
                           [0, 4], [1, 5], [2, 6], [3, 7]])
        edges = E[:, epairs].reshape(-1, 2)
        mid, midx, ukey = _edge_midpoints(v, edges)
        midx = midx.reshape(-1, 12)
        lf = _LOCAL_FACES["hex"]
        fv = E[:, lf].reshape(-1, 4)
        fkey = _row_keys(fv, nv)
        ufkey, ffirst, finv = np.unique(fkey, return_index=True, return_inverse=True)
        fcen = v[fv[ffirst]].mean(axis=1)
        fidx = (nv + len(mid) + finv).reshape(-1, 6)
        cen = v[E].mean(axis=1)
        cidx = nv + len(mid) + len(fcen) + np.arange(m.ne)
        verts = np.concatenate([v, mid, fcen, cen])
        V = E.T
        e = midx.T
        f = fidx.T   # z-, y-, x+, y+, x-, z+
        c = cidx
        # 3x3x3 lattice of vertex ids for every hex: L[i][j][k], i along x, j along y, k along z
        L = [[[None] * 3 for _ in range(3)] for _ in range(3)]
        L[0][0][0], L[2][0][0], L[2][2][0], L[0][2][0] = V[0], V[1], V[2], V[3]
        L[0][0][2], L[2][0][2], L[2][2][2], L[0][2][2] = V[4], V[5], V[6], V[7]
        L[1][0][0], L[2][1][0], L[1][2][0], L[0][1][0] = e[0], e[1], e[2], e[3]
        L[1][0][2], L[2][1][2], L[1][2][2], L[0][1][2] = e[4], e[5], e[6], e[7]
        L[0][0][1], L[2][0][1], L[2][2][1], L[0][2][1] = e[8], e[9], e[10], e[11]
        L[1][1][0], L[1][0][1], L[2][1][1], L[1][2][1], L[0][1][1], L[1][1][2] = f[0], f[1], f[2], f[3], f[4], f[5]
        L[1][1][1] = c
        kids = []
        for k in (0, 1):
            for j in (0, 1):
                for i in (0, 1):
                    kids.append(np.stack([L[i][j][k], L[i + 1][j][k], L[i + 1][j + 1][k], L[i][j + 1][k],
                                          L[i][j][k + 1], L[i + 1][j][k + 1], L[i + 1][j + 1][k + 1],
                                          L[i][j + 1][k + 1]], 1))
        ch = np.stack(kids, 1)
        # boundary quads -> 4
        bq = m.bdr
        bp = np.array([[0, 1], [1, 2], [2, 3], [3, 0]])
        bm = _lookup_mid(bq[:, bp].reshape(-1, 2), ukey, nv).reshape(-1, 4)
        bfk = _row_keys(bq, nv)
        pos = np.searchsorted(ufkey, bfk)
        bc_ = nv + len(mid) + pos
        a, b, cc, d = bq.T
        ab, bc2, cd, da = bm.T
        nb = np.stack([np.stack([a, ab, bc_, da], 1), np.stack([ab, b, bc2, bc_], 1),
                       np.stack([bc_, bc2, cc, cd], 1), np.stack([da, bc_, cd, d], 1)], 1).reshape(-1, 4)
        nbattr = np.repeat(m.bdr_attr, 4)
    else:
        raise ValueError(et)
    nchild = ch.shape[1]
    fine = Mesh(et, verts, ch.reshape(-1, ch.shape[2]).astype(np.int64),
                np.repeat(m.elem_attr, nchild).astype(np.int32), nb.astype(np.int64), nbattr.astype(np.int32))
    if et in ("tri", "tet"):
        _orient_simplices(fine)
    parent = np.repeat(np.arange(m.ne, dtype=np.int64), nchild)
    return fine, parent


def _lookup_mid(edges, ukey, nv):
    key = _row_keys(edges.reshape(-1, 2), nv)
    pos = np.searchsorted(ukey, key)
    if not (ukey[pos] == key).all():
        raise ValueError("boundary edge is not a mesh edge")
    return nv + pos


def element_volumes(m: Mesh) -> np.ndarray:
    if m.etype in ("tri", "tet"):
        return np.abs(_simplex_volume(m.verts, m.elems))
    p = m.verts[m.elems]
    ext = p.max(axis=1) - p.min(axis=1)
    return np.prod(ext, axis=1)


def element_centroids(m: Mesh) -> np.ndarray:
    return m.verts[m.elems].mean(axis=1)


def mesh_from_json(path) -> Mesh:
    """Load a mesh stored as JSON arrays (tests/golden/meshes/*.json, written by
    tests/golden/make_golden.py from the reference's mesh data files)."""
    import json
    with open(path, "r") as f:
        d = json.load(f)
    et = d["etype"]
    nvf = _FACE_NV[et]
    m = Mesh(et, np.asarray(d["verts"], np.float64), np.asarray(d["elems"], np.int64),
             np.asarray(d["elem_attr"], np.int32), np.asarray(d["bdr"], np.int64).reshape(-1, nvf),
             np.asarray(d["bdr_attr"], np.int32))
    if et in ("tri", "tet"):
        _orient_simplices(m)
    return m
