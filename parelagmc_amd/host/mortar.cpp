// P0 x P0 mortar (L2-projection) matrix between two NON-MATCHING meshes:  G[i,j] = | A_i  ∩  B_j |.
//
// Replaces, for the piecewise-constant spaces the sampler uses, the reference's ParMortarAssembler::Assemble
// (src/transfer/ParMortarAssembler.cpp:1127-1144: hash-grid candidate search, polytope intersection, composite
// quadrature of the product of the basis functions) as called by L2ProjectionPDESampler::BuildHierarchy
// (src/L2ProjectionPDESampler.cpp:488-505).  With phi == 1 on every element the quadrature collapses to the measure
// of the intersection, so this file only needs exact intersection measures:
//   * every element is split into simplices (triangle -> itself, quad -> 2, tet -> itself, hex -> 6 around the 0-6
//     diagonal), a simplex of A is clipped by the d+1 half-spaces of a simplex of B (Sutherland-Hodgman on the
//     polygon / on every face of the polyhedron, closing the cut with a cap polygon), and the measure of the convex
//     result is summed by fans from an interior point;
//   * candidate pairs come from a uniform bucket grid over the bounding boxes of B's elements.
// Setup-side, host-only code (runs once per mesh pair); the coarse levels follow by RAP with the P0 prolongators
// (src/L2ProjectionPDESampler.cpp:512-513), see fe/transfer.py.
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "pmc_host.h"

namespace {

struct P3 { double x, y, z; };
inline P3 operator-(P3 a, P3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline P3 operator+(P3 a, P3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline P3 operator*(double s, P3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline double dot(P3 a, P3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline P3 cross(P3 a, P3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double det3(P3 a, P3 b, P3 c) { return dot(a, cross(b, c)); }

using Poly = std::vector<P3>;

// ------------------------------------------------------------------------------------------- 2D
double polygon_area(const Poly& p) {
    double a = 0.0;
    for (size_t i = 0, n = p.size(); i < n; ++i) {
        const P3& u = p[i];
        const P3& v = p[(i + 1) % n];
        a += u.x * v.y - u.y * v.x;
    }
    return 0.5 * std::fabs(a);
}

// keep the part of the polygon with n . x <= d
void clip_polygon(Poly& p, P3 n, double d, double eps, Poly& tmp) {
    tmp.clear();
    for (size_t i = 0, m = p.size(); i < m; ++i) {
        const P3 a = p[i], b = p[(i + 1) % m];
        const double da = dot(n, a) - d, db = dot(n, b) - d;
        const bool ina = da <= eps, inb = db <= eps;
        if (ina) tmp.push_back(a);
        if (ina != inb) tmp.push_back(a + (da / (da - db)) * (b - a));
    }
    p.swap(tmp);
}

double tri_tri_area(const P3* A, const P3* B, double eps) {
    Poly p{A[0], A[1], A[2]}, tmp;
    // orientation of B
    const double o = (B[1].x - B[0].x) * (B[2].y - B[0].y) - (B[1].y - B[0].y) * (B[2].x - B[0].x);
    if (o == 0.0) return 0.0;
    for (int e = 0; e < 3 && p.size() >= 3; ++e) {
        const P3 u = B[e], v = B[(e + 1) % 3];
        P3 n{v.y - u.y, -(v.x - u.x), 0.0};          // outward normal of a counter-clockwise triangle
        if (o < 0.0) n = -1.0 * n;
        const double len = std::sqrt(dot(n, n));
        n = (1.0 / len) * n;
        clip_polygon(p, n, dot(n, u), eps, tmp);
    }
    return p.size() >= 3 ? polygon_area(p) : 0.0;
}

// ------------------------------------------------------------------------------------------- 3D
struct Polyhedron {
    std::vector<Poly> faces;
};

// keep the part with n . x <= d (|n| = 1); closes the cut with a cap polygon ordered around its centroid
void clip_polyhedron(Polyhedron& ph, P3 n, double d, double eps, Polyhedron& out, Poly& cap) {
    out.faces.clear();
    cap.clear();
    bool any_outside = false;
    for (const Poly& f : ph.faces) {
        Poly g;
        for (size_t i = 0, m = f.size(); i < m; ++i) {
            const P3 a = f[i], b = f[(i + 1) % m];
            const double da = dot(n, a) - d, db = dot(n, b) - d;
            const bool ina = da <= eps, inb = db <= eps;
            if (!ina) any_outside = true;
            if (ina) {
                g.push_back(a);
                if (std::fabs(da) <= eps) cap.push_back(a);
            }
            if (ina != inb) {
                const P3 q = a + (da / (da - db)) * (b - a);
                g.push_back(q);
                cap.push_back(q);
            }
        }
        if (g.size() >= 3) out.faces.push_back(std::move(g));
    }
    if (any_outside && cap.size() >= 3) {
        // order the cap points by angle in the cutting plane
        P3 c{0, 0, 0};
        for (const P3& q : cap) c = c + q;
        c = (1.0 / cap.size()) * c;
        P3 e1 = std::fabs(n.x) < 0.9 ? P3{1, 0, 0} : P3{0, 1, 0};
        e1 = e1 - dot(e1, n) * n;
        e1 = (1.0 / std::sqrt(dot(e1, e1))) * e1;
        const P3 e2 = cross(n, e1);
        std::sort(cap.begin(), cap.end(), [&](const P3& u, const P3& v) {
            return std::atan2(dot(u - c, e2), dot(u - c, e1)) < std::atan2(dot(v - c, e2), dot(v - c, e1));
        });
        out.faces.push_back(cap);
    }
    ph.faces.swap(out.faces);
}

double polyhedron_volume(const Polyhedron& ph) {
    // fans from an interior point (the vertex average): the result is convex, so all cones are disjoint
    P3 c{0, 0, 0};
    size_t cnt = 0;
    for (const Poly& f : ph.faces)
        for (const P3& q : f) { c = c + q; ++cnt; }
    if (cnt == 0) return 0.0;
    c = (1.0 / cnt) * c;
    double v = 0.0;
    for (const Poly& f : ph.faces)
        for (size_t i = 1; i + 1 < f.size(); ++i) v += std::fabs(det3(f[0] - c, f[i] - c, f[i + 1] - c));
    return v / 6.0;
}

double tet_tet_volume(const P3* A, const P3* B, double eps, Polyhedron& ph, Polyhedron& scratch, Poly& cap) {
    static const int F[4][3] = {{1, 2, 3}, {0, 3, 2}, {0, 1, 3}, {0, 2, 1}};
    ph.faces.clear();
    for (int f = 0; f < 4; ++f) ph.faces.push_back(Poly{A[F[f][0]], A[F[f][1]], A[F[f][2]]});
    for (int f = 0; f < 4; ++f) {
        const P3 a = B[F[f][0]], b = B[F[f][1]], c = B[F[f][2]];
        P3 n = cross(b - a, c - a);
        const double len = std::sqrt(dot(n, n));
        if (len == 0.0) return 0.0;
        n = (1.0 / len) * n;
        // make n point away from the opposite vertex
        if (dot(n, B[f] - a) > 0.0) n = -1.0 * n;
        clip_polyhedron(ph, n, dot(n, a), eps, scratch, cap);
        if (ph.faces.size() < 4) return 0.0;   // a solid needs at least 4 faces
    }
    return polyhedron_volume(ph);
}

// ------------------------------------------------------------------------------------------- elements -> simplices
struct MeshSimplices {
    int dim = 0, nelem = 0, per_elem = 0;         // per_elem simplices of dim+1 points each
    std::vector<P3> pts;                          // nelem * per_elem * (dim+1)
    std::vector<std::array<double, 6>> bbox;      // per element: lo xyz, hi xyz
    std::vector<double> measure;                  // per element
    std::vector<char> is_box;                     // element is an axis-aligned box (= its bounding box)
};

double simplex_measure(int dim, const P3* s) {
    if (dim == 2) return 0.5 * std::fabs((s[1].x - s[0].x) * (s[2].y - s[0].y) - (s[1].y - s[0].y) * (s[2].x - s[0].x));
    return std::fabs(det3(s[1] - s[0], s[2] - s[0], s[3] - s[0])) / 6.0;
}

MeshSimplices split(const pmc_mesh_view& m) {
    if (!(m.dim == 2 || m.dim == 3)) throw std::invalid_argument("mortar: dimension must be 2 or 3");
    if (!m.verts || !m.elems || m.nverts <= 0 || m.nelems <= 0) throw std::invalid_argument("mortar: empty mesh");
    static const int quad[2][3] = {{0, 1, 2}, {0, 2, 3}};
    // six tetrahedra sharing the 0-6 diagonal of an MFEM-ordered hexahedron (bottom 0123, top 4567)
    static const int hex[6][4] = {{0, 1, 2, 6}, {0, 2, 3, 6}, {0, 3, 7, 6}, {0, 7, 4, 6}, {0, 4, 5, 6}, {0, 5, 1, 6}};
    MeshSimplices s;
    s.dim = m.dim;
    s.nelem = m.nelems;
    const int npe = m.verts_per_elem;
    const int ns = (m.dim == 2) ? (npe == 3 ? 1 : npe == 4 ? 2 : 0) : (npe == 4 ? 1 : npe == 8 ? 6 : 0);
    if (ns == 0) throw std::invalid_argument("mortar: unsupported element type (vertices per element)");
    s.per_elem = ns;
    const int np = m.dim + 1;
    s.pts.resize((size_t)m.nelems * ns * np);
    s.bbox.resize(m.nelems);
    s.measure.assign(m.nelems, 0.0);
    auto vert = [&](int v) {
        if (v < 0 || v >= m.nverts) throw std::invalid_argument("mortar: vertex index out of range");
        const double* p = m.verts + (size_t)v * m.dim;
        return P3{p[0], p[1], m.dim == 3 ? p[2] : 0.0};
    };
    for (int e = 0; e < m.nelems; ++e) {
        const int32_t* ev = m.elems + (size_t)e * npe;
        std::array<double, 6> bb{1e300, 1e300, 1e300, -1e300, -1e300, -1e300};
        for (int k = 0; k < npe; ++k) {
            const P3 p = vert(ev[k]);
            bb[0] = std::min(bb[0], p.x); bb[1] = std::min(bb[1], p.y); bb[2] = std::min(bb[2], p.z);
            bb[3] = std::max(bb[3], p.x); bb[4] = std::max(bb[4], p.y); bb[5] = std::max(bb[5], p.z);
        }
        s.bbox[e] = bb;
        // axis-aligned box: every vertex sits on a corner of the bounding box (quadrilaterals / hexahedra only)
        bool box = (m.dim == 2 && npe == 4) || (m.dim == 3 && npe == 8);
        for (int k = 0; k < npe && box; ++k) {
            const P3 p = vert(ev[k]);
            const double c[3] = {p.x, p.y, p.z};
            for (int a = 0; a < m.dim && box; ++a) {
                const double tol = 1e-12 * std::max(bb[3 + a] - bb[a], 1e-300);
                box = std::fabs(c[a] - bb[a]) <= tol || std::fabs(c[a] - bb[3 + a]) <= tol;
            }
        }
        s.is_box.push_back(box ? 1 : 0);
        for (int t = 0; t < ns; ++t) {
            P3* dst = &s.pts[((size_t)e * ns + t) * np];
            for (int k = 0; k < np; ++k) {
                int loc = k;
                if (m.dim == 2 && npe == 4) loc = quad[t][k];
                if (m.dim == 3 && npe == 8) loc = hex[t][k];
                dst[k] = vert(ev[loc]);
            }
            s.measure[e] += simplex_measure(m.dim, dst);
        }
    }
    return s;
}

}  // namespace

struct pmc_mortar {
    int nrows = 0, ncols = 0;
    std::vector<int32_t> rowptr, colind;
    std::vector<double> vals;
    std::vector<double> measure_a, measure_b;
};

static thread_local std::string g_mortar_err;
extern "C" const char* pmc_mortar_last_error(void) { return g_mortar_err.c_str(); }

extern "C" int pmc_mortar_assemble(const pmc_mesh_view* a, const pmc_mesh_view* b, double rel_tol, pmc_mortar** out) {
    try {
        if (!a || !b || !out) throw std::invalid_argument("mortar: NULL argument");
        if (a->dim != b->dim) throw std::invalid_argument("mortar: meshes of different dimension");
        const MeshSimplices A = split(*a), B = split(*b);
        const int dim = A.dim, np = dim + 1;
        if (rel_tol <= 0.0) rel_tol = 1e-12;
        // bucket grid over B's bounding boxes
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300}, hmean[3] = {0, 0, 0};
        for (const auto& bb : B.bbox)
            for (int k = 0; k < 3; ++k) {
                lo[k] = std::min(lo[k], bb[k]);
                hi[k] = std::max(hi[k], bb[3 + k]);
                hmean[k] += bb[3 + k] - bb[k];
            }
        int ng[3] = {1, 1, 1};
        double cell[3] = {1, 1, 1};
        for (int k = 0; k < dim; ++k) {
            hmean[k] /= B.nelem;
            const double ext = std::max(hi[k] - lo[k], 1e-300);
            ng[k] = (int)std::min(512.0, std::max(1.0, std::floor(ext / std::max(hmean[k], 1e-300))));
            cell[k] = ext / ng[k];
        }
        auto cell_of = [&](double x, int k) {
            const int c = (int)std::floor((x - lo[k]) / cell[k]);
            return std::min(ng[k] - 1, std::max(0, c));
        };
        std::vector<std::vector<int>> bucket((size_t)ng[0] * ng[1] * ng[2]);
        for (int j = 0; j < B.nelem; ++j) {
            const auto& bb = B.bbox[j];
            for (int cz = (dim == 3 ? cell_of(bb[2], 2) : 0); cz <= (dim == 3 ? cell_of(bb[5], 2) : 0); ++cz)
                for (int cy = cell_of(bb[1], 1); cy <= cell_of(bb[4], 1); ++cy)
                    for (int cx = cell_of(bb[0], 0); cx <= cell_of(bb[3], 0); ++cx)
                        bucket[((size_t)cz * ng[1] + cy) * ng[0] + cx].push_back(j);
        }
        std::unique_ptr<pmc_mortar> M(new pmc_mortar());
        M->nrows = A.nelem;
        M->ncols = B.nelem;
        M->measure_a = A.measure;
        M->measure_b = B.measure;
        M->rowptr.assign(A.nelem + 1, 0);
        std::vector<int> stamp(B.nelem, -1), cand;
        Polyhedron ph, scratch;
        Poly cap;
        for (int i = 0; i < A.nelem; ++i) {
            const auto& ab = A.bbox[i];
            cand.clear();
            const bool out_of_grid = ab[3] < lo[0] || ab[0] > hi[0] || ab[4] < lo[1] || ab[1] > hi[1] ||
                                     (dim == 3 && (ab[5] < lo[2] || ab[2] > hi[2]));
            if (!out_of_grid) {
                for (int cz = (dim == 3 ? cell_of(ab[2], 2) : 0); cz <= (dim == 3 ? cell_of(ab[5], 2) : 0); ++cz)
                    for (int cy = cell_of(ab[1], 1); cy <= cell_of(ab[4], 1); ++cy)
                        for (int cx = cell_of(ab[0], 0); cx <= cell_of(ab[3], 0); ++cx)
                            for (int j : bucket[((size_t)cz * ng[1] + cy) * ng[0] + cx]) {
                                if (stamp[j] == i) continue;
                                stamp[j] = i;
                                const auto& bb = B.bbox[j];
                                bool ov = true;
                                for (int k = 0; k < dim; ++k) ov = ov && ab[k] <= bb[3 + k] && bb[k] <= ab[3 + k];
                                if (ov) cand.push_back(j);
                            }
            }
            std::sort(cand.begin(), cand.end());
            const double scale = std::pow(std::max(A.measure[i], 1e-300), 1.0 / dim);
            for (int j : cand) {
                const double sc = std::min(scale, std::pow(std::max(B.measure[j], 1e-300), 1.0 / dim));
                const double eps = 1e-13 * sc;       // "on the plane" distance
                double v = 0.0;
                if (A.is_box[i] && B.is_box[j]) {
                    // two axis-aligned boxes: exact product of the interval overlaps, no clipping
                    const auto& bb = B.bbox[j];
                    v = 1.0;
                    for (int k = 0; k < dim; ++k) v *= std::max(0.0, std::min(ab[3 + k], bb[3 + k]) - std::max(ab[k], bb[k]));
                } else
                for (int s = 0; s < A.per_elem; ++s)
                    for (int t = 0; t < B.per_elem; ++t) {
                        const P3* sa = &A.pts[((size_t)i * A.per_elem + s) * np];
                        const P3* sb = &B.pts[((size_t)j * B.per_elem + t) * np];
                        v += (dim == 2) ? tri_tri_area(sa, sb, eps) : tet_tet_volume(sa, sb, eps, ph, scratch, cap);
                    }
                if (v > rel_tol * std::min(A.measure[i], B.measure[j])) {
                    M->colind.push_back(j);
                    M->vals.push_back(v);
                }
            }
            M->rowptr[i + 1] = (int32_t)M->colind.size();
        }
        *out = M.release();
        return PMC_OK;
    } catch (const std::exception& e) {
        g_mortar_err = e.what();
        return PMC_ERR_INVALID;
    }
}

extern "C" int64_t pmc_mortar_nnz(const pmc_mortar* m) { return m ? (int64_t)m->colind.size() : -1; }

extern "C" int pmc_mortar_get(const pmc_mortar* m, int32_t* rowptr, int32_t* colind, double* vals, double* measure_a,
                              double* measure_b) {
    if (!m) return PMC_ERR_INVALID;
    if (rowptr) std::memcpy(rowptr, m->rowptr.data(), sizeof(int32_t) * m->rowptr.size());
    if (colind) std::memcpy(colind, m->colind.data(), sizeof(int32_t) * m->colind.size());
    if (vals) std::memcpy(vals, m->vals.data(), sizeof(double) * m->vals.size());
    if (measure_a) std::memcpy(measure_a, m->measure_a.data(), sizeof(double) * m->measure_a.size());
    if (measure_b) std::memcpy(measure_b, m->measure_b.data(), sizeof(double) * m->measure_b.size());
    return PMC_OK;
}

extern "C" void pmc_mortar_destroy(pmc_mortar* m) { delete m; }
