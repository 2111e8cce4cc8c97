// Element-local elimination of the sampler's mixed system: the setup step behind the reference's "Hybridization" solver.
//
// The reference hands A[i] and the de Rham sequence of the level to prec_factory->BuildSolver(A[i], state)
// (/root/reference/src/PDESampler.cpp:302-318, extra parameters :307-311: "RescaleIteration", the L2 mass weight alpha);
// ParELAG's HybridHdivL2 then eliminates (u, s) element by element.  This file is that step for a caller of libpmc.so:
// from what BuildHierarchy holds - the element decomposition of the u-mass matrix (exactly as pmc_darcy_level takes it),
// B without boundary elimination, diag(W), alpha - to the arrays pmc_sampler_create_hybrid consumes:
//
//     [[X, y], [y^T, z]]_e = [[M_e, b_e^T], [b_e, -alpha w_e]]^-1          one dense (n_fe + 1)^2 inverse per element
//     H = sum_e C_e X_e C_e^T        G = sum_e C_e y_e        z_diag[e] = z_e
//
// with C_e = sign(B[e, f]) (+1 for the element whose outward normal is the face's global normal, -1 for the other one; a
// boundary face has one element and its multiplier imposes u.n = 0 there, src/PDESampler.cpp:210-214).  Host code only
// (setup, outside every timed region); threads over element ranges.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>
#include <thread>

#include "common.hpp"

namespace pmc {

static constexpr int kMaxFe = 12;   // faces per element (tet 4, hex 6; agglomerated elements are not hybridized here)

struct HybridSystem {
    int n_lambda = 0, n_s = 0;
    HostCsr H, G;
    std::vector<double> z, w;
    HostCsr P;
    bool has_P = false;
};

// in-place inverse of the symmetric indefinite (n x n, n <= kMaxFe + 1) matrix a by Gauss-Jordan with partial pivoting;
// false if singular to working precision
static bool invert_small(double* a, int n) {
    int piv[kMaxFe + 1];
    double scale = 0.0;
    for (int i = 0; i < n * n; ++i) scale = std::max(scale, std::fabs(a[i]));
    if (!(scale > 0.0) || !std::isfinite(scale)) return false;
    for (int k = 0; k < n; ++k) {
        int p = k;
        double best = std::fabs(a[k * n + k]);
        for (int i = k + 1; i < n; ++i)
            if (std::fabs(a[i * n + k]) > best) { best = std::fabs(a[i * n + k]); p = i; }
        if (!(best > 1e-14 * scale)) return false;
        piv[k] = p;
        if (p != k)
            for (int j = 0; j < n; ++j) std::swap(a[k * n + j], a[p * n + j]);
        const double d = 1.0 / a[k * n + k];
        a[k * n + k] = 1.0;
        for (int j = 0; j < n; ++j) a[k * n + j] *= d;
        for (int i = 0; i < n; ++i) {
            if (i == k) continue;
            const double f = a[i * n + k];
            if (f == 0.0) continue;
            a[i * n + k] = 0.0;
            for (int j = 0; j < n; ++j) a[i * n + j] -= f * a[k * n + j];
        }
    }
    for (int k = n - 1; k >= 0; --k)
        if (piv[k] != k)
            for (int i = 0; i < n; ++i) std::swap(a[i * n + k], a[i * n + piv[k]]);
    return true;
}

template <class F>
static void parallel_ranges(int n, F&& f) {
    int nt = (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u);
    if (n < 20000) nt = 1;
    if (nt == 1) {
        f(0, n);
        return;
    }
    std::vector<std::thread> th;
    std::vector<std::exception_ptr> err(nt);
    for (int t = 0; t < nt; ++t)
        th.emplace_back([&, t] {
            try {
                f((int)((int64_t)n * t / nt), (int)((int64_t)n * (t + 1) / nt));
            } catch (...) {
                err[t] = std::current_exception();
            }
        });
    for (auto& x : th) x.join();
    for (auto& e : err)
        if (e) std::rethrow_exception(e);
}

ElementInverses element_inverses(const HostCsr& Mp, const HostCsr& B, const int32_t* c_ptr, const int32_t* c_elem,
                                 const double* c_val, const double* corner, const char* who) {
    const std::string w(who);
    const int nu = Mp.nrows, ns = B.nrows;
    PMC_REQUIRE(Mp.ncols == nu && B.ncols == nu, w + ": operator shapes");
    PMC_REQUIRE(c_ptr && c_elem && c_val, w + ": NULL contribution array");
    // element -> faces: the row of B (no boundary elimination: every face of the element is listed)
    int nfe_max = 0;
    for (int e = 0; e < ns; ++e) {
        const int len = B.rowptr[e + 1] - B.rowptr[e];
        PMC_REQUIRE(len >= 2 && len <= kMaxFe, w + ": an element (row of B) must list 2 ... 12 faces");
        nfe_max = std::max(nfe_max, len);
    }
    for (int64_t t = 0; t < B.nnz(); ++t)
        PMC_REQUIRE(B.vals[t] != 0.0, w + ": B stores an explicit zero (eliminated columns? hand over B as assembled)");
    // element mass matrices, global face orientation: Me[e][a][b] = sum of the contributions of element e to entry
    // (face a, face b) of the u-mass matrix
    const int m = nfe_max;
    std::vector<double> Me((size_t)ns * m * m, 0.0);
    const int64_t nnzM = Mp.nnz();
    PMC_REQUIRE(c_ptr[0] == 0, w + ": c_ptr[0] != 0");
    // rows of M are independent of each other only per ELEMENT pair, and two rows may write the same element block (at
    // different (a, b)): writes never collide, so rows can be split over threads
    parallel_ranges(nu, [&](int r0, int r1) {
        for (int i = r0; i < r1; ++i)
            for (int p = Mp.rowptr[i]; p < Mp.rowptr[i + 1]; ++p) {
                const int j = Mp.colind[p];
                for (int t = c_ptr[p]; t < c_ptr[p + 1]; ++t) {
                    const int e = c_elem[t];
                    PMC_REQUIRE(e >= 0 && e < ns, w + ": c_elem out of range");
                    const int b0 = B.rowptr[e], len = B.rowptr[e + 1] - b0;
                    int la = -1, lb = -1;
                    for (int q = 0; q < len; ++q) {
                        if (B.colind[b0 + q] == i) la = q;
                        if (B.colind[b0 + q] == j) lb = q;
                    }
                    PMC_REQUIRE(la >= 0 && lb >= 0, w + ": a mass contribution names an element that does not own the face");
                    Me[((size_t)e * m + la) * m + lb] += c_val[t];
                }
            }
    });
    PMC_REQUIRE(c_ptr[nnzM] >= nnzM, w + ": fewer contributions than stored entries");
    // local inverses
    ElementInverses out;
    out.m = m;
    out.X.resize((size_t)ns * m * m);
    out.Y.resize((size_t)ns * m);
    out.z.resize(ns);
    std::vector<double>&X = out.X, &Y = out.Y;
    std::vector<int> bad(1, -1);
    parallel_ranges(ns, [&](int e0, int e1) {
        double a[(kMaxFe + 1) * (kMaxFe + 1)];
        for (int e = e0; e < e1; ++e) {
            const int b0 = B.rowptr[e], len = B.rowptr[e + 1] - b0, n = len + 1;
            for (int p = 0; p < len; ++p) {
                for (int q = 0; q < len; ++q) a[p * n + q] = Me[((size_t)e * m + p) * m + q];
                a[p * n + len] = a[len * n + p] = B.vals[b0 + p];
            }
            a[len * n + len] = corner ? corner[e] : 0.0;
            if (!invert_small(a, n)) {
                bad[0] = e;
                continue;
            }
            for (int p = 0; p < len; ++p) {
                // C_e = sign of the element's B entry
                const double cp = B.vals[b0 + p] > 0.0 ? 1.0 : -1.0;
                for (int q = 0; q < len; ++q) {
                    const double cq = B.vals[b0 + q] > 0.0 ? 1.0 : -1.0;
                    // symmetrised: the inverse of a symmetric matrix, rounded asymmetrically by the elimination order
                    X[((size_t)e * m + p) * m + q] = cp * cq * 0.5 * (a[p * n + q] + a[q * n + p]);
                }
                Y[(size_t)e * m + p] = cp * 0.5 * (a[p * n + len] + a[len * n + p]);
            }
            out.z[e] = a[len * n + len];
        }
    });
    PMC_REQUIRE(bad[0] < 0, w + ": a local saddle-point matrix is singular (element " + std::to_string(bad[0]) + ")");
    return out;
}

static std::unique_ptr<HybridSystem> hybrid_build(const pmc_hybrid_elements& in, double alpha) {
    PMC_REQUIRE(in.n_u > 0 && in.n_s > 0 && alpha > 0.0 && std::isfinite(alpha), "pmc_hybrid_build: sizes / alpha");
    PMC_REQUIRE(in.c_ptr && in.c_elem && in.c_val && in.w_diag, "pmc_hybrid_build: NULL array");
    const int nu = in.n_u, ns = in.n_s;
    HostCsr Mp = csr_from_c(in.M_pattern, false, "pmc_hybrid_elements.M_pattern");
    HostCsr B = csr_from_c(in.B, true, "pmc_hybrid_elements.B");
    PMC_REQUIRE(Mp.nrows == nu && Mp.ncols == nu && B.nrows == ns && B.ncols == nu, "pmc_hybrid_build: operator shapes");
    auto sys = std::make_unique<HybridSystem>();
    sys->n_lambda = nu;
    sys->n_s = ns;
    sys->w.assign(in.w_diag, in.w_diag + ns);
    if (in.P.rowptr) {
        sys->P = csr_from_c(in.P, true, "pmc_hybrid_elements.P");
        PMC_REQUIRE(sys->P.nrows == ns, "pmc_hybrid_build: P must have n_s rows");
        sys->has_P = true;
    }
    std::vector<double> corner(ns);
    for (int e = 0; e < ns; ++e) {
        PMC_REQUIRE(in.w_diag[e] > 0.0, "pmc_hybrid_build: w_diag must be positive");
        corner[e] = -alpha * in.w_diag[e];
    }
    std::vector<int> face_count(nu, 0);
    for (int64_t t = 0; t < B.nnz(); ++t)
        PMC_REQUIRE(++face_count[B.colind[t]] <= 2, "pmc_hybrid_build: a face belongs to more than two elements");
    for (int f = 0; f < nu; ++f) PMC_REQUIRE(face_count[f] >= 1, "pmc_hybrid_build: a face belongs to no element");
    ElementInverses inv = element_inverses(Mp, B, in.c_ptr, in.c_elem, in.c_val, corner.data(), "pmc_hybrid_build");
    const int m = inv.m;
    const std::vector<double>&X = inv.X, &Y = inv.Y;
    sys->z = inv.z;
    // face -> (element, local index), at most two per face, in element order
    std::vector<int> fe((size_t)nu * 2, -1), fl((size_t)nu * 2, -1);
    for (int e = 0; e < ns; ++e)
        for (int p = B.rowptr[e]; p < B.rowptr[e + 1]; ++p) {
            const int f = B.colind[p], s = fe[2 * (size_t)f] < 0 ? 0 : 1;
            fe[2 * (size_t)f + s] = e;
            fl[2 * (size_t)f + s] = p - B.rowptr[e];
        }
    // G: row f = its elements in increasing order (they were entered in element order)
    HostCsr& G = sys->G;
    G.nrows = nu;
    G.ncols = ns;
    G.rowptr.assign(nu + 1, 0);
    for (int f = 0; f < nu; ++f) G.rowptr[f + 1] = G.rowptr[f] + face_count[f];
    G.colind.resize(G.rowptr[nu]);
    G.vals.resize(G.rowptr[nu]);
    // H: row f = union of the faces of its (one or two) elements, sorted, the shared face f itself merged
    HostCsr& H = sys->H;
    H.nrows = H.ncols = nu;
    H.rowptr.assign(nu + 1, 0);
    std::vector<int> rowlen(nu);
    parallel_ranges(nu, [&](int f0, int f1) {
        int cols[2 * kMaxFe];
        for (int f = f0; f < f1; ++f) {
            int n = 0;
            for (int s = 0; s < 2; ++s) {
                const int e = fe[2 * (size_t)f + s];
                if (e < 0) continue;
                for (int p = B.rowptr[e]; p < B.rowptr[e + 1]; ++p) cols[n++] = B.colind[p];
            }
            std::sort(cols, cols + n);
            rowlen[f] = (int)(std::unique(cols, cols + n) - cols);
        }
    });
    for (int f = 0; f < nu; ++f) H.rowptr[f + 1] = H.rowptr[f] + rowlen[f];
    H.colind.resize(H.rowptr[nu]);
    H.vals.assign(H.rowptr[nu], 0.0);
    parallel_ranges(nu, [&](int f0, int f1) {
        int cols[2 * kMaxFe];
        for (int f = f0; f < f1; ++f) {
            int n = 0;
            for (int s = 0; s < 2; ++s) {
                const int e = fe[2 * (size_t)f + s];
                if (e < 0) continue;
                for (int p = B.rowptr[e]; p < B.rowptr[e + 1]; ++p) cols[n++] = B.colind[p];
            }
            std::sort(cols, cols + n);
            n = (int)(std::unique(cols, cols + n) - cols);
            const int h0 = H.rowptr[f];
            std::copy(cols, cols + n, H.colind.begin() + h0);
            int g = G.rowptr[f];
            for (int s = 0; s < 2; ++s) {
                const int e = fe[2 * (size_t)f + s];
                if (e < 0) continue;
                const int la = fl[2 * (size_t)f + s], b0 = B.rowptr[e], len = B.rowptr[e + 1] - b0;
                for (int q = 0; q < len; ++q) {
                    const int pos = (int)(std::lower_bound(cols, cols + n, B.colind[b0 + q]) - cols);
                    H.vals[h0 + pos] += X[((size_t)e * m + la) * m + q];
                }
                G.colind[g] = e;
                G.vals[g] = Y[(size_t)e * m + la];
                ++g;
            }
        }
    });
    return sys;
}

}  // namespace pmc

struct pmc_hybrid_system : pmc::HybridSystem {};

template <class F>
static int hb_guarded(F&& f) {
    try {
        f();
        return PMC_OK;
    } catch (const pmc::Error& e) {
        pmc::set_last_error(e.what());
        return e.code;
    } catch (const std::bad_alloc&) {
        pmc::set_last_error("host allocation failed");
        return PMC_ERR_INTERNAL;
    } catch (const std::exception& e) {
        pmc::set_last_error(e.what());
        return PMC_ERR_INTERNAL;
    }
}

static pmc_csr view_of(const pmc::HostCsr& a) {
    return pmc_csr{a.nrows, a.ncols, a.rowptr.data(), a.colind.data(), a.vals.data()};
}

extern "C" {

int pmc_hybrid_build(const pmc_hybrid_elements* in, double alpha, pmc_hybrid_system** out) {
    return hb_guarded([&] {
        PMC_REQUIRE(in != nullptr && out != nullptr, "pmc_hybrid_build: NULL argument");
        *out = nullptr;
        auto sys = pmc::hybrid_build(*in, alpha);
        *out = static_cast<pmc_hybrid_system*>(sys.release());
    });
}

int pmc_hybrid_system_level(const pmc_hybrid_system* sys, pmc_hybrid_level* view) {
    return hb_guarded([&] {
        PMC_REQUIRE(sys != nullptr && view != nullptr, "pmc_hybrid_system_level: NULL argument");
        std::memset(view, 0, sizeof(*view));
        view->n_lambda = sys->n_lambda;
        view->n_s = sys->n_s;
        view->H = view_of(sys->H);
        view->G = view_of(sys->G);
        view->z_diag = sys->z.data();
        view->w_diag = sys->w.data();
        if (sys->has_P) view->P = view_of(sys->P);
    });
}

void pmc_hybrid_system_destroy(pmc_hybrid_system* sys) { delete static_cast<pmc::HybridSystem*>(sys); }

int pmc_sampler_create_hybrid_from_elements(pmc_ctx* ctx, int nlevels, const pmc_hybrid_elements* levels, double alpha,
                                            double matern_g, int lognormal, const pmc_solver_opts* opts, pmc_sampler** out) {
    std::vector<pmc_hybrid_system*> sys;
    int rc = hb_guarded([&] {
        PMC_REQUIRE(ctx != nullptr && levels != nullptr && out != nullptr && nlevels >= 1,
                    "pmc_sampler_create_hybrid_from_elements: NULL argument / nlevels");
        *out = nullptr;
    });
    if (rc != PMC_OK) return rc;
    std::vector<pmc_hybrid_level> hl((size_t)nlevels);
    for (int i = 0; i < nlevels && rc == PMC_OK; ++i) {
        pmc_hybrid_system* s = nullptr;
        rc = pmc_hybrid_build(&levels[i], alpha, &s);
        if (rc == PMC_OK) {
            sys.push_back(s);
            rc = pmc_hybrid_system_level(s, &hl[(size_t)i]);
        }
    }
    if (rc == PMC_OK) rc = pmc_sampler_create_hybrid(ctx, nlevels, hl.data(), alpha, matern_g, lognormal, opts, out);
    for (auto* s : sys) pmc_hybrid_system_destroy(s);   // the sampler has copied what it keeps
    return rc;
}

}  // extern "C"
