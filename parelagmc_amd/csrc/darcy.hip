// Mixed Darcy forward solve on the device: DarcySolver::SolveFwd (reference:
// src/DarcySolver.cpp:416-437 -> assemble :472-520 -> solve :562-649).  The reference re-assembles
// M(k) and REBUILDS the whole solver (incl. AMG setup) for every realization (:568-601,
// src/DarcySolver.hpp:34); here all symbolic work is done once at create time and a realization
// only refreshes values on fixed patterns (K12-K14 of SURVEY.md 2.3).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <numeric>

#include "handles.hpp"

namespace pmc {

namespace {

struct Triple { int r, c, idx; double w; };

// pattern + contribution lists from a bag of (row, col, idx, w) tuples; `extra` pattern entries
// (row, col) are added with empty lists.  Output CSR has sorted columns; lists are in CSR order.
struct Symbolic {
    HostCsr pat;                // vals unused (zeros)
    std::vector<int> ptr, idx;  // ptr: nnz+1
    std::vector<double> w;
};

Symbolic build_symbolic(int n, std::vector<Triple>& t) {
    std::sort(t.begin(), t.end(), [](const Triple& a, const Triple& b) {
        if (a.r != b.r) return a.r < b.r;
        if (a.c != b.c) return a.c < b.c;
        return a.idx < b.idx;
    });
    Symbolic s;
    s.pat.nrows = s.pat.ncols = n;
    s.pat.rowptr.assign(n + 1, 0);
    s.ptr.push_back(0);
    size_t i = 0;
    while (i < t.size()) {
        size_t j = i;
        while (j < t.size() && t[j].r == t[i].r && t[j].c == t[i].c) {
            if (t[j].idx >= 0) { s.idx.push_back(t[j].idx); s.w.push_back(t[j].w); }
            ++j;
        }
        s.pat.colind.push_back(t[i].c);
        s.pat.rowptr[t[i].r + 1]++;
        s.ptr.push_back((int)s.idx.size());
        i = j;
    }
    std::partial_sum(s.pat.rowptr.begin(), s.pat.rowptr.end(), s.pat.rowptr.begin());
    s.pat.vals.assign(s.pat.colind.size(), 0.0);
    return s;
}

// re-order CSR-nnz-indexed lists into SELL slot order
void lists_to_slots(const Sell& S, const Symbolic& sym, std::vector<int>& ptr, std::vector<int>& idx,
                    std::vector<double>& w) {
    ptr.assign(S.nslots + 1, 0);
    idx.clear();
    w.clear();
    idx.reserve(sym.idx.size());
    w.reserve(sym.w.size());
    for (int64_t s = 0; s < S.nslots; ++s) {
        const int p = S.h_src[s];
        if (p >= 0)
            for (int t = sym.ptr[p]; t < sym.ptr[p + 1]; ++t) { idx.push_back(sym.idx[t]); w.push_back(sym.w[t]); }
        ptr[s + 1] = (int)idx.size();
    }
}

std::vector<int> nnz_to_slot(const Sell& S, int64_t nnz) {
    std::vector<int> m(nnz, -1);
    for (int64_t s = 0; s < S.nslots; ++s)
        if (S.h_src[s] >= 0) m[S.h_src[s]] = (int)s;
    return m;
}

HostCsr drop_columns(const HostCsr& B, const unsigned char* mask) {
    HostCsr o;
    o.nrows = B.nrows;
    o.ncols = B.ncols;
    o.rowptr.assign(B.nrows + 1, 0);
    for (int i = 0; i < B.nrows; ++i) {
        for (int p = B.rowptr[i]; p < B.rowptr[i + 1]; ++p)
            if (!mask[B.colind[p]]) { o.colind.push_back(B.colind[p]); o.vals.push_back(B.vals[p]); }
        o.rowptr[i + 1] = (int)o.colind.size();
    }
    return o;
}

struct SetupClock {   // PMC_VERBOSE=1: setup phase timings on stderr
    const bool on = getenv("PMC_VERBOSE") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(const char* what, int level) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[pmc] darcy setup L%d %-28s %8.2f s\n", level, what, std::chrono::duration<double>(now - t).count());
        t = now;
    }
};

std::vector<int> diag_slots(const Sell& S, const HostCsr& pat) {
    std::vector<int> ds(pat.nrows, -1);
    std::vector<int> n2s = nnz_to_slot(S, pat.nnz());
    for (int e = 0; e < pat.nrows; ++e)
        for (int p = pat.rowptr[e]; p < pat.rowptr[e + 1]; ++p)
            if (pat.colind[p] == e) ds[e] = n2s[p];
    for (int v : ds) PMC_REQUIRE(v >= 0, "darcy: Schur pattern lacks a diagonal entry");
    return ds;
}

// chain of one MC level: own Schur lists `own` (over diag(M)), prolongators from the k == 1 operator K1
// plain: plain aggregation by coupling magnitude (indicator prolongators) instead of smoothed aggregation; galerkin_scale: the
// coarse operators are that factor times P^T S P (over-correction of a piecewise-constant prolongator, LAB_NOTES 10.15)
std::unique_ptr<DarcyChain> build_chain(const Symbolic& own, const HostCsr& K1, const pmc_solver_opts& o, hipStream_t st,
                                        bool plain = false, double galerkin_scale = 1.0, double ratio_scale = 1.0) {
    int passes0 = 3, passes1 = 3;
    if (const char* e = lab_env("PMC_DARCY_HYB_PASSES0")) passes0 = atoi(e);
    if (const char* e = lab_env("PMC_DARCY_HYB_PASSES1")) passes1 = atoi(e);
    std::vector<AmgLevelHost> lvh = plain ? agg_hierarchy(K1, passes0, passes1, /*theta=*/0.25, /*min_size=*/256,
                                                          /*max_levels=*/14)
                                          : sa_hierarchy(K1, std::vector<double>(), /*passes=*/2, /*theta=*/0.25,
                                                         /*min_size=*/256, /*max_levels=*/14);
    SetupClock clk;
    clk.lap("  sa_hierarchy", (int)lvh.size());
    std::unique_ptr<DarcyChain> ch(new DarcyChain());
    Multigrid& mg = ch->mg;
    mg.smooth_degree = o.mg_smooth_degree;
    mg.smooth_ratio = ratio_scale * o.mg_smooth_ratio;
    mg.coarse_degree = o.mg_coarse_degree;
    mg.coarse_ratio = o.mg_coarse_ratio;
    mg.f32_intermediates = o.precond_storage != PMC_STORAGE_FP64;
    const int nl = (int)lvh.size();
    mg.L.resize(nl);
    ch->cl.resize(nl);
    Symbolic sym = own;                      // lists of the current level; idx refers to the source array of its refresh
    for (int j = 0; j < nl; ++j) {
        MgLevel& m = mg.L[j];
        DarcyChainLevel& c = ch->cl[j];
        const HostCsr& pat = sym.pat;
        m.n = pat.nrows;
        m.bv = true;
        m.f32 = o.precond_storage != PMC_STORAGE_FP64;   // fp32 copies of the per-realization values for the V-cycle kernels
        m.lmax = 1.0;                        // dinv carries the per-realization Gershgorin bound (k::gersh_scale_bv)
        sell_build(m.S, pat, false, true, st);   // per-realization values: sized by Darcy::ensure for the width in use
        {
            std::vector<int> ptr, idx;
            std::vector<double> w;
            lists_to_slots(m.S, sym, ptr, idx, w);
            c.ptr.upload(ptr, st);
            c.idx.upload(idx, st);
            c.w.upload(w, st);
            c.diag_slot.upload(diag_slots(m.S, pat), st);
        }
        PMC_HIP(hipStreamSynchronize(st));
        if (j + 1 == nl) break;
        const HostCsr& P = lvh[j].P;
        PMC_REQUIRE(P.nrows == m.n, "darcy: aggregation prolongator of the wrong height");
        sell_build(m.P, P, true, false, st);
        sell_build(m.Pt, csr_transpose(P), true, false, st);
        // S_{j+1} = P^T S_j P as lists over the SELL slots of S_j
        std::vector<int> n2s = nnz_to_slot(m.S, pat.nnz());
        std::vector<Triple> tr;
        tr.reserve((size_t)pat.nnz() * 4);
        for (int e = 0; e < pat.nrows; ++e)
            for (int p = pat.rowptr[e]; p < pat.rowptr[e + 1]; ++p) {
                const int e2 = pat.colind[p];
                for (int a = P.rowptr[e]; a < P.rowptr[e + 1]; ++a)
                    for (int b = P.rowptr[e2]; b < P.rowptr[e2 + 1]; ++b)
                        tr.push_back({P.colind[a], P.colind[b], n2s[p], galerkin_scale * P.vals[a] * P.vals[b]});
            }
        const size_t ntr = tr.size();
        Symbolic next = build_symbolic(P.ncols, tr);
        PMC_HIP(hipStreamSynchronize(st));
        if (clk.on)
            fprintf(stderr, "[pmc]   chain level %d: n %d nnz %lld, P nnz %lld, %zu triples -> coarse n %d nnz %lld\n", j, m.n,
                    (long long)pat.nnz(), (long long)P.nnz(), ntr, P.ncols, (long long)next.pat.nnz());
        clk.lap("  chain level symbolic", j);
        sym = std::move(next);
    }
    for (int j = 0; j < nl; ++j) {
        mg.L[j].S.h_src.clear(); mg.L[j].S.h_src.shrink_to_fit();
        mg.L[j].S.h_cols.clear(); mg.L[j].S.h_cols.shrink_to_fit();
    }
    mg.enable_bv_tail();
    mg.build_tails(st);
    return ch;
}

}  // namespace


Darcy::Darcy(Ctx& c, int nlevels_, int n_mc_, const pmc_darcy_level* in, bool kdiv, const pmc_solver_opts& o, bool hybrid_)
    : ctx(c), nlevels(nlevels_), n_mc(n_mc_), k_divides(kdiv), hybrid(hybrid_), opts(o) {
    SetupClock clk;
    PMC_REQUIRE(nlevels >= 1 && n_mc >= 1 && n_mc <= nlevels, "darcy: need 1 <= n_mc_levels <= nlevels");
    PMC_REQUIRE(in != nullptr, "darcy: levels is NULL");
    ctx.activate();
    hipStream_t st = ctx.stream;
    lv.resize(nlevels);
    mg.L.resize(nlevels);
    mg.smooth_degree = o.mg_smooth_degree;
    mg.smooth_ratio = o.mg_smooth_ratio;
    mg.coarse_degree = o.mg_coarse_degree;
    mg.coarse_ratio = o.mg_coarse_ratio;
    mg.f32_intermediates = o.precond_storage != PMC_STORAGE_FP64;

    std::vector<Symbolic> schur(nlevels);          // own (rediscretised) Schur lists per level
    std::vector<HostCsr> Pl(nlevels);
    for (int l = 0; l < nlevels; ++l) {
        const pmc_darcy_level& L = in[l];
        DarcyLevel& d = lv[l];
        PMC_REQUIRE(L.n_u > 0 && L.n_p > 0, "darcy level: empty block");
        PMC_REQUIRE(L.c_ptr && L.c_elem && L.c_val && L.rhs && L.ess_mask && L.ess_data && L.obs,
                    "darcy level: NULL array");
        d.n_u = L.n_u;
        d.n_p = L.n_p;
        HostCsr Mp = csr_from_c(L.M_pattern, false, "darcy M_pattern");
        PMC_REQUIRE(Mp.nrows == L.n_u && Mp.ncols == L.n_u, "darcy M_pattern: wrong shape");
        const int64_t nnzM = Mp.nnz();
        PMC_REQUIRE(L.c_ptr[0] == 0, "darcy c_ptr[0] != 0");
        for (int64_t p = 0; p < nnzM; ++p) PMC_REQUIRE(L.c_ptr[p + 1] >= L.c_ptr[p], "darcy c_ptr not monotone");
        const int ncontrib = L.c_ptr[nnzM];
        for (int t = 0; t < ncontrib; ++t) PMC_REQUIRE(L.c_elem[t] >= 0 && L.c_elem[t] < L.n_p, "darcy c_elem out of range");
        bool has_diag_all = true;
        for (int i = 0; i < Mp.nrows && has_diag_all; ++i) {
            bool f = false;
            for (int p = Mp.rowptr[i]; p < Mp.rowptr[i + 1]; ++p) f |= (Mp.colind[p] == i);
            has_diag_all = f;
        }
        PMC_REQUIRE(has_diag_all, "darcy M_pattern must store the diagonal");
        sell_build(d.M, Mp, false, true, st);
        d.slot_src.upload(d.M.h_src, st);
        clk.lap("checks + M pattern", l);
        {
            // element-grouped layout: expand the contribution lists per row and group them by element
            std::vector<std::vector<std::pair<int, std::vector<std::pair<int, double>>>>> rows(Mp.nrows);
            bool ok = true;
            int gw = 1;
            for (int i = 0; i < Mp.nrows && ok; ++i) {
                auto& groups = rows[i];
                if (L.ess_mask[i]) {
                    groups.push_back({L.n_p, {{i, 1.0}}});       // identity row: constant-one coefficient row
                    continue;
                }
                for (int p = Mp.rowptr[i]; p < Mp.rowptr[i + 1]; ++p) {
                    const int j = Mp.colind[p];
                    for (int t = L.c_ptr[p]; t < L.c_ptr[p + 1]; ++t) {
                        const int e = L.c_elem[t];
                        size_t gi = 0;
                        while (gi < groups.size() && groups[gi].first != e) ++gi;
                        if (gi == groups.size()) groups.push_back({e, {}});
                        groups[gi].second.push_back({j, L.ess_mask[j] ? 0.0 : L.c_val[t]});
                    }
                }
                if (groups.size() > 2) ok = false;
                for (auto& gr : groups) gw = std::max(gw, (int)gr.second.size());
            }
            if (ok && gw <= 16) {
                HostCsr Meg;
                Meg.nrows = Meg.ncols = Mp.nrows;
                Meg.rowptr.resize(Mp.nrows + 1);
                Meg.colind.assign((size_t)Mp.nrows * 2 * gw, 0);
                Meg.vals.assign((size_t)Mp.nrows * 2 * gw, 0.0);
                std::vector<int> e12((size_t)Mp.nrows * 2, L.n_p);
                for (int i = 0; i < Mp.nrows; ++i) {
                    Meg.rowptr[i] = i * 2 * gw;
                    for (int q = 0; q < 2 * gw; ++q) Meg.colind[(size_t)i * 2 * gw + q] = i;   // padding: own (cached) entry, weight 0
                    for (size_t gi = 0; gi < rows[i].size(); ++gi) {
                        e12[2 * i + gi] = rows[i][gi].first;
                        for (size_t q = 0; q < rows[i][gi].second.size(); ++q) {
                            Meg.colind[(size_t)i * 2 * gw + gi * gw + q] = rows[i][gi].second[q].first;
                            Meg.vals[(size_t)i * 2 * gw + gi * gw + q] = rows[i][gi].second[q].second;
                        }
                    }
                }
                Meg.rowptr[Mp.nrows] = Mp.nrows * 2 * gw;
                sell_build(d.Meg, Meg, true, false, st);
                d.eg_e12.upload(e12, st);
                d.eg_gw = gw;
                d.has_eg = true;
                PMC_HIP(hipStreamSynchronize(st));
            }
        }
        clk.lap("element-grouped M", l);
        if (o.cheb_ratio_M > 1.0) {
            d.ratio_M = o.cheb_ratio_M;
        } else {
            // interval of the M-block smoother from M(k == 1) after the essential elimination; the element-wise lower
            // bound behind it does not depend on the coefficient (M(k) and its l1 diagonal are the same positive
            // combination of element matrices)
            HostCsr M1 = Mp;
            for (int i = 0; i < M1.nrows; ++i)
                for (int p = M1.rowptr[i]; p < M1.rowptr[i + 1]; ++p) {
                    const int j = M1.colind[p];
                    double v = 0.0;
                    for (int t = L.c_ptr[p]; t < L.c_ptr[p + 1]; ++t) v += L.c_val[t];
                    if (L.ess_mask[i] || L.ess_mask[j]) v = (i == j) ? 1.0 : 0.0;
                    M1.vals[p] = v;
                }
            std::vector<double> l1(M1.nrows);
            for (int i = 0; i < M1.nrows; ++i) {
                double sabs = 0.0;
                for (int p = M1.rowptr[i]; p < M1.rowptr[i + 1]; ++p) sabs += std::fabs(M1.vals[p]);
                l1[i] = 1.0 / sabs;
            }
            d.ratio_M = mass_block_ratio(M1, l1);
        }
        d.c_ptr.upload(L.c_ptr, nnzM + 1, st);
        d.c_elem.upload(L.c_elem, ncontrib, st);
        d.c_val.upload(L.c_val, ncontrib, st);
        clk.lap("M-block interval (Lanczos)", l);
        d.ess.upload(L.ess_mask, L.n_u, st);
        d.ess_data.upload(L.ess_data, L.n_u, st);
        d.rhs_u0.upload(L.rhs, L.n_u, st);
        d.obs.upload(L.obs, L.n_u + L.n_p, st);
        {
            std::vector<int> rows;
            std::vector<double> w;
            for (int i = 0; i < L.n_u + L.n_p; ++i)
                if (L.obs[i] != 0.0) { rows.push_back(i); w.push_back(L.obs[i]); }
            d.n_obs = (int)rows.size();
            d.obs_rows.upload(rows, st);
            d.obs_w.upload(w, st);
        }

        HostCsr Bfull = csr_from_c(L.B, true, "darcy B");
        PMC_REQUIRE(Bfull.nrows == L.n_p && Bfull.ncols == L.n_u, "darcy B: wrong shape");
        csr_sort_rows(Bfull);
        // p-block of rhs_bc: rhs_p - B[:,ess] ess_data  (EliminateRowCol, DarcySolver.cpp:498)
        std::vector<double> rp(L.n_p);
        for (int e = 0; e < L.n_p; ++e) {
            double s = L.rhs[L.n_u + e];
            for (int p = Bfull.rowptr[e]; p < Bfull.rowptr[e + 1]; ++p)
                if (L.ess_mask[Bfull.colind[p]]) s -= Bfull.vals[p] * L.ess_data[Bfull.colind[p]];
            rp[e] = s;
        }
        d.rhs_p.upload(rp, st);
        HostCsr B = drop_columns(Bfull, L.ess_mask);
        HostCsr Bt = csr_transpose(B);
        sell_build(d.B, B, true, false, st);
        sell_build(d.Bt, Bt, true, false, st);
        d.nnz = nnzM + 2 * B.nnz();

        // symbolic S = B diag(M)^-1 B^T
        std::vector<Triple> tr;
        tr.reserve((size_t)B.nnz() * 3);
        for (int e = 0; e < B.nrows; ++e) {
            tr.push_back({e, e, -1, 0.0});
            for (int p = B.rowptr[e]; p < B.rowptr[e + 1]; ++p) {
                const int f = B.colind[p];
                for (int q = Bt.rowptr[f]; q < Bt.rowptr[f + 1]; ++q)
                    tr.push_back({e, Bt.colind[q], f, B.vals[p] * Bt.vals[q]});
            }
        }
        schur[l] = build_symbolic(L.n_p, tr);
        clk.lap("uploads + symbolic Schur", l);
        if (l < n_mc && o.mg_coarsening != 0) {
            // algebraic hierarchy of this MC level, prolongators from the k == 1 operator (c(1) = 1 either way)
            std::vector<double> dM1(L.n_u, 1.0);
            for (int i = 0; i < Mp.nrows; ++i)
                for (int p = Mp.rowptr[i]; p < Mp.rowptr[i + 1]; ++p)
                    if (Mp.colind[p] == i) {
                        double sdiag = 0.0;
                        for (int t = L.c_ptr[p]; t < L.c_ptr[p + 1]; ++t) sdiag += L.c_val[t];
                        dM1[i] = sdiag;
                    }
            HostCsr K1 = schur_host(B, Bt, dM1, nullptr);
            if (l == 0) anisotropy = csr_anisotropy(K1);
            if (o.mg_coarsening == 1 || anisotropy > 10.0) {
                if ((int)chains.size() < n_mc) chains.resize(n_mc);
                chains[l] = build_chain(schur[l], K1, o, st);
            }
            clk.lap("algebraic chain", l);
        }
        if (l + 1 < nlevels) {
            Pl[l] = csr_from_c(L.P, true, "darcy P");
            PMC_REQUIRE(Pl[l].nrows == L.n_p && Pl[l].ncols == in[l + 1].n_p, "darcy P: wrong shape");
        }
        PMC_HIP(hipStreamSynchronize(st));
    }

    if (hybrid) {
        hyb.resize(n_mc);
        for (int l = 0; l < n_mc; ++l) {
            build_hybrid(l, in[l]);
            clk.lap("hybridized system", l);
        }
    }
    clk.lap("levels done", -1);
    // Level patterns: S_l pattern = own pattern U Galerkin image of level l-1's pattern.
    std::vector<HostCsr> pattern(nlevels);
    pattern[0] = schur[0].pat;
    std::vector<Symbolic> galerkin(nlevels);   // galerkin[l+1]: lists over CSR nnz of pattern[l]
    for (int l = 0; l + 1 < nlevels; ++l) {
        const HostCsr& S = pattern[l];
        const HostCsr& P = Pl[l];
        std::vector<Triple> tr;
        for (int e = 0; e < S.nrows; ++e)
            for (int p = S.rowptr[e]; p < S.rowptr[e + 1]; ++p) {
                const int e2 = S.colind[p];
                for (int a = P.rowptr[e]; a < P.rowptr[e + 1]; ++a)
                    for (int b = P.rowptr[e2]; b < P.rowptr[e2 + 1]; ++b)
                        tr.push_back({P.colind[a], P.colind[b], p, 0.5 * P.vals[a] * P.vals[b]});
            }
        const HostCsr& own = schur[l + 1].pat;
        for (int e = 0; e < own.nrows; ++e)
            for (int p = own.rowptr[e]; p < own.rowptr[e + 1]; ++p) tr.push_back({e, own.colind[p], -1, 0.0});
        galerkin[l + 1] = build_symbolic(own.nrows, tr);
        pattern[l + 1] = galerkin[l + 1].pat;
    }

    for (int l = 0; l < nlevels; ++l) {
        DarcyLevel& d = lv[l];
        MgLevel& m = mg.L[l];
        m.n = d.n_p;
        m.bv = true;
        m.f32 = o.precond_storage != PMC_STORAGE_FP64;
        m.lmax = 2.0 * 1.0001;   // weakly diagonally dominant M-matrix: spec(D^-1 S) in (0, 2]
        sell_build(m.S, pattern[l], false, true, st);   // per-realization values: sized by Darcy::ensure for the width in use
        // own Schur lists mapped onto the (possibly larger) level pattern
        {
            // map own CSR nnz -> level-pattern CSR nnz by (row, col) search
            const HostCsr& own = schur[l].pat;
            const HostCsr& pat = pattern[l];
            Symbolic onpat;
            onpat.ptr.assign(pat.nnz() + 1, 0);
            std::vector<int> own_of(pat.nnz(), -1);
            for (int e = 0; e < own.nrows; ++e)
                for (int p = own.rowptr[e]; p < own.rowptr[e + 1]; ++p) {
                    auto b = pat.colind.begin() + pat.rowptr[e], en = pat.colind.begin() + pat.rowptr[e + 1];
                    auto it = std::lower_bound(b, en, own.colind[p]);
                    PMC_REQUIRE(it != en && *it == own.colind[p], "darcy: Schur pattern mismatch");
                    own_of[it - pat.colind.begin()] = p;
                }
            for (int64_t q = 0; q < pat.nnz(); ++q) {
                const int p = own_of[q];
                if (p >= 0)
                    for (int t = schur[l].ptr[p]; t < schur[l].ptr[p + 1]; ++t) {
                        onpat.idx.push_back(schur[l].idx[t]);
                        onpat.w.push_back(schur[l].w[t]);
                    }
                onpat.ptr[q + 1] = (int)onpat.idx.size();
            }
            std::vector<int> ptr, idx;
            std::vector<double> w;
            lists_to_slots(m.S, onpat, ptr, idx, w);
            d.s_ptr.upload(ptr, st);
            d.s_idx.upload(idx, st);
            d.s_w.upload(w, st);
        }
        // diagonal slots
        d.s_diag_slot.upload(diag_slots(m.S, pattern[l]), st);
        if (l + 1 < nlevels) {
            HostCsr Pt = csr_transpose(Pl[l]);
            sell_build(m.P, Pl[l], true, false, st);
            sell_build(m.Pt, Pt, true, false, st);
            m.p_oct = csr_is_oct_injection(Pl[l]);
        }
        PMC_HIP(hipStreamSynchronize(st));
    }
    // Galerkin lists: indices are CSR nnz of the finer pattern -> convert to the finer level's SELL slots
    for (int l = 0; l + 1 < nlevels; ++l) {
        std::vector<int> n2s = nnz_to_slot(mg.L[l].S, pattern[l].nnz());
        Symbolic g = galerkin[l + 1];
        for (int& v : g.idx) v = n2s[v];
        std::vector<int> ptr, idx;
        std::vector<double> w;
        lists_to_slots(mg.L[l + 1].S, g, ptr, idx, w);
        lv[l + 1].g_ptr.upload(ptr, st);
        lv[l + 1].g_idx.upload(idx, st);
        lv[l + 1].g_w.upload(w, st);
        PMC_HIP(hipStreamSynchronize(st));
    }
    clk.lap("geometric Galerkin lists", -1);
    mg.enable_bv_tail();
    mg.build_tails(st);
    for (int l = 0; l < nlevels; ++l) {   // host mirrors no longer needed
        mg.L[l].S.h_src.clear(); mg.L[l].S.h_src.shrink_to_fit();
        mg.L[l].S.h_cols.clear(); mg.L[l].S.h_cols.shrink_to_fit();
        lv[l].M.h_src.clear(); lv[l].M.h_src.shrink_to_fit();
        lv[l].M.h_cols.clear(); lv[l].M.h_cols.shrink_to_fit();
    }
}

double Darcy::operator_bytes(int level, int nb) const {
    // ALGORITHMIC bytes of one launch of the u-rows [M(k) | B^T] x with the fused <x, Ax>, every operand
    // once.  Element-grouped M(k): 12 B per stored (column, shared element entry) slot + the two coefficient rows per dof
    // (8 B), the coefficient table once ((n_p + 1) x nb doubles); otherwise 4 B per column index + 8 nb B per value.  B^T: 12 B
    // per nonzero.  4 B per row of slice offsets.  Vectors: x_u and x_p read, y_u written (the dot takes x_u from the same read).
    const DarcyLevel& d = lv[level];
    // the input is a preconditioned vector (fp32 or fp64 storage), the result fp64
    const double V = 8.0 * nb, Z = (opts.precond_storage == PMC_STORAGE_FP64 ? 8.0 : 4.0) * nb;
    double b = 12.0 * (double)d.Bt.nnz + 4.0 * d.n_u + Z * ((double)d.n_u + d.n_p) + V * d.n_u;
    if (use_eg(d)) b += 12.0 * (double)d.Meg.nslots + 8.0 * d.n_u + V * (d.n_p + 1.0);
    else b += (4.0 + V) * (double)d.M.nnz;
    return b;
}

double Darcy::poly_bytes(int level, int nb) const {
    // one launch of z_u = D^-1 (c0 r - c1 M(k) D^-1 r) on the element-grouped matrix: 12 B per stored slot, the two
    // coefficient rows per dof (8 B), 4 B per row, the coefficient table once; vectors: r and the per-realization l1
    // diagonal are gathered (each row once algorithmically), z is written in its storage
    const DarcyLevel& d = lv[level];
    const double V = 8.0 * nb, Z = (opts.precond_storage == PMC_STORAGE_FP64 ? 8.0 : 4.0) * nb;
    if (!use_eg(d)) return (4.0 + V) * (double)d.M.nnz + 4.0 * d.n_u + (2.0 * V + Z) * d.n_u;
    return 12.0 * (double)d.Meg.nslots + 12.0 * d.n_u + V * (d.n_p + 1.0) + (2.0 * V + Z) * d.n_u;
}

bool Darcy::use_eg(const DarcyLevel& d) const {
    static const bool off = lab_env("PMC_DARCY_NO_EG") != nullptr;   // laboratory A/B switch: materialise M(k)
    return d.has_eg && (opts.cheb_degree_M == 2 || opts.cheb_degree_M == 0) && !off;   // 0 (automatic) = 2 here: the element-grouped form is degree 2
}

void Darcy::ensure(int level, int nb) {
    DarcyLevel& d = lv[level];
    const size_t n = (size_t)d.n_u + d.n_p;
    // per-realization values of the Schur hierarchy this solve walks, at the width it runs with (a large level is solved
    // 16 at a time: it never pays for the 256 columns of the small ones)
    auto size_values = [nb, this](Multigrid& g, size_t first) {
        g.ensure_bv_tail_width(ctx.stream, nb);
        for (size_t l = first; l < g.L.size(); ++l) {
            MgLevel& m = g.L[l];
            m.vals_bv.ensure((size_t)m.S.nslots * nb);
            if (m.f32) {
                m.vals32.ensure((size_t)m.S.nslots * nb);
                m.scaled32.ensure((size_t)m.S.nslots * nb);
            } else {
                m.vals_scaled.ensure((size_t)m.S.nslots * nb);
            }
            m.dinv.ensure((size_t)m.n * nb);
        }
    };
    if (level < (int)chains.size() && chains[level]) size_values(chains[level]->mg, 0);
    else size_values(mg, (size_t)level);
    d.coef.ensure((size_t)(d.n_p + 1) * nb);     // + the constant-one row of the element-grouped layout
    if (!use_eg(d)) {
        d.mvals.ensure((size_t)d.M.nslots * nb);
        d.mvals_scaled.ensure((size_t)d.M.nslots * nb);
    }
    sol_compact.ensure((size_t)std::max(d.n_obs, 1) * nb);
    d.diagM.ensure((size_t)d.n_u * nb);
    d.l1invM.ensure((size_t)d.n_u * nb);
    d.rhs_bc.ensure(n * nb);
    sol.ensure(n * nb);
    cx.ensure((size_t)d.n_u * nb);
    cd.ensure((size_t)d.n_u * nb);
    stage_k.ensure((size_t)d.n_p * nb);
    stage_sol.ensure(n * nb);
    qpartial.ensure((size_t)dot_capacity((int)n, nb) * nb);
    qout.ensure(kMaxBatch);
}

void Darcy::set_observations(int level, const pmc_csr* Gc) {
    PMC_REQUIRE(level >= 0 && level < n_mc && Gc != nullptr, "set_observations: bad arguments");
    DarcyLevel& d = lv[level];
    HostCsr G = csr_from_c(*Gc, true, "observation functionals");
    PMC_REQUIRE(G.ncols == d.n_p && G.nrows >= 1, "set_observations: Gobs must be nobs x n_p");
    ctx.activate();
    hipStream_t st = ctx.stream;
    // union of supp(obs) and the pressure dofs any g_i touches, as sorted row indices of the full solution vector
    std::vector<double> obs_h((size_t)d.n_u + d.n_p);
    PMC_HIP(hipMemcpyAsync(obs_h.data(), d.obs.p, sizeof(double) * obs_h.size(), hipMemcpyDeviceToHost, st));
    PMC_HIP(hipStreamSynchronize(st));
    std::vector<char> mark(obs_h.size(), 0);
    for (size_t i = 0; i < obs_h.size(); ++i) mark[i] = obs_h[i] != 0.0;
    for (int c : G.colind) mark[(size_t)d.n_u + c] = 1;
    std::vector<int> rows, pos(obs_h.size(), -1);
    std::vector<double> w;
    for (size_t i = 0; i < mark.size(); ++i)
        if (mark[i]) { pos[i] = (int)rows.size(); rows.push_back((int)i); w.push_back(obs_h[i]); }
    std::vector<double> norm(G.nrows);
    for (int i = 0; i < G.nrows; ++i) {
        double s = 0.0;
        for (int p = G.rowptr[i]; p < G.rowptr[i + 1]; ++p) s += G.vals[p];
        PMC_REQUIRE(s != 0.0, "set_observations: an observation functional sums to zero");
        norm[i] = 1.0 / s;                      // G_i = <g_i, p> / sum(g_i)
    }
    HostCsr Gcmp = G;
    Gcmp.ncols = (int)rows.size();
    for (int& c : Gcmp.colind) c = pos[(size_t)d.n_u + c];
    sell_build(d.Gobs, Gcmp, true, false, st);
    d.g_rows.upload(rows, st);
    d.g_obs_w.upload(w, st);
    d.g_norm.upload(norm, st);
    d.n_gobs = G.nrows;
    d.n_grows = (int)rows.size();
    PMC_HIP(hipStreamSynchronize(st));
}

void Darcy::compute_G(int level, int nbatch, const double* kf, double* G, double* C, double* Q, int memspace,
                      pmc_stats* stats) {
    PMC_REQUIRE(level >= 0 && level < n_mc, "ComputeG: level out of range");
    PMC_REQUIRE(nbatch >= 1 && kf != nullptr && G != nullptr, "ComputeG: bad arguments");
    DarcyLevel& d = lv[level];
    PMC_REQUIRE(d.n_gobs > 0, "ComputeG: no observation functionals set on this level");
    ctx.activate();
    hipStream_t st = ctx.stream;
    std::vector<double> q(kMaxBatch);
    int done = 0;
    while (done < nbatch) {
        int nb = batch_width((size_t)d.n_u + d.n_p, true, ctx.device);
        while (nb > nbatch - done) nb >>= 1;
        const double* k_d = kf + (size_t)done * d.n_p;
        if (memspace == PMC_MEM_HOST) {
            ensure(level, nb);
            PMC_HIP(hipMemcpyAsync(stage_k.p, k_d, sizeof(double) * d.n_p * nb, hipMemcpyHostToDevice, st));
            k_d = stage_k.p;
        }
        solve_chunk(level, nb, k_d, q.data(), nullptr, stats ? stats + done : nullptr, 0, 0, G + (size_t)done * d.n_gobs);
        for (int b = 0; b < nb; ++b) {
            if (Q) Q[done + b] = q[b];
            if (C) C[done + b] = (double)((size_t)d.n_u + d.n_p);
        }
        done += nb;
    }
}

void Darcy::solve_chunk(int level, int nb, const double* k_d, double* Q_host, double* sol_d, pmc_stats* stats, int row0,
                        int nrows, double* G_host) {
    hipStream_t st = ctx.stream;
    DarcyLevel& d = lv[level];
    const int n_u = d.n_u, n_p = d.n_p, n = n_u + n_p;
    ensure(level, nb);
    if (stats) ctx.phase_mark(0);
    // K12/K13: M(k), elimination, rhs_bc
    const bool eg = use_eg(d);
    k::darcy_coef(st, nb, n_p, k_d, k_divides, d.coef.p);
    if (eg) k::fill(st, (size_t)nb, d.coef.p + (size_t)n_p * nb, 1.0);
    SellView Mpat = view(d.M);
    k::darcy_assemble(st, nb, Mpat, d.slot_src.p, d.c_ptr.p, d.c_elem.p, d.c_val.p, d.coef.p, d.ess.p, d.ess_data.p,
                      d.rhs_u0.p, eg ? nullptr : d.mvals.p, d.diagM.p, d.l1invM.p, d.rhs_bc.p);
    k::broadcast(st, nb, n_p, d.rhs_p.p, d.rhs_bc.p + (size_t)n_u * nb);
    // K14: Schur complement values on the level hierarchy
    // what the V-cycle kernels read of a level: S D^-1 (one-pass smoothers) and S itself - in fp32 when the level says so
    auto scaled_copies = [&](MgLevel& m) {
        if (m.f32) k::scale_cols_bv32(st, nb, m.S.nslots, m.S.cols.p, m.vals_bv.p, m.dinv.p, m.scaled32.p, m.vals32.p);
        else k::scale_cols_bv(st, nb, m.S.nslots, m.S.cols.p, m.vals_bv.p, m.dinv.p, m.vals_scaled.p);
    };
    DarcyChain* chain = (level < (int)chains.size()) ? chains[level].get() : nullptr;
    if (!eg) k::scale_cols_bv(st, nb, d.M.nslots, d.M.cols.p, d.mvals.p, d.l1invM.p, d.mvals_scaled.p);
    if (chain) {
        gwork.ensure(kMaxBatch);
        for (size_t j = 0; j < chain->cl.size(); ++j) {
            MgLevel& m = chain->mg.L[j];
            DarcyChainLevel& c = chain->cl[j];
            const double* src = (j == 0) ? d.diagM.p : chain->mg.L[j - 1].vals_bv.p;
            k::refresh(st, nb, m.S.nslots, c.ptr.p, c.idx.p, c.w.p, src, j == 0, m.vals_bv.p);
            k::diag_inv(st, nb, m.n, c.diag_slot.p, m.vals_bv.p, m.dinv.p);
            k::gersh_scale_bv(st, nb, view_bv(m.S, m.vals_bv.p), m.dinv.p, gwork.p);
            scaled_copies(m);
        }
        chain->mg.refresh_bv_tail(st, nb);
    } else {
        MgLevel& m = mg.L[level];
        k::refresh(st, nb, m.S.nslots, d.s_ptr.p, d.s_idx.p, d.s_w.p, d.diagM.p, true, m.vals_bv.p);
        k::diag_inv(st, nb, m.n, d.s_diag_slot.p, m.vals_bv.p, m.dinv.p);
        scaled_copies(m);
        for (int l = level; l + 1 < nlevels; ++l) {
            MgLevel& f = mg.L[l];
            MgLevel& c = mg.L[l + 1];
            DarcyLevel& dc = lv[l + 1];
            k::refresh(st, nb, c.S.nslots, dc.g_ptr.p, dc.g_idx.p, dc.g_w.p, f.vals_bv.p, false, c.vals_bv.p);
            k::diag_inv(st, nb, c.n, dc.s_diag_slot.p, c.vals_bv.p, c.dinv.p);
            scaled_copies(c);
        }
        mg.refresh_bv_tail(st, nb, level);
    }
    if (stats) ctx.phase_mark(1);   // "Darcy: Build Solver" ends here: everything below is the solve
    // operator [M(k) Bt; B 0] and block-diagonal preconditioner
    const SellView Mv = view_bv(d.M, d.mvals.p);
    const SellView Bv = view(d.B), Btv = view(d.Bt);
    const EgView Mg{d.n_u, d.Meg.nslices, d.eg_gw, d.Meg.cols.p, d.Meg.vals.p, d.eg_e12.p};
    const double* coefp = d.coef.p;
    LinOp A;
    A.n = n;
    A.n0 = n_u;
    // in-loop event brackets are not recorded while the MINRES iterations may be captured into a hipGraph (as the sampler's)
    const bool timing_ok = opts.use_graph == 0;
    OpTimer* tm = &op_timer;
    OpTimer* tp = &poly_timer;
    // (the Darcy solves start from zero: only the product from a preconditioned vector is ever needed)
    A.apply_z = [=](const Lanes& L, int nb_, zvec x, double* y, double* partial, double* partial2) {
        // u-rows: M(k) x_u + B^T x_p in one pass; p-rows: B x_u (beside it on the second stream); <x, Ax> fused into both
        const zvec xp = x + (size_t)n_u * nb_;
        const bool timed = timing_ok && tm->on && partial != nullptr;      // the in-loop launches (fused dot) only
        if (timed) tm->begin(L.main);                          // timed: the p-rows follow on the same stream, not beside it
        else L.fork();
        const int nu_blk = eg ? k::eg_pair_spmm_z(L.main, nb_, Mg, coefp, x, Btv, xp, y, partial, x)
                              : k::pair_spmm_z(L.main, nb_, Mv, x, Btv, xp, y, partial, x);
        if (timed) tm->end(L.main);
        const int np_blk = k::spmm_z(timed ? L.main : L.side(), nb_, Bv, x, y + (size_t)n_u * nb_, partial2, xp);
        if (!timed) L.join();
        return k::DotParts{partial, nu_blk, partial2, np_blk};
    };
    ChebParams cpM{opts.cheb_degree_M > 0 ? opts.cheb_degree_M : 2, 1.0, d.ratio_M, d.mvals_scaled.p};
    const double* l1 = d.l1invM.p;
    if (!eg && !cheb_fused(cpM, true)) cx2.ensure((size_t)n_u * nb);
    double* cxp = cx.p;
    double* cx2p = cx2.p;
    double* cdp = cd.p;
    Multigrid* mgp = chain ? &chain->mg : &mg;
    const int mg_l0 = chain ? 0 : level;
    PrecFn prec = [=](const Lanes& L, int nb_, const double* r, zvec z, double* dot_partial, double* dot_partial2) {
        // independent diagonal blocks: V-cycle of the S-block on the main stream, the M-block polynomial on the second
        // stream beside the V-cycle's coarse levels (see the sampler's preconditioner)
        int nblk_u = 0;
        auto m_block = [&]() {
            if (eg && tp->on && timing_ok) {
                // timed: on the main stream, bracketed by events (an empty bracket behind it), not beside the tail
                double c0, c1;
                cheb2_coefficients(cpM.lmax, cpM.ratio, &c0, &c1);
                tp->begin(L.main);
                nblk_u = k::eg_poly2_z(L.main, nb_, Mg, coefp, l1, r, z, c0, c1, dot_partial2);
                tp->end(L.main);
                return;
            }
            L.fork();
            if (eg) {
                double c0, c1;
                cheb2_coefficients(cpM.lmax, cpM.ratio, &c0, &c1);
                nblk_u = k::eg_poly2_z(L.side(), nb_, Mg, coefp, l1, r, z, c0, c1, dot_partial2);
            } else {
                nblk_u = cheb_apply_z(L.side(), nb_, Mv, l1, true, cpM, r, z, cxp, cx2p, cdp, dot_partial2);
            }
        };
        const int nblk_s = mgp->vcycle_z(L.main, nb_, mg_l0, r + (size_t)n_u * nb_, z + (size_t)n_u * nb_, dot_partial, m_block);
        L.join();
        return k::DotParts{dot_partial, nblk_s, dot_partial2, nblk_u};
    };
    // SolveFwd only needs Q = <obs, sol>: unless the solution itself is requested, MINRES maintains just the rows in
    // the support of obs (compact w / x vectors)
    const bool gmode = G_host != nullptr;
    const bool compact = gmode || ((sol_d == nullptr) && d.n_obs > 0 && d.n_obs < n);
    const int ncomp = gmode ? d.n_grows : d.n_obs;
    const int* comp_rows = gmode ? d.g_rows.p : d.obs_rows.p;
    const double* comp_w = gmode ? d.g_obs_w.p : d.obs_w.p;
    sol_compact.ensure((size_t)std::max(ncomp, 1) * nb);
    GraphHint hint;
    hint.key = hash_mix(hash_mix(hash_mix(0xda, (uint64_t)level + 1), (uint64_t)nb), gmode ? 3 : (compact ? 1 : 2));
    hint.sig = mgp->signature(mg_l0);
    for (const void* p : {(const void*)cx.p, (const void*)cd.p, (const void*)cx2.p, (const void*)d.mvals.p, (const void*)d.mvals_scaled.p,
                          (const void*)d.l1invM.p, (const void*)d.rhs_bc.p})
        hint.sig = hash_ptr(hint.sig, p);
    MinresResult res = compact ? minres_solve(ctx, nb, A, prec, d.rhs_bc.p, sol_compact.p, true, opts, work, 0, ncomp,
                                              comp_rows, hint)
                               : minres_solve(ctx, nb, A, prec, d.rhs_bc.p, sol.p, true, opts, work, 0, n, nullptr, hint);
    if (stats) {
        ctx.phase_mark(2);
        for (int kcol = 0; kcol < nb; ++kcol) stats[kcol] = res.col[kcol];
        ctx.phase_report(stats, nb);
    }
    if (op_timer.on) op_timer.harvest();   // minres_solve has synchronised the stream
    if (poly_timer.on) poly_timer.harvest();
    // K15: Q = <obs, sol>
    const int qblocks = compact ? k::wdot(st, nb, ncomp, comp_w, sol_compact.p, qpartial.p)
                                : k::wdot(st, nb, n, d.obs.p, sol.p, qpartial.p);
    k::reduce_final(st, nb, qblocks, qpartial.p, qout.p);
    PMC_HIP(hipMemcpyAsync(ctx.h_scal, qout.p, sizeof(double) * nb, hipMemcpyDeviceToHost, st));
    if (sol_d) k::deinterleave(st, nb, nrows, sol.p + (size_t)row0 * nb, nullptr, nullptr, false, sol_d);
    if (gmode) {
        // G_i = <g_i, p> / sum(g_i) for every realization (BayesianInverseProblem::ComputeG), on the compact solution
        gtmp.ensure((size_t)d.n_gobs * nb);
        gout.ensure((size_t)d.n_gobs * nb);
        k::spmm(st, nb, view(d.Gobs), sol_compact.p, gtmp.p, false, nullptr, nullptr);
        k::deinterleave(st, nb, d.n_gobs, gtmp.p, nullptr, d.g_norm.p, false, gout.p);
        PMC_HIP(hipMemcpyAsync(G_host, gout.p, sizeof(double) * d.n_gobs * nb, hipMemcpyDeviceToHost, st));
    }
    PMC_HIP(hipStreamSynchronize(st));
    for (int kcol = 0; kcol < nb; ++kcol) Q_host[kcol] = ctx.h_scal[kcol];
}

// Setup of the hybridized form of one level (see DarcyHybrid; numpy twin: parelagmc_amd/fe/darcy_hybrid.py).
void Darcy::build_hybrid(int level, const pmc_darcy_level& L) {
    hipStream_t st = ctx.stream;
    const int nu = L.n_u, np = L.n_p;
    HostCsr Mp = csr_from_c(L.M_pattern, false, "darcy M_pattern");
    HostCsr B = csr_from_c(L.B, true, "darcy B");
    csr_sort_rows(B);
    ElementInverses inv = element_inverses(Mp, B, L.c_ptr, L.c_elem, L.c_val, nullptr, "pmc_darcy_create_hybrid");
    const int m = inv.m;
    // face -> its (one or two) elements in element order; multipliers: interior faces and essential boundary faces
    std::vector<int> fe((size_t)nu * 2, -1), fl((size_t)nu * 2, -1), cnt(nu, 0);
    for (int e = 0; e < np; ++e)
        for (int p = B.rowptr[e]; p < B.rowptr[e + 1]; ++p) {
            const int f = B.colind[p];
            PMC_REQUIRE(cnt[f] < 2, "pmc_darcy_create_hybrid: a face belongs to more than two elements");
            fe[2 * (size_t)f + cnt[f]] = e;
            fl[2 * (size_t)f + cnt[f]] = p - B.rowptr[e];
            ++cnt[f];
        }
    std::vector<int> lam_of(nu, -1);
    int nl = 0;
    for (int f = 0; f < nu; ++f) {
        PMC_REQUIRE(cnt[f] >= 1, "pmc_darcy_create_hybrid: a face belongs to no element");
        PMC_REQUIRE(!(cnt[f] == 2 && L.ess_mask[f]), "pmc_darcy_create_hybrid: an essential dof on an interior face");
        if (cnt[f] == 2 || L.ess_mask[f]) lam_of[f] = nl++;
    }
    PMC_REQUIRE(nl > 0, "pmc_darcy_create_hybrid: no multipliers (every face carries a pressure condition)");
    auto hy = std::make_unique<DarcyHybrid>();
    hy->n_lambda = nl;
    // signed element data: sgn = C_e, fs = C_e f_e with rhs_u of a face given to its first element
    auto sgn = [&](int e, int q) { return B.vals[B.rowptr[e] + q] > 0.0 ? 1.0 : -1.0; };
    auto fs = [&](int e, int q) {
        const int f = B.colind[B.rowptr[e] + q];
        return fe[2 * (size_t)f] == e ? sgn(e, q) * L.rhs[f] : 0.0;
    };
    // H(kappa) as lists over the elements, R, b0, back-substitution operators
    std::vector<Triple> tr;
    tr.reserve((size_t)np * m * m);
    HostCsr R, UL, PL;
    R.nrows = nl; R.ncols = np;
    UL.nrows = nu; UL.ncols = nl;
    PL.nrows = np; PL.ncols = nl;
    std::vector<std::vector<std::pair<int, double>>> Rrows(nl);
    std::vector<double> b0(nl, 0.0), U0(nu, 0.0), ug(nu, 0.0), P0(np, 0.0), zg(np, 0.0);
    std::vector<int> owner(nu, 0);
    UL.rowptr.assign(nu + 1, 0);
    PL.rowptr.assign(np + 1, 0);
    std::vector<std::vector<std::pair<int, double>>> ULrows(nu);
    for (int e = 0; e < np; ++e) {
        const int b0e = B.rowptr[e], len = B.rowptr[e + 1] - b0e;
        const double g = L.rhs[nu + e];
        const double* Xe = &inv.X[(size_t)e * m * m];
        const double* Ye = &inv.Y[(size_t)e * m];
        zg[e] = -inv.z[e] * g;
        for (int p = 0; p < len; ++p) {
            const int f = B.colind[b0e + p];
            double xf = 0.0;                          // sum_q Xs[p][q] fs[q] = c_p (X_e f_e)[p]
            for (int q = 0; q < len; ++q) xf += Xe[p * m + q] * fs(e, q);
            P0[e] += Ye[p] * fs(e, p);
            const int lp = lam_of[f];
            if (lp >= 0) {
                for (int q = 0; q < len; ++q) {
                    const int lq = lam_of[B.colind[b0e + q]];
                    if (lq >= 0) tr.push_back({lp, lq, e, Xe[p * m + q]});
                }
                if (xf != 0.0) Rrows[lp].push_back({e, xf});
                b0[lp] += Ye[p] * g;
                PL.colind.push_back(lp);
                PL.vals.push_back(Ye[p]);
            }
            if (fe[2 * (size_t)f] == e) {             // this element reports the face's flux
                owner[f] = e;
                U0[f] = sgn(e, p) * xf;
                ug[f] = sgn(e, p) * Ye[p] * g;
                for (int q = 0; q < len; ++q) {
                    const int lq = lam_of[B.colind[b0e + q]];
                    if (lq >= 0) ULrows[f].push_back({lq, sgn(e, p) * Xe[p * m + q]});
                }
                if (lp >= 0 && cnt[f] == 1) b0[lp] -= sgn(e, p) * L.ess_data[f];   // essential face: c u = c d
            }
        }
        PL.rowptr[e + 1] = (int)PL.colind.size();
    }
    auto rows_to_csr = [](HostCsr& A, std::vector<std::vector<std::pair<int, double>>>& rows) {
        A.rowptr.assign(rows.size() + 1, 0);
        for (size_t i = 0; i < rows.size(); ++i) {
            std::sort(rows[i].begin(), rows[i].end());
            for (auto& cv : rows[i]) { A.colind.push_back(cv.first); A.vals.push_back(cv.second); }
            A.rowptr[i + 1] = (int)A.colind.size();
        }
    };
    rows_to_csr(R, Rrows);
    rows_to_csr(UL, ULrows);
    csr_sort_rows(PL);
    Symbolic own = build_symbolic(nl, tr);
    HostCsr K1 = own.pat;
    for (int64_t p = 0; p < K1.nnz(); ++p) {
        double v = 0.0;
        for (int t = own.ptr[p]; t < own.ptr[p + 1]; ++t) v += own.w[t];
        K1.vals[p] = v;
    }
    {   // element-grouped layout of H(kappa): per row two groups of m slots (its one or two elements), shared entries
        const int gw = m;
        PMC_REQUIRE(gw <= 16, "pmc_darcy_create_hybrid: more than 16 faces per element");
        HostCsr Heg;
        Heg.nrows = Heg.ncols = nl;
        Heg.rowptr.resize(nl + 1);
        Heg.colind.assign((size_t)nl * 2 * gw, 0);
        Heg.vals.assign((size_t)nl * 2 * gw, 0.0);
        std::vector<int> e12((size_t)nl * 2, np);        // np: the constant-one coefficient row (padding groups, weight 0)
        for (int i = 0; i <= nl; ++i) Heg.rowptr[i] = i * 2 * gw;
        for (int f = 0; f < nu; ++f) {
            const int lp = lam_of[f];
            if (lp < 0) continue;
            for (int q = 0; q < 2 * gw; ++q) Heg.colind[(size_t)lp * 2 * gw + q] = lp;
            for (int sidx = 0; sidx < cnt[f]; ++sidx) {
                const int e = fe[2 * (size_t)f + sidx], p = fl[2 * (size_t)f + sidx];
                const int b0e = B.rowptr[e], len = B.rowptr[e + 1] - b0e;
                e12[2 * (size_t)lp + sidx] = e;
                for (int q = 0; q < len; ++q) {
                    const int lq = lam_of[B.colind[b0e + q]];
                    if (lq < 0) continue;
                    Heg.colind[(size_t)lp * 2 * gw + sidx * gw + q] = lq;
                    Heg.vals[(size_t)lp * 2 * gw + sidx * gw + q] = inv.X[((size_t)e * m + p) * m + q];
                }
            }
        }
        sell_build(hy->Heg, Heg, true, false, st);
        hy->eg_e12.upload(e12, st);
        hy->eg_gw = gw;
        hy->no_rows.upload(std::vector<int>((size_t)hy->Heg.nslices + 1, 0), st);
        HostCsr I;
        I.nrows = I.ncols = nl;
        I.rowptr.resize(nl + 1);
        I.colind.resize(nl);
        I.vals.assign(nl, 1.0);
        for (int i = 0; i <= nl; ++i) I.rowptr[i] = i;
        for (int i = 0; i < nl; ++i) I.colind[i] = i;
        sell_build(hy->ident, I, true, false, st);
    }
    double scale = 0.5;
    if (const char* e = lab_env("PMC_DARCY_HYB_SCALE")) scale = atof(e);
    hy->chain = build_chain(own, K1, opts, st, /*plain=*/true, scale, /*ratio_scale=*/2.0);
    sell_build(hy->R, R, true, false, st);
    sell_build(hy->UL, UL, true, false, st);
    sell_build(hy->PL, PL, true, false, st);
    hy->b0.upload(b0, st);
    hy->U0.upload(U0, st);
    hy->ug.upload(ug, st);
    hy->P0.upload(P0, st);
    hy->zg.upload(zg, st);
    hy->owner.upload(owner, st);
    PMC_HIP(hipStreamSynchronize(st));
    hyb[level] = std::move(hy);
}

// SolveFwd through the hybridized form: H(kappa) lambda = R kappa + b_0 by MINRES with one V-cycle of the per-realization
// aggregation hierarchy, then the element-local back-substitution into the full solution vector (what Q, the returned
// solution and the pressure block are taken from, exactly as after the saddle-point solve).
void Darcy::solve_chunk_hybrid(int level, int nb, const double* k_d, double* Q_host, double* sol_d, pmc_stats* stats, int row0,
                               int nrows) {
    hipStream_t st = ctx.stream;
    DarcyLevel& d = lv[level];
    DarcyHybrid& hy = *hyb[level];
    const int n_u = d.n_u, n_p = d.n_p, n = n_u + n_p, nl = hy.n_lambda;
    DarcyChain& ch = *hy.chain;
    // sizes at this launch width
    ch.mg.ensure_bv_tail_width(st, nb);
    for (MgLevel& m : ch.mg.L) {
        m.vals_bv.ensure((size_t)m.S.nslots * nb);
        if (m.f32) {
            m.vals32.ensure((size_t)m.S.nslots * nb);
            m.scaled32.ensure((size_t)m.S.nslots * nb);
        } else {
            m.vals_scaled.ensure((size_t)m.S.nslots * nb);
        }
        m.dinv.ensure((size_t)m.n * nb);
    }
    hy.coef.ensure((size_t)(n_p + 1) * nb);     // + the constant-one row of the element-grouped layout
    hy.rhs.ensure((size_t)nl * nb);
    hy.lam.ensure((size_t)nl * nb);
    hy.tu.ensure((size_t)n_u * nb);
    hy.tp.ensure((size_t)n_p * nb);
    sol.ensure((size_t)n * nb);
    qpartial.ensure((size_t)dot_capacity(n, nb) * nb);
    qout.ensure(kMaxBatch);
    gwork.ensure(kMaxBatch);
    static const bool eg_off = lab_env("PMC_DARCY_HYB_NO_EG") != nullptr;   // laboratory A/B: generic V-cycle on explicit values
    const bool eg_cycle = ch.mg.L.size() >= 2 && ch.mg.smooth_degree == 2 && !eg_off;
    if (eg_cycle) {
        hy.negcoef.ensure((size_t)(n_p + 1) * nb);
        hy.vx.ensure((size_t)nl * nb);
        hy.vres.ensure((size_t)nl * nb);
        hy.vd.ensure((size_t)nl * nb);
        hy.vxc.ensure((size_t)ch.mg.L[1].n * nb);
        ch.mg.L[1].ensure(nb);
    }
    if (stats) ctx.phase_mark(0);
    // kappa = 1 / c(k); operators of the hierarchy; right-hand side
    k::darcy_coef(st, nb, n_p, k_d, !k_divides, hy.coef.p);
    k::fill(st, (size_t)nb, hy.coef.p + (size_t)n_p * nb, 1.0);
    if (eg_cycle) k::scale(st, (size_t)(n_p + 1) * nb, hy.coef.p, -1.0, hy.negcoef.p);
    for (size_t j = 0; j < ch.cl.size(); ++j) {
        MgLevel& m = ch.mg.L[j];
        DarcyChainLevel& c = ch.cl[j];
        const double* src = (j == 0) ? hy.coef.p : ch.mg.L[j - 1].vals_bv.p;
        k::refresh(st, nb, m.S.nslots, c.ptr.p, c.idx.p, c.w.p, src, false, m.vals_bv.p);
        k::diag_inv(st, nb, m.n, c.diag_slot.p, m.vals_bv.p, m.dinv.p);
        k::gersh_scale_bv(st, nb, view_bv(m.S, m.vals_bv.p), m.dinv.p, gwork.p);
        if (j == 0 && eg_cycle) continue;      // the finest level is smoothed in element-grouped form: no scaled copies of it
        if (m.f32) k::scale_cols_bv32(st, nb, m.S.nslots, m.S.cols.p, m.vals_bv.p, m.dinv.p, m.scaled32.p, m.vals32.p);
        else k::scale_cols_bv(st, nb, m.S.nslots, m.S.cols.p, m.vals_bv.p, m.dinv.p, m.vals_scaled.p);
    }
    ch.mg.refresh_bv_tail(st, nb);
    k::broadcast(st, nb, nl, hy.b0.p, hy.rhs.p);
    k::spmm(st, nb, view(hy.R), hy.coef.p, hy.rhs.p, true, nullptr, nullptr);
    if (stats) ctx.phase_mark(1);
    const EgView Hg{nl, hy.Heg.nslices, hy.eg_gw, hy.Heg.cols.p, hy.Heg.vals.p, hy.eg_e12.p};
    SellView none;                    // the element-grouped operator kernel adds a second, shared-value operator: none here
    none.nrows = nl;
    none.nslices = hy.Heg.nslices;
    none.slice_off = hy.no_rows.p;
    none.ncols_hint = nl;
    const double* coefp = hy.coef.p;
    LinOp A;
    A.n = nl;
    A.n0 = 0;
    A.apply_z = [Hg, none, coefp](const Lanes& L, int nb_, zvec x, double* y, double* partial, double*) {
        return k::DotParts{partial, k::eg_pair_spmm_z(L.main, nb_, Hg, coefp, x, none, x, y, partial, x)};
    };
    Multigrid* mgp = &ch.mg;
    PrecFn prec;
    if (!eg_cycle) {
        prec = [mgp](const Lanes& L, int nb_, const double* r, zvec z, double* dot_partial, double*) {
            const int nblk = mgp->vcycle_z(L.main, nb_, 0, r, z, dot_partial);
            return k::DotParts{dot_partial, nblk, nullptr, 0};
        };
    } else {
        // V(1,1) with the finest level in element-grouped form: x = p2(H) r; res = r - H x; x += P B_1 P^T res;
        // x += p2(H)(r - H x) - every pass over H(kappa) is the operator kernel (shared element entries + two coefficient rows
        // per multiplier) or its one-pass polynomial form, never the explicit per-realization values
        double c0, c1;
        cheb2_coefficients(1.0, ch.mg.smooth_ratio, &c0, &c1);
        const SellView Iv = view(hy.ident), Ptv = view(ch.mg.L[0].Pt), Pv = view(ch.mg.L[0].P);
        const double* negc = hy.negcoef.p;
        const double* dinv0 = ch.mg.L[0].dinv.p;
        double *vx = hy.vx.p, *vres = hy.vres.p, *vd = hy.vd.p, *vxc = hy.vxc.p, *rc = ch.mg.L[1].r.p;
        prec = [=](const Lanes& L, int nb_, const double* r, zvec z, double* dot_partial, double*) {
            hipStream_t s = L.main;
            k::eg_poly2(s, nb_, Hg, coefp, dinv0, r, vx, c0, c1, nullptr);
            k::eg_pair_spmm(s, nb_, Hg, negc, vx, Iv, r, vres, nullptr, nullptr);
            k::spmm(s, nb_, Ptv, vres, rc, false, nullptr, nullptr);
            mgp->vcycle(s, nb_, 1, rc, vxc);
            k::spmm(s, nb_, Pv, vxc, vx, true, nullptr, nullptr);
            k::eg_pair_spmm(s, nb_, Hg, negc, vx, Iv, r, vres, nullptr, nullptr);
            k::eg_poly2(s, nb_, Hg, coefp, dinv0, vres, vd, c0, c1, nullptr);
            k::spmm(s, nb_, Iv, vd, vx, true, nullptr, nullptr);
            const int nblk = k::convert_z(s, nb_, nl, vx, z, r, dot_partial);
            return k::DotParts{dot_partial, dot_partial ? nblk : 0, nullptr, 0};
        };
    }
    work.want_r32 = false;
    GraphHint hint;
    hint.key = hash_mix(hash_mix(hash_mix(0xdb, (uint64_t)level + 1), (uint64_t)nb), 7);
    hint.sig = hash_ptr(hash_ptr(mgp->signature(0), hy.rhs.p), hy.lam.p);
    MinresResult res = minres_solve(ctx, nb, A, prec, hy.rhs.p, hy.lam.p, true, opts, work, 0, nl, nullptr, hint);
    if (stats) {
        ctx.phase_mark(2);
        for (int kcol = 0; kcol < nb; ++kcol) stats[kcol] = res.col[kcol];
        ctx.phase_report(stats, nb);
    }
    // element-local back-substitution
    k::spmm(st, nb, view(hy.UL), hy.lam.p, hy.tu.p, false, nullptr, nullptr);
    k::darcy_backsub_u(st, nb, n_u, hy.owner.p, hy.coef.p, hy.U0.p, hy.ug.p, hy.tu.p, sol.p);
    k::spmm(st, nb, view(hy.PL), hy.lam.p, hy.tp.p, false, nullptr, nullptr);
    k::darcy_backsub_p(st, nb, n_p, hy.coef.p, hy.P0.p, hy.zg.p, hy.tp.p, sol.p + (size_t)n_u * nb);
    // Q = <obs, sol>
    const int qblocks = k::wdot(st, nb, n, d.obs.p, sol.p, qpartial.p);
    k::reduce_final(st, nb, qblocks, qpartial.p, qout.p);
    PMC_HIP(hipMemcpyAsync(ctx.h_scal, qout.p, sizeof(double) * nb, hipMemcpyDeviceToHost, st));
    if (sol_d) k::deinterleave(st, nb, nrows, sol.p + (size_t)row0 * nb, nullptr, nullptr, false, sol_d);
    PMC_HIP(hipStreamSynchronize(st));
    for (int kcol = 0; kcol < nb; ++kcol) Q_host[kcol] = ctx.h_scal[kcol];
}

void Darcy::solve_fwd(int level, int nbatch, const double* kf, double* Q, double* C, double* sol_out, int memspace,
                      pmc_stats* stats, int sol_kind) {
    PMC_REQUIRE(level >= 0 && level < n_mc, "SolveFwd: level out of range");
    PMC_REQUIRE(nbatch >= 1 && kf != nullptr && Q != nullptr, "SolveFwd: bad arguments");
    ctx.activate();
    hipStream_t st = ctx.stream;
    DarcyLevel& d = lv[level];
    const size_t n = (size_t)d.n_u + d.n_p;
    int done = 0;
    while (done < nbatch) {
        int nb = batch_width((size_t)d.n_u + d.n_p, true, ctx.device);
        while (nb > nbatch - done) nb >>= 1;
        const double* k_d = kf + (size_t)done * d.n_p;
        const int row0 = (sol_kind == 2) ? d.n_u : 0;
        const size_t nout = (sol_kind == 2) ? (size_t)d.n_p : n;
        double* sol_d = sol_out ? sol_out + (size_t)done * nout : nullptr;
        auto chunk = [&](const double* kd, double* q, double* so, pmc_stats* stt) {
            if (hybrid) solve_chunk_hybrid(level, nb, kd, q, so, stt, row0, (int)nout);
            else solve_chunk(level, nb, kd, q, so, stt, row0, (int)nout, nullptr);
        };
        if (memspace == PMC_MEM_HOST) {
            ensure(level, nb);
            PMC_HIP(hipMemcpyAsync(stage_k.p, k_d, sizeof(double) * d.n_p * nb, hipMemcpyHostToDevice, st));
            chunk(stage_k.p, Q + done, sol_d ? stage_sol.p : nullptr, stats ? stats + done : nullptr);
            if (sol_d) {
                PMC_HIP(hipMemcpyAsync(sol_d, stage_sol.p, sizeof(double) * nout * nb, hipMemcpyDeviceToHost, st));
                PMC_HIP(hipStreamSynchronize(st));
            }
        } else {
            chunk(k_d, Q + done, sol_d, stats ? stats + done : nullptr);
        }
        if (C)
            for (int b = 0; b < nb; ++b) C[done + b] = (double)n;   // global true dofs (DarcySolver.cpp:429)
        done += nb;
    }
}

}  // namespace pmc
