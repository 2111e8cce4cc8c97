// Host-side sparse helpers (setup only, outside every timed region): CSR copies, transpose,
// block assembly and the CSR -> SELL-64 re-layout done once when a handle is created.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>

#include "common.hpp"

namespace pmc {

HostCsr csr_from_c(const pmc_csr& c, bool need_vals, const char* what) {
    PMC_REQUIRE(c.nrows >= 0 && c.ncols >= 0, std::string(what) + ": negative dimension");
    PMC_REQUIRE(c.rowptr != nullptr || c.nrows == 0, std::string(what) + ": rowptr is NULL");
    HostCsr a;
    a.nrows = c.nrows;
    a.ncols = c.ncols;
    a.rowptr.assign(c.nrows + 1, 0);
    if (c.nrows) std::copy(c.rowptr, c.rowptr + c.nrows + 1, a.rowptr.begin());
    PMC_REQUIRE(a.rowptr[0] == 0, std::string(what) + ": rowptr[0] != 0");
    for (int i = 0; i < c.nrows; ++i)
        PMC_REQUIRE(a.rowptr[i + 1] >= a.rowptr[i], std::string(what) + ": rowptr not monotone");
    const int64_t nnz = a.rowptr[c.nrows];
    PMC_REQUIRE(nnz == 0 || c.colind != nullptr, std::string(what) + ": colind is NULL");
    a.colind.assign(c.colind, c.colind + nnz);
    for (int64_t p = 0; p < nnz; ++p)
        PMC_REQUIRE(a.colind[p] >= 0 && a.colind[p] < c.ncols, std::string(what) + ": column index out of range");
    if (need_vals) {
        PMC_REQUIRE(nnz == 0 || c.vals != nullptr, std::string(what) + ": vals is NULL");
        a.vals.assign(c.vals, c.vals + nnz);
    } else if (c.vals) {
        a.vals.assign(c.vals, c.vals + nnz);
    } else {
        a.vals.assign(nnz, 1.0);
    }
    return a;
}

HostCsr csr_transpose(const HostCsr& a) {
    HostCsr t;
    t.nrows = a.ncols;
    t.ncols = a.nrows;
    t.rowptr.assign(t.nrows + 1, 0);
    for (int c : a.colind) t.rowptr[c + 1]++;
    std::partial_sum(t.rowptr.begin(), t.rowptr.end(), t.rowptr.begin());
    t.colind.resize(a.colind.size());
    t.vals.resize(a.colind.size());
    std::vector<int> next(t.rowptr.begin(), t.rowptr.end() - 1);
    for (int i = 0; i < a.nrows; ++i)
        for (int p = a.rowptr[i]; p < a.rowptr[i + 1]; ++p) {
            const int q = next[a.colind[p]]++;
            t.colind[q] = i;
            t.vals[q] = a.vals[p];
        }
    return t;
}

void csr_sort_rows(HostCsr& a) {
    std::vector<std::pair<int, double>> tmp;
    for (int i = 0; i < a.nrows; ++i) {
        const int b = a.rowptr[i], e = a.rowptr[i + 1];
        bool sorted = true;
        for (int p = b + 1; p < e; ++p)
            if (a.colind[p] < a.colind[p - 1]) { sorted = false; break; }
        if (sorted) continue;
        tmp.clear();
        for (int p = b; p < e; ++p) tmp.emplace_back(a.colind[p], a.vals[p]);
        std::sort(tmp.begin(), tmp.end(), [](const auto& x, const auto& y) { return x.first < y.first; });
        for (int p = b; p < e; ++p) {
            a.colind[p] = tmp[p - b].first;
            a.vals[p] = tmp[p - b].second;
        }
    }
}

std::vector<double> csr_diag(const HostCsr& a) {
    std::vector<double> d(a.nrows, 0.0);
    for (int i = 0; i < a.nrows; ++i)
        for (int p = a.rowptr[i]; p < a.rowptr[i + 1]; ++p)
            if (a.colind[p] == i) d[i] += a.vals[p];
    return d;
}

HostCsr csr_block2x2(const HostCsr& M, const HostCsr& Bt, const HostCsr& B, const double* d11) {
    const int nu = M.nrows, ns = B.nrows;
    PMC_REQUIRE(M.ncols == nu && Bt.nrows == nu && Bt.ncols == ns && B.ncols == nu, "block operator: shape mismatch");
    HostCsr A;
    A.nrows = A.ncols = nu + ns;
    A.rowptr.assign(nu + ns + 1, 0);
    const int64_t nnz = M.nnz() + Bt.nnz() + B.nnz() + (d11 ? ns : 0);
    PMC_REQUIRE(nnz < (int64_t)2147483647, "block operator exceeds int32 nonzeros");
    A.colind.reserve(nnz);
    A.vals.reserve(nnz);
    for (int i = 0; i < nu; ++i) {
        for (int p = M.rowptr[i]; p < M.rowptr[i + 1]; ++p) { A.colind.push_back(M.colind[p]); A.vals.push_back(M.vals[p]); }
        for (int p = Bt.rowptr[i]; p < Bt.rowptr[i + 1]; ++p) { A.colind.push_back(nu + Bt.colind[p]); A.vals.push_back(Bt.vals[p]); }
        A.rowptr[i + 1] = (int)A.colind.size();
    }
    for (int i = 0; i < ns; ++i) {
        for (int p = B.rowptr[i]; p < B.rowptr[i + 1]; ++p) { A.colind.push_back(B.colind[p]); A.vals.push_back(B.vals[p]); }
        if (d11) { A.colind.push_back(nu + i); A.vals.push_back(d11[i]); }
        A.rowptr[nu + i + 1] = (int)A.colind.size();
    }
    return A;
}

bool csr_is_oct_injection(const HostCsr& P) {
    if (P.nrows != 8 * P.ncols || P.nrows == 0) return false;
    for (int i = 0; i < P.nrows; ++i)
        if (P.rowptr[i + 1] - P.rowptr[i] != 1 || P.colind[P.rowptr[i]] != i / 8 || P.vals[P.rowptr[i]] != 1.0) return false;
    return true;
}

void sell_build(Sell& S, const HostCsr& A, bool upload_vals, bool keep_src, hipStream_t st, bool diag_last) {
    PMC_REQUIRE(A.nrows == 0 || A.ncols > 0, "SELL: matrix with rows but no columns");
    // position of the diagonal entry of every row (diagonal-last order only when all rows have one)
    std::vector<int> dpos;
    if (diag_last && A.nrows == A.ncols) {
        dpos.assign(A.nrows, -1);
        for (int r = 0; r < A.nrows && diag_last; ++r) {
            for (int p = A.rowptr[r]; p < A.rowptr[r + 1]; ++p)
                if (A.colind[p] == r) dpos[r] = p;
            if (dpos[r] < 0) diag_last = false;
        }
    } else {
        diag_last = false;
    }
    S.diag_last = diag_last;
    S.nrows = A.nrows;
    S.ncols = A.ncols;
    S.nnz = A.nnz();
    S.nslices = (A.nrows + 63) / 64;
    S.h_slice_off.assign(S.nslices + 1, 0);
    int64_t total = 0;
    for (int s = 0; s < S.nslices; ++s) {
        int w = 0;
        for (int r = s * 64; r < std::min(A.nrows, (s + 1) * 64); ++r) w = std::max(w, A.rowptr[r + 1] - A.rowptr[r]);
        S.h_slice_off[s] = (int)total;
        total += (int64_t)w * 64;
        PMC_REQUIRE(total < (int64_t)2147483647, "SELL storage exceeds int32 slots");
    }
    S.h_slice_off[S.nslices] = (int)total;
    S.nslots = total;
    S.h_cols.assign(total, 0);
    std::vector<double> hv(upload_vals ? total : 0, 0.0);
    S.h_src.clear();
    if (keep_src) S.h_src.assign(total, -1);
    for (int s = 0; s < S.nslices; ++s) {
        const int off = S.h_slice_off[s];
        const int w = (S.h_slice_off[s + 1] - off) / 64;
        for (int lane = 0; lane < 64; ++lane) {
            const int r = s * 64 + lane;
            const int pad_col = r < A.ncols ? r : 0;   // padding gathers a nearby (cached) entry, times 0
            const int b = r < A.nrows ? A.rowptr[r] : 0, e = r < A.nrows ? A.rowptr[r + 1] : 0;
            for (int j = 0; j < w; ++j) {
                const int slot = off + j * 64 + lane;
                if (b + j < e) {
                    int p = b + j;                     // CSR order, or: the diagonal entry moved behind the others
                    if (diag_last) p = (p == e - 1) ? dpos[r] : (p >= dpos[r] ? p + 1 : p);
                    S.h_cols[slot] = A.colind[p];
                    if (upload_vals) hv[slot] = A.vals[p];
                    if (keep_src) S.h_src[slot] = p;
                } else {
                    S.h_cols[slot] = pad_col;
                }
            }
        }
    }
    S.slice_off.upload(S.h_slice_off, st);
    S.cols.upload(S.h_cols, st);
    if (upload_vals) S.vals.upload(hv, st);
    PMC_HIP(hipStreamSynchronize(st));  // host staging vectors go out of scope
    if (!keep_src) { S.h_cols.clear(); S.h_cols.shrink_to_fit(); }
}

std::vector<double> sell_scaled_values(const Sell& S, const HostCsr& A, const std::vector<double>& colscale) {
    PMC_REQUIRE((int64_t)S.h_src.size() == S.nslots, "sell_scaled_values: matrix was built without its slot map");
    std::vector<double> v(S.nslots, 0.0);
    for (int64_t s = 0; s < S.nslots; ++s) {
        const int p = S.h_src[s];
        if (p >= 0) v[s] = A.vals[p] * colscale[A.colind[p]];
    }
    return v;
}

void sell_schedule_two_blocks(Sell& S, int n0, hipStream_t st, const HostCsr* A) {
    // slices [0, s0) lie (mostly) in the first row block, [s0, nslices) in the second
    const int s0 = (n0 + 63) / 64, s1 = S.nslices - s0;
    if (s0 <= 0 || s1 <= 0) return;
    std::vector<std::pair<double, int>> key(S.nslices);
    for (int k = 0; k < s0; ++k) key[k] = {k + 0.5, k};
    for (int j = 0; j < s1; ++j) {
        // default: by relative position; with the matrix at hand: right behind the first-block slices its rows actually
        // reference (their mean), so that both kinds of rows of one mesh region gather the same x rows at the same time.
        // (The proportional rule drifts: dofs per element vary over the mesh, and at 596 k rows the two blocks were up
        // to 80 slices apart - median distance between two uses of an x row 65 slices, 15 with the matrix-based key.)
        double pos = (j + 0.5) / s1 * s0;
        if (A) {
            double sum = 0.0;
            int64_t cnt = 0;
            for (int r = std::max(n0, (s0 + j) * 64); r < std::min(A->nrows, (s0 + j + 1) * 64); ++r)   // the rows of slice s0 + j
                for (int p = A->rowptr[r]; p < A->rowptr[r + 1]; ++p)
                    if (A->colind[p] < n0) { sum += A->colind[p] / 64.0; ++cnt; }
            if (cnt) pos = sum / cnt + 0.5;
        }
        key[s0 + j] = {pos + 1e-6, s0 + j};
    }
    std::stable_sort(key.begin(), key.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
    std::vector<int> order(S.nslices);
    for (int i = 0; i < S.nslices; ++i) order[i] = key[i].second;
    S.sched.upload(order, st);
    PMC_HIP(hipStreamSynchronize(st));
    S.h_sched = order;
}

}  // namespace pmc

// ---- algebraic coarsening of the Schur complement (setup, host) ---------------------------------------------------
namespace pmc {

// C = P^T A P for a piecewise-constant P given as agg[i] = coarse index of fine row i (values 1).
HostCsr csr_galerkin_agg(const HostCsr& A, const std::vector<int>& agg, int nc) {
    std::vector<std::vector<std::pair<int, double>>> rows(nc);
    // accumulate per coarse row with a marker array
    std::vector<int> marker(nc, -1);
    std::vector<int> members_ptr(nc + 1, 0);
    for (int i = 0; i < A.nrows; ++i) members_ptr[agg[i] + 1]++;
    for (int c = 0; c < nc; ++c) members_ptr[c + 1] += members_ptr[c];
    std::vector<int> members(A.nrows), fill(members_ptr.begin(), members_ptr.end() - 1);
    for (int i = 0; i < A.nrows; ++i) members[fill[agg[i]]++] = i;
    HostCsr C;
    C.nrows = C.ncols = nc;
    C.rowptr.assign(nc + 1, 0);
    std::vector<std::pair<int, double>> row;
    for (int c = 0; c < nc; ++c) {
        row.clear();
        for (int m = members_ptr[c]; m < members_ptr[c + 1]; ++m) {
            const int i = members[m];
            for (int p = A.rowptr[i]; p < A.rowptr[i + 1]; ++p) {
                const int cj = agg[A.colind[p]];
                if (marker[cj] < 0) { marker[cj] = (int)row.size(); row.emplace_back(cj, A.vals[p]); }
                else row[marker[cj]].second += A.vals[p];
            }
        }
        std::sort(row.begin(), row.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
        for (auto& cv : row) { C.colind.push_back(cv.first); C.vals.push_back(cv.second); marker[cv.first] = -1; }
        C.rowptr[c + 1] = (int)C.colind.size();
    }
    return C;
}

// One pass of pairwise matching along the strongest negative coupling (Notay-style): every unmatched row joins its
// strongest unmatched neighbour j with -a_ij >= theta * max_k(-a_ik), otherwise it stays alone.  Returns the number of
// aggregates; agg[i] = aggregate of row i.  On anisotropic operators this coarsens along the strong direction only.
int pairwise_match(const HostCsr& A, double theta, std::vector<int>& agg, bool allow_weak, bool join_singletons) {
    const int n = A.nrows;
    agg.assign(n, -1);
    // visit rows by increasing number of strong neighbours... plain index order keeps the mesh ordering's locality
    int nc = 0;
    for (int i = 0; i < n; ++i) {
        if (agg[i] >= 0) continue;
        double smax = 0.0;
        for (int p = A.rowptr[i]; p < A.rowptr[i + 1]; ++p)
            if (A.colind[p] != i) smax = std::max(smax, -A.vals[p]);
        int best = -1, weak = -1;
        double bval = 0.0, wval = 0.0;
        for (int p = A.rowptr[i]; p < A.rowptr[i + 1]; ++p) {
            const int j = A.colind[p];
            if (j == i || agg[j] >= 0) continue;
            const double s = -A.vals[p];
            if (s >= theta * smax && s > bval) { bval = s; best = j; }
            if (s > wval) { wval = s; weak = j; }
        }
        // isotropic / coarse operators: rather pair with the strongest FREE neighbour than stay a singleton (dense
        // smoothed-aggregation stencils leave few strong neighbours unmatched and the coarsening would stall)
        if (best < 0 && allow_weak) best = weak;
        agg[i] = nc;
        if (best >= 0) agg[best] = nc;
        ++nc;
    }
    if (!join_singletons) return nc;
    // a row whose neighbours were all taken before its turn joins the PAIR of its strongest neighbour (aggregates of at
    // most 3): without this 2-6 % of the rows of a face-based operator stay alone in every pass and the coarsening stalls
    std::vector<int> size(nc, 0);
    for (int i = 0; i < n; ++i) size[agg[i]]++;
    for (int i = 0; i < n; ++i) {
        if (size[agg[i]] != 1) continue;
        int best = -1;
        double bval = 0.0;
        for (int p = A.rowptr[i]; p < A.rowptr[i + 1]; ++p) {
            const int j = A.colind[p];
            if (j == i || size[agg[j]] != 2) continue;
            const double s = -A.vals[p];
            if (s > bval) { bval = s; best = j; }
        }
        if (best < 0) continue;
        size[agg[i]] = 0;
        agg[i] = agg[best];
        size[agg[best]] = 3;
    }
    std::vector<int> remap(nc, -1);
    int nc2 = 0;
    for (int c = 0; c < nc; ++c)
        if (size[c] > 0) remap[c] = nc2++;
    for (int& a : agg) a = remap[a];
    return nc2;
}

// `passes` rounds of pairwise matching (aggregates of up to 2^passes rows).  K = the coupling part of the operator that
// steers the matching; returns agg and the number of aggregates.
int aggregate_rows(const HostCsr& K, int passes, double theta, std::vector<int>& agg, bool allow_weak, bool join_singletons) {
    std::vector<int> cur;
    int nc = pairwise_match(K, theta, cur, allow_weak, join_singletons);
    agg = cur;
    HostCsr Kc = K;
    for (int pass = 1; pass < passes; ++pass) {
        Kc = csr_galerkin_agg(Kc, cur, nc);
        std::vector<int> nxt;
        const int nc2 = pairwise_match(Kc, theta, nxt, allow_weak, join_singletons);
        if (nc2 == nc) break;
        for (int& a : agg) a = nxt[a];
        cur = nxt;
        nc = nc2;
    }
    return nc;
}

HostCsr prolongator_from_agg(const std::vector<int>& agg, int nc) {
    HostCsr P;
    P.nrows = (int)agg.size();
    P.ncols = nc;
    P.rowptr.resize(P.nrows + 1);
    P.colind.resize(P.nrows);
    P.vals.assign(P.nrows, 1.0);
    for (int i = 0; i < P.nrows; ++i) { P.rowptr[i] = i; P.colind[i] = agg[i]; }
    P.rowptr[P.nrows] = P.nrows;
    return P;
}

// Renumbering of the fine rows of an aggregation level so that the restriction P^T res can be taken inside the residual
// kernel: every aggregate becomes a run of CONSECUTIVE rows that lies inside one 64-row SELL slice (= one wavefront), which
// then sums its own aggregates from the tile it has just computed (k::vc_residual_restrict_agg32) - the separate product with
// P^T (one more read of the residual, 32 us per iteration at 400 k multipliers) disappears.  Aggregates keep their coarse
// numbers and, up to the packing, their order (the coarse ids follow the fine ordering in windows of 512: mesh locality of
// the gathers stays); a slice is filled exactly by taking the next aggregates in order and, when the next one does not
// fit, the EARLIEST subset of the following `window` aggregates that fills the remaining rows (subset sum over at most 63
// rows).  Returns new2old (new row i holds old row new2old[i]) and the segment tables, or an empty vector when some slice
// cannot be filled exactly (the caller then keeps the separate restriction).
std::vector<int> agg_pack_rows(const HostCsr& P, std::vector<int>& seg_ptr, std::vector<int>& seg_cid,
                               std::vector<int>& seg_pos) {
    const int n = P.nrows, nc = P.ncols;
    seg_ptr.clear(); seg_cid.clear(); seg_pos.clear();
    std::vector<int> size(nc, 0), start(nc + 1, 0);
    for (int i = 0; i < n; ++i) {
        if (P.rowptr[i + 1] - P.rowptr[i] != 1 || P.vals[P.rowptr[i]] != 1.0) return {};
        ++size[P.colind[P.rowptr[i]]];
    }
    for (int c = 0; c < nc; ++c) {
        if (size[c] < 1 || size[c] > 64) return {};
        start[c + 1] = start[c] + size[c];
    }
    std::vector<int> members(n), fill(nc, 0);
    for (int i = 0; i < n; ++i) {          // rows of an aggregate in increasing (old) order
        const int c = P.colind[P.rowptr[i]];
        members[start[c] + fill[c]++] = i;
    }
    std::vector<int> new2old;
    new2old.reserve(n);
    std::vector<char> placed(nc, 0);
    const int nslices = (n + 63) / 64;
    seg_ptr.assign(1, 0);
    constexpr int window = 96;
    // aggregates are taken in the order of their FIRST fine row: the new numbering then follows the old one (mesh
    // neighbours stay index neighbours for the gathers of the 50-100 MB fine vectors); their coarse ids - sorted by coarse-row
    // length inside windows of 512 - only scatter accesses to the 10 x smaller coarse vectors (taken in coarse-id order
    // instead, the finest level's kernels lost 2-5 % each and their HBM traffic rose from 1.16 to 1.27 x algorithmic)
    std::vector<int> order(nc);
    for (int c = 0; c < nc; ++c) order[c] = c;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return members[start[a]] < members[start[b]]; });
    int next = 0;                          // first position in `order` not yet placed
    auto place = [&](int c, int pos) {
        placed[c] = 1;
        seg_cid.push_back(c);
        seg_pos.push_back((pos << 8) | size[c]);
        for (int q = 0; q < size[c]; ++q) new2old.push_back(members[start[c] + q]);
    };
    for (int sl = 0; sl < nslices; ++sl) {
        const int cap = std::min(64, n - sl * 64);
        while (next < nc && placed[order[next]]) ++next;
        // candidates: the next unplaced aggregates in order
        int cand[window], ncand = 0;
        for (int c = next; c < nc && ncand < window; ++c)
            if (!placed[order[c]]) cand[ncand++] = order[c];
        // 0/1 subset sum over the candidates in order: a sum keeps the FIRST way it was reached, so the subset found for `cap`
        // prefers the earliest aggregates
        int from[65], via[65];
        for (int t = 0; t <= 64; ++t) { from[t] = -2; via[t] = -1; }
        from[0] = -1;
        for (int k = 0; k < ncand && from[cap] == -2; ++k) {
            const int sz = size[cand[k]];
            for (int t = cap; t >= sz; --t)
                if (from[t] == -2 && from[t - sz] != -2) {
                    from[t] = t - sz;
                    via[t] = k;
                }
        }
        if (from[cap] == -2) return {};
        bool take[window] = {false};
        for (int t = cap; t > 0; t = from[t]) take[via[t]] = true;
        int used = 0;
        for (int k = 0; k < ncand; ++k)
            if (take[k]) {
                place(cand[k], used);
                used += size[cand[k]];
            }
        seg_ptr.push_back((int)seg_cid.size());
    }
    if ((int)new2old.size() != n) return {};
    return new2old;
}

// rows (and, sym, columns) of A renumbered: row i of the result is row new2old[i] of A
HostCsr csr_permute(const HostCsr& A, const std::vector<int>& new2old, bool rows, bool cols) {
    HostCsr B;
    B.nrows = A.nrows;
    B.ncols = A.ncols;
    std::vector<int> old2new;
    if (cols) {
        PMC_REQUIRE((int)new2old.size() == A.ncols, "csr_permute: column permutation size");
        old2new.resize(A.ncols);
        for (int i = 0; i < A.ncols; ++i) old2new[new2old[i]] = i;
    }
    if (rows) PMC_REQUIRE((int)new2old.size() == A.nrows, "csr_permute: row permutation size");
    B.rowptr.assign(A.nrows + 1, 0);
    B.colind.resize(A.colind.size());
    B.vals.resize(A.vals.size());
    int at = 0;
    for (int i = 0; i < A.nrows; ++i) {
        const int src = rows ? new2old[i] : i;
        for (int p = A.rowptr[src]; p < A.rowptr[src + 1]; ++p, ++at) {
            B.colind[at] = cols ? old2new[A.colind[p]] : A.colind[p];
            B.vals[at] = A.vals[p];
        }
        B.rowptr[i + 1] = at;
    }
    csr_sort_rows(B);
    return B;
}

// Every row of A cut into 2^sl consecutive pieces of (nearly) equal length: row i of A is the sum of the rows
// (i << sl) .. (i << sl) + 2^sl - 1 of the result.  A narrow launch walks 2^sl times as many wavefronts through slices that
// are 2^sl times shorter (the row-split kernels of the V-cycle add the pieces with a shuffle tree, see SellView::split_log2).
HostCsr csr_split_rows(const HostCsr& A, int sl) {
    const int S = 1 << sl;
    PMC_REQUIRE(sl >= 0 && sl <= 4 && (int64_t)A.nrows * S < (int64_t)INT32_MAX, "csr_split_rows: split factor");
    HostCsr B;
    B.nrows = A.nrows * S;
    B.ncols = A.ncols;
    B.colind = A.colind;
    B.vals = A.vals;
    B.rowptr.assign((size_t)B.nrows + 1, 0);
    for (int i = 0; i < A.nrows; ++i) {
        const int p0 = A.rowptr[i], len = A.rowptr[i + 1] - p0;
        for (int q = 0; q <= S; ++q)   // piece q holds the entries [len q / S, len (q + 1) / S)
            if (q > 0 || i == 0) B.rowptr[(size_t)i * S + q] = p0 + (int)((int64_t)len * q / S);
    }
    B.rowptr[B.nrows] = A.rowptr[A.nrows];
    return B;
}

// Plain (unsmoothed) aggregation hierarchy of an SPD operator whose off-diagonal entries have either sign (the hybridized
// sampler's multiplier system: positive couplings across right / obtuse dihedral angles): the matching is steered by the
// MAGNITUDE of the couplings, the prolongator is the aggregates' indicator (an injection - the V-cycle folds the coarse
// correction into the post-smoothing, MgLevel::has_sp), coarse operators are Galerkin sums.
std::vector<AmgLevelHost> agg_hierarchy(const HostCsr& A0, int passes0, int passes, double theta, int min_size, int max_levels) {
    std::vector<AmgLevelHost> out;
    HostCsr A = A0;
    for (int lvl = 0;; ++lvl) {
        AmgLevelHost L;
        const int n = A.nrows;
        if (n <= min_size || lvl + 1 >= max_levels) { L.S = std::move(A); out.push_back(std::move(L)); break; }
        HostCsr Ka = A;
        for (int i = 0; i < n; ++i)
            for (int p = Ka.rowptr[i]; p < Ka.rowptr[i + 1]; ++p)
                if (Ka.colind[p] != i) Ka.vals[p] = -std::fabs(Ka.vals[p]);
        std::vector<int> agg;
        const int nc = aggregate_rows(Ka, lvl == 0 ? passes0 : passes, theta, agg, true, true);
        if (getenv("PMC_VERBOSE"))
            fprintf(stderr, "[pmc]   aggregation level %d: n %d, %.1f entries/row -> %d rows\n", lvl, n, (double)A.nnz() / n, nc);
        if (nc * 10 > n * 9 || nc < 1) { L.S = std::move(A); out.push_back(std::move(L)); break; }
        HostCsr Ac = csr_galerkin_agg(A, agg, nc);
        {
            // The numbering of the aggregates is free: inside windows of 512 consecutive aggregates (consecutive in the fine
            // ordering, so the locality of the gathers stays) sort them by the length of their coarse row.  A SELL slice of
            // 64 rows is as wide as its longest row - irregular aggregates left 44 % padding on the first coarse level.
            std::vector<int> order(nc), newid(nc);
            for (int c = 0; c < nc; ++c) order[c] = c;
            for (int w0 = 0; w0 < nc; w0 += 512) {
                const int w1 = std::min(nc, w0 + 512);
                std::stable_sort(order.begin() + w0, order.begin() + w1, [&](int a, int b) {
                    return Ac.rowptr[a + 1] - Ac.rowptr[a] > Ac.rowptr[b + 1] - Ac.rowptr[b];
                });
            }
            for (int c = 0; c < nc; ++c) newid[order[c]] = c;
            for (int& a : agg) a = newid[a];
            Ac = csr_galerkin_agg(A, agg, nc);
        }
        L.P = prolongator_from_agg(agg, nc);
        L.S = std::move(A);
        out.push_back(std::move(L));
        A = std::move(Ac);
    }
    return out;
}

}  // namespace pmc

// ---- smoothed-aggregation hierarchy for the Schur complement (setup, host) -------------------------------------------
namespace pmc {

HostCsr csr_spgemm(const HostCsr& A, const HostCsr& B) {
    PMC_REQUIRE(A.ncols == B.nrows, "spgemm: shape mismatch");
    HostCsr C;
    C.nrows = A.nrows;
    C.ncols = B.ncols;
    C.rowptr.assign(A.nrows + 1, 0);
    std::vector<int> marker(B.ncols, -1);
    std::vector<std::pair<int, double>> row;
    for (int i = 0; i < A.nrows; ++i) {
        row.clear();
        for (int p = A.rowptr[i]; p < A.rowptr[i + 1]; ++p) {
            const int k = A.colind[p];
            const double a = A.vals[p];
            for (int q = B.rowptr[k]; q < B.rowptr[k + 1]; ++q) {
                const int j = B.colind[q];
                if (marker[j] < 0) { marker[j] = (int)row.size(); row.emplace_back(j, a * B.vals[q]); }
                else row[marker[j]].second += a * B.vals[q];
            }
        }
        std::sort(row.begin(), row.end(), [](const auto& x, const auto& y) { return x.first < y.first; });
        for (auto& cv : row) { C.colind.push_back(cv.first); C.vals.push_back(cv.second); marker[cv.first] = -1; }
        PMC_REQUIRE(C.colind.size() < (size_t)2147483647, "spgemm: result exceeds int32 nonzeros");
        C.rowptr[i + 1] = (int)C.colind.size();
    }
    return C;
}

static HostCsr csr_add_diag(const HostCsr& K, const std::vector<double>& w) {
    if (w.empty()) return K;
    HostCsr S = K;
    for (int i = 0; i < S.nrows; ++i) {
        bool found = false;
        for (int p = S.rowptr[i]; p < S.rowptr[i + 1]; ++p)
            if (S.colind[p] == i) { S.vals[p] += w[i]; found = true; }
        PMC_REQUIRE(found, "Schur coupling matrix lacks a diagonal entry");
    }
    return S;
}

// median over rows of (largest / smallest) nonzero off-diagonal magnitude: ~1 for isotropic meshes, >> 1 for
// stretched cells
double csr_anisotropy(const HostCsr& K) {
    std::vector<double> ratio;
    ratio.reserve(K.nrows);
    for (int i = 0; i < K.nrows; ++i) {
        double mx = 0.0, mn = 1e300;
        for (int p = K.rowptr[i]; p < K.rowptr[i + 1]; ++p) {
            if (K.colind[p] == i) continue;
            const double a = std::fabs(K.vals[p]);
            if (a == 0.0) continue;
            mx = std::max(mx, a);
            mn = std::min(mn, a);
        }
        if (mx > 0.0) ratio.push_back(mx / mn);
    }
    if (ratio.empty()) return 1.0;
    std::nth_element(ratio.begin(), ratio.begin() + ratio.size() / 2, ratio.end());
    return ratio[ratio.size() / 2];
}

// Smoothed aggregation (Vanek, Mandel, Brezina) on S = diag(w) + K: pairwise-matching aggregates of up to 2^passes rows
// along strong couplings, tentative piecewise-constant prolongator smoothed by one damped Jacobi step with the
// strength-filtered operator, Galerkin coarse operators.  K and diag(w) are coarsened separately so the caller can
// keep treating them differently.  levels[0] holds the input operator.
std::vector<AmgLevelHost> sa_hierarchy(const HostCsr& K0, const std::vector<double>& w0, int passes, double theta,
                                       int min_size, int max_levels) {
    std::vector<AmgLevelHost> out;
    HostCsr K = K0;
    bool iso0 = false;
    HostCsr Wm;           // mass part as a general matrix on coarse levels (diagonal on level 0)
    bool w_is_diag = true;
    std::vector<double> w = w0;
    for (int lvl = 0;; ++lvl) {
        AmgLevelHost L;
        if (w_is_diag) {
            L.S = csr_add_diag(K, w);
        } else {
            // S = Wm + K: union pattern via spgemm-free merge (both sorted)
            HostCsr S;
            S.nrows = S.ncols = K.nrows;
            S.rowptr.assign(K.nrows + 1, 0);
            for (int i = 0; i < K.nrows; ++i) {
                int a = K.rowptr[i], b = Wm.rowptr[i];
                const int ae = K.rowptr[i + 1], be = Wm.rowptr[i + 1];
                while (a < ae || b < be) {
                    const int ca = a < ae ? K.colind[a] : 2147483647, cb = b < be ? Wm.colind[b] : 2147483647;
                    if (ca == cb) { S.colind.push_back(ca); S.vals.push_back(K.vals[a++] + Wm.vals[b++]); }
                    else if (ca < cb) { S.colind.push_back(ca); S.vals.push_back(K.vals[a++]); }
                    else { S.colind.push_back(cb); S.vals.push_back(Wm.vals[b++]); }
                }
                S.rowptr[i + 1] = (int)S.colind.size();
            }
            L.S = std::move(S);
        }
        const int n = L.S.nrows;
        if (n <= min_size || lvl + 1 >= max_levels) { out.push_back(std::move(L)); break; }
        // aggregates of 2^passes rows on the finest level (follows the strong direction of stretched cells), one more
        // matching pass below it: once the coupling is isotropic a smoothed prolongator over aggregates of 4 widens the
        // stencil faster than the level shrinks (measured 7 -> 14 -> 33 -> 82 -> 123 entries per row; with aggregates
        // of 8 it stays at ~30) and the per-realization Galerkin lists of the Darcy hierarchy grow with stencil x |P|^2
        std::vector<int> agg;
        // isotropic coupling on the finest level as well (median strongest/weakest ratio <= 10): aggregates of 8 from
        // the start, 16 below
        int extra = (lvl == 0 ? 0 : 1) + (lvl == 0 && csr_anisotropy(K) <= 10.0 ? 1 : 0) + (lvl > 0 && iso0 ? 1 : 0);
        if (lvl == 0) iso0 = csr_anisotropy(K) <= 10.0;
        bool weak = iso0;
        if (iso0) {   // tuning overrides (isotropic problems only)
            if (const char* e = lab_env(lvl == 0 ? "PMC_SA_ISO_PASSES0" : "PMC_SA_ISO_PASSES1")) extra = atoi(e) - passes;
            if (const char* e = lab_env("PMC_SA_ISO_WEAK")) weak = atoi(e) != 0;
        }
        const int nc = aggregate_rows(K, passes + extra, theta, agg, weak);
        if (getenv("PMC_VERBOSE"))
            fprintf(stderr, "[pmc]   SA level %d: n %d, %.1f entries/row, aggregates of 2^%d -> %d rows\n", lvl, n,
                    (double)L.S.nnz() / n, passes + extra, nc);
        if (nc * 10 > n * 9 || nc < 1) { out.push_back(std::move(L)); break; }      // coarsening stalled
        // strength-filtered operator (weak off-diagonals lumped onto the diagonal), damped-Jacobi smoothing of P_tent
        const HostCsr& S = L.S;
        std::vector<double> df(n, 0.0), rowabs(n, 0.0);
        std::vector<char> keep(S.colind.size(), 1);
        for (int i = 0; i < n; ++i) {
            double smax = 0.0;
            for (int p = S.rowptr[i]; p < S.rowptr[i + 1]; ++p)
                if (S.colind[p] != i) smax = std::max(smax, -S.vals[p]);
            for (int p = S.rowptr[i]; p < S.rowptr[i + 1]; ++p) {
                if (S.colind[p] == i) { df[i] += S.vals[p]; continue; }
                if (-S.vals[p] < theta * smax) { keep[p] = 0; df[i] += S.vals[p]; }
            }
        }
        double lam = 0.0;
        for (int i = 0; i < n; ++i) {
            double s = std::fabs(df[i]);
            for (int p = S.rowptr[i]; p < S.rowptr[i + 1]; ++p)
                if (S.colind[p] != i && keep[p]) s += std::fabs(S.vals[p]);
            PMC_REQUIRE(df[i] > 0.0, "smoothed aggregation: filtered operator lost its positive diagonal");
            lam = std::max(lam, s / df[i]);
        }
        const double omega = 4.0 / (3.0 * lam);
        HostCsr P;
        P.nrows = n;
        P.ncols = nc;
        P.rowptr.assign(n + 1, 0);
        {
            std::vector<int> marker(nc, -1);
            std::vector<std::pair<int, double>> row;
            for (int i = 0; i < n; ++i) {
                row.clear();
                auto add = [&](int c, double v) {
                    if (marker[c] < 0) { marker[c] = (int)row.size(); row.emplace_back(c, v); }
                    else row[marker[c]].second += v;
                };
                add(agg[i], 1.0 - omega);            // (I - omega D_f^-1 S_f): diagonal part = 1 - omega * df/df
                for (int p = S.rowptr[i]; p < S.rowptr[i + 1]; ++p)
                    if (S.colind[p] != i && keep[p]) add(agg[S.colind[p]], -omega * S.vals[p] / df[i]);
                std::sort(row.begin(), row.end(), [](const auto& x, const auto& y) { return x.first < y.first; });
                for (auto& cv : row) { P.colind.push_back(cv.first); P.vals.push_back(cv.second); marker[cv.first] = -1; }
                P.rowptr[i + 1] = (int)P.colind.size();
            }
        }
        const HostCsr Pt = csr_transpose(P);
        HostCsr Kc = csr_spgemm(Pt, csr_spgemm(K, P));
        HostCsr Wc;
        if (w_is_diag) {
            if (!w.empty()) {
                HostCsr WP = P;
                for (int i = 0; i < n; ++i)
                    for (int p = WP.rowptr[i]; p < WP.rowptr[i + 1]; ++p) WP.vals[p] *= w[i];
                Wc = csr_spgemm(Pt, WP);
            }
        } else {
            Wc = csr_spgemm(Pt, csr_spgemm(Wm, P));
        }
        L.P = std::move(P);
        out.push_back(std::move(L));
        K = std::move(Kc);
        if (!w.empty() || !w_is_diag) {
            Wm = std::move(Wc);
            w_is_diag = false;
            w.clear();
            if (Wm.nrows == 0) { w_is_diag = true; }     // no mass part at all (Darcy)
        }
    }
    return out;
}


// Smallest eigenvalue of D^-1 M (D^-1 = dinv > 0, M symmetric positive definite) by `steps` Lanczos iterations on the
// symmetrically scaled matrix (three-term recurrence, no stored basis) and bisection on the tridiagonal matrix.  The
// Ritz value approaches the eigenvalue from above; it sizes the Chebyshev interval of the M-block smoother, where an
// over-estimate only weakens the polynomial on the few modes below it (it stays positive definite).
double lanczos_lambda_min_scaled(const HostCsr& M, const std::vector<double>& dinv, int steps) {
    const int n = M.nrows;
    if (n == 0) return 1.0;
    std::vector<double> sd(n), v(n), vold(n, 0.0), w(n), t(n);
    for (int i = 0; i < n; ++i) sd[i] = std::sqrt(dinv[i]);
    uint64_t state = 0x9e3779b97f4a7c15ull;
    double nrm = 0.0;
    for (int i = 0; i < n; ++i) {
        state = state * 6364136223846793005ull + 1442695040888963407ull;
        v[i] = (double)((state >> 11) & 0xfffff) / 1048576.0 - 0.5;
        nrm += v[i] * v[i];
    }
    nrm = std::sqrt(nrm);
    for (double& x : v) x /= nrm;
    std::vector<double> al, be;
    double beta = 0.0;
    steps = std::min(steps, n);
    for (int k = 0; k < steps; ++k) {
        for (int i = 0; i < n; ++i) t[i] = sd[i] * v[i];
        for (int i = 0; i < n; ++i) {
            double acc = 0.0;
            for (int p = M.rowptr[i]; p < M.rowptr[i + 1]; ++p) acc += M.vals[p] * t[M.colind[p]];
            w[i] = sd[i] * acc;
        }
        double alpha = 0.0;
        for (int i = 0; i < n; ++i) alpha += w[i] * v[i];
        for (int i = 0; i < n; ++i) w[i] -= alpha * v[i] + beta * vold[i];
        al.push_back(alpha);
        double b2 = 0.0;
        for (int i = 0; i < n; ++i) b2 += w[i] * w[i];
        beta = std::sqrt(b2);
        if (beta < 1e-14 * std::fabs(alpha) || k + 1 == steps) break;
        be.push_back(beta);
        for (int i = 0; i < n; ++i) { vold[i] = v[i]; v[i] = w[i] / beta; }
    }
    // smallest eigenvalue of the tridiagonal (al, be): bisection with the Sturm count
    const int m = (int)al.size();
    double lo = 1e300, hi = -1e300;
    for (int i = 0; i < m; ++i) {
        const double r = (i > 0 ? std::fabs(be[i - 1]) : 0.0) + (i + 1 < m ? std::fabs(be[i]) : 0.0);
        lo = std::min(lo, al[i] - r);
        hi = std::max(hi, al[i] + r);
    }
    auto count_below = [&](double x) {
        int cnt = 0;
        double q = al[0] - x;
        if (q < 0) ++cnt;
        for (int i = 1; i < m; ++i) {
            if (q == 0.0) q = 1e-300;
            q = al[i] - x - be[i - 1] * be[i - 1] / q;
            if (q < 0) ++cnt;
        }
        return cnt;
    };
    for (int it = 0; it < 200 && hi - lo > 1e-12 * std::max(1.0, std::fabs(hi)); ++it) {
        const double mid = 0.5 * (lo + hi);
        if (count_below(mid) >= 1) hi = mid; else lo = mid;
    }
    return 0.5 * (lo + hi);
}

// Chebyshev interval ratio lambda_max / lambda_min of the l1-scaled M-block (lambda_max <= 1 by construction)
double mass_block_ratio(const HostCsr& M, const std::vector<double>& l1inv) {
    const double lmin = lanczos_lambda_min_scaled(M, l1inv, 40);
    const double ratio = 1.0 / std::max(lmin, 1.0 / 64.0);
    return std::min(64.0, std::max(1.5, ratio));
}
}  // namespace pmc
