// Host-side sparse helpers (setup only, outside every timed region): CSR copies, transpose,
// block assembly and the CSR -> SELL-64 re-layout done once when a handle is created.
#include <algorithm>
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

void sell_build(Sell& S, const HostCsr& A, bool upload_vals, bool keep_src, hipStream_t st) {
    PMC_REQUIRE(A.nrows == 0 || A.ncols > 0, "SELL: matrix with rows but no columns");
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
                    S.h_cols[slot] = A.colind[b + j];
                    if (upload_vals) hv[slot] = A.vals[b + j];
                    if (keep_src) S.h_src[slot] = b + j;
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

void sell_schedule_two_blocks(Sell& S, int n0, hipStream_t st) {
    // slices [0, s0) lie (mostly) in the first row block, [s0, nslices) in the second
    const int s0 = (n0 + 63) / 64, s1 = S.nslices - s0;
    if (s0 <= 0 || s1 <= 0) return;
    std::vector<std::pair<double, int>> key(S.nslices);
    for (int k = 0; k < s0; ++k) key[k] = {(k + 0.5) / s0, k};
    for (int j = 0; j < s1; ++j) key[s0 + j] = {(j + 0.5) / s1, s0 + j};
    std::stable_sort(key.begin(), key.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
    std::vector<int> order(S.nslices);
    for (int i = 0; i < S.nslices; ++i) order[i] = key[i].second;
    S.sched.upload(order, st);
    PMC_HIP(hipStreamSynchronize(st));
}

}  // namespace pmc
