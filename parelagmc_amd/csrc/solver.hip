// Host orchestration: Chebyshev smoother, level V-cycle, batched preconditioned MINRES.
#include "solver.hpp"

#include <atomic>
#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <cstring>

namespace pmc {

double* cheb_apply(hipStream_t st, int nb, const SellView& A, const double* dinv, bool dinv_bv, const ChebParams& cp,
                   const double* r, double* xa, double* xb, double* d, bool zero_guess, double* dot_partial,
                   int* dot_blocks, zvec zlast) {
    if (cp.degree < 1) throw Error(PMC_ERR_INVALID, "Chebyshev degree must be >= 1");
    const double lmax = cp.lmax, lmin = cp.lmax / cp.ratio;
    const double theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin);
    const double sigma = theta / delta;
    double rho_old = 1.0 / sigma;
    if (cheb_fused(cp, zero_guess)) {
        const double rho1 = 1.0 / (2.0 * sigma - rho_old);
        const double c0 = (1.0 + rho1 * rho_old) / theta + 2.0 * rho1 / delta;
        const double c1 = 2.0 * rho1 / (delta * theta);
        SellView As = A;
        As.vals = cp.scaled_vals;
        int nblk;
        if (zero_guess) {
            nblk = zlast ? k::poly2_z(st, nb, As, dinv, dinv_bv, r, zlast, c0, c1, dot_partial)
                         : k::poly2(st, nb, As, dinv, dinv_bv, r, xa, c0, c1, dot_partial);
        } else {
            // x2 = x0 + p2(r - A x0): residual into the work vector, polynomial of it accumulated onto x0 (in place, or
            // into the typed result)
            k::residual(st, nb, A, r, xa, d);
            nblk = zlast ? k::poly2_z(st, nb, As, dinv, dinv_bv, d, zlast, c0, c1, dot_partial, xa, r)
                         : k::poly2(st, nb, As, dinv, dinv_bv, d, xa, c0, c1, dot_partial, xa, r);
        }
        if (dot_blocks) *dot_blocks = dot_partial ? nblk : 0;
        return zlast ? nullptr : xa;
    }
    double* cur = xa;
    double* oth = xb;
    int step = 0;
    int nblk = 0;
    double* dp = (cp.degree == 1 && !zlast) ? dot_partial : nullptr;
    if (zero_guess) {
        nblk = k::cheb_first(st, nb, A.nrows, dinv, dinv_bv, r, d, cur, 1.0 / theta, dp);
        step = 1;
    } else {
        nblk = k::cheb_step(st, nb, A, dinv, dinv_bv, r, cur, d, oth, 0.0, 1.0 / theta, dp);
        std::swap(cur, oth);
        step = 1;
    }
    for (; step < cp.degree; ++step) {
        const double rho = 1.0 / (2.0 * sigma - rho_old);
        dp = (step == cp.degree - 1) ? dot_partial : nullptr;
        if (zlast && step == cp.degree - 1) {
            nblk = k::cheb_step_z(st, nb, A, dinv, dinv_bv, r, cur, d, zlast, rho * rho_old, 2.0 * rho / delta, dp);
            cur = nullptr;                      // the result is in zlast
            break;
        }
        nblk = k::cheb_step(st, nb, A, dinv, dinv_bv, r, cur, d, oth, rho * rho_old, 2.0 * rho / delta, dp);
        std::swap(cur, oth);
        rho_old = rho;
    }
    if (zlast && cur) {                         // degree 1: no typed kernel, one rounding pass with the fused dot
        nblk = k::convert_z(st, nb, A.nrows, cur, zlast, r, dot_partial);
        cur = nullptr;
    }
    if (dot_blocks) *dot_blocks = dot_partial ? nblk : 0;
    return cur;
}

int cheb_apply_z(hipStream_t st, int nb, const SellView& A, const double* dinv, bool dinv_bv, const ChebParams& cp,
                 const double* r, zvec z, double* xa, double* xb, double* d, double* dot_partial) {
    if (!cheb_fused(cp, true) && (!xa || !xb || !d)) throw Error(PMC_ERR_INTERNAL, "cheb_apply_z: scratch vectors missing");
    int nblk = 0;
    cheb_apply(st, nb, A, dinv, dinv_bv, cp, r, xa, xb, d, true, dot_partial, &nblk, z);
    return nblk;
}

int cheb_post_from_residual(hipStream_t st, int nb, const SellView& A, const double* dinv, bool dinv_bv,
                            const ChebParams& cp, const double* r, const double* res, double* x, const int* parent,
                            const double* xc, double* dot_partial, zvec zout) {
    if (!(cp.degree == 2 && cp.scaled_vals)) throw Error(PMC_ERR_INTERNAL, "fused post-smoothing needs degree 2");
    const double lmax = cp.lmax, lmin = cp.lmax / cp.ratio;
    const double theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin);
    const double sigma = theta / delta;
    const double rho_old = 1.0 / sigma;
    const double rho1 = 1.0 / (2.0 * sigma - rho_old);
    const double c0 = (1.0 + rho1 * rho_old) / theta + 2.0 * rho1 / delta;
    const double c1 = 2.0 * rho1 / (delta * theta);
    SellView As = A;
    As.vals = cp.scaled_vals;
    if (zout) return k::poly2_z(st, nb, As, dinv, dinv_bv, res, zout, c0, c1, dot_partial, x, r, parent, xc);
    return k::poly2(st, nb, As, dinv, dinv_bv, res, x, c0, c1, dot_partial, x, r, parent, xc);
}

void MgLevel::ensure(int nb) {
    const size_t need = (size_t)n * nb;
    r.ensure(need);
    xa.ensure(need);
    xb.ensure(need);
    d.ensure(need);
    res.ensure(need);
}

void Multigrid::enable_bv_tail(int max_rows) {
    bv_tail_width = kGroup;
    for (MgLevel& m : L) {
        if (!m.bv || m.n > max_rows) continue;
        m.vals_t.alloc((size_t)m.S.nslots * bv_tail_width);
        m.scaled_t.alloc((size_t)m.S.nslots * bv_tail_width);
        m.dinv_t.alloc((size_t)m.n * bv_tail_width);
    }
}

void Multigrid::ensure_bv_tail_width(hipStream_t st, int nb) {
    if (bv_tail_width == 0 || nb <= bv_tail_width) return;
    PMC_HIP(hipStreamSynchronize(st));           // nothing in flight reads the old copies
    bv_tail_width = nb;
    for (MgLevel& m : L) {
        if (!m.vals_t.p) continue;
        m.vals_t.alloc((size_t)m.S.nslots * nb);
        m.scaled_t.alloc((size_t)m.S.nslots * nb);
        m.dinv_t.alloc((size_t)m.n * nb);
    }
    build_tails(st);
}

void Multigrid::refresh_bv_tail(hipStream_t st, int nb, int first_level) {
    for (size_t l = (size_t)first_level; l < L.size(); ++l) {   // levels finer than the one being solved hold stale values
        MgLevel& m = L[l];
        if (!m.vals_t.p) continue;
        k::transpose_bv(st, nb, (size_t)m.S.nslots, m.vals_bv.p, m.vals_t.p);
        if (m.f32) k::transpose_bv32(st, nb, (size_t)m.S.nslots, m.scaled32.p, m.scaled_t.p);
        else k::transpose_bv(st, nb, (size_t)m.S.nslots, m.vals_scaled.p, m.scaled_t.p);
        k::transpose_bv(st, nb, (size_t)m.n, m.dinv.p, m.dinv_t.p);
    }
}

void Multigrid::build_tails(hipStream_t st) {
    const int nl = (int)L.size();
    tail.clear();
    tail.resize(nl);
    tail_lds.assign(nl, 0);
    for (int l0 = 0; l0 < nl; ++l0) {
        // per-realization values are interleaved [slot][nb]: one workgroup per column would pull 16x the bytes it uses
        // through its L1 (measured: slower than the separate kernels), so the tail is for shared-value hierarchies
        // ... unless the levels carry transposed copies (enable_bv_tail): bv = 2
        if (L[l0].bv && !L[l0].vals_t.p) continue;
        TailParams tp{};
        tp.bv = L[l0].bv ? 2 : 0;
        tp.smooth_degree = smooth_degree;
        tp.smooth_ratio = smooth_ratio;
        size_t off = 0;
        int cnt = 0;
        bool ok = true;
        for (int l = l0; l < nl; ++l) {
            if (cnt == 8) { ok = false; break; }
            const MgLevel& m = L[l];
            if (m.bv != L[l0].bv || (m.bv && !m.vals_t.p)) { ok = false; break; }
            TailLevelDev& d = tp.lev[cnt++];
            d.n = m.n;
            d.nslices = m.S.nslices;
            d.nslots = (int)m.S.nslots;
            d.slice_off = m.S.slice_off.p;
            d.cols = m.S.cols.p;
            d.vals = m.bv ? m.vals_t.p : m.S.vals.p;
            d.vals_scaled = m.bv ? m.scaled_t.p : m.vals_scaled.p;
            d.dinv = m.bv ? m.dinv_t.p : m.dinv.p;
            d.lmax = m.lmax;
            d.lds_off = (int)off;
            off += 3 * (size_t)m.n;
            const bool last = m.is_last || l == nl - 1;
            d.ainv = last ? m.ainv.p : nullptr;
            d.last_degree = last ? (m.is_last ? m.last_degree : coarse_degree) : 0;
            d.last_ratio = last ? (m.is_last ? m.last_ratio : coarse_ratio) : 1.0;
            if (last) break;
            d.p_nslices = m.P.nslices; d.p_off = m.P.slice_off.p; d.p_cols = m.P.cols.p; d.p_vals = m.P.vals.p;
            d.pt_nslices = m.Pt.nslices; d.pt_off = m.Pt.slice_off.p; d.pt_cols = m.Pt.cols.p; d.pt_vals = m.Pt.vals.p;
        }
        if (!ok || off > kTailLdsDoubles) continue;
        tp.nlev = cnt;
        tail[l0].alloc(1);
        PMC_HIP(hipMemcpyAsync(tail[l0].p, &tp, sizeof(tp), hipMemcpyHostToDevice, st));
        PMC_HIP(hipStreamSynchronize(st));
        tail_lds[l0] = off;
    }
}

double* Multigrid::cycle(hipStream_t st, int nb, int l, int l0, const double* r, double* target, zvec ztarget,
                         double* dot_partial, int* dot_blocks, const std::function<void()>* side) {
    MgLevel& lv = L[l];
    lv.ensure(nb);
    const bool last = (l == (int)L.size() - 1) || lv.is_last;
    struct ClearR32 {   // whatever path the top level takes, a copy offered for THIS cycle is not seen by the next one
        Multigrid* m; bool top;
        ~ClearR32() { if (top) m->r32_top = nullptr; }
    } clear_r32{this, l == l0};
    // A launch of few realizations (the drop-in path: one per call) runs the LDS tail on as many compute units as it has
    // realizations, and a tail that starts at a level of several thousand rows with 17-27 entries each is bound by ONE
    // unit's L2 port (LAB_NOTES 9.16: 142 us per cycle - half of a one-realization Eval of the hybridized sampler).  Such
    // a level runs as kernels then and the tail starts one level further down.
    // ... and it ends on the first level that carries a dense inverse: x = A^-1 r by n wavefronts (k::dense_apply) instead of
    // one workgroup cycling through the remaining levels (573 rows at 400 k multipliers: 25 -> 4 us per cycle)
    if (l > l0 && nb <= dense_nb && lv.dense_inv.p && !target && !ztarget && !dot_partial) {
        if (side && *side) (*side)();
        k::dense_apply(st, nb, lv.n, lv.dense_inv.p, r, lv.xa.p);
        return lv.xa.p;
    }
    const bool tail_later = nb <= tail_later_nb && lv.n > 4096 && !last && l + 1 < (int)tail.size() && tail[l + 1].p;
    const bool tail_here = use_tail && l < (int)tail.size() && tail[l].p && !tail_later;
    const bool f32_shared = !last && !lv.bv && lv.has_sp && (lv.p_oct || f32_any_injection) && smooth_degree == 2 &&
                            lv.vals_scaled.p && f32_intermediates;
    const bool f32_bv = !last && lv.bv && lv.f32 && lv.p_oct && smooth_degree == 2 && lv.scaled32.p && f32_intermediates;
    if (ztarget && target) throw Error(PMC_ERR_INTERNAL, "V-cycle: two result buffers");
    const bool ends_here = tail_here || last;
    if (side && *side && ends_here) (*side)();   // beside the bottom of the V: the least parallel kernels of the cycle
    if (tail_here) {
        if (ztarget) {
            const int nblk = k::mg_tail_z(st, nb, tail[l].p, tail_lds[l], r, ztarget, dot_partial);
            if (dot_blocks) *dot_blocks = nblk;
            return nullptr;
        }
        double* out = target ? target : lv.xa.p;
        const int nblk = k::mg_tail(st, nb, tail[l].p, tail_lds[l], r, out, dot_partial);
        if (dot_blocks) *dot_blocks = nblk;
        return out;
    }
    const SellView A = lv.sview();
    // Shared-value level with an injection prolongator over groups of 8 (uniform refinement) and the one-pass degree-2
    // smoothers: the iterate and the residuals of the level - vectors that live only inside this application of the
    // preconditioner - are kept in fp32 (k::vc_* kernels; the buffers xb / res hold them).  Input, output, coarse vectors and
    // all arithmetic stay fp64.
    if (f32_shared) {
        double c0, c1;
        cheb2_coefficients(lv.lmax, smooth_ratio, &c0, &c1);
        SellView As = A;
        As.vals = lv.vals_scaled.p;
        float* xf = reinterpret_cast<float*>(lv.xb.p);
        float* resf = reinterpret_cast<float*>(lv.res.p);
        double* out = target ? target : lv.xa.p;
        // top level of a cycle inside the MINRES loop of an aggregation hierarchy: the fp32 copy of r, if the caller has one
        const float* r32 = (l == l0 && !lv.p_oct) ? r32_top : nullptr;
        if (l == l0) r32_top = nullptr;
        // inner level of a launch of at most 8 realizations: the row-split forms of the four kernels (MgLevel::S_split)
        if (l > l0 && nb <= dense_nb && lv.split_log2 > 0 && !lv.p_oct && !lv.p_agg && !dot_partial) {
            const SellView Asp = view_split(lv.S_split, lv.split_log2);
            SellView Assp = Asp;
            Assp.vals = lv.scaled_split.p;
            const SellView SPv = lv.sp_split_log2 > 0 ? view_split(lv.SP_split, lv.sp_split_log2) : view(lv.SP);
            k::vc_presmooth32(st, nb, Assp, lv.dinv.p, r, xf, c0, c1);
            MgLevel& lcs = L[l + 1];
            lcs.ensure(nb);
            k::vc_residual32(st, nb, Asp, r, xf, resf);
            k::spmm_z(st, nb, view(lv.Pt), zvec(resf, true), lcs.r.p, nullptr, zvec());
            double* xcs = cycle(st, nb, l + 1, l0, lcs.r.p, nullptr, zvec(), nullptr, nullptr, side);
            k::vc_residual_coarse32(st, nb, SPv, resf, xcs);
            if (ztarget) k::vc_postsmooth32_z(st, nb, Assp, lv.dinv.p, resf, xf, ztarget, c0, c1, r, lv.parent.p, xcs, nullptr);
            else k::vc_postsmooth32(st, nb, Assp, lv.dinv.p, resf, xf, out, c0, c1, r, lv.parent.p, xcs, nullptr);
            if (dot_blocks) *dot_blocks = 0;
            return ztarget ? nullptr : out;
        }
        if (r32) k::vc_presmooth32_r32(st, nb, As, lv.dinv.p, r32, xf, c0, c1);
        else k::vc_presmooth32(st, nb, As, lv.dinv.p, r, xf, c0, c1);
        MgLevel& lc = L[l + 1];
        lc.ensure(nb);
        if (lv.p_oct) {
            k::vc_residual_restrict8_32(st, nb, A, r, xf, resf, lc.r.p);
        } else if (lv.p_agg) {
            if (r32) k::vc_residual_restrict_agg32_r32(st, nb, A, r32, xf, resf, lc.r.p, lv.seg_ptr.p, lv.seg_cid.p, lv.seg_pos.p);
            else k::vc_residual_restrict_agg32(st, nb, A, r, xf, resf, lc.r.p, lv.seg_ptr.p, lv.seg_cid.p, lv.seg_pos.p);
        } else if (r32) {
            k::vc_residual32_r32(st, nb, A, r32, xf, resf);
            k::spmm_z(st, nb, view(lv.Pt), zvec(resf, true), lc.r.p, nullptr, zvec());
        } else {
            k::vc_residual32(st, nb, A, r, xf, resf);
            k::spmm_z(st, nb, view(lv.Pt), zvec(resf, true), lc.r.p, nullptr, zvec());
        }
        double* xc = cycle(st, nb, l + 1, l0, lc.r.p, nullptr, zvec(), nullptr, nullptr, side);
        k::vc_residual_coarse32(st, nb, view(lv.SP), resf, xc);
        const bool timed = smooth_timer && smooth_timer->on && l == l0;
        if (timed) smooth_timer->begin(st);
        const int nblk = ztarget ? k::vc_postsmooth32_z(st, nb, As, lv.dinv.p, resf, xf, ztarget, c0, c1, r, lv.parent.p, xc, dot_partial)
                                 : k::vc_postsmooth32(st, nb, As, lv.dinv.p, resf, xf, out, c0, c1, r, lv.parent.p, xc, dot_partial);
        if (timed) smooth_timer->end(st);
        if (dot_blocks) *dot_blocks = dot_partial ? nblk : 0;
        return ztarget ? nullptr : out;
    }
    // The same for a per-realization level with fp32 values (Darcy): pre-smoothing into an fp32 iterate, restriction of its
    // residual without storing the fine residual (nothing reads it: there is no S P for per-realization values), the coarse
    // correction added to the fp32 iterate, residual (fp32) and post-smoothing from it.
    if (f32_bv) {
        double c0, c1;
        cheb2_coefficients(lv.lmax, smooth_ratio, &c0, &c1);
        SellView As = A;
        As.vals = lv.scaled_ptr();
        float* xf = reinterpret_cast<float*>(lv.xb.p);
        float* df = reinterpret_cast<float*>(lv.d.p);
        double* out = target ? target : lv.xa.p;
        k::vc_presmooth32_bv(st, nb, As, lv.dinv.p, r, xf, c0, c1);
        MgLevel& lc = L[l + 1];
        lc.ensure(nb);
        k::vc_restrict8_32_bv(st, nb, A, r, xf, lc.r.p);
        double* xc = cycle(st, nb, l + 1, l0, lc.r.p, nullptr, zvec(), nullptr, nullptr, side);
        k::vc_prolong8_32(st, nb, lv.n, xf, xc);
        k::vc_residual32_bv(st, nb, A, r, xf, df);
        const int nblk = ztarget ? k::vc_postsmooth32_bv_z(st, nb, As, lv.dinv.p, df, xf, ztarget, c0, c1, r, dot_partial)
                                 : k::vc_postsmooth32_bv(st, nb, As, lv.dinv.p, df, xf, out, c0, c1, r, dot_partial);
        if (dot_blocks) *dot_blocks = dot_partial ? nblk : 0;
        return ztarget ? nullptr : out;
    }
    const int last_deg = lv.is_last ? lv.last_degree : coarse_degree;
    const double last_rat = lv.is_last ? lv.last_ratio : coarse_ratio;
    const double* sv = lv.scaled_ptr();   // shared (sampler) or per-realization (Darcy: fp32 storage) column-scaled values
    const ChebParams cp_last{last_deg, lv.lmax, last_rat, sv};
    const ChebParams cp_smooth{smooth_degree, lv.lmax, smooth_ratio, sv};
    const int flips = last ? cheb_flips(cp_last, true) : cheb_flips(cp_smooth, true) + cheb_flips(cp_smooth, false);
    double* start = lv.xa.p;
    double* other = lv.xb.p;
    if (target) {
        if (flips % 2 == 0) start = target; else other = target;
    }
    if (last) {
        const ChebParams& cp = cp_last;
        return cheb_apply(st, nb, A, lv.dinv.p, lv.bv, cp, r, start, other, lv.d.p, true, dot_partial, dot_blocks, ztarget);
    }
    const ChebParams& cp = cp_smooth;
    double* x = cheb_apply(st, nb, A, lv.dinv.p, lv.bv, cp, r, start, other, lv.d.p, true);
    double* oth = (x == start) ? other : start;
    MgLevel& lc = L[l + 1];
    lc.ensure(nb);
    if (lv.p_oct) {
        k::residual_restrict8(st, nb, A, r, x, lv.res.p, lc.r.p);
    } else {
        k::residual(st, nb, A, r, x, lv.res.p);
        k::spmm(st, nb, view(lv.Pt), lv.res.p, lc.r.p, false, nullptr, nullptr);
    }
    double* xc = cycle(st, nb, l + 1, l0, lc.r.p, nullptr, zvec(), nullptr, nullptr, side);
    if (lv.has_sp && !lv.bv && cheb_fused(cp, false)) {
        // r - S (x + P xc) = res - (S P) xc, in place; then x <- x + P xc + p2(that residual) in one pass
        k::residual(st, nb, view(lv.SP), lv.res.p, xc, lv.res.p);
        const int nblk = cheb_post_from_residual(st, nb, A, lv.dinv.p, lv.bv, cp, r, lv.res.p, x, lv.parent.p, xc,
                                                 dot_partial, ztarget);
        if (dot_blocks) *dot_blocks = dot_partial ? nblk : 0;
        return ztarget ? nullptr : x;
    }
    k::spmm(st, nb, view(lv.P), xc, x, true, nullptr, nullptr);
    return cheb_apply(st, nb, A, lv.dinv.p, lv.bv, cp, r, x, oth, lv.d.p, false, dot_partial, dot_blocks, ztarget);
}

int Multigrid::vcycle(hipStream_t st, int nb, int l0, const double* r, double* xout, double* dot_partial,
                      const std::function<void()>& side) {
    int nblk = 0;
    double* res = cycle(st, nb, l0, l0, r, xout, zvec(), dot_partial, &nblk, side ? &side : nullptr);
    if (res != xout) throw Error(PMC_ERR_INTERNAL, "V-cycle result landed in the wrong buffer");
    return nblk;
}

int Multigrid::vcycle_z(hipStream_t st, int nb, int l0, const double* r, zvec zout, double* dot_partial,
                        const std::function<void()>& side) {
    int nblk = 0;
    cycle(st, nb, l0, l0, r, nullptr, zout, dot_partial, &nblk, side ? &side : nullptr);
    return nblk;
}

MinresWork::~MinresWork() {
    for (auto& kv : graphs)
        if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
}

uint64_t Multigrid::signature(int l0) const {
    uint64_t h = 0x1234;
    for (int l = l0; l < (int)L.size(); ++l) {
        const MgLevel& m = L[l];
        h = hash_ptr(h, m.r.p); h = hash_ptr(h, m.xa.p); h = hash_ptr(h, m.xb.p); h = hash_ptr(h, m.d.p);
        h = hash_ptr(h, m.SP.vals.p); h = hash_ptr(h, m.parent.p);
        h = hash_ptr(h, m.res.p); h = hash_ptr(h, m.vals_bv.p); h = hash_ptr(h, m.vals_scaled.p); h = hash_ptr(h, m.dinv.p);
        h = hash_ptr(h, m.vals32.p); h = hash_ptr(h, m.scaled32.p);
        // the LDS tail descriptors and the transposed copies they point at are re-allocated when a wider batch arrives
        // (ensure_bv_tail_width): a graph captured before that must not be replayed
        h = hash_ptr(h, m.vals_t.p); h = hash_ptr(h, m.scaled_t.p); h = hash_ptr(h, m.dinv_t.p);
        if (l < (int)tail.size()) h = hash_ptr(h, tail[l].p);
    }
    return h;
}

void MinresWork::ensure(int n, int nb, bool z32) {
    const size_t need = (size_t)n * nb;
    v0.ensure(need); v1.ensure(need); u0.ensure(need, z32); u1.ensure(need, z32);
    w0.ensure(need); w1.ensure(need); q.ensure(need);
    // two segments each (see k::DotParts): [0, cap) and [cap, 2 cap)
    partial.ensure((size_t)2 * dot_capacity(n, nb) * nb);
    partial_op.ensure((size_t)2 * dot_capacity(n, nb) * nb);
    if (!state.p) state.alloc(1);
    stage.ensure(k::scal_stage_doubles());
}

// y = a*x + b*y with host scalars, via lincomb on a tiny device constant block would need a copy;
// keep it simple with a dedicated kernel.
__global__ void axpby_kernel(size_t n, double a, const double* __restrict__ x, double b, double* __restrict__ y) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) y[i] = a * x[i] + b * y[i];
}
static void axpby(hipStream_t st, size_t n, double a, const double* x, double b, double* y) {
    if (!n) return;
    const unsigned g = (unsigned)std::min<size_t>((n + 255) / 256, 4096);
    axpby_kernel<<<g, 256, 0, st>>>(n, a, x, b, y);
    PMC_HIP(hipGetLastError());
    count_kernel_launches(1);
}

// Rows x batch width from which a solve uses both streams of its handle (PMC_SPLIT_MIN overrides; 0 = never split)
static size_t split_threshold() {
    static const size_t v = [] {
        const char* e = lab_env("PMC_SPLIT_MIN");
        return e ? (size_t)atoll(e) : (size_t)1500000;
    }();
    return v;
}

static bool stage_on() {
    static const bool v = [] {
        const char* e = lab_env("PMC_SCAL_STAGE");
        return !e || atoi(e) != 0;
    }();
    return v;
}
// Rows up to which a SAMPLER level is solved 32 realizations per launch, from the memory of the device: a solve keeps
// ~110 bytes per row and realization (five fp64 Krylov vectors, the ring of preconditioned vectors, right-hand side,
// solution, the V-cycle's level vectors, staging), and a GPU is expected to carry up to eight handles at once (four manager
// lanes, each with a sampler and a Darcy solver); half the memory is left to the operators and the other plugin.  288 GB:
// 5.1 M rows (config 5's 6.46 M-row level therefore runs 16 wide, as measured necessary in round 3).  Deterministic: the
// TOTAL memory is asked, not what happens to be free, so every lane and every rank of a farm picks the same width.
static size_t sampler_wide_rows(int device) {
    // per DEVICE (the handle's, never "the current one": a manager lane thread that has not activated its context would ask
    // device 0 - and create a primary context there), cached; hipDeviceTotalMem needs no current device
    static std::atomic<size_t> cache[64];
    const int slot = device & 63;
    size_t v = cache[slot].load(std::memory_order_relaxed);
    if (v == 0) {
        size_t total_b = 0;
        hipDevice_t dev;
        if (hipDeviceGet(&dev, device) != hipSuccess || hipDeviceTotalMem(&total_b, dev) != hipSuccess || total_b == 0)
            v = (size_t)5000000;
        else
            v = (size_t)((double)total_b * 0.5 / (110.0 * 32.0 * 8.0));
        cache[slot].store(v, std::memory_order_relaxed);
    }
    return v;
}

int batch_width(size_t rows, bool darcy, int device) {
    static const auto lim = [](const char* name, size_t dflt) {
        const char* e = lab_env(name);
        return e ? (size_t)atoll(e) : dflt;
    };
    // Large levels: the sampler's kernels move more per gather at 32 realizations per launch (config 2, four lanes: 1 741 ->
    // 1 794 samples/s, one lane 1 237 -> 1 358; beyond 5 M rows the vectors of four lanes beside a Darcy solver no longer fit comfortably: config 5); the
    // element-grouped Darcy kernels sweep a slice in two passes at that width and gain nothing, and a Darcy level takes
    // several times the memory per realization
    // PMC_WIDE_ROWS: limit of the 32-wide launches for Darcy levels, and for sampler levels too when it is set (0 = always
    // 16); PMC_S_WIDE_ROWS: the sampler's own limit
    static const size_t l32 = lim("PMC_WIDE_ROWS", 300000),
                        l32s_lab = lim("PMC_S_WIDE_ROWS", lab_env("PMC_WIDE_ROWS") ? l32 : 0),
                        l64 = lim("PMC_W64_ROWS", 150000), l128 = lim("PMC_W128_ROWS", 40000), l256 = lim("PMC_W256_ROWS", 20000),
                        // sampler levels up to 500 k rows: two column groups per launch (config 4's 314 k-row level, four lanes:
                        // hybridized 5 735 -> 6 441, saddle-point 1 474 -> 1 512 realizations/s; a 600 k-row level gains 30 % with
                        // one lane and nothing with four, LAB_NOTES 9.15)
                        // ... and round 5, with the restriction fused and the V-cycle kernels at three waves per SIMD: the
                        // 596 k-row level of config 2 gains at EVERY lane count (hybridized: one lane 2 100 -> 2 720, two
                        // 2 856 -> 3 288, four 3 315 -> 3 482 samples/s; saddle-point: one lane +8.5 %, four +0.4 %): 700 k
                        l64s = lim("PMC_S_W64_ROWS", lab_env("PMC_W64_ROWS") ? l64 : 700000);
    const size_t l32s = l32s_lab ? l32s_lab : (darcy ? 0 : sampler_wide_rows(device));
    if (rows > (darcy ? l32 : l32s)) return 16;
    if (rows <= l256) return 256;
    if (rows <= l128) return 128;
    if (rows <= (darcy ? l64 : l64s)) return 64;
    return 32;
}

// PMC_WX_DEFER=0: one w / x update launch per iteration (A/B switch for k::minres_wx_deferred)
static bool wx_defer_on() {
    static const bool v = [] {
        const char* e = lab_env("PMC_WX_DEFER");
        return !e || atoi(e) != 0;
    }();
    return v;
}

// PMC_LATE_WX=0 keeps the w / x update inside its own iteration also on two streams (A/B switch)
static bool late_wx() {
    static const bool v = [] {
        const char* e = lab_env("PMC_LATE_WX");
        return !e || atoi(e) != 0;
    }();
    return v;
}

MinresResult minres_solve(Ctx& ctx, int nb, const LinOp& A, const PrecFn& prec, const double* b, double* x,
                          bool zero_guess, const pmc_solver_opts& o, MinresWork& w, int x_row0, int x_nrows,
                          const int* x_rows, GraphHint hint) {
    hipStream_t st = ctx.stream;
    const int n = A.n;
    const size_t len = (size_t)n * nb;
    // A handle that has the GPU to itself spreads a solve over two streams; when other handles (manager lanes, bench
    // streams) share the GPU their kernels already fill the gaps and the second stream only adds event traffic
    // (measured at config 2: one lane 826 -> 930 samples/s with the split, four lanes 1328 -> 1273).
    const bool split = o.two_streams == 1 || (o.two_streams == 0 && split_threshold() > 0 && len >= split_threshold() &&
                                              Ctx::contexts_on_device(ctx.device) == 1);
    const Lanes L = ctx.lanes(split);
    const bool z32 = o.precond_storage != PMC_STORAGE_FP64;
    w.ensure(n, nb, z32);
    k::MinresState* S = w.state.p;
    double* v0 = w.v0.p; double* v1 = w.v1.p; zvec u0 = w.u0.v(z32); zvec u1 = w.u1.v(z32);
    double* w0 = w.w0.p; double* w1 = w.w1.p; double* q = w.q.p;

    // v1 = b - A x0
    if (x_rows && !zero_guess) throw Error(PMC_ERR_INTERNAL, "minres: compact solution needs a zero initial guess");
    if (!A.apply_z || (!zero_guess && !A.apply)) throw Error(PMC_ERR_INTERNAL, "minres: operator closures missing");
    if (zero_guess) {
        k::fill(st, x_rows ? (size_t)x_nrows * nb : len, x, 0.0);
        k::copy(st, len, b, v1);
    } else {
        A.apply(L, nb, x, v1, nullptr, nullptr);
        axpby(st, len, 1.0, b, -1.0, v1);
    }
    if (x_row0 < 0 || x_nrows < 0 || (!x_rows && x_row0 + x_nrows > n))
        throw Error(PMC_ERR_INTERNAL, "minres: bad solution row range");
    const size_t xoff = (size_t)x_row0 * nb;
    const size_t seg2 = (size_t)dot_capacity(n, nb) * nb;
    const bool r32 = w.want_r32 && z32;
    if (r32) {
        w.r32.ensure(len);
        k::convert_z(st, nb, n, v1, zvec(reinterpret_cast<double*>(w.r32.p), true), nullptr, nullptr);
        w.r32_valid = true;
    }
    k::DotParts dp = prec(L, nb, v1, u1, w.partial.p, w.partial.p + seg2);
    if (dp.total() == 0) dp = k::DotParts{w.partial.p, k::dot_z(st, nb, n, v1, u1, w.partial.p)};
    const int every = o.check_every > 0 ? o.check_every : 1;
    const bool graphs = hint.key != 0 && o.use_graph != 0 && every == 2;
    const bool late = L.split && !graphs && A.n0 > 0 && A.n0 < n && late_wx();
    // w / x updates of kWxDefer iterations in one pass (see k::minres_wx_deferred): whenever the update is a plain vector
    // kernel on this stream - not the compact index-list update of the Darcy solves (a few rows), not the two-stream
    // schedule (its update already runs beside other work) and not inside a captured graph
    const bool defer = !graphs && !late && !x_rows && wx_defer_on();
    k::minres_init(st, nb, S, dp, o.rel_tol, o.abs_tol, defer ? k::kWxDefer : 1);
    k::fill(st, len, v0, 0.0);
    k::fill(st, len, w0, 0.0);
    k::fill(st, len, w1, 0.0);

    auto coef = [&](size_t off) { return reinterpret_cast<const double*>(reinterpret_cast<const char*>(S) + off); };
    const double* cV0 = coef(offsetof(k::MinresState, cV));
    const double* cV1 = cV0 + kMaxBatch;
    const double* cV2 = cV1 + kMaxBatch;
    const double* cW0 = coef(offsetof(k::MinresState, cW));
    const double* cW1 = cW0 + kMaxBatch;
    const double* cW2 = cW1 + kMaxBatch;
    const double* cW3 = cW2 + kMaxBatch;
    const int* d_nactive = reinterpret_cast<const int*>(reinterpret_cast<const char*>(S) + offsetof(k::MinresState, n_active));

    auto poll = [&]() {
        PMC_HIP(hipMemcpyAsync(ctx.h_flag, d_nactive, sizeof(int), hipMemcpyDeviceToHost, st));
        PMC_HIP(hipStreamSynchronize(st));
        return *ctx.h_flag;
    };

    MinresResult out;
    zvec u2;
    if (late) {
        w.u2.ensure(len, z32);
        u2 = w.u2.v(z32);
    }
    const bool timing = w.op_timer.on && !(hint.key != 0 && o.use_graph != 0);
    // q = A u, d1 = <u, A u>.  The product for iteration i+1 is issued right after the preconditioner of iteration i has
    // written u (both blocks of u are then the most recently written data on the chip), before the scalar recurrences
    // and the w / x update of iteration i, which do not depend on it.
    k::DotParts dp_op;
    auto apply_op = [&](zvec u) {
        if (timing) w.op_timer.begin(st);
        dp_op = A.apply_z(L, nb, u, q, w.partial_op.p, w.partial_op.p + seg2);
        if (timing) w.op_timer.end(st);   // + an empty bracket: what one event record costs on this stream
    };
    // one MINRES iteration with explicit roles of the ping-pong vectors; on entry q = A u1_ and its dot are in place
    // (the first scalar step of an iteration - alpha and the Lanczos coefficients from <u, Au> - has already been done: by
    // the prologue for the first iteration, by the fused scalar launch of the previous iteration otherwise)
    //
    // Two-stream schedule (`late`): the w / x update of an iteration needs nothing but that iteration's scalars, and
    // nothing needs it before the solve ends, so it is issued one iteration late on the second stream, behind the u-rows
    // of the next Lanczos update and ahead of the M-block of the preconditioner; the first stream carries the s-rows of
    // the update and the V-cycle, whose coarse levels leave most of the chip idle.  Same kernels, same arguments, same
    // results; the preconditioned vector the update reads must survive one more iteration, hence three of them.
    struct PendingWx { zvec u; double *w0 = nullptr, *w1 = nullptr; } pend;
    auto wx = [&](hipStream_t s, zvec u_, double* w0_, double* w1_) {
        if (x_rows) k::minres_wx_idx(s, nb, x_nrows, x_rows, cW0, u_, cW1, w0_, cW2, w1_, cW3, x);
        else k::minres_wx(s, nb, x_nrows, cW0, u_ + xoff, cW1, w0_, cW2, w1_, cW3, x + xoff);
    };
    auto flush_wx = [&](hipStream_t s) {
        if (pend.u) wx(s, pend.u, pend.w0, pend.w1);
        pend.u = zvec();
    };
    auto iteration = [&](zvec u0_, zvec u1_, double* v0_, double* v1_, double* w0_, double* w1_, bool last) {
        if (late) {
            const size_t off = (size_t)A.n0 * nb;
            L.fork();
            k::lincomb3(L.aux, nb, A.n0, cV0, q, cV1, v1_, cV2, v0_);
            flush_wx(L.aux);
            k::lincomb3(st, nb, n - A.n0, cV0, q + off, cV1, v1_ + off, cV2, v0_ + off);
        } else {
            k::lincomb3(st, nb, n, cV0, q, cV1, v1_, cV2, v0_, r32 ? w.r32.p : nullptr);
            w.r32_valid = r32;
        }
        k::DotParts d2 = prec(L, nb, v0_, u0_, w.partial.p, w.partial.p + seg2);   // joins the second stream
        if (d2.total() == 0) d2 = k::DotParts{w.partial.p, k::dot_z(st, nb, n, v0_, u0_, w.partial.p)};
        if (!last) {
            apply_op(u0_);
            k::minres_scal21(st, nb, S, d2, dp_op, stage_on() ? w.stage.p : nullptr);
        } else {
            k::minres_scal2(st, nb, S, d2);
        }
        if (defer) return;                       // the caller queues the update (k::WxDeferred)
        if (late && !last) pend = PendingWx{u1_, w0_, w1_};
        else wx(st, u1_, w0_, w1_);
    };
    // two iterations return every vector to its original role
    auto pair = [&]() {
        iteration(u0, u1, v0, v1, w0, w1, false);
        iteration(u1, u0, v1, v0, w1, w0, false);
    };
    int it = 0;
    int n_active = poll();
    // Convergence polls drain the stream; iteration counts of one configuration barely move between batches, so the polls
    // start a few iterations before the count the previous solve of this configuration needed.  Converged columns are
    // frozen on the device (zero update coefficients), so iterating past convergence never changes a result.
    const uint64_t hint_key = hint.key ? hash_mix(hint.key, zero_guess ? 1 : 2) : 0;
    int first_poll = 0;
    if (hint_key) {
        auto f = w.iter_hint.find(hint_key);
        if (f != w.iter_hint.end()) first_poll = std::max(0, f->second - 4);
    }
    const int poll_step = graphs ? 2 : every;
    const int mid_poll = first_poll >= 8 ? (first_poll / 2 / poll_step) * poll_step : -1;   // see the sparse poll below
    if (n_active > 0 && o.max_iter > 0) {
        apply_op(u1);
        k::minres_scal1(st, nb, S, dp_op);
    }
    if (graphs) {
        // hipGraph path: the first pair runs eagerly (it also performs every lazy allocation), later pairs replay one
        // captured graph - ~70 kernel launches become a single hipGraphLaunch, which is what the launch-bound small
        // levels need; the convergence poll stays between replays.
        if (n_active > 0 && it + 2 <= o.max_iter) {
            pair();
            it += 2;
            n_active = poll();
        }
        if (n_active > 0 && it + 2 <= o.max_iter) {
            uint64_t sig = hash_mix(hint.sig, (uint64_t)nb);
            sig = hash_mix(sig, (uint64_t)n);
            for (const void* p : {(const void*)b, (const void*)x, (const void*)v0, (const void*)v1, (const void*)u0.p,
                                  (const void*)u1.p, (const void*)w0, (const void*)w1, (const void*)q,
                                  (const void*)w.partial.p, (const void*)w.partial_op.p, (const void*)S, (const void*)w.stage.p,
                                  (const void*)x_rows, (const void*)(r32 ? w.r32.p : nullptr)})
                sig = hash_ptr(sig, p);
            sig = hash_mix(sig, ((uint64_t)x_row0 << 32) ^ (uint64_t)x_nrows);
            sig = hash_mix(sig, L.split ? 7 : 3);
            MinresWork::GraphEntry& ge = w.graphs[hint.key];
            if (ge.exec && ge.sig != sig) {
                PMC_HIP(hipGraphExecDestroy(ge.exec));
                ge.exec = nullptr;
            }
            if (!ge.exec) {
                hipGraph_t graph = nullptr;
                PMC_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
                try {
                    pair();
                } catch (...) {
                    (void)hipStreamEndCapture(st, &graph);
                    if (graph) (void)hipGraphDestroy(graph);
                    throw;
                }
                PMC_HIP(hipStreamEndCapture(st, &graph));
                PMC_HIP(hipGraphInstantiate(&ge.exec, graph, nullptr, nullptr, 0));
                PMC_HIP(hipGraphDestroy(graph));
                ge.sig = sig;
            }
            while (n_active > 0 && it + 2 <= o.max_iter) {
                PMC_HIP(hipGraphLaunch(ge.exec, st));
                it += 2;
                if (it >= first_poll || it == mid_poll) n_active = poll();
            }
        }
        if (n_active > 0 && it < o.max_iter) {   // odd max_iter: one last eager iteration
            iteration(u0, u1, v0, v1, w0, w1, true);
            ++it;
            n_active = poll();
        }
    } else if (defer) {
        // ring of kWxDefer + 1 preconditioned vectors: the pending iterations' z stay alive until their updates are flushed
        constexpr int R = k::kWxDefer + 1;
        if ((int)w.ring.size() < R - 2) w.ring.resize(R - 2);
        zvec ub[R];
        ub[0] = u1;
        ub[1] = u0;
        for (int j = 2; j < R; ++j) {
            w.ring[(size_t)j - 2].ensure(len, z32);
            ub[j] = w.ring[(size_t)j - 2].v(z32);
        }
        int c = 0;                                   // ub[c] holds the preconditioned vector of the current Lanczos vector
        k::WxDeferred pending{};
        pending.f32 = z32;
        auto flush = [&]() {
            k::minres_wx_deferred(st, nb, x_nrows, S, pending, w0, w1, x + xoff);
            pending.cnt = 0;
        };
        while (n_active > 0 && it < o.max_iter) {
            ++it;
            iteration(ub[(c + 1) % R], ub[c], v0, v1, w0, w1, it == o.max_iter);
            pending.u[pending.cnt] = (ub[c] + xoff).p;
            pending.slot[pending.cnt] = (it - 1) % k::kWxDefer;
            if (++pending.cnt == k::kWxDefer) flush();
            c = (c + 1) % R;
            std::swap(v0, v1);
            if ((it % every == 0 && (it >= first_poll || it == mid_poll)) || it == o.max_iter) n_active = poll();
        }
        flush();
    } else {
        while (n_active > 0 && it < o.max_iter) {
            ++it;
            iteration(u0, u1, v0, v1, w0, w1, it == o.max_iter);
            if (late) {                      // u1 (read by the pending update) stays untouched for one more iteration
                zvec t = u1;
                u1 = u0;
                u0 = u2;
                u2 = t;
            } else {
                std::swap(u0, u1);
            }
            std::swap(v0, v1);
            std::swap(w0, w1);
            // one sparse poll half way to the hinted count: a solve that converges much earlier than its predecessor
            // (warm start after a cold solve, an easier batch) stops there instead of running blind up to the hint
            if ((it % every == 0 && (it >= first_poll || it == mid_poll)) || it == o.max_iter) n_active = poll();
        }
        flush_wx(st);
    }
    out.iterations = it;
    static_assert(sizeof(k::MinresState) <= Ctx::kHostScratch * sizeof(double), "pinned scratch too small");
    PMC_HIP(hipMemcpyAsync(ctx.h_scal, S, sizeof(k::MinresState), hipMemcpyDeviceToHost, st));
    PMC_HIP(hipStreamSynchronize(st));
    if (timing) w.op_timer.harvest();
    k::MinresState hs;
    std::memcpy(&hs, ctx.h_scal, sizeof(hs));
    int max_it = 0;
    for (int kcol = 0; kcol < nb; ++kcol) {
        pmc_stats& s = out.col[kcol];
        s.solve_ms = s.setup_ms = 0.0;   // filled by the caller from its phase events
        s.iterations = hs.iters[kcol];
        max_it = std::max(max_it, s.iterations);
        s.initial_norm = hs.eta0[kcol];
        s.final_norm = std::fabs(hs.eta[kcol]);
        s.converged = (hs.flag[kcol] == 0 && hs.active[kcol] == 0 && s.final_norm <= hs.goal[kcol]) ? 1 : 0;
        if (hs.flag[kcol] != 0) s.converged = -1;   // indefinite preconditioner / NaN
    }
    if (hint_key) w.iter_hint[hint_key] = max_it;
    return out;
}

double gershgorin_scaled(const HostCsr& A, const std::vector<double>& diag, double* lmin) {
    double lmax = 0.0, lo = 2.0;
    for (int i = 0; i < A.nrows; ++i) {
        double s = 0.0;
        for (int p = A.rowptr[i]; p < A.rowptr[i + 1]; ++p) s += std::fabs(A.vals[p]);
        lmax = std::max(lmax, s / diag[i]);
        lo = std::min(lo, 2.0 - s / diag[i]);
    }
    if (lmin) *lmin = lo;
    return lmax;
}

}  // namespace pmc
