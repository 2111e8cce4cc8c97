// Krylov solver and preconditioner building blocks (host orchestration of kernels.hip).
#pragma once
#include <functional>
#include <map>
#include <vector>

#include "kernels.hpp"

namespace pmc {

// Fixed-degree Chebyshev polynomial in D^-1 A (SPD A, positive diagonal scaling D): a fixed SPD
// linear operator, i.e. a legal MINRES preconditioner block / multigrid smoother.  Replaces the
// reference's sequential hypre l1-Gauss-Seidel (CreateSamplerParameterList.hpp:80-93) - the XML
// schema itself lists Chebyshev / l1-Jacobi as hypre smoother alternatives (example_parameters.xml:789-801).
struct ChebParams {
    int degree = 3;
    double lmax = 1.0;   // upper bound of spec(D^-1 A)
    double ratio = 8.0;  // interval [lmax/ratio, lmax]
    // column-scaled values A D^-1 on A's pattern (shared-value matrices only); enables the one-pass
    // degree-2 kernel for zero initial guesses
    const double* scaled_vals = nullptr;
};
// coefficients of the one-pass degree-2 Chebyshev polynomial on [lmax/ratio, lmax]: x2 = dinv (c0 r - c1 A dinv r)
inline void cheb2_coefficients(double lmax, double ratio, double* c0, double* c1) {
    const double lmin = lmax / ratio;
    const double theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin), sigma = theta / delta;
    const double rho_old = 1.0 / sigma, rho1 = 1.0 / (2.0 * sigma - rho_old);
    *c0 = (1.0 + rho1 * rho_old) / theta + 2.0 * rho1 / delta;
    *c1 = 2.0 * rho1 / (delta * theta);
}
// degree 2 with column-scaled values runs as one polynomial pass (on r from a zero guess, on the residual otherwise)
inline bool cheb_fused(const ChebParams& cp, bool /*zero_guess*/) { return cp.degree == 2 && cp.scaled_vals; }
// Runs `degree` steps.  xa holds the initial guess (ignored when zero_guess); the iterate ping-pongs
// between xa and xb; returns the buffer holding the result.  d is work space.
// dot_partial != nullptr: the last step also writes per-block partials of <r, result>; *dot_blocks gets their count.
double* cheb_apply(hipStream_t st, int nb, const SellView& A, const double* dinv, bool dinv_bv, const ChebParams& cp,
                   const double* r, double* xa, double* xb, double* d, bool zero_guess,
                   double* dot_partial = nullptr, int* dot_blocks = nullptr, zvec zlast = zvec());
// zlast non-null: the last kernel writes the result there in zvec storage (rounded before the fused dot) and null is
// returned (degree 1 has no typed kernel: one extra rounding pass)
// The same from a zero guess with the result in zvec storage (a preconditioner block of a MINRES solve): the last kernel
// writes z itself; earlier steps of an unfused polynomial iterate in the fp64 scratch xa / xb.  Returns the partial-block
// count of <r, z> (dot_partial != nullptr).
int cheb_apply_z(hipStream_t st, int nb, const SellView& A, const double* dinv, bool dinv_bv, const ChebParams& cp,
                 const double* r, zvec z, double* xa, double* xb, double* d, double* dot_partial);
// Post-smoothing of a V-cycle level from an already formed residual `res` = r - A (x + P xc) without x + P xc in memory:
// x <- x + xc[parent] + p2(res); degree 2 with scaled values only.  Returns the partial-block count of <r, x>.
int cheb_post_from_residual(hipStream_t st, int nb, const SellView& A, const double* dinv, bool dinv_bv,
                            const ChebParams& cp, const double* r, const double* res, double* x, const int* parent,
                            const double* xc, double* dot_partial, zvec zout = zvec());
// number of buffer flips cheb_apply performs
inline int cheb_flips(const ChebParams& cp, bool zero_guess) {
    if (cheb_fused(cp, zero_guess)) return 0;
    return zero_guess ? cp.degree - 1 : cp.degree;
}

// One level of the Schur-complement multigrid hierarchy.
struct MgLevel {
    int n = 0;
    Sell S;                      // pattern (+ shared values for the sampler)
    DevBuf<double> dinv;         // shared: n ; batched: n*nb
    DevBuf<double> vals_bv;      // batched values (Darcy): nslots*nb, sized by the owner for the width in use
    DevBuf<double> vals_scaled;  // S D^-1 on S's pattern (shared, or nslots*nb when bv): one-pass pre-smoothing
    // small per-realization levels: column-major copies [nb][...] for the LDS tail kernel (allocated by
    // Multigrid::enable_bv_tail / ensure_bv_tail_width, refreshed by refresh_bv_tail after the level's values changed)
    DevBuf<double> vals_t, scaled_t, dinv_t;
    // f32 (bv levels of a preconditioner hierarchy): the SpMM-type kernels of the V-cycle read fp32 copies of the values
    // (vals32) and of the column-scaled values (scaled32) - half the bytes of their dominant stream; vals_bv (fp64) stays the
    // master the Galerkin refresh and the diagonals are computed from, and vals_scaled is not kept at all.
    DevBuf<float> vals32, scaled32;
    bool f32 = false;
    bool bv = false;
    double lmax = 2.0;
    // Coarsest-level treatment: `is_last` levels end the recursion with a Chebyshev solve of degree
    // last_degree on [lmax/last_ratio, lmax].  The last supplied level always is one; a level whose
    // diagonally scaled spectrum is provably narrow (reaction-dominated sampler levels) is one too, so
    // the V-cycle does not descend into launch-latency-bound tiny levels.
    bool is_last = false;
    int last_degree = 12;
    double last_ratio = 100.0;
    Sell P, Pt;                  // to/from the next coarser level (absent on the last)
    // Injection-type prolongators (one unit entry per row: P0 on nested meshes) with shared values: the coarse
    // correction never materialises on this level.  SP = S P lets the post-smoothing residual be formed from the
    // pre-restriction one, r - S (x + P xc) = res - SP xc, and `parent` adds xc[parent[i]] inside the smoother pass.
    Sell SP;
    DevBuf<int> parent;
    bool has_sp = false;
    // P has exactly the children 8 i .. 8 i + 7 of parent i with unit weights (uniform refinement with contiguous
    // children): the restriction is fused into the residual kernel (k::residual_restrict8)
    bool p_oct = false;
    // aggregation level whose rows were renumbered by agg_pack_rows (every aggregate a run of consecutive rows inside one
    // slice): the restriction is fused into the residual kernel through per-slice segment tables
    bool p_agg = false;
    DevBuf<int> seg_ptr, seg_cid, seg_pos;
    DevBuf<double> r, xa, xb, d, res;
    DevBuf<double> ainv;         // last level of a small shared-value hierarchy: dense inverse (see TailLevelDev::ainv)
    // an inner level of a few hundred rows: dense inverse for launches of at most Multigrid::dense_nb realizations, which
    // end their cycle there with one multi-workgroup matrix-vector product instead of the one-workgroup LDS tail
    DevBuf<double> dense_inv;
    // the same launches on the inner levels above it: the operator (values: the level's own and S D^-1) and S P with every row
    // cut into 2^split_log2 / 2^sp_split_log2 pieces (csr_split_rows; 0 = not built) for the row-split V-cycle kernels
    int split_log2 = 0, sp_split_log2 = 0;
    Sell S_split, SP_split;
    DevBuf<double> scaled_split;
    void ensure(int nb);
    SellView sview() const { return bv ? (f32 ? view_bv32(S, vals32.p) : view_bv(S, vals_bv.p)) : view(S); }
    // column-scaled values in the storage sview() uses
    const double* scaled_ptr() const { return (bv && f32) ? reinterpret_cast<const double*>(scaled32.p) : vals_scaled.p; }
};

struct OpTimer;
struct Multigrid {
    std::vector<MgLevel> L;      // [0] finest
    int smooth_degree = 2;
    double smooth_ratio = 4.0;
    int coarse_degree = 12;
    double coarse_ratio = 100.0;
    // Small-level tail: tail[l] (if non-empty) describes levels l.. that together fit one workgroup's LDS; the V-cycle
    // then finishes in a single kernel from level l on.  Call build_tails() once all levels are set up.
    std::vector<DevBuf<TailParams>> tail;
    std::vector<size_t> tail_lds;
    bool use_tail = true;
    // pmc_solver_opts.precond_storage: the intermediates of a level (iterate, residuals) of the structured hierarchies live
    // in fp32 (k::vc_* kernels) or - false - in fp64 like everything else
    bool f32_intermediates = true;
    // the fp32-intermediate kernels also on levels whose indicator prolongator is NOT over groups of 8 consecutive rows
    // (aggregation hierarchies: the restriction is then a separate product with P^T)
    bool f32_any_injection = false;
    // launches of at most this many realizations start the LDS tail one level further down when the tail's first level has
    // more than 4 096 rows (0 = never).  Set by the owner for the hierarchy it was measured on - the aggregation hierarchy of
    // the hybridized sampler (LAB_NOTES 9.16); the structured and per-realization (Darcy) hierarchies keep 0, so a
    // realization of a ragged remainder chunk takes the same kernels as one of a full launch there.
    int tail_later_nb = 0;
    int dense_nb = 0;            // see MgLevel::dense_inv (0: never)
    // fp32 copy of the right-hand side of the NEXT cycle's top level (set by the caller right before vcycle_z, consumed and
    // cleared by it): the pre-smoothing and residual kernels of that level gather / read it instead of the fp64 vector.  The
    // MINRES loop provides it for free-standing cost of one more fp32 stream in the Lanczos update (MinresWork::r32).
    const float* r32_top = nullptr;
    // would a cycle from level l0 at width nb read such a copy?  (the fp32-intermediate path of a shared-value level whose
    // restriction is not the 8-children tree, not inside the LDS tail)
    bool top_reads_r32(int l0, int nb) const {
        if (l0 < 0 || l0 + 1 >= (int)L.size()) return false;
        const MgLevel& lv = L[(size_t)l0];
        const bool tail_later = nb <= tail_later_nb && lv.n > 4096 && l0 + 1 < (int)tail.size() && tail[(size_t)l0 + 1].p;
        const bool tail_here = use_tail && l0 < (int)tail.size() && tail[(size_t)l0].p && !tail_later;
        return !tail_here && !lv.is_last && !lv.bv && lv.has_sp && !lv.p_oct && f32_any_injection && smooth_degree == 2 &&
               lv.vals_scaled.p && f32_intermediates;
    }
    // in-situ timing (HIP events) of the top level's post-smoothing kernel - the largest single kernel of a cycle on an
    // aggregation hierarchy; set per solve by the owner, null = off
    OpTimer* smooth_timer = nullptr;
    void build_tails(hipStream_t st);
    // per-realization hierarchies: give every level of at most max_rows rows transposed value copies so that
    // build_tails can include them; refresh_bv_tail(nb) re-fills the copies (call after every numeric refresh)
    void enable_bv_tail(int max_rows = 8192);
    // the transposed copies hold bv_tail_width realizations; a wider batch re-allocates them and rebuilds the tail
    // descriptors (which carry their addresses) - once per handle and width
    int bv_tail_width = 0;
    void ensure_bv_tail_width(hipStream_t st, int nb);
    void refresh_bv_tail(hipStream_t st, int nb, int first_level = 0);
    // hash of the work-buffer pointers a V-cycle from level l0 touches (for GraphHint::sig)
    uint64_t signature(int l0) const;
    // x = V(r) starting at level l0 with zero initial guess; result written to xout (n(l0)*nb).
    // dot_partial != nullptr: also per-block partials of <r, xout>; returns their count.
    // side (optional) is called exactly once, just before the kernels of the bottom level of the V are enqueued (the LDS
    // tail: one workgroup per realization): work issued from it on another stream then runs beside the least parallel
    // part of the cycle instead of competing with the bandwidth-bound fine levels or stretching the short kernels of
    // the intermediate ones (measured: a 9 us level-1 residual takes 37 us next to the M-block polynomial).
    int vcycle(hipStream_t st, int nb, int l0, const double* r, double* xout, double* dot_partial = nullptr,
               const std::function<void()>& side = nullptr);
    // the same with the result in zvec storage (S-block of a MINRES preconditioner): the last kernel of the cycle writes it
    int vcycle_z(hipStream_t st, int nb, int l0, const double* r, zvec zout, double* dot_partial = nullptr,
                 const std::function<void()>& side = nullptr);

  private:
    // ztarget non-null (top level of vcycle_z only): the result goes there and the return value is null
    double* cycle(hipStream_t st, int nb, int l, int l0, const double* r, double* target, zvec ztarget, double* dot_partial,
                  int* dot_blocks, const std::function<void()>* side);
};

// Abstract pieces MINRES needs.
struct LinOp {
    int n = 0;
    // rows [0, n0) and [n0, n) are the two diagonal blocks of the preconditioner (0: unknown); a solve that owns both
    // streams then updates the two blocks of the Lanczos vector on the stream that preconditions them
    int n0 = 0;
    // y = A x ; when dot_partial != nullptr also per-block partials of <x, A x>; returns the number of
    // partial blocks written (none when dot_partial == nullptr); two row blocks on two streams may use a segment each
    // (independent row blocks may run between L.fork() and L.join(); ordered on L.main again on return)
    std::function<k::DotParts(const Lanes& L, int nb, const double* x, double* y, double* dot_partial,
                              double* dot_partial2)> apply;
    // the same product from a preconditioned vector (zvec storage): every product inside the MINRES loop
    std::function<k::DotParts(const Lanes& L, int nb, zvec x, double* y, double* dot_partial,
                              double* dot_partial2)> apply_z;
};
// z = B^-1 r.  When dot_partial != nullptr the preconditioner may fuse <r, z> into its last kernels and
// return the number of partial blocks it wrote (0 = not computed, the solver then runs a separate dot).
// The preconditioner receives both streams of the solve: independent blocks may run between L.fork() and L.join(); on
// return everything must be ordered on L.main again.
// dot_partial / dot_partial2 (each dot_capacity(n) blocks) receive the fused <r, z>: return where the partials are
// (total() == 0: not computed, the solver then runs a separate dot).
// z is a zvec (see kernels.hpp); <r, z> is the inner product with the STORED values.
using PrecFn = std::function<k::DotParts(const Lanes& L, int nb, const double* r, zvec z, double* dot_partial,
                                         double* dot_partial2)>;

// Caller-side identity of one solver configuration for hipGraph reuse: `key` names the configuration (handle, level,
// batch width, ...), `sig` hashes every device pointer the caller's operator / preconditioner closures use, so a
// reallocation anywhere invalidates the cached graph.  key == 0: never use graphs.
struct GraphHint {
    uint64_t key = 0, sig = 0;
};
inline uint64_t hash_mix(uint64_t h, uint64_t v) {
    h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
    return h;
}
inline uint64_t hash_ptr(uint64_t h, const void* p) { return hash_mix(h, (uint64_t)(uintptr_t)p); }

// In-situ kernel timing with HIP events on the launching stream: begin / end bracket one launch, and an EMPTY bracket is
// recorded right behind it (what the event pair itself costs on that stream); harvest() - after the stream has been
// synchronised - accumulates both.
struct OpTimer {
    bool on = false;
    std::vector<hipEvent_t> ev;
    size_t used = 0;
    double ms = 0.0, gap_ms = 0.0;
    int64_t launches = 0;
    OpTimer() = default;
    OpTimer(const OpTimer&) = delete;
    OpTimer& operator=(const OpTimer&) = delete;
    ~OpTimer() {
        for (hipEvent_t e : ev) (void)hipEventDestroy(e);
    }
    void begin(hipStream_t st) {
        while (ev.size() < used + 3) {
            hipEvent_t e;
            PMC_HIP(hipEventCreate(&e));
            ev.push_back(e);
        }
        PMC_HIP(hipEventRecord(ev[used], st));
    }
    void end(hipStream_t st) {
        PMC_HIP(hipEventRecord(ev[used + 1], st));
        PMC_HIP(hipEventRecord(ev[used + 2], st));
        used += 3;
    }
    void harvest() {
        for (size_t e = 0; e + 2 < used; e += 3) {
            float a = 0.f, g = 0.f;
            PMC_HIP(hipEventElapsedTime(&a, ev[e], ev[e + 1]));
            PMC_HIP(hipEventElapsedTime(&g, ev[e + 1], ev[e + 2]));
            ms += a;
            gap_ms += g;
            ++launches;
        }
        used = 0;
    }
    void clear() { ms = gap_ms = 0.0; launches = 0; }
};

struct MinresWork {
    DevBuf<double> v0, v1, w0, w1, q, partial;
    ZBuf u0, u1;                             // preconditioned vectors (storage: pmc_solver_opts.precond_storage)
    DevBuf<double> stage;                    // first-stage sums of the scalar kernel (k::minres_scal21)
    ZBuf u2;                                 // third preconditioned vector: only when the w / x update runs one iteration late
    std::vector<ZBuf> ring;                  // ring of the deferred w / x update (with u0, u1: kWxDefer + 1 vectors)
    DevBuf<double> partial_op;               // partials of the operator's fused <u, Au> (the preconditioner's live in `partial`)
    // want_r32 (set by the owner of the solve): every vector handed to the preconditioner is also kept in fp32 (r32); the
    // solver raises r32_valid right before each preconditioner call, the preconditioner's closure consumes the flag - a
    // preconditioner applied outside the loop (pmc_sampler_apply_preconditioner) never sees a stale copy
    DevBuf<float> r32;
    bool want_r32 = false, r32_valid = false;
    std::map<uint64_t, int> iter_hint;       // per solver configuration: iterations its previous solve needed
    DevBuf<k::MinresState> state;
    struct GraphEntry {
        uint64_t sig = 0;
        hipGraphExec_t exec = nullptr;
    };
    std::map<uint64_t, GraphEntry> graphs;   // two MINRES iterations per graph, see minres_solve
    // optional in-situ timing of the operator launches (K5) with HIP events on the solve's own stream
    OpTimer op_timer;
    MinresWork() = default;
    MinresWork(const MinresWork&) = delete;
    MinresWork& operator=(const MinresWork&) = delete;
    ~MinresWork();
    void ensure(int n, int nb, bool z32);
};

struct MinresResult {
    pmc_stats col[kMaxBatch];
    int iterations = 0;   // iterations executed (max over columns)
};

// Preconditioned MINRES on nb right-hand sides at once.  x holds the initial guess on entry when
// !zero_guess.  b, x: n*nb interleaved device vectors.
// Only rows [x_row0, x_row0 + x_nrows) of the solution are updated (the samplers need the s-block only,
// which saves two thirds of the w / x vector traffic); pass 0, A.n for the full solution.  With x_rows != nullptr
// (device index list of x_nrows rows, zero initial guess only) x is a COMPACT [x_nrows][nb] vector holding just those
// rows of the solution (Darcy: the support of the observation functional).
MinresResult minres_solve(Ctx& ctx, int nb, const LinOp& A, const PrecFn& prec, const double* b, double* x,
                          bool zero_guess, const pmc_solver_opts& o, MinresWork& w, int x_row0, int x_nrows,
                          const int* x_rows = nullptr, GraphHint hint = GraphHint());

// Gershgorin bounds of spec(D^-1 A) for D = diag(A) on the host (setup): returns lmax, sets *lmin
// (may be <= 0 when a row is not strictly diagonally dominant)
double gershgorin_scaled(const HostCsr& A, const std::vector<double>& diag, double* lmin = nullptr);

}  // namespace pmc
