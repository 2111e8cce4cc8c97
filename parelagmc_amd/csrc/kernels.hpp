// Launchers of the CDNA4 kernels (kernels.hip).  All vectors are "interleaved" batches:
// row i, column (realization) k of a batch of nb lives at v[i*nb + k], nb in {1,2,4,8,16,32} or - as 2, 4, 8 column groups
// of 32 handled by one launch (gridDim.y) - 64, 128, 256.
#pragma once
#include "common.hpp"

namespace pmc {

// The PRECONDITIONED Krylov vectors z = B^-1 v of the MINRES solves, in the storage pmc_solver_opts.precond_storage asks
// for: fp32 (default) or fp64.  They are the method's search directions: whatever is stored IS the direction, every inner
// product and recurrence uses the stored values, so rounding them to fp32 perturbs the preconditioner by 6e-8 relative and
// nothing else (iteration counts and converged fields unchanged to 1e-12, true residuals at full size:
// tests/test_gpu_round4.py); each z is written once and read three times per iteration.  Lanczos vectors, operator products,
// w, x and every scalar are fp64 whatever the option says.  The launchers below take the vector with its storage type at
// run time and pick the kernel instantiation (the fp64 ones are the kernels every other caller uses anyway).
struct zvec {
    void* p = nullptr;
    bool f32 = true;
    zvec() = default;
    zvec(void* p_, bool f32_) : p(p_), f32(f32_) {}
    size_t elem_bytes() const { return f32 ? 4 : 8; }
    zvec operator+(size_t elems) const { return zvec(static_cast<char*>(p) + elems * elem_bytes(), f32); }
    explicit operator bool() const { return p != nullptr; }
    template <typename T>
    T* as() const { return static_cast<T*>(p); }
};
// device buffer behind a zvec: `count` entries in either storage
struct ZBuf {
    DevBuf<float> b;
    void ensure(size_t count, bool f32) { b.ensure(f32 ? count : 2 * count); }
    zvec v(bool f32) const { return zvec(b.p, f32); }
};

// Device view of a SELL-64 matrix.  vals == shared values (nslots) or, when bv, per-realization
// values (nslots*NB, interleaved like vectors).
struct SellView {
    int nrows = 0, nslices = 0;
    const int* slice_off = nullptr;
    const int* cols = nullptr;
    const double* vals = nullptr;
    bool bv = false;
    const int* sched = nullptr;   // optional slice processing order
    int tag = 0;                  // 1 = block saddle-point operator (own kernel instantiation / profile row)
    int ncols_hint = 0;           // number of columns (rows of x): kernels with 32-bit gather offsets check it
    bool diag_last = false;       // see Sell::diag_last
    bool f32 = false;             // bv only: vals points at fp32 values (per-realization values of a PRECONDITIONER matrix)
    // > 0: the matrix was built from csr_split_rows(A, split_log2) - nrows counts the pieces; the k::vc_* launchers then
    // start their row-split instantiations (launches of at most 8 realizations), which add the pieces of a row
    int split_log2 = 0;
};
inline SellView view(const Sell& S) {
    return {S.nrows, S.nslices, S.slice_off.p, S.cols.p, S.vals.p, false, S.sched.p, 0, S.ncols, S.diag_last};
}
inline SellView view_split(const Sell& S, int sl) {
    SellView v = view(S);
    v.split_log2 = sl;
    return v;
}
inline SellView view_bv(const Sell& S, const double* vals) {
    return {S.nrows, S.nslices, S.slice_off.p, S.cols.p, vals, true, S.sched.p, 0, S.ncols};
}
inline SellView view_bv32(const Sell& S, const float* vals) {
    SellView v = view_bv(S, reinterpret_cast<const double*>(vals));
    v.f32 = true;
    return v;
}

// Element-grouped SELL view of a per-realization mass matrix M(k) = sum_e c(k_e) M_e: every row stores its entries in
// two groups of `gw` slice columns, one per element the row's dof belongs to; the values of a group are the SHARED
// element-matrix entries w, and the realization enters only through the two per-row coefficients coef[e12[row]][k].
// A row product is  c1 * sum_g1 w x  +  c2 * sum_g2 w x : the matrix costs 12 B per stored entry for the whole batch
// instead of 8 B per entry per realization.
struct EgView {
    int nrows = 0, nslices = 0, gw = 0;
    const int* cols = nullptr;     // [nslices][2 gw][64]
    const double* w = nullptr;     // same layout
    const int* e12 = nullptr;      // [nrows][2] coefficient rows (n_elem = constant-one row for eliminated dofs)
};

// capacity (in blocks of nb doubles) of a partial-sum buffer for (fused) dots over nrows rows of a batch of nb columns:
// allocate dot_capacity(nrows, nb) * nb doubles
int dot_capacity(int nrows, int nb);
// kernels this process has launched through the library so far (all handles, all threads): launch-rate diagnostics
uint64_t kernel_launch_count();
void count_kernel_launches(int n);

namespace k {

// The w / x updates of up to kWxDefer consecutive MINRES iterations are applied in ONE pass over the vectors
// (minres_wx_deferred): nothing reads w or x before the solve ends, and every update needs only its own iteration's
// coefficients and preconditioned vector.  One pass costs (kWxDefer + 5) vector streams instead of 6 kWxDefer.
// (round 5: 8 instead of 4 - per iteration (8 x 4 + 6 x 8) / 8 = 10 bytes per entry instead of 16 with fp32-stored u; the ring
// of preconditioned vectors grows from 5 to 9)
#ifndef PMC_WX_DEFER_N
#define PMC_WX_DEFER_N 8
#endif
static constexpr int kWxDefer = PMC_WX_DEFER_N;
struct WxDeferred {
    const void* u[kWxDefer];    // preconditioned vectors of the pending iterations (rows of the maintained block), oldest first
    int slot[kWxDefer];          // coefficient set (MinresState::cW ring) of each
    int cnt;
    bool f32;                    // storage of the u vectors
};

// Device-resident MINRES scalars, one entry per batch column.
struct MinresState {
    double beta[kMaxBatch], beta_old[kMaxBatch], eta[kMaxBatch], eta0[kMaxBatch];
    double gamma0[kMaxBatch], gamma1[kMaxBatch], sigma0[kMaxBatch], sigma1[kMaxBatch];
    double goal[kMaxBatch], alpha[kMaxBatch], delta[kMaxBatch], rho2[kMaxBatch], rho3[kMaxBatch];
    double cV[3][kMaxBatch];  // v_new = cV0*q + cV1*v1 + cV2*v0
    // w_new = cW0*u1 + cW1*w0 + cW2*w1 ; x += cW3*w_new.  Iteration i (0-based) writes set i % ring: ring == 1 when the
    // update follows its iteration at once, kWxDefer when updates are deferred
    double cW[kWxDefer][4][kMaxBatch];
    int active[kMaxBatch], iters[kMaxBatch], flag[kMaxBatch];
    int n_active, it, ring;
};

// y = A x (accumulate=false) or y += A x.  If dot_partial != nullptr (accumulate must be false) also
// writes per-block partial sums of <dot_with, A x>; returns the number of partial blocks written.
int spmm(hipStream_t st, int nb, const SellView& A, const double* x, double* y, bool accumulate,
          double* dot_partial, const double* dot_with);
// the same product from a vector in zvec storage (shared values, no accumulation): the operator products of the solver loop
int spmm_z(hipStream_t st, int nb, const SellView& A, zvec x, double* y, double* dot_partial, zvec dot_with);
// out = r - A x and coarse[i] = sum of out over the rows 8 i .. 8 i + 7 (restriction with the transpose of an
// "8 consecutive children, unit weights" prolongator); A.nrows must be a multiple of 8
void residual_restrict8(hipStream_t st, int nb, const SellView& A, const double* r, const double* x, double* out,
                        double* coarse);
// out = r - A x
void residual(hipStream_t st, int nb, const SellView& A, const double* r, const double* x, double* out);
// Chebyshev / Jacobi step:  d = a*d + b*dinv.*(r - A xin);  xout = xin + d   (xin != xout).
// dot_partial != nullptr: also per-block partials of <r, xout>; returns the number of blocks written.
int cheb_step(hipStream_t st, int nb, const SellView& A, const double* dinv, bool dinv_bv, const double* r,
              const double* xin, double* d, double* xout, double a, double b, double* dot_partial = nullptr);
// the LAST step of a polynomial whose result is a preconditioned Krylov vector: as cheb_step, the iterate goes to zvec
// storage (rounded before the fused dot), d is left as it was
int cheb_step_z(hipStream_t st, int nb, const SellView& A, const double* dinv, bool dinv_bv, const double* r,
                const double* xin, double* d, zvec zout, double a, double b, double* dot_partial = nullptr);
// one-pass degree-2 polynomial from a zero guess: xout = dinv.*(c0 r - c1 As r), As = A D^-1 (shared values)
int poly2(hipStream_t st, int nb, const SellView& As, const double* dinv, bool dinv_bv, const double* r, double* xout,
          double c0, double c1, double* dot_partial = nullptr, const double* xadd = nullptr,
          const double* dot_with = nullptr, const int* padd_idx = nullptr, const double* padd_x = nullptr);
// ... into zvec storage (the value stored is the one the fused <r, xout> uses)
int poly2_z(hipStream_t st, int nb, const SellView& As, const double* dinv, bool dinv_bv, const double* r, zvec xout,
            double c0, double c1, double* dot_partial = nullptr, const double* xadd = nullptr,
            const double* dot_with = nullptr, const int* padd_idx = nullptr, const double* padd_x = nullptr);
// V-cycle level with fp32 intermediates (shared values; see vc_poly2_kernel): pre-smoothing from zero into an fp32 iterate,
// residual + restriction over groups of 8 rows from it (fp32 residual, fp64 coarse right-hand side), res -= (S P) xc, and the
// post-smoothing x + xc[parent] + p2(res) -> fp64 result with the fused <r, result>
void vc_presmooth32(hipStream_t st, int nb, const SellView& As, const double* dinv, const double* r, float* xout, double c0,
                    double c1);
void vc_residual_restrict8_32(hipStream_t st, int nb, const SellView& A, const double* r, const float* x, float* out,
                              double* coarse);
void vc_residual32(hipStream_t st, int nb, const SellView& A, const double* r, const float* x, float* out);   // no restriction
// the V-cycle's first two kernels of a level from the fp32 copy of its right-hand side (written by k::lincomb3)
void vc_presmooth32_r32(hipStream_t st, int nb, const SellView& As, const double* dinv, const float* r32, float* xout,
                        double c0, double c1);
void vc_residual32_r32(hipStream_t st, int nb, const SellView& A, const float* r32, const float* x, float* out);
void vc_residual_restrict_agg32_r32(hipStream_t st, int nb, const SellView& A, const float* r32, const float* x, float* out,
                                    double* coarse, const int* seg_ptr, const int* seg_cid, const int* seg_pos);
// residual + restriction of an aggregation level renumbered by agg_pack_rows (segments per slice: seg_ptr / seg_cid / seg_pos)
void vc_residual_restrict_agg32(hipStream_t st, int nb, const SellView& A, const double* r, const float* x, float* out,
                                double* coarse, const int* seg_ptr, const int* seg_cid, const int* seg_pos);
void vc_residual_coarse32(hipStream_t st, int nb, const SellView& SP, float* res, const double* xc);
int vc_postsmooth32(hipStream_t st, int nb, const SellView& As, const double* dinv, const float* res, const float* x,
                    double* xout, double c0, double c1, const double* r, const int* parent, const double* xc, double* dot_partial);
int vc_postsmooth32_z(hipStream_t st, int nb, const SellView& As, const double* dinv, const float* res, const float* x,
                      zvec xout, double c0, double c1, const double* r, const int* parent, const double* xc, double* dot_partial);
// ... and for a level with per-realization fp32 values and diagonals (Darcy; no S P: the coarse correction is added to the
// fp32 iterate, then residual and post-smoothing): the fine residual of the restriction is never stored
void vc_presmooth32_bv(hipStream_t st, int nb, const SellView& As, const double* dinv, const double* r, float* xout, double c0,
                       double c1);
void vc_restrict8_32_bv(hipStream_t st, int nb, const SellView& A, const double* r, const float* x, double* coarse);
void vc_prolong8_32(hipStream_t st, int nb, int n, float* x, const double* xc);
void vc_residual32_bv(hipStream_t st, int nb, const SellView& A, const double* r, const float* x, float* out);
int vc_postsmooth32_bv(hipStream_t st, int nb, const SellView& As, const double* dinv, const float* res, const float* x,
                       double* xout, double c0, double c1, const double* r, double* dot_partial);
int vc_postsmooth32_bv_z(hipStream_t st, int nb, const SellView& As, const double* dinv, const float* res, const float* x,
                         zvec xout, double c0, double c1, const double* r, double* dot_partial);
// y = A1 x1 + A2 x2 (A1 per-realization values, A2 shared values, same rows); optional fused dot
int pair_spmm(hipStream_t st, int nb, const SellView& A1, const double* x1, const SellView& A2, const double* x2, double* y,
              double* dot_partial, const double* dot_with);
int pair_spmm_z(hipStream_t st, int nb, const SellView& A1, zvec x1, const SellView& A2, zvec x2, double* y,
                double* dot_partial, zvec dot_with);
// out[slot][k] = vals[slot][k] * colscale[cols[slot]][k]
void scale_cols_bv(hipStream_t st, int nb, int64_t nslots, const int* cols, const double* vals, const double* colscale,
                   double* out);
// the fp32 copies the preconditioner kernels read: out_scaled = (float)(vals * colscale[cols]), out_vals = (float)vals
void scale_cols_bv32(hipStream_t st, int nb, int64_t nslots, const int* cols, const double* vals, const double* colscale,
                     float* out_scaled, float* out_vals);
// MINRES w/x update on an index list of rows (w0, w1, x compact [nsel][nb]; u full)
void minres_wx_idx(hipStream_t st, int nb, int nsel, const int* rows, const double* c0, zvec u, const double* c1,
                   double* w0, const double* c2, const double* w1, const double* c3, double* x);
// first step from a zero guess: d = b*dinv.*r; x = d
int cheb_first(hipStream_t st, int nb, int n, const double* dinv, bool dinv_bv, const double* r, double* d,
               double* x, double b, double* dot_partial = nullptr);
// both return the number of partial blocks written
int dot(hipStream_t st, int nb, int n, const double* a, const double* b, double* partial);
int wdot(hipStream_t st, int nb, int n, const double* w, const double* x, double* partial);
int dot_z(hipStream_t st, int nb, int n, const double* a, zvec b, double* partial);
// out = in rounded to zvec storage; dot_partial != nullptr: partials of <r, out>.  Returns the partial-block count.
int convert_z(hipStream_t st, int nb, int n, const double* in, zvec out, const double* r, double* dot_partial);
void reduce_final(hipStream_t st, int nb, int nblocks, const double* partial, double* out);
void lincomb3(hipStream_t st, int nb, int n, const double* c0, const double* a, const double* c1, const double* b,
              const double* c2, double* y, float* y32 = nullptr);
void minres_wx(hipStream_t st, int nb, int n, const double* c0, zvec u, const double* c1, double* w0,
               const double* c2, const double* w1, const double* c3, double* x);
void fill(hipStream_t st, size_t n, double* x, double v);
void copy(hipStream_t st, size_t n, const double* src, double* dst);
void scale(hipStream_t st, size_t n, const double* in, double a, double* out);   // out = a in

// per-block partial sums of one dot product, in up to two segments (n blocks of nb doubles each): kernels that run side
// by side on two streams write a segment each
struct DotParts {
    const double* p1 = nullptr;
    int n1 = 0;
    const double* p2 = nullptr;
    int n2 = 0;
    int total() const { return n1 + n2; }
};
void minres_init(hipStream_t st, int nb, MinresState* s, const DotParts& d, double rel_tol, double abs_tol, int ring = 1);
// the pending w / x updates of B.cnt iterations on n rows: for j < cnt  w = cW0 u_j + cW1 w0 + cW2 w1; x += cW3 w;
// (w0, w1) <- (w1, w).  w0 is the OLDER direction on entry and on return (no role swap by the caller).
void minres_wx_deferred(hipStream_t st, int nb, int n, const MinresState* s, const WxDeferred& B, double* w0, double* w1,
                        double* x);
void minres_scal1(hipStream_t st, int nb, MinresState* s, const DotParts& d);
void minres_scal2(hipStream_t st, int nb, MinresState* s, const DotParts& d);
// scal2 of this iteration followed by scal1 of the next one (d1 = partials of the next operator product) in one launch
// stage (scal_stage_doubles() doubles, may be null): scratch of the multi-block first reduction stage for long lists
void minres_scal21(hipStream_t st, int nb, MinresState* s, const DotParts& d2, const DotParts& d1, double* stage = nullptr);
size_t scal_stage_doubles();

// realization b of the batch = generator realization first_id + b * id_stride
void normal_fill(hipStream_t st, int n, int nbatch, uint64_t seed, uint64_t first_id, uint32_t stream, double mean,
                 double sigma, double* out, uint64_t id_stride = 1);
// out[i*nb+k] = scale * in[k*n+i] * (w ? w[i] : 1)
void interleave(hipStream_t st, int nb, int n, const double* in, const double* w, double scale, double* out,
                const int* src = nullptr);
// out[k*m+i] = post(rowscale[i] * in[(idx?idx[i]:i)*nb + k])
void deinterleave(hipStream_t st, int nb, int m, const double* in, const int* idx, const double* rowscale, bool do_exp,
                  double* out);
void broadcast(hipStream_t st, int nb, int n, const double* a, double* out);

void darcy_coef(hipStream_t st, int nb, int n, const double* kfield, bool k_divides, double* coef);
void darcy_assemble(hipStream_t st, int nb, const SellView& Mp, const int* slot_src, const int* c_ptr, const int* c_elem,
                    const double* c_val, const double* coef, const unsigned char* ess, const double* ess_data,
                    const double* rhs0, double* mvals, double* diag, double* l1inv, double* rhs_bc);
// y = M(k) x1 + A2 x2 over the same rows with M(k) element-grouped (the u-rows [M(k) | B^T] of the Darcy operator);
// dot_partial != nullptr: partials of <dot_with, y>.  Returns the partial-block count.
int eg_pair_spmm(hipStream_t st, int nb, const EgView& M, const double* coef, const double* x1, const SellView& A2,
                 const double* x2, double* y, double* dot_partial, const double* dot_with);
int eg_pair_spmm_z(hipStream_t st, int nb, const EgView& M, const double* coef, zvec x1, const SellView& A2,
                   zvec x2, double* y, double* dot_partial, zvec dot_with);
// one-pass degree-2 Chebyshev polynomial of D^-1 M(k) from a zero guess (see sell_poly2_kernel) on the element-grouped
// matrix: xout = dinv (c0 r - c1 M(k) (dinv r)); dot_partial != nullptr: partials of <r, xout>
int eg_poly2(hipStream_t st, int nb, const EgView& M, const double* coef, const double* dinv, const double* r, double* xout,
             double c0, double c1, double* dot_partial);
int eg_poly2_z(hipStream_t st, int nb, const EgView& M, const double* coef, const double* dinv, const double* r, zvec xout,
               double c0, double c1, double* dot_partial);

// per-realization Gershgorin scaling of dinv (batched values): afterwards spec(diag(dinv) S) lies in (0, 1] for every
// realization; gwork = kMaxBatch doubles of scratch
void gersh_scale_bv(hipStream_t st, int nb, const SellView& S, double* dinv, double* gwork);
void refresh(hipStream_t st, int nb, int64_t nslots, const int* ptr, const int* idx, const double* w, const double* src,
             bool recip, double* out);
// back-substitution of the hybridized Darcy system (DarcyHybrid; interleaved [row][nb] vectors)
void darcy_backsub_u(hipStream_t st, int nb, int n_u, const int* owner, const double* kappa, const double* U0, const double* ug,
                     const double* t, double* out);
void darcy_backsub_p(hipStream_t st, int nb, int n_p, const double* kappa, const double* P0, const double* zg, const double* t,
                     double* out);
void diag_inv(hipStream_t st, int nb, int n, const int* diag_slot, const double* vals, double* dinv);

}  // namespace k
}  // namespace pmc

// ---- small-level V-cycle tail: every level from some level on fits one workgroup's LDS ---------------
namespace pmc {
struct TailLevelDev {
    int n, nslices;
    int nslots;                 // total SELL slots of the level (column stride of transposed per-realization values)
    const int* slice_off;
    const int* cols;
    const double* vals;         // shared [nslots], per-realization [nslots][nb] (bv 1) or transposed [nb][nslots] (bv 2)
    const double* vals_scaled;  // S D^-1 on the same pattern, may be null
    const double* dinv;         // [n], [n][nb] (bv 1) or [nb][n] (bv 2)
    int p_nslices, pt_nslices;  // transfers to / from the next coarser level (unused on the last tail level)
    const int *p_off, *p_cols, *pt_off, *pt_cols;
    const double *p_vals, *pt_vals;
    double lmax;
    int last_degree;            // > 0: this level ends the recursion with a Chebyshev solve of that degree
    double last_ratio;
    int lds_off;                // offset (in doubles) of this level's [r | x | d] block
    // last tail level only, may be null: the dense inverse of its (shared-value, symmetric) operator, n x n - the recursion
    // then ends with x = A^-1 r in one pass instead of a many-step Chebyshev solve whose every step re-reads the matrix behind
    // two barriers (a 64-row level with 30 entries per row: ~40 of the tail's 146 us per cycle, all of it latency)
    const double* ainv;
};
static constexpr size_t kTailLdsDoubles = (160 * 1024 - 1024) / 8;
struct TailParams {
    int nlev, bv, smooth_degree;
    double smooth_ratio;
    TailLevelDev lev[8];
};
// Everything the persistent small-level solver of the sampler needs (device pointers; by value in the kernel arguments)
struct MiniSamplerParams {
    int n_u, n_s;
    const int *a_off, *a_cols;      // block operator A (SELL-64, shared values)
    const double* a_vals;
    const int *m_off, *m_cols;      // M with column-scaled values M D^-1 and the l1 inverse: one-pass degree-2 polynomial
    const double *m_scaled, *m_dinv;
    double mc0, mc1;
    const TailParams* tail;         // V-cycle of the Schur block from this level on, all in LDS
    int max_iter;
    double rel_tol, abs_tol;
    int x_row0, x_nrows;            // maintained rows of the solution
    size_t scratch_per_col;         // doubles of scratch per realization: 5 n + 3 x_nrows
};
namespace k {
// Whole preconditioned MINRES solves of nb realizations, one workgroup each (see mini_sampler_kernel).  b, x: interleaved
// [row][nb] vectors as in minres_solve; stats: nb device entries.
void mini_sampler_solve(hipStream_t st, int nb, const MiniSamplerParams& P, size_t lds_doubles, const double* b, double* x,
                        bool zero_guess, double* scratch, pmc_stats* stats);
// One workgroup per batch column runs the whole V-cycle over the tail levels in LDS.  r, xout: interleaved
// vectors of the first tail level.  dot_partial != nullptr: writes <r, xout> per column as ONE partial block.
int mg_tail(hipStream_t st, int nb, const TailParams* dev_params, size_t lds_doubles, const double* r, double* xout,
            double* dot_partial, bool out32 = false);
inline int mg_tail_z(hipStream_t st, int nb, const TailParams* dev_params, size_t lds_doubles, const double* r, zvec xout,
                     double* dot_partial) {
    return mg_tail(st, nb, dev_params, lds_doubles, r, xout.as<double>(), dot_partial, xout.f32);
}
// x = ainv r, ainv dense n x n row-major (symmetric), r / x interleaved [row][nb], nb <= 8 (see dense_apply_kernel)
void dense_apply(hipStream_t st, int nb, int n, const double* ainv, const double* r, double* x);
// out[k][i] = in[i][k]: per-realization values of a small level re-laid column-major for the tail kernel, whose
// workgroup k then streams only its own realization's values
void transpose_bv(hipStream_t st, int nb, size_t count, const double* in, double* out);
void transpose_bv32(hipStream_t st, int nb, size_t count, const float* in, double* out);   // fp32 source
}  // namespace k
}  // namespace pmc
