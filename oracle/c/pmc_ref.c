/* ORACLE / CPU BASELINE (test infrastructure, never linked into the product).
 *
 * Plain-C restatement of the reference's default per-realization solve, used (a) as an
 * iterative cross-check of the direct-solve oracle and (b) as the "port" CPU baseline that
 * bench.py times on the GPU box's host cores.
 *
 * Algorithm = the reference's "MINRES-BJ-GS" entry
 * (/root/reference/examples/example_helpers/CreateSamplerParameterList.hpp:46-113):
 *   MINRES (<= 300 its, rel 1e-6, abs 1e-12, preconditioned-residual stopping rule)
 *   preconditioner = block-diagonal:
 *     A00^-1 ~ 3 sweeps of symmetric Gauss-Seidel on M   (hypre "L1 Gauss-Seidel", :80-93; on one
 *              process hypre's l1 hybrid smoother reduces to plain symmetric GS)
 *     A11^-1 ~ one multigrid V(1,1)-cycle with symmetric GS smoothing on
 *              S = alpha W + B diag(M)^-1 B^T (the "S Type = Diagonal" Schur complement, :76; explicit
 *              form in src/DarcySolver_Legacy.cpp:496-504).  BoomerAMG (:95-113) is not available, so
 *              the V-cycle runs over the caller's nested level hierarchy instead of algebraic levels.
 * The operator is [M B^T; B -alpha W] (src/PDESampler.cpp:279-284).
 *
 * Samples are farmed over OpenMP threads, one whole realization per thread - the most
 * favourable CPU layout and the mirror of the GPU sample farm.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    int nrows, ncols;
    const int* rp;
    const int* ci;
    const double* v;
} csr_t;

typedef struct {
    int n_u, n_s;
    csr_t M, B, Bt, S; /* S includes alpha*W */
    const double* aw;  /* alpha * diag(W), n_s */
    csr_t P, Pt;       /* to / from the next coarser level (unused on the last) */
} level_t;

static void spmv(const csr_t* A, const double* x, double* y) {
    for (int i = 0; i < A->nrows; ++i) {
        double s = 0.0;
        for (int p = A->rp[i]; p < A->rp[i + 1]; ++p) s += A->v[p] * x[A->ci[p]];
        y[i] = s;
    }
}
static void spmv_add(const csr_t* A, const double* x, double* y, double alpha) {
    for (int i = 0; i < A->nrows; ++i) {
        double s = 0.0;
        for (int p = A->rp[i]; p < A->rp[i + 1]; ++p) s += A->v[p] * x[A->ci[p]];
        y[i] += alpha * s;
    }
}
static double dot(int n, const double* a, const double* b) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
}

/* one forward + one backward Gauss-Seidel sweep on A x = b */
static void sym_gs(const csr_t* A, const double* b, double* x) {
    const int n = A->nrows;
    for (int i = 0; i < n; ++i) {
        double s = b[i], d = 1.0;
        for (int p = A->rp[i]; p < A->rp[i + 1]; ++p) {
            const int c = A->ci[p];
            if (c == i) d = A->v[p]; else s -= A->v[p] * x[c];
        }
        x[i] = s / d;
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = b[i], d = 1.0;
        for (int p = A->rp[i]; p < A->rp[i + 1]; ++p) {
            const int c = A->ci[p];
            if (c == i) d = A->v[p]; else s -= A->v[p] * x[c];
        }
        x[i] = s / d;
    }
}

typedef struct {
    double *r, *x, *t; /* per level work */
} mgwork_t;

static void vcycle(int nlevels, const level_t* lv, mgwork_t* w, int l, const double* r, double* x) {
    const csr_t* S = &lv[l].S;
    const int n = S->nrows;
    memset(x, 0, sizeof(double) * n);
    if (l == nlevels - 1) {
        for (int it = 0; it < 4; ++it) sym_gs(S, r, x);
        return;
    }
    sym_gs(S, r, x);
    double* res = w[l].t;
    spmv(S, x, res);
    for (int i = 0; i < n; ++i) res[i] = r[i] - res[i];
    spmv(&lv[l].Pt, res, w[l + 1].r);
    vcycle(nlevels, lv, w, l + 1, w[l + 1].r, w[l + 1].x);
    spmv_add(&lv[l].P, w[l + 1].x, x, 1.0);
    sym_gs(S, r, x);
}

static void apply_A(const level_t* L, const double* x, double* y) {
    const int nu = L->n_u, ns = L->n_s;
    spmv(&L->M, x, y);
    spmv_add(&L->Bt, x + nu, y, 1.0);
    spmv(&L->B, x, y + nu);
    for (int i = 0; i < ns; ++i) y[nu + i] -= L->aw[i] * x[nu + i];
}

static void apply_prec(int nlevels, const level_t* lv, mgwork_t* w, int level, const double* r, double* z) {
    const level_t* L = &lv[level];
    memset(z, 0, sizeof(double) * L->n_u);
    for (int s = 0; s < 3; ++s) sym_gs(&L->M, r, z);
    vcycle(nlevels, lv, w, level, r + L->n_u, z + L->n_u);
}

/* preconditioned MINRES, zero initial guess; returns iterations (negative: not converged) */
static int minres(int nlevels, const level_t* lv, mgwork_t* mw, int level, const double* b, double* x, int max_iter,
                  double rel_tol, double abs_tol, double* work) {
    const level_t* L = &lv[level];
    const int n = L->n_u + L->n_s;
    double *v0 = work, *v1 = work + n, *u1 = work + 2 * n, *q = work + 3 * n, *w0 = work + 4 * n, *w1 = work + 5 * n;
    memset(x, 0, sizeof(double) * n);
    memset(v0, 0, sizeof(double) * n);
    memset(w0, 0, sizeof(double) * n);
    memset(w1, 0, sizeof(double) * n);
    memcpy(v1, b, sizeof(double) * n);
    apply_prec(nlevels, lv, mw, level, v1, u1);
    double beta = sqrt(dot(n, v1, u1));
    double eta = beta, gamma0 = 1.0, gamma1 = 1.0, sigma0 = 0.0, sigma1 = 0.0;
    const double goal = fmax(rel_tol * eta, abs_tol);
    if (eta <= goal) return 0;
    for (int it = 1; it <= max_iter; ++it) {
        const double ib = 1.0 / beta;
        for (int i = 0; i < n; ++i) { v1[i] *= ib; u1[i] *= ib; }
        apply_A(L, u1, q);
        const double alpha = dot(n, u1, q);
        for (int i = 0; i < n; ++i) v0[i] = q[i] - alpha * v1[i] - beta * v0[i];
        const double delta = gamma1 * alpha - gamma0 * sigma1 * beta;
        const double rho3 = sigma0 * beta;
        const double rho2 = sigma1 * alpha + gamma0 * gamma1 * beta;
        apply_prec(nlevels, lv, mw, level, v0, q);
        const double beta_new = sqrt(fmax(dot(n, v0, q), 0.0));
        const double rho1 = hypot(delta, beta_new);
        for (int i = 0; i < n; ++i) w0[i] = (u1[i] - rho3 * w0[i] - rho2 * w1[i]) / rho1;
        gamma0 = gamma1;
        gamma1 = delta / rho1;
        const double step = gamma1 * eta;
        for (int i = 0; i < n; ++i) x[i] += step * w0[i];
        sigma0 = sigma1;
        sigma1 = beta_new / rho1;
        eta = -sigma1 * eta;
        /* rotate: u1 <- q, v0 <-> v1, w0 <-> w1 */
        double* t;
        memcpy(u1, q, sizeof(double) * n);
        t = v0; v0 = v1; v1 = t;
        t = w0; w0 = w1; w1 = t;
        beta = beta_new;
        if (fabs(eta) <= goal) return it;
        if (beta == 0.0) return it;
    }
    return -max_iter;
}

/* Solves [M Bt; B -aW] sol = rhs for `nsamples` right-hand sides (each n_u+n_s long), farmed over
 * `nthreads` OpenMP threads.  iters[i] receives the MINRES iteration count of sample i. */
int pmc_ref_solve_batch(int nlevels, const level_t* lv, int level, int nsamples, const double* rhs, double* sol,
                        int max_iter, double rel_tol, double abs_tol, int nthreads, int* iters) {
    if (nlevels < 1 || level < 0 || level >= nlevels || nsamples < 0) return -1;
    const int n = lv[level].n_u + lv[level].n_s;
    int fail = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel
    {
        double* work = (double*)malloc(sizeof(double) * 6 * (size_t)n);
        mgwork_t* mw = (mgwork_t*)calloc((size_t)nlevels, sizeof(mgwork_t));
        int ok = work != NULL && mw != NULL;
        for (int l = level; ok && l < nlevels; ++l) {
            const size_t m = (size_t)lv[l].n_s;
            mw[l].r = (double*)malloc(sizeof(double) * m);
            mw[l].x = (double*)malloc(sizeof(double) * m);
            mw[l].t = (double*)malloc(sizeof(double) * m);
            ok = mw[l].r && mw[l].x && mw[l].t;
        }
        if (!ok) {
#pragma omp atomic write
            fail = 1;
        } else {
#pragma omp for schedule(dynamic, 1)
            for (int i = 0; i < nsamples; ++i) {
                const int it = minres(nlevels, lv, mw, level, rhs + (size_t)i * n, sol + (size_t)i * n, max_iter, rel_tol,
                                      abs_tol, work);
                if (iters) iters[i] = it;
            }
        }
        if (mw)
            for (int l = level; l < nlevels; ++l) { free(mw[l].r); free(mw[l].x); free(mw[l].t); }
        free(mw);
        free(work);
    }
    return fail ? -2 : 0;
}

int pmc_ref_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
