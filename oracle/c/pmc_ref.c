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


/* ------------------------------------------------------------------------------------------------
 * Darcy leg of the per-realization work, as the reference performs it for EVERY sample
 * (/root/reference/src/DarcySolver.cpp:472-520 assemble, :562-649 BuildForwardSolver + solve):
 *   1. M(k) = sum_e c(k_e) M_e  (ComputeMassOperator(uform, k), :479)
 *   2. EliminateRowCol on the essential u-dofs, right-hand side fix-up (:487-498)
 *   3. REBUILD the solver: block operator [M(k) B^T; B 0], block-diagonal preconditioner with
 *      3 x symmetric GS on M(k) and a V-cycle on S(k) = B diag(M(k))^-1 B^T - the Schur complement and every
 *      coarse operator of its hierarchy are re-assembled numerically (the reference's BoomerAMG setup, which also
 *      builds the coarse grids, costs more than this Galerkin refresh on the caller's nested levels)
 *   4. MINRES, Q = <obs, sol> (src/Utilities.cpp:411-420)
 * B and B^T arrive with the essential columns / rows already removed and the k-independent part of the right-hand side
 * fix-up applied (setup, done once by the caller).
 */
typedef struct {
    int n_u, n_p;
    csr_t Mpat;            /* pattern of M (values ignored) */
    const int* c_ptr;      /* per stored nonzero: contributions c_ptr[p]..c_ptr[p+1] */
    const int* c_elem;
    const double* c_val;
    csr_t B, Bt;           /* essential columns / rows removed */
    const unsigned char* ess;
    const double* ess_data;
    const double* rhs;     /* n_u + n_p, before the M(k)-dependent fix-up */
    const double* obs;     /* n_u + n_p */
    csr_t Spat;            /* pattern of S on this level (values ignored) */
    const int* parent;     /* p-dof of the next coarser level owning each p-dof (injection prolongator); NULL on the last */
    csr_t P, Pt;
} dlevel_t;

typedef struct {
    int nlev;              /* levels of the V-cycle from the solved level on */
    csr_t M;               /* M(k) after elimination (own values) */
    const dlevel_t* L;     /* solved level */
    level_t* mg;           /* per V-cycle level: S with own values, P, Pt (the sampler's level_t reused; n_u/M/B unused) */
    mgwork_t* mw;
} dsys_t;

static void darcy_apply_A(const dsys_t* s, const double* x, double* y) {
    const int nu = s->L->n_u;
    spmv(&s->M, x, y);
    spmv_add(&s->L->Bt, x + nu, y, 1.0);
    spmv(&s->L->B, x, y + nu);
}
static void darcy_apply_prec(const dsys_t* s, const double* r, double* z) {
    const int nu = s->L->n_u;
    memset(z, 0, sizeof(double) * nu);
    for (int it = 0; it < 3; ++it) sym_gs(&s->M, r, z);
    vcycle(s->nlev, s->mg, s->mw, 0, r + nu, z + nu);
}

/* MINRES as in minres() above, on the Darcy system */
static int darcy_minres(const dsys_t* s, const double* b, double* x, int max_iter, double rel_tol, double abs_tol,
                        double* work) {
    const int n = s->L->n_u + s->L->n_p;
    double *v0 = work, *v1 = work + n, *u1 = work + 2 * n, *q = work + 3 * n, *w0 = work + 4 * n, *w1 = work + 5 * n;
    memset(x, 0, sizeof(double) * n);
    memset(v0, 0, sizeof(double) * n);
    memset(w0, 0, sizeof(double) * n);
    memset(w1, 0, sizeof(double) * n);
    memcpy(v1, b, sizeof(double) * n);
    darcy_apply_prec(s, v1, u1);
    double beta = sqrt(fmax(dot(n, v1, u1), 0.0));
    double eta = beta, gamma0 = 1.0, gamma1 = 1.0, sigma0 = 0.0, sigma1 = 0.0;
    const double goal = fmax(rel_tol * eta, abs_tol);
    if (eta <= goal) return 0;
    for (int it = 1; it <= max_iter; ++it) {
        const double ib = 1.0 / beta;
        for (int i = 0; i < n; ++i) { v1[i] *= ib; u1[i] *= ib; }
        darcy_apply_A(s, u1, q);
        const double alpha = dot(n, u1, q);
        for (int i = 0; i < n; ++i) v0[i] = q[i] - alpha * v1[i] - beta * v0[i];
        const double delta = gamma1 * alpha - gamma0 * sigma1 * beta;
        const double rho3 = sigma0 * beta;
        const double rho2 = sigma1 * alpha + gamma0 * gamma1 * beta;
        darcy_apply_prec(s, v0, q);
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

static int find_col(const csr_t* A, int row, int col) {
    for (int p = A->rp[row]; p < A->rp[row + 1]; ++p)
        if (A->ci[p] == col) return p;
    return -1;
}

/* Solves the Darcy system of `level` for nsamples coefficient fields k (n_p each, sample-major) and returns
 * Q[i] = <obs, sol_i>; sol (optional) receives the solutions.  nlevels dlevel_t entries, finest first. */
int pmc_ref_darcy_batch(int nlevels, const dlevel_t* lv, int level, int k_divides, int nsamples, const double* kf, double* Q,
                        double* sol, int max_iter, double rel_tol, double abs_tol, int nthreads, int* iters) {
    if (nlevels < 1 || level < 0 || level >= nlevels || nsamples < 0) return -1;
    const dlevel_t* L = &lv[level];
    const int nu = L->n_u, np = L->n_p, n = nu + np;
    const int nlev = nlevels - level;
    int fail = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel
    {
        double* work = (double*)malloc(sizeof(double) * 6 * (size_t)n);
        double* rhs = (double*)malloc(sizeof(double) * (size_t)n);
        double* x = (double*)malloc(sizeof(double) * (size_t)n);
        double* coef = (double*)malloc(sizeof(double) * (size_t)np);
        double* mval = (double*)malloc(sizeof(double) * (size_t)L->Mpat.rp[nu]);
        double* dinv = (double*)malloc(sizeof(double) * (size_t)nu);
        level_t* mg = (level_t*)calloc((size_t)nlev, sizeof(level_t));
        mgwork_t* mw = (mgwork_t*)calloc((size_t)nlev, sizeof(mgwork_t));
        double** sval = (double**)calloc((size_t)nlev, sizeof(double*));
        int ok = work && rhs && x && coef && mval && dinv && mg && mw && sval;
        for (int l = 0; ok && l < nlev; ++l) {
            const dlevel_t* D = &lv[level + l];
            const size_t m = (size_t)D->n_p;
            sval[l] = (double*)malloc(sizeof(double) * (size_t)D->Spat.rp[D->n_p]);
            mw[l].r = (double*)malloc(sizeof(double) * m);
            mw[l].x = (double*)malloc(sizeof(double) * m);
            mw[l].t = (double*)malloc(sizeof(double) * m);
            ok = sval[l] && mw[l].r && mw[l].x && mw[l].t;
            if (!ok) break;
            mg[l].n_u = 0;
            mg[l].n_s = D->n_p;
            mg[l].S = D->Spat;
            mg[l].S.v = sval[l];
            mg[l].P = D->P;
            mg[l].Pt = D->Pt;
        }
        if (!ok) {
#pragma omp atomic write
            fail = 1;
        } else {
            dsys_t sys;
            sys.nlev = nlev;
            sys.M = L->Mpat;
            sys.M.v = mval;
            sys.L = L;
            sys.mg = mg;
            sys.mw = mw;
#pragma omp for schedule(dynamic, 1)
            for (int i = 0; i < nsamples; ++i) {
                const double* k = kf + (size_t)i * np;
                for (int e = 0; e < np; ++e) coef[e] = k_divides ? 1.0 / k[e] : k[e];
                /* 1. M(k) */
                const int nnzM = L->Mpat.rp[nu];
                for (int p = 0; p < nnzM; ++p) {
                    double v = 0.0;
                    for (int t = L->c_ptr[p]; t < L->c_ptr[p + 1]; ++t) v += coef[L->c_elem[t]] * L->c_val[t];
                    mval[p] = v;
                }
                /* 2. EliminateRowCol(ess_dofs, ess_data, rhs) */
                memcpy(rhs, L->rhs, sizeof(double) * n);
                for (int r = 0; r < nu; ++r) {
                    const int re = L->ess[r];
                    for (int p = L->Mpat.rp[r]; p < L->Mpat.rp[r + 1]; ++p) {
                        const int c = L->Mpat.ci[p];
                        if (re || L->ess[c]) {
                            if (!re) rhs[r] -= mval[p] * L->ess_data[c];
                            mval[p] = (c == r) ? 1.0 : 0.0;
                        }
                    }
                    if (re) rhs[r] = L->ess_data[r];
                }
                /* 3. rebuild the preconditioner: S(k) = B diag(M(k))^-1 B^T and its Galerkin coarse operators */
                for (int r = 0; r < nu; ++r) {
                    const int p = find_col(&L->Mpat, r, r);
                    dinv[r] = 1.0 / mval[p];
                }
                {
                    const csr_t* S = &mg[0].S;
                    memset(sval[0], 0, sizeof(double) * (size_t)S->rp[np]);
                    for (int e = 0; e < np; ++e)
                        for (int p = L->B.rp[e]; p < L->B.rp[e + 1]; ++p) {
                            const int f = L->B.ci[p];
                            const double w = L->B.v[p] * dinv[f];
                            for (int q2 = L->Bt.rp[f]; q2 < L->Bt.rp[f + 1]; ++q2)
                                sval[0][find_col(S, e, L->Bt.ci[q2])] += w * L->Bt.v[q2];
                        }
                }
                for (int l = 0; l + 1 < nlev; ++l) {
                    const dlevel_t* D = &lv[level + l];
                    const csr_t* Sf = &mg[l].S;
                    const csr_t* Sc = &mg[l + 1].S;
                    memset(sval[l + 1], 0, sizeof(double) * (size_t)Sc->rp[Sc->nrows]);
                    for (int r = 0; r < Sf->nrows; ++r) {
                        const int I = D->parent[r];
                        for (int p = Sf->rp[r]; p < Sf->rp[r + 1]; ++p)
                            sval[l + 1][find_col(Sc, I, D->parent[Sf->ci[p]])] += 0.5 * sval[l][p];
                    }
                }
                /* 4. solve, Q */
                const int it = darcy_minres(&sys, rhs, x, max_iter, rel_tol, abs_tol, work);
                if (iters) iters[i] = it;
                Q[i] = dot(n, L->obs, x);
                if (sol) memcpy(sol + (size_t)i * n, x, sizeof(double) * n);
            }
        }
        if (mw)
            for (int l = 0; l < nlev; ++l) { free(mw[l].r); free(mw[l].x); free(mw[l].t); }
        if (sval)
            for (int l = 0; l < nlev; ++l) free(sval[l]);
        free(sval); free(mw); free(mg); free(dinv); free(mval); free(coef); free(x); free(rhs); free(work);
    }
    return fail ? -2 : 0;
}


/* ------------------------------------------------------------------------------------------------
 * The reference's OTHER sampler solver, "Hybridization"
 * (/root/reference/examples/example_parameterlists/example_parameters.xml:200-212: "the hybridized system is then solved by
 * BoomerAMG-preconditioned CG, after that the solution of the original system is obtained by back substitution";
 * selected in src/PDESampler.cpp:289-311, applied in :383-389 / :451-480):
 *   rhs_lambda = G f,   PCG on H lambda = rhs_lambda with ONE AMG V(1,1)-cycle as preconditioner,   s = z f - G^T lambda.
 * BoomerAMG is not available: the V-cycle runs over a smoothed-aggregation hierarchy of H built by the caller
 * (oracle/cport.py::HybridCPort: greedy aggregation below, damped-Jacobi prolongator smoothing, Galerkin products),
 * symmetric Gauss-Seidel smoothing - the standard AMG of the same class.  Stopping rule = MFEM's CGSolver:
 * sqrt(<B r, r>) <= max(rel * initial, abs).  [UNVERIFIED: the XML entry sets no tolerances; the values of the other library
 * entries (1e-6 / 1e-12) are used.]
 */

/* Greedy aggregation (Vanek, Mandel, Brezina 1996) on the strength graph |a_ij| >= theta * sqrt(a_ii a_jj).
 * agg[i] receives the aggregate of row i; returns the number of aggregates. */
int pmc_ref_aggregate(int n, const int* rp, const int* ci, const double* v, double theta, int* agg) {
    double* d = (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    if (!d) return -1;
    for (int i = 0; i < n; ++i) {
        d[i] = 1.0;
        for (int p = rp[i]; p < rp[i + 1]; ++p)
            if (ci[p] == i) d[i] = fabs(v[p]);
        agg[i] = -1;
    }
    int na = 0;
#define STRONG(i, p) (ci[p] != (i) && fabs(v[p]) >= theta * sqrt(d[i] * d[ci[p]]))
    /* phase 1: a row whose strong neighbourhood is untouched becomes the root of a new aggregate */
    for (int i = 0; i < n; ++i) {
        if (agg[i] >= 0) continue;
        int free_nb = 1, has_nb = 0;
        for (int p = rp[i]; p < rp[i + 1]; ++p)
            if (STRONG(i, p)) {
                has_nb = 1;
                if (agg[ci[p]] >= 0) { free_nb = 0; break; }
            }
        if (!free_nb || !has_nb) continue;
        agg[i] = na;
        for (int p = rp[i]; p < rp[i + 1]; ++p)
            if (STRONG(i, p)) agg[ci[p]] = na;
        ++na;
    }
    /* phase 2: remaining rows join the aggregate of their strongest aggregated neighbour (of phase 1) */
    int* tmp = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    if (!tmp) { free(d); return -1; }
    memcpy(tmp, agg, sizeof(int) * (size_t)n);
    for (int i = 0; i < n; ++i) {
        if (tmp[i] >= 0) continue;
        double best = -1.0;
        for (int p = rp[i]; p < rp[i + 1]; ++p)
            if (STRONG(i, p) && tmp[ci[p]] >= 0 && fabs(v[p]) > best) { best = fabs(v[p]); agg[i] = tmp[ci[p]]; }
    }
    free(tmp);
    /* phase 3: what is left forms aggregates with its remaining strong neighbours (or stays alone) */
    for (int i = 0; i < n; ++i) {
        if (agg[i] >= 0) continue;
        agg[i] = na;
        for (int p = rp[i]; p < rp[i + 1]; ++p)
            if (STRONG(i, p) && agg[ci[p]] < 0) agg[ci[p]] = na;
        ++na;
    }
#undef STRONG
    free(d);
    return na;
}

typedef struct {
    int n_lambda, n_s;
    csr_t G, Gt;        /* n_lambda x n_s and its transpose */
    const double* z;    /* n_s */
} hsys_t;

/* preconditioned CG on mg[0].S (= H), zero initial guess; returns iterations (negative: not converged) */
static int pcg(int nlev, const level_t* mg, mgwork_t* mw, const double* b, double* x, int max_iter, double rel_tol,
               double abs_tol, double* work) {
    const csr_t* H = &mg[0].S;
    const int n = H->nrows;
    double *r = work, *zz = work + n, *p = work + 2 * n, *q = work + 3 * n;
    memset(x, 0, sizeof(double) * n);
    memcpy(r, b, sizeof(double) * n);
    vcycle(nlev, mg, mw, 0, r, zz);
    memcpy(p, zz, sizeof(double) * n);
    double nom = dot(n, zz, r);
    if (!(nom >= 0.0)) return -1;
    const double goal = fmax(rel_tol * rel_tol * nom, abs_tol * abs_tol);
    if (nom <= goal) return 0;
    for (int it = 1; it <= max_iter; ++it) {
        spmv(H, p, q);
        const double den = dot(n, p, q);
        if (!(den > 0.0)) return -it;
        const double alpha = nom / den;
        for (int i = 0; i < n; ++i) { x[i] += alpha * p[i]; r[i] -= alpha * q[i]; }
        vcycle(nlev, mg, mw, 0, r, zz);
        const double betanom = dot(n, zz, r);
        if (betanom <= goal) return it;
        const double beta = betanom / nom;
        for (int i = 0; i < n; ++i) p[i] = zz[i] + beta * p[i];
        nom = betanom;
    }
    return -max_iter;
}

/* For nsamples right-hand sides f (n_s each): lambda = H^-1 G f by PCG-AMG, s = z f - G^T lambda.  mg: nlev levels of the AMG
 * hierarchy of H (level_t with S = operator of the level, P / Pt its transfer to / from the next coarser one; n_s = rows). */
int pmc_ref_hybrid_batch(int nlev, const level_t* mg, const hsys_t* sys, int nsamples, const double* f, double* s_out,
                         int max_iter, double rel_tol, double abs_tol, int nthreads, int* iters) {
    if (nlev < 1 || nsamples < 0) return -1;
    const int n = sys->n_lambda, ns = sys->n_s;
    int fail = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel
    {
        double* work = (double*)malloc(sizeof(double) * 6 * (size_t)n);
        mgwork_t* mw = (mgwork_t*)calloc((size_t)nlev, sizeof(mgwork_t));
        int ok = work != NULL && mw != NULL;
        for (int l = 0; ok && l < nlev; ++l) {
            const size_t m = (size_t)mg[l].S.nrows;
            mw[l].r = (double*)malloc(sizeof(double) * m);
            mw[l].x = (double*)malloc(sizeof(double) * m);
            mw[l].t = (double*)malloc(sizeof(double) * m);
            ok = mw[l].r && mw[l].x && mw[l].t;
        }
        if (!ok) {
#pragma omp atomic write
            fail = 1;
        } else {
            double *b = work + 4 * (size_t)n, *lam = work + 5 * (size_t)n;
#pragma omp for schedule(dynamic, 1)
            for (int i = 0; i < nsamples; ++i) {
                const double* fi = f + (size_t)i * ns;
                double* si = s_out + (size_t)i * ns;
                spmv(&sys->G, fi, b);
                const int it = pcg(nlev, mg, mw, b, lam, max_iter, rel_tol, abs_tol, work);
                if (iters) iters[i] = it;
                spmv(&sys->Gt, lam, si);
                for (int e = 0; e < ns; ++e) si[e] = sys->z[e] * fi[e] - si[e];
            }
        }
        if (mw)
            for (int l = 0; l < nlev; ++l) { free(mw[l].r); free(mw[l].x); free(mw[l].t); }
        free(mw);
        free(work);
    }
    return fail ? -2 : 0;
}
