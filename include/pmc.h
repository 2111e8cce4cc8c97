/* pmc.h - C ABI of the MI355X-native ParELAGMC hot path (libpmc.so).
 *
 * Drop-in boundary for the per-realization work that ParELAGMC performs behind its two
 * plugin interfaces (all paths relative to /root/reference):
 *     MLSampler::Sample / Eval               src/MLSampler.hpp:33-52
 *     PhysicalMLSolver::SolveFwd             src/PhysicalMLSolver.hpp:33-47
 * as called, and only called, from MLMC_Manager::InitRun (src/MLMC_Manager.cpp:113-173) and
 * MC_Manager::InitRun (src/MC_Manager.cpp:82-116).
 *
 * Conventions
 *   - plain C types only: opaque handles, pointers + sizes, int32 indices, fp64 values;
 *   - every function returns PMC_OK (0) or a negative error code; no exception crosses the
 *     boundary; pmc_last_error() returns a thread-local message for the last failure;
 *   - operators are passed as host CSR arrays (exactly what ParELAG's SparseMatrix /
 *     HypreParMatrix diag blocks hold) and are copied/re-laid-out on the device at create
 *     time; callers keep ownership of everything they pass in;
 *   - level 0 is the FINEST level (ParELAG convention); P of level i maps level i+1 -> i;
 *   - vectors may live in host or device memory (`memspace`); batched vectors are
 *     sample-major: sample b occupies [b*n, (b+1)*n);
 *   - one handle per GPU; calls on one ctx are serialised by the caller (the reference's
 *     objects are not re-entrant either: src/DarcySolver.hpp:238).
 */
#ifndef PMC_H_
#define PMC_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PMC_OK 0
#define PMC_ERR_INVALID (-1)  /* bad argument (PARELAG_ASSERT / PARELAG_TEST_FOR_EXCEPTION sites) */
#define PMC_ERR_DEVICE (-2)   /* HIP runtime error, no GPU, out of memory */
#define PMC_ERR_COMM (-3)     /* RCCL failure */
#define PMC_ERR_INTERNAL (-4)

enum pmc_memspace { PMC_MEM_HOST = 0, PMC_MEM_DEVICE = 1 };
enum pmc_projection { PMC_PROJ_NONE = 0, PMC_PROJ_GATHER = 1, PMC_PROJ_L2 = 2 };
/* pmc_solver_opts.precond_storage: how data that lives INSIDE one application of the preconditioner z = B^-1 v is stored -
 * the preconditioned vectors z themselves, the V-cycle's iterates / residuals on a level, the per-realization values of the
 * Darcy Schur-complement hierarchy.  Operators, Lanczos vectors, products A z, directions, solution, every inner product and
 * ALL arithmetic are fp64 either way (the reference is fp64 end to end, src/PDESampler.cpp:279-333); with FP32 the solver is
 * MINRES with the fixed symmetric preconditioner fl32(B^-1 .), converges to the same solution at the same tolerance and
 * moves a quarter fewer bytes per iteration. */
enum pmc_storage { PMC_STORAGE_FP32 = 0, PMC_STORAGE_FP64 = 1 };

typedef struct pmc_ctx pmc_ctx;
typedef struct pmc_sampler pmc_sampler;
typedef struct pmc_darcy pmc_darcy;

/* CSR matrix, host pointers, sorted or unsorted column indices. */
typedef struct pmc_csr {
    int32_t nrows, ncols;
    const int32_t* rowptr; /* nrows+1 */
    const int32_t* colind; /* nnz */
    const double* vals;    /* nnz */
} pmc_csr;

/* Linear-solver options.  Defaults restate the reference's "MINRES-BJ-GS" entry
 * (examples/example_helpers/CreateSamplerParameterList.hpp:54-66): MINRES, 300 iterations,
 * rel 1e-6, abs 1e-12, block-diagonal preconditioner.  The two diagonal blocks are
 * GPU-native replacements inside the reference's own configuration space:
 * a fixed Chebyshev/l1-Jacobi polynomial on M (instead of 3 sequential l1-GS sweeps) and one
 * V-cycle over the caller's level hierarchy on S = aW + B diag(M)^-1 B^T (instead of
 * BoomerAMG), both fixed SPD linear operators as MINRES requires. */
typedef struct pmc_solver_opts {
    int32_t abi_version;      /* = PMC_ABI_VERSION; written by pmc_solver_opts_default, checked by the create functions: a caller
                                 compiled against another layout of this struct / pmc_stats is refused instead of misread   */
    int32_t max_iter;
    double rel_tol;
    double abs_tol;
    int32_t cheb_degree_M;    /* polynomial degree on the M block; 0 (default) = automatic: 2, or 4 on sampler levels whose
                                 measured Chebyshev interval exceeds 16 (badly shaped cells)                                */
    double cheb_ratio_M;      /* Chebyshev interval lambda_max/lambda_min of the l1-scaled M-block; <= 0 (default): measured at
                                 create time by a host Lanczos run on M (Darcy: on M(k == 1))                              */
    int32_t mg_smooth_degree; /* Chebyshev pre/post smoothing degree per level (default 2) */
    double mg_smooth_ratio;   /* smoothing interval [lmax/ratio, lmax] (default 8) */
    int32_t mg_coarse_degree; /* polynomial degree on the coarsest level (default 12) */
    double mg_coarse_ratio;   /* (default 100) */
    int32_t check_every;      /* iterations between host convergence polls (default 2) */
    int32_t use_graph;        /* replay pairs of MINRES iterations as one hipGraph (default 0; needs check_every 2) */
    double schur_scale;       /* gamma in S = aW + gamma * B diag(M)^-1 B^T (default 1): diag(M) under-/over-estimates M by
                                 the spectrum of diag(M)^-1 M, gamma recentres that interval                              */
    int32_t mg_coarsening;    /* hierarchy of the Schur-complement V-cycle: 0 = the caller's levels (P of the level structs),
                                 1 = smoothed aggregation built internally from S itself (what BoomerAMG does in the reference:
                                 strength-based, robust on stretched cells), 2 = choose 1 when the cells are strongly anisotropic
                                 (median strongest/weakest coupling per row > 10), else 0 (default 2)                       */
    int32_t mini_max_rows;    /* sampler levels of at most this many rows (n_u + n_s) whose Schur V-cycle fits the LDS tail are
                                 solved by ONE persistent workgroup per realization - the whole MINRES solve in a single
                                 kernel launch - instead of ~7 launches per iteration (default 6000; 0 = never)           */
    int32_t two_streams;      /* the two diagonal blocks of the preconditioner on two HIP streams of the handle: 0 = automatic
                                 (default: only when the handle is the only one on its device and the level has at least
                                 ~1.5 M rows x realizations; with several handles per GPU their kernels already fill the
                                 gaps), 1 = always, 2 = never.  Results do not depend on it.                                 */
    int32_t precond_storage;  /* enum pmc_storage (ABI 3): PMC_STORAGE_FP32 (default) or PMC_STORAGE_FP64 = everything fp64    */
} pmc_solver_opts;

/* Per-realization solver report; the reference returns -1 for iteration counts
 * (src/PDESampler.hpp:142-145, src/DarcySolver.hpp:104-107) and is silent on non-convergence. */
typedef struct pmc_stats {
    int32_t iterations;  /* iterations until THIS realization met the tolerance (the batch keeps iterating until its last one has) */
    int32_t converged;   /* 1 converged, 0 iteration cap reached, -1 breakdown (non-finite data / indefinite preconditioner) */
    double initial_norm; /* preconditioned residual norm before the first iteration */
    double final_norm;   /* |eta| at exit */
    /* Device time (HIP events on the handle's stream) of the launch group this realization was solved in, divided by the
     * realizations of that group, so that summing over realizations gives device milliseconds.  The reference's per-level
     * timers: solve_ms = "Sampler: Mult" (src/PDESampler.cpp:328-333) / "Darcy: Mult" (src/DarcySolver.cpp:231-236) - the
     * Krylov solve; setup_ms = what precedes it per realization: right-hand side and initial guess (sampler), M(k),
     * elimination and the Schur-complement hierarchy refresh = "Darcy: Build Solver" (src/DarcySolver.cpp:238-242). */
    double solve_ms;
    double setup_ms;
} pmc_stats;

/* One level of the SPDE sampler hierarchy = the blocks PDESampler::BuildHierarchy assembles
 * (src/PDESampler.cpp:232-284): A = [M B^T; B -alpha*W]. */
typedef struct pmc_sampler_level {
    int32_t n_u, n_s;
    pmc_csr M;            /* n_u x n_u, SPD, essential rows/cols -> identity (:236-241)     */
    pmc_csr B;            /* n_s x n_u, = W*D with essential columns zeroed (:243-246)       */
    const double* w_diag; /* n_s, diag(W) > 0; w_sqrt = sqrt(w_diag) (:248-254)              */
    pmc_csr P;            /* n_s(level) x n_s(level+1) = ComputeTrueP(sform) (:189-193);
                             ignored (may be zeroed) on the last level                        */
} pmc_sampler_level;

/* One level of the HYBRIDIZED sampler: the reference's alternative solver of the same system ("Hybridization" in the
 * sampler's parameter list, src/PDESampler.cpp:291,307-311,383-389; ParELAG's HybridHdivL2 does the element-local
 * elimination).  Eliminating (u, s) element by element from [M B^T; B -alpha*W][u; s] = [0; f] leaves one Lagrange
 * multiplier per face:   H lambda = G f,   s = z_diag .* f - G^T lambda   (exact, not an approximation). */
typedef struct pmc_hybrid_level {
    int32_t n_lambda, n_s;
    pmc_csr H;             /* n_lambda x n_lambda, SPD                                              */
    pmc_csr G;             /* n_lambda x n_s                                                        */
    const double* z_diag;  /* n_s, the (s, s) entry of the local inverses (negative)                */
    const double* w_diag;  /* n_s, diag(W) > 0 as in pmc_sampler_level                              */
    pmc_csr P;             /* n_s(level) x n_s(level+1) = ComputeTrueP(sform); ignored on the last  */
} pmc_hybrid_level;

/* What PDESampler::BuildHierarchy holds for one level BEFORE it eliminates boundary rows - the input of the element-local
 * elimination (pmc_hybrid_build).  The reference hands A[i] plus the level's de Rham sequence to
 * prec_factory->BuildSolver(A[i], state) (src/PDESampler.cpp:302-318; extra parameter "L2MassWeight" = alpha, :307-311) and
 * ParELAG's HybridHdivL2 reads the element matrices from the sequence; here they arrive as the same element decomposition of
 * the u-mass matrix that pmc_darcy_level carries for ComputeMassOperator(uform, k). */
typedef struct pmc_hybrid_elements {
    int32_t n_u, n_s;
    pmc_csr M_pattern;       /* n_u x n_u, sparsity of the u-mass matrix (vals ignored, may be NULL)                */
    const int32_t* c_ptr;    /* nnz(M)+1: contributions of stored entry p are c_ptr[p] .. c_ptr[p+1]                */
    const int32_t* c_elem;   /* element of each contribution                                                         */
    const double* c_val;     /* element-matrix value (global face orientation)                                       */
    pmc_csr B;               /* n_s x n_u as assembled, NO boundary elimination: row e lists every face of element e
                                with the sign of its outward normal against the face's global normal (:232-234)     */
    const double* w_diag;    /* n_s, diag(W) > 0                                                                     */
    pmc_csr P;               /* n_s(level) x n_s(level+1) = ComputeTrueP(sform); ignored (may be zeroed) on the last */
} pmc_hybrid_elements;
typedef struct pmc_hybrid_system pmc_hybrid_system;

/* One level of the Darcy hierarchy = what DarcySolver precomputes (src/DarcySolver.cpp:
 * 194-227 B/Bt/P, :297-319 obs, :360-384 ess_data, :386-414 rhs) plus the element
 * decomposition of the mass matrix that ComputeMassOperator(uform,k) re-assembles per sample
 * (:479):  M(k).vals[p] = sum_{t in c_ptr[p]..c_ptr[p+1]} coef(k[c_elem[t]]) * c_val[t]. */
typedef struct pmc_darcy_level {
    int32_t n_u, n_p;
    pmc_csr M_pattern;       /* n_u x n_u, sparsity of M(k) (vals ignored, may be NULL)       */
    const int32_t* c_ptr;    /* nnz(M)+1                                                      */
    const int32_t* c_elem;   /* element (entry of k) of each contribution                      */
    const double* c_val;     /* unit-coefficient element-matrix value                          */
    pmc_csr B;               /* n_p x n_u, no boundary elimination (:203-207)                 */
    const double* rhs;       /* n_u+n_p                                                       */
    const uint8_t* ess_mask; /* n_u, 1 = essential u-dof (:487-492)                           */
    const double* ess_data;  /* n_u, essential values (only read where ess_mask)              */
    const double* obs;       /* n_u+n_p, observation functional (:297-319)                    */
    pmc_csr P;               /* n_p(level) x n_p(level+1), P0 prolongator; ignored on last    */
} pmc_darcy_level;

/* ---- library / context ---------------------------------------------------------------- */
int pmc_version(void);
#define PMC_ABI_VERSION 3 /* layout of pmc_solver_opts / pmc_stats; 2: abi_version field, solve_ms / setup_ms; 3: precond_storage */
int pmc_abi_version(void); /* the library's PMC_ABI_VERSION */
/* bytes per entry of the PRECONDITIONED Krylov vectors inside the MINRES solves of a handle (4: PMC_STORAGE_FP32, 8:
 * PMC_STORAGE_FP64; pmc_krylov_z_bytes: of the default options).  The byte counts pmc_sampler_apply_operator reports are for
 * fp64 input; the launches inside the solver loop read their input vector at this width. */
int pmc_krylov_z_bytes(void);
int pmc_sampler_krylov_z_bytes(const pmc_sampler* s);
int pmc_darcy_krylov_z_bytes(const pmc_darcy* d);
/* kernels launched by this process through the library so far (all handles, all host threads): launch-rate diagnostics */
uint64_t pmc_kernel_launches(void);
const char* pmc_last_error(void);
void pmc_solver_opts_default(pmc_solver_opts* opts);

/* pmc_ctx_create_abi refuses a caller compiled against another PMC_ABI_VERSION (another layout of pmc_solver_opts /
 * pmc_stats) before any handle exists - also callers that pass opts == NULL (defaults) and a pmc_stats array later.  C and
 * C++ callers get it through the macro below; binders that cannot use macros (ctypes, cgo) call it directly.  The plain
 * pmc_ctx_create symbol stays exported and performs no such check: the create functions then check pmc_solver_opts only. */
int pmc_ctx_create(int device_id, pmc_ctx** out);
int pmc_ctx_create_abi(int device_id, int abi_version, pmc_ctx** out);
#ifndef PMC_NO_ABI_CHECK_MACRO
#define pmc_ctx_create(device_id, out) pmc_ctx_create_abi((device_id), PMC_ABI_VERSION, (out))
#endif
void pmc_ctx_destroy(pmc_ctx* ctx);
int pmc_ctx_synchronize(pmc_ctx* ctx);
/* hipStream_t all work of this ctx is enqueued on (for callers that record their own events) */
void* pmc_ctx_stream(pmc_ctx* ctx);
/* elapsed device milliseconds between two points on the ctx stream (HIP events) */
int pmc_timer_start(pmc_ctx* ctx);
int pmc_timer_stop(pmc_ctx* ctx, double* ms);

/* device memory helpers so non-HIP callers can keep xi / s / k resident in HBM */
int pmc_malloc(pmc_ctx* ctx, size_t bytes, void** dptr);
int pmc_free(pmc_ctx* ctx, void* dptr);
int pmc_memcpy_h2d(pmc_ctx* ctx, void* dst, const void* src, size_t bytes);
int pmc_memcpy_d2h(pmc_ctx* ctx, void* dst, const void* src, size_t bytes);

/* ---- NormalDistributionSampler (src/NormalDistributionSampler.cpp:17-37) ---------------- */
/* Seed the counter-based generator.  (nparts, mypart) restate NormalDistributionSampler::Split
 * (src/NormalDistributionSampler.cpp:21-24, a leap-frog of the stream): part `mypart` owns the generator's
 * realizations mypart, mypart + nparts, ...; the ids handed to pmc_*_sample / pmc_normal_fill are LOCAL to the part
 * (local id i = generator realization i * nparts + mypart), so parts with the same seed and different mypart never
 * share a realization, and the union over the parts is exactly the unsplit stream.  (1, 0) = no split. */
int pmc_rng_seed(pmc_ctx* ctx, uint64_t seed, int nparts, int mypart);
/* out[b*n + i] = mean + sqrt(sigma2) * Phi^-1(u),  b < nbatch, realization id first_id+b */
int pmc_normal_fill(pmc_ctx* ctx, double mean, double sigma2, uint64_t first_sample_id, uint32_t stream,
                    int nbatch, int n, double* out, int memspace);

/* ---- PDESampler / EmbeddedPDESampler / L2ProjectionPDESampler -------------------------- */
/* nlevels >= n_mc_levels >= 1: levels [n_mc_levels, nlevels) are never sampled on, they only
 * deepen the V-cycle of the Schur-complement preconditioner. */
int pmc_sampler_create(pmc_ctx* ctx, int nlevels, int n_mc_levels, const pmc_sampler_level* levels,
                       double alpha, double matern_g, int lognormal, const pmc_solver_opts* opts,
                       pmc_sampler** out);
/* The same sampler with the hybridized solver (pmc_hybrid_level): MINRES on H with one V-cycle of an internal aggregation
 * multigrid as preconditioner.  Every level is a Monte Carlo level.  The handle behaves like any other pmc_sampler
 * (Sample / Eval / projections / batch width); init_s / use_init are accepted and ignored (the multiplier has no
 * counterpart of a coarse field), pmc_sampler_mult / _apply_preconditioner / _apply_operator act on multiplier vectors
 * of n_lambda entries, pmc_sampler_nnz reports nnz(H).  Of pmc_solver_opts the Krylov fields, precond_storage,
 * mg_smooth_degree / mg_coarse_* apply; the smoothing interval of the aggregation hierarchy is [lmax / (2 mg_smooth_ratio),
 * lmax] (default 16); cheb_*_M, schur_scale, mg_coarsening and mini_max_rows have no role. */
int pmc_sampler_create_hybrid(pmc_ctx* ctx, int nlevels, const pmc_hybrid_level* levels, double alpha, double matern_g,
                              int lognormal, const pmc_solver_opts* opts, pmc_sampler** out);
int pmc_sampler_is_hybrid(const pmc_sampler* s);
/* The element-local elimination itself (host code, setup): one dense (n_fe + 1)^2 inverse per element,
 * [[X, y], [y^T, z]]_e = [[M_e, b_e^T], [b_e, -alpha w_e]]^-1, H = sum_e C_e X_e C_e^T, G = sum_e C_e y_e, C_e = sign(B[e, f]).
 * pmc_hybrid_system_level fills `view` with pointers into the system (valid until pmc_hybrid_system_destroy) - the arrays
 * pmc_sampler_create_hybrid takes.  pmc_sampler_create_hybrid_from_elements = build every level, create, destroy: the one call
 * that stands where the reference's `if (if_solver_hybridization)` branch builds its solver (src/PDESampler.cpp:302-318). */
int pmc_hybrid_build(const pmc_hybrid_elements* level, double alpha, pmc_hybrid_system** out);
int pmc_hybrid_system_level(const pmc_hybrid_system* sys, pmc_hybrid_level* view);
void pmc_hybrid_system_destroy(pmc_hybrid_system* sys);
int pmc_sampler_create_hybrid_from_elements(pmc_ctx* ctx, int nlevels, const pmc_hybrid_elements* levels, double alpha,
                                            double matern_g, int lognormal, const pmc_solver_opts* opts, pmc_sampler** out);
void pmc_sampler_destroy(pmc_sampler* s);
/* Output map of the embedded variants.  PMC_PROJ_GATHER: s = sbar[gather_idx]
 * (src/EmbeddedPDESampler.cpp:552-556); PMC_PROJ_L2: s = inv_w_orig .* (Gt sbar)
 * (src/L2ProjectionPDESampler.cpp:738-750). */
int pmc_sampler_set_projection(pmc_sampler* s, int level, int kind, const pmc_csr* Gt,
                               const int32_t* gather_idx, const double* inv_w_orig, int orig_size);
int pmc_sampler_num_levels(const pmc_sampler* s);
int pmc_sampler_xi_size(const pmc_sampler* s, int level);     /* size Sample() fills            */
int pmc_sampler_sample_size(const pmc_sampler* s, int level); /* SampleSize(): size of Eval's s */
int64_t pmc_sampler_nnz(const pmc_sampler* s, int level);     /* GetNNZ()                       */
/* Realizations of `level` ONE launch of the solver kernels carries (16 on large levels, 32 on small ones, 64 / 128 / 256 -
 * column groups of 32 - on the smallest): pmc_sampler_eval cuts any nbatch into chunks of this width, so callers that
 * can choose (the managers' realizations per plugin call) should hand over multiples of it. */
int pmc_sampler_batch_width(const pmc_sampler* s, int level);
/* GetTrueP(level) (src/MLSampler.hpp:85-87, src/PDESampler.hpp:153-156): the prolongator of the s-space from level+1 to
 * level as handed over at create time; the pointers stay valid for the life of the handle.  Error on the last level. */
int pmc_sampler_true_p(const pmc_sampler* s, int level, pmc_csr* out);
/* Sample(level, xi): xi ~ N(0, 1) of pmc_sampler_xi_size(level) entries per realization */
int pmc_sampler_sample(pmc_sampler* s, int level, uint64_t first_sample_id, int nbatch, double* xi,
                       int memspace);
/* Eval(level, xi, s, embed_s, use_init) (src/PDESampler.cpp:411-535).
 *   xi_level <= level : level xi was drawn on (the reference infers it from xi.Size(), :419)
 *   s                 : out, nbatch x sample_size(level); exp() applied if lognormal
 *   init_s/init_level : in, Gaussian field on a coarser-or-equal level used as the initial
 *                       guess when use_init != 0 (:498-510); ignored otherwise (may be NULL)
 *   embed_s_out       : out (may be NULL), Gaussian field on the sampler mesh at `level`
 *                       (:527); may alias init_s
 *   stats             : out (may be NULL), nbatch entries */
int pmc_sampler_eval(pmc_sampler* s, int level, int xi_level, int nbatch, const double* xi, double* s_out,
                     const double* init_s, int init_level, int use_init, double* embed_s_out, int memspace,
                     pmc_stats* stats);

/* invA[level]->Mult(rhs, sol) (src/PDESampler.cpp:397,521), the narrowest seam of the reference: the whole linear solve
 * A [u; s] = rhs on FULL vectors of n_u + n_s entries per realization (sample-major), every row of the solution maintained.
 * use_sol_as_guess != 0 = mfem::Solver::iterative_mode (:510): sol holds the initial guess on entry.  pmc_sampler_eval is this
 * solve with the right-hand side, warm start and output maps of Eval around it (and only the s-rows maintained); this entry
 * exists for callers that keep the reference's Eval and replace just the solver, and for true-residual checks. */
int pmc_sampler_mult(pmc_sampler* s, int level, int nbatch, const double* rhs, double* sol, int use_sol_as_guess,
                     int memspace, pmc_stats* stats);

/* z = B^-1 r: ONE application of the block-diagonal preconditioner the MINRES solves of `level` use (the reference's
 * "BJ-GS" block, examples/example_helpers/CreateSamplerParameterList.hpp:68-113), fp64 in and out, full vectors, nbatch one
 * of the level's launch widths.  Diagnostics: sqrt(<r, B^-1 r>) of a TRUE residual r = b - A x is the quantity whose
 * recurrence estimate pmc_stats.final_norm reports. */
int pmc_sampler_apply_preconditioner(pmc_sampler* s, int level, int nbatch, const double* r, double* z, int memspace);

/* y = A x with A = [M B^T; B -alpha W] of `level` (src/PDESampler.cpp:279-284; the oper->Mult inside the
 * Krylov loop, kernel K5) for nbatch in {1,2,4,8,16,32,64,128,256} vectors of n_u+n_s entries each.  The SpMM kernel is
 * launched `repeat` >= 1 times between two HIP events on the ctx stream; avg_ms (may be NULL) receives the
 * mean kernel duration, bytes (may be NULL) the algorithmic bytes of ONE launch:
 * 12 nnz + 4 nrows + nbatch * 8 * (nrows + ncols). */
int pmc_sampler_apply_operator(pmc_sampler* s, int level, int nbatch, const double* x, double* y, int memspace,
                               int repeat, double* avg_ms, double* bytes);

/* In-situ timing of the K5 launches inside the MINRES loop of pmc_sampler_eval: with on != 0 every operator launch is
 * bracketed by HIP events on the solve's own stream (adds two event records per iteration; hipGraph replay is not
 * timed).  pmc_sampler_operator_time returns and clears the accumulated kernel time [ms] and launch count. */
int pmc_sampler_set_operator_timing(pmc_sampler* s, int on);
int pmc_sampler_operator_time(pmc_sampler* s, double* total_ms, int64_t* launches);
/* Behind every timed launch an EMPTY event bracket is recorded as well: returns and clears the sum of those [ms] - what
 * the event pair itself adds to a bracket on that stream (call before pmc_sampler_operator_time clears the count). */
int pmc_sampler_operator_event_overhead(pmc_sampler* s, double* total_ms);
/* Hybridized samplers, while pmc_sampler_set_operator_timing is on: the post-smoothing kernel of the finest level of the
 * multiplier V-cycle (k::vc_postsmooth32: the largest single kernel of an iteration) is bracketed the same way.
 * pmc_sampler_smoother_bytes: algorithmic bytes of one such launch = 12 B per entry of H + 12 B per row + nbatch x
 * ((4 + 4 + 8 + z) n_lambda + 8 n_coarse): residual and pre-smoothed iterate (fp32) and r read, z written, the coarse
 * correction read once.  Both report 0 for a saddle-point sampler. */
int pmc_sampler_smoother_time(pmc_sampler* s, double* total_ms, int64_t* launches, double* event_overhead_ms);
int pmc_sampler_smoother_bytes(const pmc_sampler* s, int level, int nbatch, double* bytes);
/* Sizes of the V-cycle hierarchy the sampler runs on `level` (diagnostics: what scripts/collect_profiles.py prices the
 * per-kernel roofline table with): info[0] = rows, [1] = entries of the level operator, [2] = its stored SELL slots,
 * [3] = entries of S P (0 when the coarse correction is not folded), [4] = slots of S P, [5] = flags: bit 0 = the level runs
 * inside the LDS tail kernel (launches of more than 8 realizations), bit 1 = launches of at most 8 realizations end their
 * cycle on this level with an exact dense solve, bits 4.. = log2 of the pieces its rows are cut into for such launches (0: not
 * split), [6] = 1 when its restriction is fused into the residual kernel.
 * Returns the number of V-cycle levels through *nvlevels; vlevel out of range is an error. */
int pmc_sampler_vcycle_info(const pmc_sampler* s, int level, int vlevel, int* nvlevels, int64_t info[7]);

/* ---- DarcySolver ------------------------------------------------------------------------ */
int pmc_darcy_create(pmc_ctx* ctx, int nlevels, int n_mc_levels, const pmc_darcy_level* levels,
                     int k_divides, const pmc_solver_opts* opts, pmc_darcy** out);
/* The same handle with SolveFwd through the HYBRIDIZED form of the mixed system - the reference's "Hybridization" branch of
 * DarcySolver (src/DarcySolver.cpp:586,619: the solver factory eliminates flux and pressure element by element).  Same level
 * structs: the element-local inverses are formed inside the library from the contribution lists of M (as pmc_hybrid_build does
 * for the sampler), one Lagrange multiplier per interior / essential face, H(k) lambda = rhs(k) with H linear in the
 * realization's coefficients solved by MINRES + one V-cycle of a per-realization aggregation hierarchy, then the element-local
 * back-substitution; Q, the returned solution and the pressure block are those of the saddle-point solve to the solver
 * tolerance.  Needs every face to belong to at most two elements and essential dofs on boundary faces only.  ComputeG keeps
 * the saddle-point path. */
int pmc_darcy_create_hybrid(pmc_ctx* ctx, int nlevels, int n_mc_levels, const pmc_darcy_level* levels,
                            int k_divides, const pmc_solver_opts* opts, pmc_darcy** out);
void pmc_darcy_destroy(pmc_darcy* d);
int pmc_darcy_num_dofs(const pmc_darcy* d, int level); /* GetGlobalNumberOfDofs() */
int pmc_darcy_num_pressure_dofs(const pmc_darcy* d, int level); /* GetSizeOfStochasticData(): entries of k */
int64_t pmc_darcy_nnz(const pmc_darcy* d, int level);  /* GetNNZ()                */
int pmc_darcy_batch_width(const pmc_darcy* d, int level);  /* as pmc_sampler_batch_width */
/* In-situ timing of the dominant kernel of the Darcy operator inside the MINRES loop of SolveFwd (solver->Mult,
 * src/DarcySolver.cpp:629-631): the u-rows y_u = M(k) x_u + B^T x_p with the fused <x, Ax> (eg_pair_spmm_kernel).  With
 * on != 0 every such launch is bracketed by HIP events on the solve's stream and an empty bracket is recorded behind it;
 * while timing, the p-rows (B x_u) follow on the same stream instead of running beside it.  pmc_darcy_operator_time returns
 * and clears the accumulated bracket time, the launch count and the sum of the empty brackets [ms];
 * pmc_darcy_operator_bytes gives the ALGORITHMIC bytes of one such launch for nbatch realizations: element-grouped M(k)
 * 12 B per stored slot + 8 B per dof (coefficient rows) + the coefficient table (n_p + 1) x nbatch x 8, B^T 12 B per nonzero,
 * 4 B per row, vectors nbatch (z (n_u + n_p) + 8 n_u) with z = pmc_krylov_z_bytes(). */
int pmc_darcy_set_operator_timing(pmc_darcy* d, int on);
int pmc_darcy_operator_time(pmc_darcy* d, double* total_ms, int64_t* launches, double* event_overhead_ms);
int pmc_darcy_operator_bytes(const pmc_darcy* d, int level, int nbatch, double* bytes);
/* The same for the other large gather kernel of a Darcy iteration, the M-block polynomial of the preconditioner
 * z_u = D^-1 (c0 r - c1 M(k) D^-1 r) (eg_poly2_kernel; the reference's A00^-1 block, three l1-Gauss-Seidel sweeps on M(k),
 * CreateSamplerParameterList.hpp:80-93): timed by the same pmc_darcy_set_operator_timing switch (the launch then runs on
 * the solve's main stream instead of beside the V-cycle's bottom).  Algorithmic bytes: 12 B per stored slot of the
 * element-grouped matrix + 12 B per dof + the coefficient table + nbatch ((8 + 8 + z) n_u): r and the per-realization l1
 * diagonal read, z written. */
int pmc_darcy_poly_time(pmc_darcy* d, double* total_ms, int64_t* launches, double* event_overhead_ms);
int pmc_darcy_poly_bytes(const pmc_darcy* d, int level, int nbatch, double* bytes);
/* SolveFwd(level, k, Q, C) (src/DarcySolver.cpp:416-437).  k: nbatch x n_p(level) in
 * `memspace`; Q, C: host arrays of nbatch; sol_out (may be NULL): nbatch x (n_u+n_p) in
 * `memspace` (SolveFwd_RtnPressure, :439-470, reads its p-block). */
int pmc_darcy_solve_fwd(pmc_darcy* d, int level, int nbatch, const double* k, double* Q, double* C,
                        double* sol_out, int memspace, pmc_stats* stats);

/* SolveFwd_RtnPressure(level, k, P, C, Q, compute_Q) (src/DarcySolver.cpp:439-470): the pressure block of the
 * solution, nbatch x n_p(level) in `memspace`; Q (host, may be NULL) is only written when compute_Q != 0. */
int pmc_darcy_solve_fwd_pressure(pmc_darcy* d, int level, int nbatch, const double* k, double* p_out, double* C,
                                 double* Q, int compute_Q, int memspace, pmc_stats* stats);

/* Bayesian observation operator (src/BayesianInverseProblem.cpp:178-186, ComputeG): Gobs is nobs x n_p(level), row i
 * = the observation functional g_obs_i (e.g. the indicator of the cells around an observation point, restricted to the
 * level).  pmc_darcy_compute_G solves like SolveFwd and returns G[b*nobs + i] = <g_i, p_b> / sum(g_i) (host array) plus
 * Q and C (host, may be NULL); only the rows of the solution that Q and G read are maintained by the Krylov solver. */
int pmc_darcy_set_observations(pmc_darcy* d, int level, const pmc_csr* Gobs);
int pmc_darcy_num_observations(const pmc_darcy* d, int level);
int pmc_darcy_compute_G(pmc_darcy* d, int level, int nbatch, const double* k, double* G, double* C, double* Q,
                        int memspace, pmc_stats* stats);

/* ---- MLMC accumulators across GPUs (new: the reference's manager is serial,
 *      src/MLMC_Manager.hpp:24) ----------------------------------------------------------- */
int pmc_comm_unique_id(void* id128);                                        /* 128 bytes    */
int pmc_comm_init(pmc_ctx* ctx, const void* id128, int nranks, int rank);   /* RCCL over xGMI */
int pmc_comm_destroy(pmc_ctx* ctx);
/* in-place SUM all-reduce of a small host buffer (the nlevels x 9 sums table + counts) */
int pmc_allreduce_sum_f64(pmc_ctx* ctx, double* host_buf, int n);

#ifdef __cplusplus
}
#endif
#endif /* PMC_H_ */
