/* pmc_host.h - C ABI of libpmc_host.so: the host-side Monte Carlo managers.
 *
 * Restates, in C++ behind a C surface, the two outer loops that drive the hot path in the
 * reference (paths relative to /root/reference):
 *     MLMC_Manager::Run / InitRun / computeNSamplesMSE   src/MLMC_Manager.cpp:103-214,300-401
 *     MC_Manager::Run / InitRun / computeNSamplesMSE     src/MC_Manager.cpp:82-146,194-239
 * They call ONLY the MLSampler / PhysicalMLSolver plugin surface (src/MLSampler.hpp:33-52,
 * src/PhysicalMLSolver.hpp:33-47), here either the device objects of pmc.h or user callbacks
 * (any other sampler/solver, and CPU-side tests of the managers).
 *
 * New relative to the reference ("serial multi-level MC manager", src/MLMC_Manager.hpp:24):
 * realizations of one InitRun round are sharded over ranks (one rank per GPU) in blocks of
 * `batch` consecutive sample ids and the nlevels x 9 sums table is SUM-all-reduced once per
 * round before computeNSamplesMSE.
 */
#ifndef PMC_HOST_H_
#define PMC_HOST_H_

#include "pmc.h"

#ifdef __cplusplus
extern "C" {
#endif

#define PMC_MLMC_NVAR 9 /* Y2,Y,ABSY,Q2,Q,ABSQ,C,Y3,Y4 - enum order of src/MLMC_Manager.hpp:65 */

typedef struct pmc_mlmc pmc_mlmc;

/* in-place SUM all-reduce of a host buffer over the sample-farm ranks */
typedef int (*pmc_reduce_fn)(double* buf, int n, void* user);

/* plugin callbacks (host pointers, batched sample-major like pmc.h) */
typedef int (*pmc_cb_sample)(void* user, int level, uint64_t first_id, int nbatch, double* xi);
typedef int (*pmc_cb_eval)(void* user, int level, int xi_level, int nbatch, const double* xi, double* s,
                           const double* init_s, int init_level, int use_init, double* embed_s_out);
typedef int (*pmc_cb_solve)(void* user, int level, int nbatch, const double* k, double* Q, double* C);

typedef struct pmc_plugin_callbacks {
    void* user;
    pmc_cb_sample sample;
    pmc_cb_eval eval;
    pmc_cb_solve solve_fwd;
    const int32_t* xi_size;     /* nlevels: size Sample() fills      */
    const int32_t* sample_size; /* nlevels: size of Eval's s         */
    const int32_t* ndofs;       /* nlevels: GetGlobalNumberOfDofs()  */
} pmc_plugin_callbacks;

/* "Problem parameters" read by the managers (src/MLMC_Manager.cpp:30-36) */
typedef struct pmc_mlmc_params {
    double eps2;              /* "Mean square error", default 0.001; < 0 = automatic        */
    double ratio;             /* "MSE splitting ratio", default 0.5                         */
    int32_t init_nsamples;    /* "Number of samples", default 10                            */
    const int32_t* array_nsamples; /* "Array number of samples" (nlevels) or NULL           */
    int32_t wall_time;        /* public member wallTime (src/MLMC_Manager.hpp:61), default 1 */
    int32_t batch;            /* upper limit of realizations per plugin call (1..256), default 256.  The managers cut a level's realizations into calls of at most this many, at most what the plugins prefer for the level (pmc_sampler_batch_width: 16 on large levels ... 256 on the smallest), cut so that every rank of a farm gets a share - independent of the lane count, so a realization's result does not depend on how many lanes share a GPU */
    int32_t max_rounds;       /* safety bound on the adaptive loop, default 1000            */
    const char* log_file;     /* "Output filename for MC managers" or NULL                  */
} pmc_mlmc_params;
void pmc_mlmc_params_default(pmc_mlmc_params* p);

typedef struct pmc_mlmc_result {
    int32_t nlevels;
    double estimate, eps2, actual_mse, estimator_variance, bias2, alpha, alpha_abs, beta, gamma;
    /* arrays of nlevels, owned by the manager, valid until the next call */
    const double *eY, *eABSY, *eQ, *eABSQ, *eC, *varY, *varQ, *consistency, *kurtosis, *VC, *cost;
    const double* sums;            /* nlevels x PMC_MLMC_NVAR, row-major */
    const int64_t* nsamples;       /* per level, global                   */
    const int64_t* nsamples_missing;
    const double* level_seconds;   /* wall time spent per level on this rank */
} pmc_mlmc_result;

/* manager over the device sampler + solver of pmc.h (device-resident vectors, no PCIe traffic) */
int pmc_mlmc_create(pmc_ctx* ctx, pmc_sampler* sampler, pmc_darcy* solver, int nlevels,
                    const pmc_mlmc_params* params, pmc_mlmc** out);
/* add a lane: another (ctx, sampler, solver) triple built from the same operators, i.e. another HIP stream
 * that processes this rank's realizations concurrently (device-handle managers only) */
int pmc_mlmc_add_lane(pmc_mlmc* m, pmc_ctx* ctx, pmc_sampler* sampler, pmc_darcy* solver);
/* manager over arbitrary plugins */
int pmc_mlmc_create_callbacks(int nlevels, const pmc_plugin_callbacks* cb, const pmc_mlmc_params* params,
                              pmc_mlmc** out);
void pmc_mlmc_destroy(pmc_mlmc* m);
/* sample-farm layout; reduce == NULL with nranks > 1 uses pmc_allreduce_sum_f64 (RCCL) of the ctx */
int pmc_mlmc_set_farm(pmc_mlmc* m, int nranks, int rank, pmc_reduce_fn reduce, void* user);
int pmc_mlmc_run(pmc_mlmc* m);                               /* MLMC_Manager::Run      */
int pmc_mlmc_reset(pmc_mlmc* m);                             /* zero sums and counters */
/* resume: add the realizations recorded in a per-sample log (params.log_file of an earlier run; columns level, Y, Q,
 * Q_c, cost as in the reference's MLMC.dat, src/MLMC_Manager.cpp:106-108) to the sums table and counters */
int pmc_mlmc_replay_log(pmc_mlmc* m, const char* path, int64_t* nread);
int pmc_mlmc_init_run(pmc_mlmc* m, const int32_t* nsamples); /* MLMC_Manager::InitRun  */
int pmc_mlmc_result_get(pmc_mlmc* m, pmc_mlmc_result* out);
/* the farm's one collective: wall milliseconds this rank has spent inside the SUM all-reduce of the accumulators (waiting
 * for the slowest rank included) and the number of reductions so far - one per InitRun round (the place the serial
 * reference would need it: src/MLMC_Manager.cpp:178, before computeNSamplesMSE) */
int pmc_mlmc_farm_times(pmc_mlmc* m, double* allreduce_ms, int64_t* reductions);
/* the table MLMC_Manager::ShowMe prints after every InitRun (src/MLMC_Manager.cpp:216-297), same labels / widths /
 * precision, into buf (NUL-terminated, truncated to cap); *needed (may be NULL) receives the full size incl. the NUL */
int pmc_mlmc_show_me(pmc_mlmc* m, char* buf, size_t cap, size_t* needed);
/* the per-level timers the reference keeps in parelag::TimeManager and prints at the end of a run (examples/MLMC.cpp:275):
 * "Sampler: Mult -- Level i" (src/PDESampler.cpp:328-333), "Darcy: Build Solver -- Level i", "Darcy: Mult -- Level i"
 * (src/DarcySolver.cpp:231-243) - device milliseconds (HIP events, pmc_stats.solve_ms / setup_ms) summed over the lanes of
 * this rank since the manager was created, as text (same calling convention as pmc_mlmc_show_me) and as numbers */
int pmc_mlmc_print_timers(pmc_mlmc* m, char* buf, size_t cap, size_t* needed);
int pmc_mlmc_phase_times(pmc_mlmc* m, int level, double* sampler_mult_ms, double* darcy_setup_ms, double* darcy_mult_ms,
                         int64_t* sampler_realizations, int64_t* darcy_realizations);
const char* pmc_host_last_error(void);

/* BayesianInverseProblem::ComputeLikelihood / ComputeLikelihoodAndQ / ComputeR (src/BayesianInverseProblem.cpp:188-218)
 * for nbatch realizations:  G = ComputeG(k) on the device (pmc_darcy_compute_G),
 *   likelihood = exp(-|G - G_obs|^2 / (2 noise)),   R = Q * likelihood.
 * likelihood, C: host arrays of nbatch; Q, R: host arrays or NULL. */
int pmc_bayes_likelihood(pmc_darcy* solver, int level, int nbatch, const double* k, int memspace, const double* G_obs,
                         int nobs, double noise, double* likelihood, double* C, double* Q, double* R);

/* ---- ML_BayesRatio_Manager / SL_BayesRatio_Manager (src/ML_BayesRatio_Manager.hpp:315-728, src/SL_BayesRatio_Manager.hpp)
 * Multilevel ratio estimator E[Q * likelihood] / E[likelihood]: per realization two INDEPENDENT prior draws, one for
 * Z = likelihood and one for R = Q * likelihood, each evaluated on level l and (same xi) on level l+1; the same
 * variance / bias / sample-allocation statistics as MLMC_Manager, computed for R and Z and combined by max.  nlevels == 1
 * is the single-level manager.  Sums use the reference's enum order (:66-69), 20 columns per level. */
#define PMC_RATIO_NVAR 20
typedef struct pmc_ratio pmc_ratio;
/* likelihood and R = Q*likelihood of nbatch realizations of k (host pointers) */
typedef int (*pmc_cb_likelihood)(void* user, int level, int nbatch, const double* k, double* likelihood, double* R,
                                 double* C);
typedef struct pmc_ratio_result {
    int32_t nlevels;
    double R_estimate, Z_estimate, ratio_estimate, eps2, actual_mse, estimator_variance, estimator_variance_R,
        estimator_variance_Z, bias2, bias2_R, bias2_Z, alpha_R, alpha_abs_R, beta_R, alpha_Z, alpha_abs_Z, beta_Z, gamma;
    const double *eR, *varR, *eYR, *varYR, *eABS_YR, *eZ, *varZ, *eYZ, *varYZ, *eABS_YZ, *eC, *cost;
    const double* sums;          /* nlevels x PMC_RATIO_NVAR */
    const int64_t *nsamples, *nsamples_missing;
    /* Ratio columns (q = r/z, y = q - q_c): what the *_Splitting managers estimate and allocate samples by */
    double alpha, alpha_abs, beta;
    const double *eRatio, *varRatio, *eYRatio, *varYRatio, *eABS_YRatio;
} pmc_ratio_result;
/* device version: prior = sampler, forward problem = solver with observation functionals set on every level */
int pmc_ratio_create(pmc_ctx* ctx, pmc_sampler* sampler, pmc_darcy* solver, int nlevels, const double* G_obs, int nobs,
                     double noise, const pmc_mlmc_params* params, pmc_ratio** out);
/* plugin version: cb->sample / cb->eval give the prior, `like` the likelihood and ratio integrand */
int pmc_ratio_create_callbacks(int nlevels, const pmc_plugin_callbacks* cb, pmc_cb_likelihood like,
                               const pmc_mlmc_params* params, pmc_ratio** out);
void pmc_ratio_destroy(pmc_ratio* m);
int pmc_ratio_set_farm(pmc_ratio* m, int nranks, int rank, pmc_reduce_fn reduce, void* user);
/* on != 0: ML_BayesRatio_Splitting_Manager / SL_BayesRatio_Splitting_Manager (src/ML_BayesRatio_Splitting_Manager.hpp:
 * 297-432 InitRun, :595-737 computeNSamplesMSE): estimate E[R/Z] by the level differences r/z - r_c/z_c; variance,
 * bias and sample allocation follow the Ratio columns; ratio_estimate = sum_l E[Y_Ratio,l] */
int pmc_ratio_set_splitting(pmc_ratio* m, int on);
int pmc_ratio_run(pmc_ratio* m);
int pmc_ratio_init_run(pmc_ratio* m, const int32_t* nsamples);
int pmc_ratio_result_get(pmc_ratio* m, pmc_ratio_result* out);

/* ---- P0 x P0 mortar matrix between two NON-MATCHING meshes:  G[i,j] = | A_i  n  B_j |  (setup side, host only).
 * Replaces ParMortarAssembler::Assemble (src/transfer/ParMortarAssembler.cpp:1127-1144) as used for Gt by
 * L2ProjectionPDESampler::BuildHierarchy (src/L2ProjectionPDESampler.cpp:488-505) with A = original mesh, B = enlarged
 * mesh; the coarser levels follow by RAP(orig_Ps, Gt, Ps) (:512-513).  Elements: triangles / quadrilaterals (dim 2),
 * tetrahedra / hexahedra in MFEM vertex order (dim 3); they are split into simplices and intersected exactly. */
typedef struct pmc_mesh_view {
    int32_t dim, nverts, nelems, verts_per_elem;
    const double* verts;   /* nverts x dim, row-major         */
    const int32_t* elems;  /* nelems x verts_per_elem          */
} pmc_mesh_view;
typedef struct pmc_mortar pmc_mortar;
/* rel_tol: intersections below rel_tol * min(|A_i|, |B_j|) are dropped (<= 0: 1e-12) */
int pmc_mortar_assemble(const pmc_mesh_view* a, const pmc_mesh_view* b, double rel_tol, pmc_mortar** out);
int64_t pmc_mortar_nnz(const pmc_mortar* m);
/* copies the CSR arrays (rowptr: nelems_a + 1) and the element measures of both meshes; NULL pointers are skipped */
int pmc_mortar_get(const pmc_mortar* m, int32_t* rowptr, int32_t* colind, double* vals, double* measure_a,
                   double* measure_b);
void pmc_mortar_destroy(pmc_mortar* m);
const char* pmc_mortar_last_error(void);

/* expWRegression (src/Utilities.cpp:257-283), exported for the host-logic tests */
double pmc_exp_w_regression(const double* y, const double* x, int n, int skip_n_last);

#ifdef __cplusplus
}
#endif
#endif /* PMC_HOST_H_ */
