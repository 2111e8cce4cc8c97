// Device-side objects behind the opaque C handles.
#pragma once
#include "solver.hpp"

namespace pmc {

HostCsr schur_host(const HostCsr& B, const HostCsr& Bt, const std::vector<double>& dM, const double* diag_add);
// device V-cycle hierarchy (shared values) from a host smoothed-aggregation hierarchy of diag(w) + K
std::unique_ptr<Multigrid> build_sa_chain(const HostCsr& K, const std::vector<double>& w, const pmc_solver_opts& o,
                                          hipStream_t st);

struct SamplerLevel {
    double ratio_M = 8.0;            // Chebyshev interval lambda_max/lambda_min of the l1-scaled M-block
    int n_u = 0, n_s = 0;
    int64_t nnz = 0;
    Sell A;                 // [M Bt; B -aW]
    Sell M;
    DevBuf<double> dinvM;   // 1 / l1 row sums of M
    DevBuf<double> M_scaled; // M D^-1 on M's SELL pattern
    DevBuf<double> w_sqrt;
    int proj = PMC_PROJ_NONE;
    int out_size = 0;
    Sell Gt;
    DevBuf<int> gather;
    DevBuf<double> inv_w;
    HostCsr P_host;         // ComputeTrueP(sform) as handed over (MLSampler::GetTrueP); empty on the last level
    // hybridized sampler (Sampler::hybrid): A = H on the multipliers (n_u = n_lambda), rhs = Gz (z f), s = z f - Gl lambda
    Sell Gz;                // G diag(1 / z): n_lambda x n_s
    Sell Gl;                // G^T: n_s x n_lambda
    Sell Ptz;               // diag(z) P^T of the next finer level: restriction of a finer xi that lands on z f directly
    DevBuf<double> zw_sqrt; // z .* sqrt(w)
    // internal numbering of the multipliers (aggregates of the V-cycle's finest level contiguous, agg_pack_rows): internal
    // row i is the caller's row lam_new2old[i]; empty = the caller's numbering
    DevBuf<int> lam_new2old, lam_old2new;
};

struct Sampler {
    Ctx& ctx;
    int nlevels, n_mc;
    double alpha, g;
    bool lognormal;
    pmc_solver_opts opts;
    std::vector<SamplerLevel> lv;
    Multigrid mg;                                  // caller's levels: transfers between MC levels + geometric V-cycle
    std::vector<std::unique_ptr<Multigrid>> amg;   // per MC level: internal smoothed-aggregation hierarchy (if selected)
    double anisotropy = 1.0;
    bool hybrid = false;                           // pmc_sampler_create_hybrid: multiplier system, amg[l] its V-cycle
    OpTimer vc_timer;                              // hybrid: the finest level's post-smoothing launches (with work.op_timer.on)
    MinresWork work;
    DevBuf<double> rhs, sol, tA, tB, cx, cd, cx2, stage_in, stage_out, stage_emb, mini_scratch;
    DevBuf<pmc_stats> mini_stats;

    Sampler(Ctx& c, int nlevels, int n_mc, const pmc_sampler_level* in, double alpha, double g, bool lognormal,
            const pmc_solver_opts& o);
    Sampler(Ctx& c, int nlevels, const pmc_hybrid_level* in, double alpha, double g, bool lognormal, const pmc_solver_opts& o);
    // rows of the vectors the Krylov solver of `level` iterates on
    // algorithmic bytes of one launch of the timed post-smoothing kernel (level 0 of the hybrid V-cycle of `level`)
    double smoother_bytes(int level, int nb) const;
    size_t system_rows(int level) const { return hybrid ? (size_t)lv[level].n_u : (size_t)lv[level].n_u + lv[level].n_s; }
    void set_projection(int level, int kind, const pmc_csr* Gt, const int32_t* idx, const double* inv_w, int orig_size);
    void sample(int level, uint64_t first_id, int nbatch, double* xi, int memspace);
    void eval(int level, int xi_level, int nbatch, const double* xi, double* s_out, const double* init_s, int init_level,
              bool use_init, double* emb_out, int memspace, pmc_stats* stats);
    void apply_operator(int level, int nb, const double* x, double* y, int memspace, int repeat, double* avg_ms,
                        double* bytes);
    // invA[level]->Mult(rhs, sol) on full vectors of n_u + n_s entries per realization (sample-major)
    void mult(int level, int nbatch, const double* rhs, double* sol, bool use_guess, int memspace, pmc_stats* stats);
    // z = B^-1 r: one application of the MINRES preconditioner (diagnostics: true preconditioned residual norms)
    void apply_preconditioner(int level, int nbatch, const double* r, double* z, int memspace);

  private:
    void ensure(int level, int nb);
    void solve_system(int level, int nb, bool zero_guess, int x_row0, int x_nrows, pmc_stats* stats);
    PrecFn preconditioner(int level, int nb, int degM, Multigrid* mgp, int mg_l0);
    void eval_chunk(int level, int xi_level, int nb, const double* xi_d, double* s_d, const double* init_d,
                    int init_level, bool use_init, double* emb_d, pmc_stats* stats);
};

struct DarcyLevel {
    int n_u = 0, n_p = 0, n_coef = 0;
    double ratio_M = 8.0;            // Chebyshev interval of the l1-scaled M-block (from M(k == 1) unless given)
    int64_t nnz = 0;
    Sell M;                          // pattern only; values are per-realization
    DevBuf<int> slot_src, c_ptr, c_elem;
    DevBuf<double> c_val;
    // element-grouped form of M(k) (EgView): shared element-matrix entries + two coefficient rows per dof; when present
    // (every dof belongs to at most two elements) the solve never materialises per-realization values of M
    Sell Meg;
    DevBuf<int> eg_e12;
    int eg_gw = 0;
    bool has_eg = false;
    Sell B, Bt;                      // shared +-1 values (essential columns/rows removed)
    DevBuf<unsigned char> ess;
    DevBuf<double> ess_data, rhs_u0, rhs_p, obs;
    // Schur complement refresh: S(k) = B diag(M(k))^-1 B^T on the fixed pattern of mg.L[level].S
    DevBuf<int> s_ptr, s_idx;        // per SELL slot of S: faces contributing
    DevBuf<double> s_w;
    DevBuf<int> s_diag_slot;         // per row: SELL slot of the diagonal entry
    // Galerkin coarse operator refresh: S_{l+1} = 1/2 P^T S_l P (lists of fine SELL slots per coarse slot)
    DevBuf<int> g_ptr, g_idx;
    DevBuf<double> g_w;
    // per-realization values
    DevBuf<double> coef, mvals, mvals_scaled, diagM, l1invM, rhs_bc;
    // support of the observation functional (rows with obs != 0) and its weights: when the caller does not ask for
    // the solution vector, MINRES only maintains these rows of it
    DevBuf<int> obs_rows;
    DevBuf<double> obs_w;
    int n_obs = 0;
    // Bayesian observation functionals g_i (src/BayesianInverseProblem.cpp:178-186): rows of Gobs act on the pressure
    // block.  MINRES then maintains the union of supp(obs) and supp(g_i); Gobs is stored on that compact numbering.
    int n_gobs = 0, n_grows = 0;
    Sell Gobs;
    DevBuf<int> g_rows;
    DevBuf<double> g_obs_w, g_norm;
};

// Internal algebraic hierarchy of one Monte Carlo level (mg_coarsening): smoothed-aggregation prolongators frozen at
// k == 1, operators S_0(k) = B diag(M(k))^-1 B^T and S_{j+1}(k) = P_j^T S_j(k) P_j refreshed per realization on fixed
// patterns through contribution lists (the same refresh kernel as the geometric hierarchy).
struct DarcyChainLevel {
    DevBuf<int> ptr, idx, diag_slot;   // per SELL slot: list into diag(M) (level 0, reciprocal) or the finer level's slots
    DevBuf<double> w;
};
struct DarcyChain {
    Multigrid mg;
    std::vector<DarcyChainLevel> cl;
};

// Hybridized form of one Monte Carlo level (pmc_darcy_create_hybrid: the reference's "Hybridization" branch of DarcySolver,
// src/DarcySolver.cpp:586,619; algebra: parelagmc_amd/fe/darcy_hybrid.py).  One multiplier per interior / essential face,
//     H(kappa) lambda = R kappa + b_0,     H(kappa) = sum_e kappa_e C_e X_e C_e^T     (kappa = 1 / c(k), SPD, linear in kappa)
//     u = kappa_owner (U_0 - U_L lambda) + u_g,     p = P_0 - P_L lambda - z_g / kappa.
// chain: level 0 holds H(kappa) on its fixed pattern through contribution lists over the coefficient table, the coarser
// levels the OVER-CORRECTED Galerkin products s P^T H P of a plain-aggregation hierarchy frozen at kappa == 1 (LAB_NOTES
// 10.15: s = 0.45 makes the iteration count nearly level-independent), all refreshed per realization.
struct DarcyHybrid {
    int n_lambda = 0;
    std::unique_ptr<DarcyChain> chain;
    Sell R, UL, PL;
    // element-grouped form of H(kappa) (EgView, as DarcyLevel::Meg for M(k)): the operator of the MINRES loop never reads the
    // explicit per-realization values of level 0 (1.1 GB per pass at 0.8 M multipliers x 16)
    Sell Heg;
    DevBuf<int> eg_e12, no_rows;       // no_rows: slice offsets of an operator without entries (the kernel's second operand)
    int eg_gw = 0;
    DevBuf<double> b0, U0, ug, P0, zg;
    DevBuf<int> owner;
    DevBuf<double> coef, rhs, lam, tu, tp;     // per launch: kappa [n_p][nb], right-hand side, multipliers, U_L lambda, P_L lambda
    // finest level of the V-cycle in element-grouped form (solve_chunk_hybrid): -kappa (the residual r - H x is ONE launch of
    // the operator kernel with the negated coefficients and the identity as its second operand), iterate, residual, update
    Sell ident;
    DevBuf<double> negcoef, vx, vres, vd, vxc;
};

struct Darcy {
    Ctx& ctx;
    int nlevels, n_mc;
    bool k_divides;
    bool hybrid = false;             // SolveFwd through the hybridized form (ComputeG keeps the saddle-point path)
    std::vector<std::unique_ptr<DarcyHybrid>> hyb;   // per MC level
    pmc_solver_opts opts;
    std::vector<DarcyLevel> lv;
    Multigrid mg;                    // batched values
    std::vector<std::unique_ptr<DarcyChain>> chains;   // per MC level, empty unless algebraic coarsening is selected
    double anisotropy = 1.0;
    DevBuf<double> gwork;
    MinresWork work;
    // in-loop timing of the dominant kernel of the Darcy operator: the u-rows [M(k) | B^T] (k::eg_pair_spmm with the fused
    // <x, Ax>; k::pair_spmm when the element-grouped form is not available)
    OpTimer op_timer;
    double operator_bytes(int level, int nb) const;   // algorithmic bytes of ONE such launch
    // ... and of the M-block polynomial of the preconditioner (k::eg_poly2, the other large gather kernel of an iteration):
    // timed on the same switch, on the main stream instead of beside the V-cycle's bottom
    OpTimer poly_timer;
    double poly_bytes(int level, int nb) const;
    DevBuf<double> sol, sol_compact, cx, cd, cx2, stage_k, stage_sol, qpartial, qout, gtmp, gout;
    bool use_eg(const DarcyLevel& d) const;
    void set_observations(int level, const pmc_csr* Gobs);
    void compute_G(int level, int nbatch, const double* k, double* G, double* C, double* Q, int memspace, pmc_stats* stats);

    Darcy(Ctx& c, int nlevels, int n_mc, const pmc_darcy_level* in, bool k_divides, const pmc_solver_opts& o,
          bool hybrid = false);
    // sol_kind: 0 none, 1 full solution (n_u+n_p per realization), 2 pressure block only (n_p per realization)
    void solve_fwd(int level, int nbatch, const double* k, double* Q, double* C, double* sol_out, int memspace,
                   pmc_stats* stats, int sol_kind = 1);

  private:
    void ensure(int level, int nb);
    void solve_chunk(int level, int nb, const double* k_d, double* Q_host, double* sol_d, pmc_stats* stats, int row0,
                     int nrows, double* G_host);
    void build_hybrid(int level, const pmc_darcy_level& L);
    void solve_chunk_hybrid(int level, int nb, const double* k_d, double* Q_host, double* sol_d, pmc_stats* stats, int row0,
                            int nrows);
};

}  // namespace pmc

struct pmc_sampler { pmc::Sampler impl; template <class... A> explicit pmc_sampler(A&&... a) : impl(std::forward<A>(a)...) {} };
struct pmc_darcy { pmc::Darcy impl; template <class... A> explicit pmc_darcy(A&&... a) : impl(std::forward<A>(a)...) {} };
