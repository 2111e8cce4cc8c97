// Adapter between ParELAGMC's MFEM-typed plugin interface and libpmc.
//
// Header-only.  With -DPMC_WITH_MFEM it includes <mfem.hpp>; otherwise the including translation unit must already have
// declared mfem::Vector, mfem::SparseMatrix and mfem::Array<int> with the handful of members used here (Size, SetSize,
// GetData, Height, Width, GetI, GetJ) - tests/c/mfem_shim.hpp does, so that this header is compiled and run in CI on a
// machine without MFEM.  Nothing else of MFEM or ParELAG is needed: the operators are handed over as the CSR arrays
// mfem::SparseMatrix holds (for a ParELAG hierarchy: the diagonal blocks of the single-rank HypreParMatrix objects,
// agglomerated coarse levels and their prolongators included - any CSR is accepted, see pmc_solver_opts.mg_coarsening).
//
// The two classes carry the reference's method names and argument meaning:
//   DevicePDESampler   MLSampler / PDESampler / EmbeddedPDESampler / L2ProjectionPDESampler
//                      (/root/reference/src/MLSampler.hpp:33-87, src/PDESampler.cpp:336-535)
//   DeviceDarcySolver  PhysicalMLSolver / DarcySolver (src/PhysicalMLSolver.hpp:33-62, src/DarcySolver.cpp:416-470)
// and throw std::runtime_error where the reference throws through PARELAG_TEST_FOR_EXCEPTION.
#pragma once

#ifdef PMC_WITH_MFEM
#include <mfem.hpp>
#endif

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/pmc.h"

namespace parelagmc {
namespace mfem_adapter {

inline pmc_csr as_csr(const mfem::SparseMatrix& A) {
    return pmc_csr{A.Height(), A.Width(), A.GetI(), A.GetJ(), A.GetData()};
}
inline void check(int rc, const char* what) {
    if (rc != PMC_OK) throw std::runtime_error(std::string(what) + ": " + pmc_last_error());
}

/// Blocks PDESampler::BuildHierarchy assembles on one level (src/PDESampler.cpp:232-284)
struct SamplerLevelOps {
    const mfem::SparseMatrix* M = nullptr;   ///< u-mass matrix, essential rows / columns eliminated (:236-241)
    const mfem::SparseMatrix* B = nullptr;   ///< W * D with the essential columns zeroed (:243-246)
    const mfem::Vector* w_diag = nullptr;    ///< diag(W) before the -alpha scaling (:248-258)
    const mfem::SparseMatrix* P = nullptr;   ///< ComputeTrueP(sform) to the next coarser level (:189-193); null on the last
};

/// What the same level holds BEFORE the boundary elimination - the input of the hybridized solver (the reference hands
/// A[i] and the de Rham sequence to prec_factory->BuildSolver(A[i], state), src/PDESampler.cpp:302-318; ParELAG's
/// HybridHdivL2 reads the element matrices from the sequence)
struct HybridLevelOps {
    const mfem::SparseMatrix* M_pattern = nullptr;   ///< sparsity of the u-mass matrix
    const mfem::Array<int>* c_ptr = nullptr;         ///< element decomposition, as DarcyLevelOps
    const mfem::Array<int>* c_elem = nullptr;
    const mfem::Vector* c_val = nullptr;
    const mfem::SparseMatrix* B = nullptr;           ///< W * D as assembled, NO eliminated columns (:232-234)
    const mfem::Vector* w_diag = nullptr;
    const mfem::SparseMatrix* P = nullptr;
};

class DevicePDESampler {
  public:
    /// prec_type == "Hybridization" (src/PDESampler.cpp:291,307-311): element-local elimination inside the library
    /// (pmc_hybrid_build), MINRES + aggregation V-cycle on the multiplier system; every level is a Monte Carlo level
    DevicePDESampler(int device_id, const std::vector<HybridLevelOps>& levels, double alpha, double matern_coeff,
                     bool lognormal, const pmc_solver_opts* opts = nullptr, uint64_t seed = 0)
        : nlevels_((int)levels.size()) {
        check(pmc_ctx_create(device_id, &ctx_), "pmc_ctx_create");
        try {
            check(pmc_rng_seed(ctx_, seed, 1, 0), "pmc_rng_seed");
            std::vector<pmc_hybrid_elements> lv(levels.size());
            for (size_t i = 0; i < levels.size(); ++i) {
                const HybridLevelOps& L = levels[i];
                if (!L.M_pattern || !L.c_ptr || !L.c_elem || !L.c_val || !L.B || !L.w_diag)
                    throw std::runtime_error("DevicePDESampler: hybrid level operators missing");
                lv[i] = pmc_hybrid_elements{};
                lv[i].n_u = L.M_pattern->Height();
                lv[i].n_s = L.B->Height();
                lv[i].M_pattern = as_csr(*L.M_pattern);
                lv[i].c_ptr = L.c_ptr->GetData();
                lv[i].c_elem = L.c_elem->GetData();
                lv[i].c_val = L.c_val->GetData();
                lv[i].B = as_csr(*L.B);
                lv[i].w_diag = L.w_diag->GetData();
                if (L.P) lv[i].P = as_csr(*L.P);
            }
            check(pmc_sampler_create_hybrid_from_elements(ctx_, (int)lv.size(), lv.data(), alpha, matern_coeff,
                                                          lognormal ? 1 : 0, opts, &h_),
                  "pmc_sampler_create_hybrid_from_elements");
        } catch (...) {
            pmc_ctx_destroy(ctx_);
            throw;
        }
    }
    /// n_mc_levels <= levels.size(): further levels only deepen the V-cycle of the preconditioner
    DevicePDESampler(int device_id, const std::vector<SamplerLevelOps>& levels, int n_mc_levels, double alpha,
                     double matern_coeff, bool lognormal, const pmc_solver_opts* opts = nullptr, uint64_t seed = 0)
        : nlevels_(n_mc_levels) {
        check(pmc_ctx_create(device_id, &ctx_), "pmc_ctx_create");
        try {
            check(pmc_rng_seed(ctx_, seed, 1, 0), "pmc_rng_seed");
            std::vector<pmc_sampler_level> lv(levels.size());
            for (size_t i = 0; i < levels.size(); ++i) {
                const SamplerLevelOps& L = levels[i];
                if (!L.M || !L.B || !L.w_diag) throw std::runtime_error("DevicePDESampler: level operators missing");
                lv[i] = pmc_sampler_level{};
                lv[i].n_u = L.M->Height();
                lv[i].n_s = L.B->Height();
                lv[i].M = as_csr(*L.M);
                lv[i].B = as_csr(*L.B);
                lv[i].w_diag = L.w_diag->GetData();
                if (L.P) lv[i].P = as_csr(*L.P);
            }
            check(pmc_sampler_create(ctx_, (int)lv.size(), n_mc_levels, lv.data(), alpha, matern_coeff, lognormal ? 1 : 0,
                                     opts, &h_),
                  "pmc_sampler_create");
        } catch (...) {
            pmc_ctx_destroy(ctx_);
            throw;
        }
    }
    DevicePDESampler(const DevicePDESampler&) = delete;
    DevicePDESampler& operator=(const DevicePDESampler&) = delete;
    ~DevicePDESampler() {
        pmc_sampler_destroy(h_);
        pmc_ctx_destroy(ctx_);
    }

    /// NormalDistributionSampler::Split (src/NormalDistributionSampler.cpp:21-24)
    void Split(uint64_t seed, int nparts, int mypart) { check(pmc_rng_seed(ctx_, seed, nparts, mypart), "Split"); }
    /// EmbeddedPDESampler: indices of the original-mesh elements in the embedded mesh (src/EmbeddedPDESampler.cpp:63-89)
    void SetEmbedding(int level, const mfem::Array<int>& orig_index) {
        check(pmc_sampler_set_projection(h_, level, PMC_PROJ_GATHER, nullptr, orig_index.GetData(), nullptr, orig_index.Size()),
              "SetEmbedding");
    }
    /// L2ProjectionPDESampler: Gt (original x embedded elements, src/L2ProjectionPDESampler.cpp:488-514) and the
    /// reciprocal original element volumes (:603-611)
    void SetL2Projection(int level, const mfem::SparseMatrix& Gt, const mfem::Vector& inv_orig_volume) {
        const pmc_csr g = as_csr(Gt);
        check(pmc_sampler_set_projection(h_, level, PMC_PROJ_L2, &g, nullptr, inv_orig_volume.GetData(), Gt.Height()),
              "SetL2Projection");
    }

    void BuildHierarchy() {}   // operators arrive assembled (constructor)
    void Sample(const int level, mfem::Vector& xi) {
        const int n = pmc_sampler_xi_size(h_, level);
        if (n < 0) throw std::runtime_error("Sample: level out of range");
        xi.SetSize(n);
        check(pmc_sampler_sample(h_, level, next_id_++, 1, xi.GetData(), PMC_MEM_HOST), "Sample");
    }
    void Eval(const int level, const mfem::Vector& xi, mfem::Vector& s) {
        s.SetSize(SampleSize(level));
        check(pmc_sampler_eval(h_, level, level_of(xi.Size()), 1, xi.GetData(), s.GetData(), nullptr, -1, 0, nullptr,
                               PMC_MEM_HOST, &stats_),
              "Eval");
    }
    /// u: Gaussian field on the sampler mesh; read as the initial guess when use_init (a coarser level's field is
    /// prolongated, src/PDESampler.cpp:498-510), written on return (:527)
    void Eval(const int level, const mfem::Vector& xi, mfem::Vector& s, mfem::Vector& u, bool use_init) {
        const int init_level = use_init ? level_of(u.Size()) : -1;
        std::vector<double> init;
        if (use_init) init.assign(u.GetData(), u.GetData() + u.Size());
        s.SetSize(SampleSize(level));
        u.SetSize(pmc_sampler_xi_size(h_, level));
        check(pmc_sampler_eval(h_, level, level_of(xi.Size()), 1, xi.GetData(), s.GetData(), use_init ? init.data() : nullptr,
                               init_level, use_init ? 1 : 0, u.GetData(), PMC_MEM_HOST, &stats_),
              "Eval");
    }
    int SampleSize(int level) const { return pmc_sampler_sample_size(h_, level); }
    size_t GetNNZ(int level) const { return (size_t)pmc_sampler_nnz(h_, level); }
    int GetNumIters() const { return stats_.iterations; }     // the reference returns -1 (src/PDESampler.hpp:142-145)
    pmc_csr GetTrueP(int level) const {
        pmc_csr P{};
        check(pmc_sampler_true_p(h_, level, &P), "GetTrueP");
        return P;
    }
    pmc_sampler* handle() { return h_; }
    pmc_ctx* context() { return ctx_; }

  private:
    int level_of(int size) const {                            // level_size.Find(xi.Size()), src/PDESampler.cpp:419
        for (int l = 0; l < nlevels_; ++l)
            if (pmc_sampler_xi_size(h_, l) == size) return l;
        throw std::runtime_error("DevicePDESampler: vector length matches no level");
    }
    pmc_ctx* ctx_ = nullptr;
    pmc_sampler* h_ = nullptr;
    int nlevels_;
    uint64_t next_id_ = 0;
    pmc_stats stats_{};
};

/// What DarcySolver precomputes on one level (src/DarcySolver.cpp:194-227,297-319,360-414) plus the element
/// decomposition of ComputeMassOperator(uform, k): M(k)[p] = sum_t coef(k[c_elem[t]]) c_val[t], t in c_ptr[p]..c_ptr[p+1]
struct DarcyLevelOps {
    const mfem::SparseMatrix* M_pattern = nullptr;
    const mfem::Array<int>* c_ptr = nullptr;
    const mfem::Array<int>* c_elem = nullptr;
    const mfem::Vector* c_val = nullptr;
    const mfem::SparseMatrix* B = nullptr;
    const mfem::Vector* rhs = nullptr;          ///< n_u + n_p
    const mfem::Array<int>* ess_dofs = nullptr; ///< marked essential u-dofs (list), :487-492
    const mfem::Vector* ess_data = nullptr;     ///< n_u
    const mfem::Vector* obs = nullptr;          ///< n_u + n_p
    const mfem::SparseMatrix* P = nullptr;      ///< p-space prolongator to the next coarser level; null on the last
};

class DeviceDarcySolver {
  public:
    // hybridization: the reference's "Hybridization" solver option (src/DarcySolver.cpp:586,619) - SolveFwd through the
    // element-local elimination and the multiplier system (pmc_darcy_create_hybrid); same operators, same results
    DeviceDarcySolver(pmc_ctx* ctx, const std::vector<DarcyLevelOps>& levels, int n_mc_levels, bool k_divides,
                      const pmc_solver_opts* opts = nullptr, bool hybridization = false) {
        std::vector<pmc_darcy_level> lv(levels.size());
        std::vector<std::vector<uint8_t>> mask(levels.size());
        for (size_t i = 0; i < levels.size(); ++i) {
            const DarcyLevelOps& L = levels[i];
            if (!L.M_pattern || !L.c_ptr || !L.c_elem || !L.c_val || !L.B || !L.rhs || !L.ess_data || !L.obs)
                throw std::runtime_error("DeviceDarcySolver: level operators missing");
            lv[i] = pmc_darcy_level{};
            lv[i].n_u = L.M_pattern->Height();
            lv[i].n_p = L.B->Height();
            lv[i].M_pattern = as_csr(*L.M_pattern);
            lv[i].c_ptr = L.c_ptr->GetData();
            lv[i].c_elem = L.c_elem->GetData();
            lv[i].c_val = L.c_val->GetData();
            lv[i].B = as_csr(*L.B);
            lv[i].rhs = L.rhs->GetData();
            mask[i].assign((size_t)lv[i].n_u, 0);
            if (L.ess_dofs)
                for (int j = 0; j < L.ess_dofs->Size(); ++j) mask[i][(size_t)L.ess_dofs->GetData()[j]] = 1;
            lv[i].ess_mask = mask[i].data();
            lv[i].ess_data = L.ess_data->GetData();
            lv[i].obs = L.obs->GetData();
            if (L.P) lv[i].P = as_csr(*L.P);
        }
        if (hybridization)
            check(pmc_darcy_create_hybrid(ctx, (int)lv.size(), n_mc_levels, lv.data(), k_divides ? 1 : 0, opts, &h_),
                  "pmc_darcy_create_hybrid");
        else
            check(pmc_darcy_create(ctx, (int)lv.size(), n_mc_levels, lv.data(), k_divides ? 1 : 0, opts, &h_), "pmc_darcy_create");
    }
    DeviceDarcySolver(const DeviceDarcySolver&) = delete;
    DeviceDarcySolver& operator=(const DeviceDarcySolver&) = delete;
    ~DeviceDarcySolver() { pmc_darcy_destroy(h_); }

    void SolveFwd(int ilevel, mfem::Vector& k_over_k_ref, double& Q, double& C) {
        check(pmc_darcy_solve_fwd(h_, ilevel, 1, k_over_k_ref.GetData(), &Q, &C, nullptr, PMC_MEM_HOST, nullptr), "SolveFwd");
    }
    void SolveFwd_RtnPressure(int ilevel, mfem::Vector& k_over_k_ref, mfem::Vector& P, double& C, double& Q, bool compute_Q) {
        P.SetSize(pmc_darcy_num_pressure_dofs(h_, ilevel));
        check(pmc_darcy_solve_fwd_pressure(h_, ilevel, 1, k_over_k_ref.GetData(), P.GetData(), &C, &Q, compute_Q ? 1 : 0,
                                           PMC_MEM_HOST, nullptr),
              "SolveFwd_RtnPressure");
    }
    int GetNumberOfDofs(int ilevel) const { return pmc_darcy_num_dofs(h_, ilevel); }
    int GetGlobalNumberOfDofs(int ilevel) const { return pmc_darcy_num_dofs(h_, ilevel); }
    int GetNNZ(int ilevel) const { return (int)pmc_darcy_nnz(h_, ilevel); }
    int GetSizeOfStochasticData(int ilevel) const { return pmc_darcy_num_pressure_dofs(h_, ilevel); }
    pmc_darcy* handle() { return h_; }

  private:
    pmc_darcy* h_ = nullptr;
};

}  // namespace mfem_adapter
}  // namespace parelagmc
