// Host-side C++ mirror of ParELAGMC's plugin surface and Monte Carlo managers.
//
// Same class and method names, argument meaning and error behaviour (exceptions) as the
// reference (paths relative to /root/reference):
//   MLSampler                 src/MLSampler.hpp:33-52
//   PhysicalMLSolver          src/PhysicalMLSolver.hpp:33-62
//   NormalDistributionSampler src/NormalDistributionSampler.hpp:27-64
//   PDESampler                src/PDESampler.hpp (Sample/Eval/SampleSize/GetNNZ)
//   DarcySolver               src/DarcySolver.hpp (SolveFwd/GetNumberOfDofs/GetNNZ)
//   MLMC_Manager              src/MLMC_Manager.hpp:24-181
//   MC_Manager                src/MC_Manager.hpp
// mfem::Vector is replaced by parelagmc::Vector, a (pointer, size, memory space) view that can
// live in HBM, so a whole realization (xi -> s -> Q) never crosses PCIe.  The numerical work is
// delegated to libpmc.so through include/pmc.h; nothing here touches a GPU API directly.
#pragma once

#include <cstdint>
#include <fstream>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <ostream>
#include <vector>

#include "../../include/pmc.h"
#include "../../include/pmc_host.h"

namespace parelagmc {

// A batch of `nbatch` vectors of `size` doubles each, sample-major, in host or device memory.
class Vector {
  public:
    Vector() = default;
    Vector(pmc_ctx* ctx, int memspace) : ctx_(ctx), memspace_(memspace) {}
    Vector(const Vector&) = delete;
    Vector& operator=(const Vector&) = delete;
    ~Vector();
    void SetSize(int size, int nbatch = 1);   // reallocates when the capacity is too small
    void Swap(Vector& o);                     // exchange storage (same context and memory space)
    void Adopt(double* p, int size, int nbatch) { data_ = p; size_ = size; nbatch_ = nbatch; cap_ = (size_t)size * nbatch; }
    void Release() { data_ = nullptr; cap_ = 0; }   // forget adopted storage without freeing it
    int Size() const { return size_; }
    int Batch() const { return nbatch_; }
    int MemSpace() const { return memspace_; }
    double* GetData() { return data_; }
    const double* GetData() const { return data_; }

  private:
    pmc_ctx* ctx_ = nullptr;
    int memspace_ = PMC_MEM_HOST;
    double* data_ = nullptr;
    size_t cap_ = 0;
    int size_ = 0, nbatch_ = 1;
};

/// Accumulated device time of one plugin on one level: the reference's TimeManager entries "Sampler: Mult -- Level i"
/// (src/PDESampler.cpp:328-333), "Darcy: Mult -- Level i" and "Darcy: Build Solver -- Level i" (src/DarcySolver.cpp:231-243),
/// measured with HIP events on the plugin's stream (pmc_stats.solve_ms / setup_ms).
struct PhaseTimes {
    double mult_ms = 0.0;    // Krylov solves
    double setup_ms = 0.0;   // per-realization work ahead of them (rhs / M(k), elimination, Schur hierarchy refresh)
    int64_t realizations = 0;
    void add(const pmc_stats* st, int n) {
        for (int i = 0; i < n; ++i) { mult_ms += st[i].solve_ms; setup_ms += st[i].setup_ms; }
        realizations += n;
    }
};

class MLSampler {
  public:
    virtual ~MLSampler() = default;
    /// Fill xi with nbatch realizations of white noise for `level`; ids first_id .. first_id+nbatch-1
    virtual void Sample(const int level, Vector& xi, uint64_t first_id = 0, int nbatch = 1) = 0;
    /// Evaluate random field at level with random sample xi
    virtual void Eval(const int level, const Vector& xi, Vector& s) = 0;
    /// ... storing the Gaussian field in u and optionally warm-starting from it (use_init)
    virtual void Eval(const int level, const Vector& xi, Vector& s, Vector& u, bool use_init) = 0;
    virtual int SampleSize(int level) const = 0;
    virtual size_t GetNNZ(int level) const = 0;
    /// Build the hierarchy of the sampler (src/MLSampler.hpp:64-66).  The operators arrive assembled through the C ABI,
    /// so for handle-backed samplers this only checks that the handle exists; plugins may do real work here.
    virtual void BuildHierarchy() {}
    /// Prolongator of the sample space from level+1 to level (src/MLSampler.hpp:85-87; a HypreParMatrix there, a CSR
    /// view with host pointers owned by the sampler here).  Default: not available.
    virtual pmc_csr GetTrueP(int /*level*/) const { throw std::runtime_error("GetTrueP: not provided by this sampler"); }
    /// Realizations of `level` the plugin processes at once most efficiently (0 = no preference).  The managers cut a
    /// level's realizations into plugin calls of at most this many (device plugins: 16 on large levels, up to 256 on the
    /// smallest - pmc_sampler_batch_width).
    virtual int PreferredBatch(int /*level*/) const { return 0; }
    /// Device time spent on `level` since construction / ResetPhaseTimes (zero for plugins that do not measure)
    virtual PhaseTimes GetPhaseTimes(int /*level*/) const { return PhaseTimes(); }
    virtual void ResetPhaseTimes() {}
};

class PhysicalMLSolver {
  public:
    virtual ~PhysicalMLSolver() = default;
    /// Solve and update quantity of interest Q, cost C (arrays of k.Batch() entries)
    virtual void SolveFwd(int ilevel, Vector& k_over_k_ref, double* Q, double* C) = 0;
    /// The reference's signature (src/PhysicalMLSolver.hpp:33-39): one realization, Q and C by reference
    void SolveFwd(int ilevel, Vector& k_over_k_ref, double& Q, double& C) {
        if (k_over_k_ref.Batch() != 1) throw std::invalid_argument("SolveFwd(double&, double&): one realization expected");
        SolveFwd(ilevel, k_over_k_ref, &Q, &C);
    }
    /// Solve and return the pressure block P (k.Batch() x pressure dofs), cost C and, when compute_Q, the QoI Q
    /// (src/PhysicalMLSolver.hpp:41-47).  Arrays of k.Batch() entries.
    virtual void SolveFwd_RtnPressure(int /*ilevel*/, Vector& /*k_over_k_ref*/, Vector& /*P*/, double* /*C*/, double* /*Q*/,
                                      bool /*compute_Q*/) {
        throw std::runtime_error("SolveFwd_RtnPressure: not provided by this solver");
    }
    void SolveFwd_RtnPressure(int ilevel, Vector& k_over_k_ref, Vector& P, double& C, double& Q, bool compute_Q) {
        if (k_over_k_ref.Batch() != 1) throw std::invalid_argument("SolveFwd_RtnPressure(double&, double&): one realization expected");
        SolveFwd_RtnPressure(ilevel, k_over_k_ref, P, &C, &Q, compute_Q);
    }
    virtual int GetNumberOfDofs(int ilevel) const = 0;
    virtual int GetGlobalNumberOfDofs(int ilevel) const = 0;
    virtual int GetNNZ(int ilevel) const = 0;
    /// see MLSampler::PreferredBatch
    virtual int PreferredBatch(int /*ilevel*/) const { return 0; }
    virtual PhaseTimes GetPhaseTimes(int /*ilevel*/) const { return PhaseTimes(); }
    virtual void ResetPhaseTimes() {}
};

/// Uncorrelated N(mu, sigma2) variates (counter-based generator on the device).
class NormalDistributionSampler {
  public:
    NormalDistributionSampler(pmc_ctx* ctx, double mu, double sigma2) : ctx_(ctx), mu_(mu), sigma2_(sigma2) {}
    NormalDistributionSampler(NormalDistributionSampler const&) = delete;
    NormalDistributionSampler& operator=(NormalDistributionSampler const&) = delete;
    /// Provides statistically independent random numbers to each part
    void Split(int nparts, int mypart);
    void Seed(uint64_t seed) { seed_ = seed; Split(nparts_, mypart_); }
    /// Fill v (all batch entries) with realizations first_id...
    void operator()(Vector& v, uint64_t first_id = 0, uint32_t stream = 0);

  private:
    pmc_ctx* ctx_;
    double mu_, sigma2_;
    uint64_t seed_ = 0;
    int nparts_ = 1, mypart_ = 0;
};

/// Device SPDE sampler (plain, matching-embedded or L2-projected: decided by the handle's projection).
class PDESampler : public MLSampler {
  public:
    PDESampler(pmc_ctx* ctx, pmc_sampler* handle) : ctx_(ctx), h_(handle) {}
    void Sample(const int level, Vector& xi, uint64_t first_id = 0, int nbatch = 1) override;
    void Eval(const int level, const Vector& xi, Vector& s) override;
    void Eval(const int level, const Vector& xi, Vector& s, Vector& u, bool use_init) override;
    int SampleSize(int level) const override;
    size_t GetNNZ(int level) const override;
    void BuildHierarchy() override;
    pmc_csr GetTrueP(int level) const override;
    int PreferredBatch(int level) const override { return pmc_sampler_batch_width(h_, level); }
    PhaseTimes GetPhaseTimes(int level) const override { return level < (int)times_.size() ? times_[level] : PhaseTimes(); }
    void ResetPhaseTimes() override { times_.clear(); }
    int GetNumIters() const { return last_iters_; }   // the reference returns -1 (PDESampler.hpp:142-145)

  private:
    int level_of_xi(int size) const;
    int level_of_field(int size) const;
    void record(int level, const std::vector<pmc_stats>& st) {
        if ((int)times_.size() <= level) times_.resize(level + 1);
        times_[level].add(st.data(), (int)st.size());
        last_iters_ = st.empty() ? -1 : st[0].iterations;
    }
    pmc_ctx* ctx_;
    pmc_sampler* h_;
    int last_iters_ = -1;
    std::vector<PhaseTimes> times_;
};

class DarcySolver : public PhysicalMLSolver {
  public:
    DarcySolver(pmc_ctx* ctx, pmc_darcy* handle) : ctx_(ctx), h_(handle) {}
    using PhysicalMLSolver::SolveFwd;
    using PhysicalMLSolver::SolveFwd_RtnPressure;
    void SolveFwd(int ilevel, Vector& k_over_k_ref, double* Q, double* C) override;
    void SolveFwd_RtnPressure(int ilevel, Vector& k_over_k_ref, Vector& P, double* C, double* Q, bool compute_Q) override;
    int GetSizeOfStochasticData(int ilevel) const;   // entries of k (src/DarcySolver.hpp:127-130)
    int GetNumberOfDofs(int ilevel) const override;
    int GetGlobalNumberOfDofs(int ilevel) const override;
    int GetNNZ(int ilevel) const override;
    int PreferredBatch(int ilevel) const override { return pmc_darcy_batch_width(h_, ilevel); }
    PhaseTimes GetPhaseTimes(int ilevel) const override { return ilevel < (int)times_.size() ? times_[ilevel] : PhaseTimes(); }
    void ResetPhaseTimes() override { times_.clear(); }

  private:
    void record(int level, const std::vector<pmc_stats>& st) {
        if ((int)times_.size() <= level) times_.resize(level + 1);
        times_[level].add(st.data(), (int)st.size());
    }
    pmc_ctx* ctx_;
    pmc_darcy* h_;
    std::vector<PhaseTimes> times_;
};

/// Plugins backed by C callbacks (host memory).
class CallbackSampler : public MLSampler {
  public:
    CallbackSampler(int nlevels, const pmc_plugin_callbacks& cb);
    void Sample(const int level, Vector& xi, uint64_t first_id = 0, int nbatch = 1) override;
    void Eval(const int level, const Vector& xi, Vector& s) override;
    void Eval(const int level, const Vector& xi, Vector& s, Vector& u, bool use_init) override;
    int SampleSize(int level) const override { return ssize_.at(level); }
    size_t GetNNZ(int) const override { return 0; }

  private:
    pmc_plugin_callbacks cb_;
    std::vector<int> xsize_, ssize_;
};
class CallbackSolver : public PhysicalMLSolver {
  public:
    CallbackSolver(int nlevels, const pmc_plugin_callbacks& cb);
    void SolveFwd(int ilevel, Vector& k, double* Q, double* C) override;
    int GetNumberOfDofs(int l) const override { return ndofs_.at(l); }
    int GetGlobalNumberOfDofs(int l) const override { return ndofs_.at(l); }
    int GetNNZ(int) const override { return 0; }

  private:
    pmc_plugin_callbacks cb_;
    std::vector<int> ndofs_;
};

/// Bayesian inverse problem on top of the forward solver (src/BayesianInverseProblem.hpp): observation operator G,
/// Gaussian likelihood with noise variance `noise`, ratio integrand R = Q * likelihood.  The observation functionals
/// live in the solver handle (pmc_darcy_set_observations).
class BayesianInverseProblem {
  public:
    BayesianInverseProblem(pmc_darcy* solver, double noise, std::vector<double> G_obs)
        : solver_(solver), noise_(noise), G_obs_(std::move(G_obs)) {}
    /// G (k.Batch() x size_obs_data), C and optionally Q
    void ComputeG(int ilevel, Vector& k_over_k_ref, std::vector<double>& G, double* C, double* Q, bool compute_Q);
    void ComputeLikelihood(int ilevel, Vector& k_over_k_ref, double* likelihood, double* C);
    void ComputeLikelihoodAndQ(int ilevel, Vector& k_over_k_ref, double* likelihood, double* C, double* Q);
    void ComputeR(int ilevel, Vector& k_over_k_ref, double* R, double* C);
    int SizeOfObservationalData() const { return (int)G_obs_.size(); }

  private:
    pmc_darcy* solver_;
    double noise_;
    std::vector<double> G_obs_;
};

double expWRegression(const std::vector<double>& y, const std::vector<double>& x, int skip_n_last);

/// Multi-level Monte Carlo manager: the reference's serial loop, sharded over a sample farm.
class MLMC_Manager {
  public:
    enum { Y2 = 0, Y = 1, ABSY = 2, Q2 = 3, Q = 4, ABSQ = 5, C = 6, Y3 = 7, Y4 = 8, NVAR = 9 };

    MLMC_Manager(pmc_ctx* ctx, int memspace, int nlevels, PhysicalMLSolver& pSolver, MLSampler& sampler,
                 const pmc_mlmc_params& params);
    void SetFarm(int nranks, int rank, std::function<void(double*, int)> reduce);
    /// Additional plugin pair (own context / HIP stream) working on this rank's realizations concurrently:
    /// the launch-latency-bound kernels of the small levels of one lane overlap the other lanes' work.
    void AddLane(pmc_ctx* ctx, PhysicalMLSolver& solver, MLSampler& sampler);
    /// the farm's collective: seconds this rank spent in the SUM all-reduce of the accumulators and the number of
    /// reductions (one per InitRun round, src/MLMC_Manager.cpp:178 is where the serial reference would need it)
    void FarmTimes(double* allreduce_seconds, int64_t* reductions) const {
        if (allreduce_seconds) *allreduce_seconds = reduce_seconds_;
        if (reductions) *reductions = reduce_count_;
    }
    /// Run ML simulation by sampling v_init_nsamples then the missing samples until the estimator variance target is met
    void Run();
    /// Run ML simulation using level_nsamples_init[i] samples on level i
    void InitRun(std::vector<int>& level_nsamples_init);
    void Reset();
    /// Resume: rebuild the sums table and sample counters from a per-sample log of an earlier run
    int64_t ReplayLog(const std::string& path);
    void ShowMe(std::ostream& os) const;
    /// The reference's TimeManager::Print (examples/MLMC.cpp:275) for the per-realization path: "Sampler: Mult", "Darcy: Build
    /// Solver" and "Darcy: Mult" per level, device milliseconds summed over all lanes of this rank, with realization counts
    void PrintTimers(std::ostream& os) const;
    void PhaseTimesOfLevel(int level, double* sampler_mult_ms, double* darcy_setup_ms, double* darcy_mult_ms,
                           int64_t* sampler_realizations, int64_t* darcy_realizations) const;

    bool wallTime;   // public switch, src/MLMC_Manager.hpp:61

    // results (read by pmc_mlmc_result_get)
    int nlevels;
    double eps2, ratio;
    double ml_estimator_variance, expected_discretization_error2, actualMSE;
    double alpha = 0, alphaABS = 0, beta = 0, gamma = 0;
    std::vector<double> sums, eY, eABSY, eQ, eABSQ, eC, varY, varQ, consistency, kurtosis, M, VC, cost, level_seconds;
    std::vector<int64_t> level_nsamples, level_nsamples_missing;

  private:
    void computeNSamplesMSE();
    void run_level(int ilevel, int nsamples);
    // several lanes: all levels of one InitRun round go through one task queue (finest level first), so the
    // launch-latency-bound batches of the coarse levels overlap the bandwidth-bound batches of the fine ones
    void run_round_overlapped(const std::vector<int>& level_nsamples_init);
    // realizations per plugin call on `ilevel`: at most `batch`, at most what the plugins prefer for that level
    // (PreferredBatch: 16 on large levels ... 256 on the smallest), and small enough that every RANK of a farm gets a share.
    // Independent of the number of lanes: on a given farm the blocks of realization ids - and with them every realization's
    // bits - do not depend on how many lanes share a GPU.  Identical on all ranks.
    int level_batch(int ilevel, int nsamples) const;
    double& S(int l, int v) { return sums[(size_t)l * NVAR + v]; }

    pmc_ctx* ctx_;
    int memspace_;
    PhysicalMLSolver& pSolver;
    MLSampler& sampler;
    int auto_eps2;
    std::vector<int> v_init_nsamples;
    int batch_, max_rounds_;
    int nranks_ = 1, rank_ = 0;
    std::function<void(double*, int)> reduce_;
    double reduce_seconds_ = 0.0;          // wall time this rank spent inside the farm's all-reduce (incl. waiting for the
    int64_t reduce_count_ = 0;             // slowest rank), and how many there were: one per InitRun round
    std::vector<double> pending_;          // this round's local contributions (sums + counts + seconds)
    struct Lane {
        MLSampler* sampler;
        PhysicalMLSolver* solver;
        Vector xi, sparam, init_s;
        Lane(pmc_ctx* c, int ms, MLSampler* s, PhysicalMLSolver* p)
            : sampler(s), solver(p), xi(c, ms), sparam(c, ms), init_s(c, ms) {}
    };
    std::vector<std::unique_ptr<Lane>> lanes_;
    std::ofstream logger;
    std::string log_path_;
    bool append_log_ = false;
};

/// What the ratio managers call (the reference's BayesianInverseProblem seen from ML_BayesRatio_Manager):
/// prior draws plus likelihood and ratio integrand.
class BayesRatioProblem {
  public:
    virtual ~BayesRatioProblem() = default;
    virtual void SamplePrior(int level, Vector& xi, uint64_t first_id, int nbatch) = 0;
    virtual void EvalPrior(int level, const Vector& xi, Vector& s) = 0;
    /// likelihood[b] and R[b] = Q[b] * likelihood[b]; C[b] = cost (dofs)
    virtual void ComputeLikelihoodAndR(int level, Vector& s, double* likelihood, double* R, double* C) = 0;
    virtual int GetGlobalNumberOfDofs(int level) const = 0;
    /// realizations of `level` one launch of the device plugins carries (0: no preference), see MLSampler::PreferredBatch
    virtual int PreferredBatch(int /*level*/) const { return 0; }
};

/// Multilevel (nlevels > 1) / single-level (nlevels == 1) ratio estimator, src/ML_BayesRatio_Manager.hpp.
/// SetSplitting(true) turns it into ML_BayesRatio_Splitting_Manager / SL_BayesRatio_Splitting_Manager
/// (src/ML_BayesRatio_Splitting_Manager.hpp): the same draws, but the estimated quantity is E[R/Z] with the
/// level difference r/z - r_c/z_c ("divide, then subtract"), and the variance / bias / sample allocation follow the
/// Ratio columns of the sums table instead of max(R, Z).
class ML_BayesRatio_Manager {
  public:
    enum { YZ2 = 0, YZ = 1, ABS_YZ = 2, Z2 = 3, Z = 4, ABS_Z = 5, YR2 = 6, YR = 7, ABS_YR = 8, R2 = 9, R = 10,
           ABS_R = 11, YRatio2 = 12, YRatio = 13, ABS_YRatio = 14, Ratio2 = 15, Ratio = 16, ABS_Ratio = 17, C = 18,
           T = 19, NVAR = 20 };
    ML_BayesRatio_Manager(pmc_ctx* ctx, int memspace, int nlevels, BayesRatioProblem& problem,
                          const pmc_mlmc_params& params);
    void SetFarm(int nranks, int rank, std::function<void(double*, int)> reduce);
    void SetSplitting(bool on) { splitting = on; }
    void Run();
    void InitRun(std::vector<int>& level_nsamples_init);
    void Reset();

    bool wallTime;
    bool splitting = false;
    int nlevels;
    double eps2, ratio;
    double alpha = 0, alphaABS = 0, beta = 0;   // rates of the Ratio columns (splitting manager)
    std::vector<double> eRatio, varRatio, eYRatio, varYRatio, eABS_YRatio;
    double ml_estimator_variance, ml_estimator_variance_R, ml_estimator_variance_Z;
    double expected_discretization_error2, expected_discretization_error2_R, expected_discretization_error2_Z, actualMSE;
    double alpha_R = 0, alphaABS_R = 0, beta_R = 0, alpha_Z = 0, alphaABS_Z = 0, beta_Z = 0, gamma = 0;
    std::vector<double> sums, eR, varR, eYR, varYR, eABS_YR, eZ, varZ, eYZ, varYZ, eABS_YZ, eC, M, cost, level_seconds;
    std::vector<int64_t> level_nsamples, level_nsamples_missing;

  private:
    void computeNSamplesMSE();
    void run_level(int ilevel, int nsamples);
    /// plugin-call size of a level: min(batch, what the plugins prefer there, a rank's share) - as MLMC_Manager::level_batch
    int level_batch(int ilevel, int nsamples) const;
    double& S(int l, int v) { return sums[(size_t)l * NVAR + v]; }
    BayesRatioProblem& problem;
    int auto_eps2, init_nsamples_, batch_, max_rounds_;
    int nranks_ = 1, rank_ = 0;
    std::function<void(double*, int)> reduce_;
    std::vector<double> pending_;
    Vector zxi, xi, zparam, sparam;
};

/// Single-level Monte Carlo manager (src/MC_Manager.hpp): the nlevels == 1 case of the above.
class MC_Manager : public MLMC_Manager {
  public:
    MC_Manager(pmc_ctx* ctx, int memspace, PhysicalMLSolver& pSolver, MLSampler& sampler, const pmc_mlmc_params& params)
        : MLMC_Manager(ctx, memspace, 1, pSolver, sampler, params) {}
};

}  // namespace parelagmc
