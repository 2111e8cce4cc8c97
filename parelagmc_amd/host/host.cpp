// libpmc_host.so: plugin wrappers + Monte Carlo managers (see parelagmc.hpp / include/pmc_host.h).
#include "parelagmc.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <iomanip>
#include <iostream>
#include <limits>
#include <mutex>
#include <thread>
#include <sstream>

namespace parelagmc {

namespace {
void check(int rc, const char* what) {
    if (rc != PMC_OK) throw std::runtime_error(std::string(what) + ": " + pmc_last_error());
}
double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
}  // namespace

// ---- Vector -----------------------------------------------------------------------------------
Vector::~Vector() {
    if (!data_) return;
    if (memspace_ == PMC_MEM_DEVICE) pmc_free(ctx_, data_);
    else delete[] data_;
}
void Vector::SetSize(int size, int nbatch) {
    if (size < 0 || nbatch < 1) throw std::invalid_argument("Vector::SetSize: bad size");
    const size_t need = (size_t)size * nbatch;
    if (need > cap_) {
        if (data_) {
            if (memspace_ == PMC_MEM_DEVICE) pmc_free(ctx_, data_);
            else delete[] data_;
            data_ = nullptr;
        }
        if (memspace_ == PMC_MEM_DEVICE) {
            void* p = nullptr;
            check(pmc_malloc(ctx_, need * sizeof(double), &p), "Vector::SetSize");
            data_ = static_cast<double*>(p);
        } else {
            data_ = new double[need];
        }
        cap_ = need;
    }
    size_ = size;
    nbatch_ = nbatch;
}

void Vector::Swap(Vector& o) {
    if (ctx_ != o.ctx_ || memspace_ != o.memspace_) throw std::invalid_argument("Vector::Swap: incompatible vectors");
    std::swap(data_, o.data_);
    std::swap(cap_, o.cap_);
    std::swap(size_, o.size_);
    std::swap(nbatch_, o.nbatch_);
}

// ---- NormalDistributionSampler ---------------------------------------------------------------
void NormalDistributionSampler::Split(int nparts, int mypart) {
    nparts_ = nparts;
    mypart_ = mypart;
    check(pmc_rng_seed(ctx_, seed_, nparts, mypart), "NormalDistributionSampler::Split");
}
void NormalDistributionSampler::operator()(Vector& v, uint64_t first_id, uint32_t stream) {
    check(pmc_normal_fill(ctx_, mu_, sigma2_, first_id, stream, v.Batch(), v.Size(), v.GetData(), v.MemSpace()),
          "NormalDistributionSampler");
}

// ---- PDESampler -------------------------------------------------------------------------------
int PDESampler::level_of_xi(int size) const {
    const int nl = pmc_sampler_num_levels(h_);
    for (int l = 0; l < nl; ++l)
        if (pmc_sampler_xi_size(h_, l) == size) return l;
    throw std::runtime_error("PDESampler: vector length matches no level");   // level_size.Find() == -1
}
int PDESampler::level_of_field(int size) const { return level_of_xi(size); }
void PDESampler::Sample(const int level, Vector& xi, uint64_t first_id, int nbatch) {
    const int n = pmc_sampler_xi_size(h_, level);
    if (n < 0) throw std::out_of_range("PDESampler::Sample: level");
    xi.SetSize(n, nbatch);
    check(pmc_sampler_sample(h_, level, first_id, nbatch, xi.GetData(), xi.MemSpace()), "PDESampler::Sample");
}
void PDESampler::Eval(const int level, const Vector& xi, Vector& s) {
    const int xi_level = level_of_xi(xi.Size());
    if (xi_level > level) throw std::runtime_error("PDESampler::Eval: xi_level <= level violated");
    s.SetSize(SampleSize(level), xi.Batch());
    std::vector<pmc_stats> st(xi.Batch());
    check(pmc_sampler_eval(h_, level, xi_level, xi.Batch(), xi.GetData(), s.GetData(), nullptr, -1, 0, nullptr,
                           xi.MemSpace(), st.data()),
          "PDESampler::Eval");
    record(level, st);
}
void PDESampler::Eval(const int level, const Vector& xi, Vector& s, Vector& u, bool use_init) {
    const int xi_level = level_of_xi(xi.Size());
    if (xi_level > level) throw std::runtime_error("PDESampler::Eval: xi_level <= level violated");
    int init_level = -1;
    if (use_init) {
        if (u.Batch() != xi.Batch()) throw std::runtime_error("PDESampler::Eval: init batch mismatch");
        init_level = level_of_field(u.Size());
    }
    s.SetSize(SampleSize(level), xi.Batch());
    const int n_field = pmc_sampler_xi_size(h_, level);
    std::vector<pmc_stats> st(xi.Batch());
    if (use_init && (size_t)n_field * xi.Batch() > (size_t)u.Size() * u.Batch()) {
        // u must grow (coarse -> fine): evaluate into a fresh buffer, then swap it in
        Vector tmp(ctx_, u.MemSpace());
        tmp.SetSize(n_field, xi.Batch());
        check(pmc_sampler_eval(h_, level, xi_level, xi.Batch(), xi.GetData(), s.GetData(), u.GetData(), init_level, 1,
                               tmp.GetData(), xi.MemSpace(), st.data()),
              "PDESampler::Eval");
        u.Swap(tmp);
    } else {
        const double* init = use_init ? u.GetData() : nullptr;
        u.SetSize(n_field, xi.Batch());   // capacity suffices: pointer unchanged
        check(pmc_sampler_eval(h_, level, xi_level, xi.Batch(), xi.GetData(), s.GetData(), init, init_level,
                               use_init ? 1 : 0, u.GetData(), xi.MemSpace(), st.data()),
              "PDESampler::Eval");
    }
    record(level, st);
}
void PDESampler::BuildHierarchy() {
    if (!h_ || pmc_sampler_num_levels(h_) < 1) throw std::runtime_error("PDESampler::BuildHierarchy: no device handle");
}
pmc_csr PDESampler::GetTrueP(int level) const {
    pmc_csr P{};
    check(pmc_sampler_true_p(h_, level, &P), "PDESampler::GetTrueP");
    return P;
}
int PDESampler::SampleSize(int level) const { return pmc_sampler_sample_size(h_, level); }
size_t PDESampler::GetNNZ(int level) const { return (size_t)pmc_sampler_nnz(h_, level); }

// ---- DarcySolver ------------------------------------------------------------------------------
void DarcySolver::SolveFwd(int ilevel, Vector& k, double* Q, double* C) {
    std::vector<pmc_stats> st(k.Batch());
    check(pmc_darcy_solve_fwd(h_, ilevel, k.Batch(), k.GetData(), Q, C, nullptr, k.MemSpace(), st.data()),
          "DarcySolver::SolveFwd");
    record(ilevel, st);
}
void DarcySolver::SolveFwd_RtnPressure(int ilevel, Vector& k, Vector& P, double* C, double* Q, bool compute_Q) {
    const int np = pmc_darcy_num_pressure_dofs(h_, ilevel);
    if (np < 0) throw std::out_of_range("DarcySolver::SolveFwd_RtnPressure: level");
    if (P.MemSpace() != k.MemSpace()) throw std::invalid_argument("SolveFwd_RtnPressure: k and P must share a memory space");
    P.SetSize(np, k.Batch());
    std::vector<pmc_stats> st(k.Batch());
    check(pmc_darcy_solve_fwd_pressure(h_, ilevel, k.Batch(), k.GetData(), P.GetData(), C, Q, compute_Q ? 1 : 0, k.MemSpace(),
                                       st.data()),
          "DarcySolver::SolveFwd_RtnPressure");
    record(ilevel, st);
}
int DarcySolver::GetSizeOfStochasticData(int l) const { return pmc_darcy_num_pressure_dofs(h_, l); }
int DarcySolver::GetNumberOfDofs(int l) const { return pmc_darcy_num_dofs(h_, l); }
int DarcySolver::GetGlobalNumberOfDofs(int l) const { return pmc_darcy_num_dofs(h_, l); }
int DarcySolver::GetNNZ(int l) const { return (int)pmc_darcy_nnz(h_, l); }

// ---- BayesianInverseProblem --------------------------------------------------------------------------
void BayesianInverseProblem::ComputeG(int ilevel, Vector& k, std::vector<double>& G, double* C, double* Q,
                                      bool compute_Q) {
    const int nobs = pmc_darcy_num_observations(solver_, ilevel);
    if (nobs != (int)G_obs_.size()) throw std::runtime_error("BayesianInverseProblem: observation count mismatch");
    G.resize((size_t)k.Batch() * nobs);
    std::vector<double> q(k.Batch());
    check(pmc_darcy_compute_G(solver_, ilevel, k.Batch(), k.GetData(), G.data(), C, q.data(), k.MemSpace(), nullptr),
          "BayesianInverseProblem::ComputeG");
    if (compute_Q && Q)
        for (int b = 0; b < k.Batch(); ++b) Q[b] = q[b];
}
void BayesianInverseProblem::ComputeLikelihoodAndQ(int ilevel, Vector& k, double* likelihood, double* C, double* Q) {
    std::vector<double> G;
    ComputeG(ilevel, k, G, C, Q, Q != nullptr);
    const int nobs = (int)G_obs_.size();
    for (int b = 0; b < k.Batch(); ++b) {
        double s = 0.0;
        for (int i = 0; i < nobs; ++i) {
            const double d = G[(size_t)b * nobs + i] - G_obs_[i];
            s += d * d;
        }
        likelihood[b] = std::exp((-1. / (noise_ * 2)) * s);      // BayesianInverseProblem.cpp:196
    }
}
void BayesianInverseProblem::ComputeLikelihood(int ilevel, Vector& k, double* likelihood, double* C) {
    ComputeLikelihoodAndQ(ilevel, k, likelihood, C, nullptr);
}
void BayesianInverseProblem::ComputeR(int ilevel, Vector& k, double* R, double* C) {
    std::vector<double> q(k.Batch());
    ComputeLikelihoodAndQ(ilevel, k, R, C, q.data());
    for (int b = 0; b < k.Batch(); ++b) R[b] *= q[b];
}

// ---- callback plugins ---------------------------------------------------------------------------
CallbackSampler::CallbackSampler(int nlevels, const pmc_plugin_callbacks& cb) : cb_(cb) {
    if (!cb.sample || !cb.eval || !cb.xi_size || !cb.sample_size) throw std::invalid_argument("sampler callbacks missing");
    xsize_.assign(cb.xi_size, cb.xi_size + nlevels);
    ssize_.assign(cb.sample_size, cb.sample_size + nlevels);
}
void CallbackSampler::Sample(const int level, Vector& xi, uint64_t first_id, int nbatch) {
    xi.SetSize(xsize_.at(level), nbatch);
    if (cb_.sample(cb_.user, level, first_id, nbatch, xi.GetData()) != 0) throw std::runtime_error("sample callback failed");
}
void CallbackSampler::Eval(const int level, const Vector& xi, Vector& s) {
    const int xl = (int)(std::find(xsize_.begin(), xsize_.end(), xi.Size()) - xsize_.begin());
    if (xl >= (int)xsize_.size() || xl > level) throw std::runtime_error("Eval: xi_level <= level violated");
    s.SetSize(ssize_.at(level), xi.Batch());
    if (cb_.eval(cb_.user, level, xl, xi.Batch(), xi.GetData(), s.GetData(), nullptr, -1, 0, nullptr) != 0)
        throw std::runtime_error("eval callback failed");
}
void CallbackSampler::Eval(const int level, const Vector& xi, Vector& s, Vector& u, bool use_init) {
    const int xl = (int)(std::find(xsize_.begin(), xsize_.end(), xi.Size()) - xsize_.begin());
    if (xl >= (int)xsize_.size() || xl > level) throw std::runtime_error("Eval: xi_level <= level violated");
    int il = -1;
    std::vector<double> init;
    if (use_init) {
        il = (int)(std::find(xsize_.begin(), xsize_.end(), u.Size()) - xsize_.begin());
        if (il >= (int)xsize_.size()) throw std::runtime_error("Eval: init size matches no level");
        init.assign(u.GetData(), u.GetData() + (size_t)u.Size() * u.Batch());
    }
    s.SetSize(ssize_.at(level), xi.Batch());
    u.SetSize(xsize_.at(level), xi.Batch());
    if (cb_.eval(cb_.user, level, xl, xi.Batch(), xi.GetData(), s.GetData(), use_init ? init.data() : nullptr, il,
                 use_init ? 1 : 0, u.GetData()) != 0)
        throw std::runtime_error("eval callback failed");
}
CallbackSolver::CallbackSolver(int nlevels, const pmc_plugin_callbacks& cb) : cb_(cb) {
    if (!cb.solve_fwd || !cb.ndofs) throw std::invalid_argument("solver callbacks missing");
    ndofs_.assign(cb.ndofs, cb.ndofs + nlevels);
}
void CallbackSolver::SolveFwd(int ilevel, Vector& k, double* Q, double* C) {
    if (cb_.solve_fwd(cb_.user, ilevel, k.Batch(), k.GetData(), Q, C) != 0) throw std::runtime_error("solve callback failed");
}

// ---- statistics -------------------------------------------------------------------------------
// y = x^out : weighted least-squares slope of log-ratios (src/Utilities.cpp:257-283)
double expWRegression(const std::vector<double>& y, const std::vector<double>& x, int skip_n_last) {
    const int n = (int)y.size() - 1 - skip_n_last;
    if (n < 1) return 0.0;
    double num = 0.0, den = 0.0;
    for (int i = 0; i < n; ++i) {
        const double logdy = std::log(std::fabs(y[i] / y[i + 1]));
        const double logdx = std::log(x[i] / x[i + 1]);
        const double w = std::pow(.5, i);
        num += w * logdy * logdx;
        den += w * logdx * logdx;
    }
    return num / den;
}

MLMC_Manager::MLMC_Manager(pmc_ctx* ctx, int memspace, int nlevels_, PhysicalMLSolver& pSolver_, MLSampler& sampler_,
                           const pmc_mlmc_params& p)
    : wallTime(p.wall_time != 0),
      nlevels(nlevels_),
      eps2(p.eps2),
      ratio(p.ratio),
      ml_estimator_variance(std::numeric_limits<double>::infinity()),
      expected_discretization_error2(std::numeric_limits<double>::infinity()),
      actualMSE(std::numeric_limits<double>::infinity()),
      ctx_(ctx),
      memspace_(memspace),
      pSolver(pSolver_),
      sampler(sampler_),
      auto_eps2(p.eps2 < 0 ? 1 : 0),
      batch_(p.batch),
      max_rounds_(p.max_rounds) {
    lanes_.emplace_back(new Lane(ctx, memspace, &sampler_, &pSolver_));
    if (nlevels < 1) throw std::invalid_argument("MLMC_Manager: nlevels < 1");
    if (batch_ < 1 || batch_ > 256) throw std::invalid_argument("MLMC_Manager: batch must be in 1..256");
    if (!(ratio > 0.0 && ratio < 1.0)) throw std::invalid_argument("MLMC_Manager: ratio must be in (0,1)");
    if (p.array_nsamples) v_init_nsamples.assign(p.array_nsamples, p.array_nsamples + nlevels);
    else v_init_nsamples.assign(nlevels, p.init_nsamples);
    M.resize(nlevels);
    for (int i = 0; i < nlevels; ++i) M[i] = pSolver.GetGlobalNumberOfDofs(i);
    if (p.log_file && p.log_file[0]) {
        log_path_ = p.log_file;   // opened by the first InitRun, once the rank (SetFarm) is known
    }
    Reset();
}

// In a farm every rank logs the realizations of its own shard; the reference logs on pid 0 only
// (src/MLMC_Manager.cpp:106-108), which sees every sample there.  Rank r > 0 writes "<log>.rank<r>" so that ranks
// sharing a file system never clobber each other's shard; rank 0 keeps the plain name.
static std::string shard_name(const std::string& base, int rank) {
    return rank == 0 ? base : base + ".rank" + std::to_string(rank);
}

void MLMC_Manager::SetFarm(int nranks, int rank, std::function<void(double*, int)> reduce) {
    if (nranks < 1 || rank < 0 || rank >= nranks) throw std::invalid_argument("SetFarm: bad rank");
    if (nranks > 1 && !reduce) throw std::invalid_argument("SetFarm: reduction missing");
    nranks_ = nranks;
    rank_ = rank;
    reduce_ = std::move(reduce);
    if (logger.is_open()) logger.close();   // re-opened under this rank's shard name by the next InitRun
}

void MLMC_Manager::Reset() {
    auto z = [&](std::vector<double>& v) { v.assign(nlevels, 0.0); };
    sums.assign((size_t)nlevels * NVAR, 0.0);
    z(eY); z(eABSY); z(eQ); z(eABSQ); z(eC); z(varY); z(varQ); z(consistency); z(kurtosis); z(VC); z(cost); z(level_seconds);
    level_nsamples.assign(nlevels, 0);
    level_nsamples_missing.assign(nlevels, 0);
    ml_estimator_variance = expected_discretization_error2 = actualMSE = std::numeric_limits<double>::infinity();
}

// One level of InitRun (src/MLMC_Manager.cpp:113-136 coarsest, :144-173 level pairs), for this
// rank's share of the `nsamples` new realizations, `batch_` at a time.
void MLMC_Manager::AddLane(pmc_ctx* ctx, PhysicalMLSolver& solver, MLSampler& smp) {
    lanes_.emplace_back(new Lane(ctx, memspace_, &smp, &solver));
}

int MLMC_Manager::level_batch(int ilevel, int nsamples) const {
    int b = batch_;
    for (int pref : {sampler.PreferredBatch(ilevel), pSolver.PreferredBatch(ilevel)})
        if (pref > 0) b = std::min(b, pref);
    const int share = (nsamples + nranks_ - 1) / nranks_;     // what a rank gets if the level is dealt evenly
    return std::max(1, std::min(b, share));
}

void MLMC_Manager::run_level(int ilevel, int nsamples) {
    const uint64_t base = (uint64_t)level_nsamples[ilevel];
    const int lb = level_batch(ilevel, nsamples);              // this level's plugin-call size
    const int nblocks = (nsamples + lb - 1) / lb;
    const int nlanes = (int)lanes_.size();
    std::vector<std::vector<double>> lane_sums(nlanes, std::vector<double>(NVAR, 0.0));
    std::vector<std::string> lane_err(nlanes);
    std::mutex log_mutex;
    const double t0 = now_s();
    // blocks of `lb` consecutive realization ids: dealt round-robin to ranks, then to this rank's lanes
    auto work = [&](int lane) {
        try {
            Lane& L = *lanes_[lane];
            double* psum = lane_sums[lane].data();
            std::vector<double> q(lb), c(lb), qc(lb), cc(lb);
            for (int blk = rank_ + lane * nranks_; blk < nblocks; blk += nranks_ * nlanes) {
                const int first = blk * lb;
                const int m = std::min(lb, nsamples - first);
                L.sampler->Sample(ilevel, L.xi, base + (uint64_t)first, m);
                if (ilevel == nlevels - 1) {
                    L.sampler->Eval(ilevel, L.xi, L.sparam);
                    L.solver->SolveFwd(ilevel, L.sparam, q.data(), c.data());
                    for (int b = 0; b < m; ++b) qc[b] = 0.0;
                } else {
                    L.sampler->Eval(ilevel + 1, L.xi, L.sparam, L.init_s, false);
                    L.solver->SolveFwd(ilevel + 1, L.sparam, qc.data(), cc.data());
                    L.sampler->Eval(ilevel, L.xi, L.sparam, L.init_s, true);
                    L.solver->SolveFwd(ilevel, L.sparam, q.data(), c.data());
                }
                for (int b = 0; b < m; ++b) {
                    const double y = (ilevel == nlevels - 1) ? q[b] : q[b] - qc[b];
                    const double cost_b = (ilevel == nlevels - 1) ? c[b] : c[b] + cc[b];
                    psum[Y3] += y * y * y;
                    psum[Y4] += y * y * y * y;
                    psum[Y2] += y * y;
                    psum[Y] += y;
                    psum[ABSY] += std::fabs(y);
                    psum[Q2] += q[b] * q[b];
                    psum[Q] += q[b];
                    psum[ABSQ] += std::fabs(q[b]);
                    psum[C] += cost_b;
                    if (logger.is_open()) {
                        std::lock_guard<std::mutex> lk(log_mutex);
                        logger << std::setw(14) << ilevel << ' ' << std::setw(24) << y << ' ' << std::setw(24) << q[b] << ' '
                               << std::setw(24) << qc[b] << ' ' << std::setw(14) << cost_b << "\n";
                    }
                }
            }
        } catch (const std::exception& e) {
            lane_err[lane] = e.what();
        }
    };
    if (nlanes == 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        for (int l = 0; l < nlanes; ++l) th.emplace_back(work, l);
        for (auto& t : th) t.join();
    }
    for (const auto& e : lane_err)
        if (!e.empty()) throw std::runtime_error(e);
    double* psum = pending_.data() + (size_t)ilevel * NVAR;
    for (int l = 0; l < nlanes; ++l)            // fixed order: deterministic for a fixed lane count
        for (int v = 0; v < NVAR; ++v) psum[v] += lane_sums[l][v];
    pending_[(size_t)nlevels * NVAR + ilevel] += now_s() - t0;
}

void MLMC_Manager::run_round_overlapped(const std::vector<int>& ns_init) {
    struct Task { int level, first, m; };
    std::vector<Task> tasks;
    std::vector<std::vector<double>> yv(nlevels), qv(nlevels), qcv(nlevels), cv(nlevels);
    for (int ilevel = 0; ilevel < nlevels; ++ilevel) {          // finest (longest) tasks first
        const int ns = ns_init[ilevel];
        if (ns < 0) throw std::invalid_argument("InitRun: negative sample count");
        yv[ilevel].assign(ns, 0.0); qv[ilevel].assign(ns, 0.0); qcv[ilevel].assign(ns, 0.0); cv[ilevel].assign(ns, 0.0);
        const int lb = level_batch(ilevel, ns);
        const int nblocks = (ns + lb - 1) / lb;
        for (int blk = rank_; blk < nblocks; blk += nranks_)
            tasks.push_back({ilevel, blk * lb, std::min(lb, ns - blk * lb)});
    }
    const int nlanes = (int)lanes_.size();
    std::vector<std::vector<double>> lane_seconds(nlanes, std::vector<double>(nlevels, 0.0));
    std::vector<std::string> lane_err(nlanes);
    std::atomic<size_t> next{0};
    auto work = [&](int lane) {
        try {
            Lane& L = *lanes_[lane];
            std::vector<double> q(batch_), c(batch_), qc(batch_), cc(batch_);
            for (;;) {
                const size_t ti = next.fetch_add(1);
                if (ti >= tasks.size()) break;
                const Task t = tasks[ti];
                const int ilevel = t.level;
                const double t0 = now_s();
                L.sampler->Sample(ilevel, L.xi, (uint64_t)level_nsamples[ilevel] + (uint64_t)t.first, t.m);
                if (ilevel == nlevels - 1) {
                    L.sampler->Eval(ilevel, L.xi, L.sparam);
                    L.solver->SolveFwd(ilevel, L.sparam, q.data(), c.data());
                    for (int b = 0; b < t.m; ++b) { qc[b] = 0.0; cc[b] = 0.0; }
                } else {
                    L.sampler->Eval(ilevel + 1, L.xi, L.sparam, L.init_s, false);
                    L.solver->SolveFwd(ilevel + 1, L.sparam, qc.data(), cc.data());
                    L.sampler->Eval(ilevel, L.xi, L.sparam, L.init_s, true);
                    L.solver->SolveFwd(ilevel, L.sparam, q.data(), c.data());
                }
                for (int b = 0; b < t.m; ++b) {
                    yv[ilevel][t.first + b] = (ilevel == nlevels - 1) ? q[b] : q[b] - qc[b];
                    qv[ilevel][t.first + b] = q[b];
                    qcv[ilevel][t.first + b] = qc[b];
                    cv[ilevel][t.first + b] = c[b] + cc[b];
                }
                lane_seconds[lane][ilevel] += now_s() - t0;
            }
        } catch (const std::exception& e) {
            lane_err[lane] = e.what();
            next.store(tasks.size());
        }
    };
    std::vector<std::thread> th;
    for (int l = 0; l < nlanes; ++l) th.emplace_back(work, l);
    for (auto& t : th) t.join();
    for (const auto& e : lane_err)
        if (!e.empty()) throw std::runtime_error(e);
    // accumulate in realization order (the order a single lane would produce), coarsest level first as in :110-173
    for (int ilevel = nlevels - 1; ilevel >= 0; --ilevel) {
        double* psum = pending_.data() + (size_t)ilevel * NVAR;
        const int ns = ns_init[ilevel];
        const int lb = level_batch(ilevel, ns);
        const int nblocks = (ns + lb - 1) / lb;
        for (int blk = rank_; blk < nblocks; blk += nranks_)
            for (int i = blk * lb; i < std::min(ns, (blk + 1) * lb); ++i) {
                const double y = yv[ilevel][i], q = qv[ilevel][i];
                psum[Y3] += y * y * y;
                psum[Y4] += y * y * y * y;
                psum[Y2] += y * y;
                psum[Y] += y;
                psum[ABSY] += std::fabs(y);
                psum[Q2] += q * q;
                psum[Q] += q;
                psum[ABSQ] += std::fabs(q);
                psum[C] += cv[ilevel][i];
                if (logger.is_open())
                    logger << std::setw(14) << ilevel << ' ' << std::setw(24) << y << ' ' << std::setw(24) << q << ' '
                           << std::setw(24) << qcv[ilevel][i] << ' ' << std::setw(14) << cv[ilevel][i] << "\n";
            }
        // cost model: lane-seconds spent on the level / lanes = the wall time the level would take on its own
        double sec = 0.0;
        for (int l = 0; l < nlanes; ++l) sec += lane_seconds[l][ilevel];
        pending_[(size_t)nlevels * NVAR + ilevel] += sec / nlanes;
    }
}

void MLMC_Manager::PhaseTimesOfLevel(int level, double* sampler_mult_ms, double* darcy_setup_ms, double* darcy_mult_ms,
                                     int64_t* sampler_realizations, int64_t* darcy_realizations) const {
    if (level < 0 || level >= nlevels) throw std::out_of_range("PhaseTimesOfLevel: level");
    PhaseTimes s, d;
    for (const auto& L : lanes_) {
        const PhaseTimes a = L->sampler->GetPhaseTimes(level), b = L->solver->GetPhaseTimes(level);
        s.mult_ms += a.mult_ms; s.realizations += a.realizations;
        d.mult_ms += b.mult_ms; d.setup_ms += b.setup_ms; d.realizations += b.realizations;
    }
    if (sampler_mult_ms) *sampler_mult_ms = s.mult_ms;
    if (darcy_setup_ms) *darcy_setup_ms = d.setup_ms;
    if (darcy_mult_ms) *darcy_mult_ms = d.mult_ms;
    if (sampler_realizations) *sampler_realizations = s.realizations;
    if (darcy_realizations) *darcy_realizations = d.realizations;
}

void MLMC_Manager::PrintTimers(std::ostream& os) const {
    // layout of parelag::TimeManager::Print: one line per named timer
    const std::streamsize old_prec = os.precision(6);
    os << std::string(79, '=') << '\n'
       << std::setw(44) << std::left << "Timer (device time, all lanes of this rank)" << std::setw(14) << std::right
       << "realizations" << std::setw(16) << "total [ms]" << '\n'
       << std::string(79, '-') << '\n';
    auto line = [&](const std::string& name, int64_t n, double ms) {
        os << std::setw(44) << std::left << name << std::setw(14) << std::right << n << std::setw(16) << ms << '\n';
    };
    for (int l = 0; l < nlevels; ++l) {
        PhaseTimes s, d;
        for (const auto& L : lanes_) {
            const PhaseTimes a = L->sampler->GetPhaseTimes(l), b = L->solver->GetPhaseTimes(l);
            s.mult_ms += a.mult_ms; s.setup_ms += a.setup_ms; s.realizations += a.realizations;
            d.mult_ms += b.mult_ms; d.setup_ms += b.setup_ms; d.realizations += b.realizations;
        }
        const std::string lv = " -- Level " + std::to_string(l);
        line("Sampler: Mult" + lv, s.realizations, s.mult_ms);
        line("Darcy: Build Solver" + lv, d.realizations, d.setup_ms);
        line("Darcy: Mult" + lv, d.realizations, d.mult_ms);
    }
    os << std::string(79, '=') << std::endl;
    os.precision(old_prec);
}

void MLMC_Manager::ShowMe(std::ostream& os) const {
    const int total_width = 79, name_width = 40;
    const std::streamsize old_prec = os.precision(8);
    auto scalar = [&](const char* name, double v, const char* end = "\n") {
        os << std::setw(name_width + 2) << std::left << name << std::setw(18) << std::left << v << end;
    };
    auto row = [&](const char* name, const std::vector<double>& v) {
        os << std::setw(name_width + 2) << std::left << name << std::setw(2) << std::left;
        for (size_t i = 0; i < v.size(); ++i) os << (i ? " " : "") << v[i];
        os << '\n';
    };
    double est = 0.0;
    for (double y : eY) est += y;
    std::vector<double> ns(level_nsamples.begin(), level_nsamples.end()), snnz(nlevels), pnnz(nlevels);
    for (int i = 0; i < nlevels; ++i) {
        snnz[i] = (double)sampler.GetNNZ(i);
        pnnz[i] = (double)pSolver.GetNNZ(i);
    }
    os << std::string(total_width, '=') << '\n' << "MLMC Manager Errors: " << '\n' << std::string(total_width, '-') << '\n';
    scalar("Estimate", est);
    scalar("Target MSE", eps2);
    scalar("Actual MSE", actualMSE);
    scalar("ML Estimator Variance", ml_estimator_variance);
    scalar("Estimator Bias", expected_discretization_error2);
    scalar("Alpha", alpha);
    scalar("AlphaAbs", alphaABS);
    scalar("Beta", beta);
    scalar("Gamma", gamma, "\n\n");
    row("DOFS in Forward Problem", M);
    row("C_l ", eC);
    os << '\n';
    row("NumSamples ", ns);
    os << '\n';
    row("E[Y_l] ", eY);
    row("E[|Y_l|] ", eABSY);
    row("Var[Y_l] ", varY);
    row("E[Q_l] ", eQ);
    row("E[|Q_l|] ", eABSQ);
    row("Var[Q_l] ", varQ);
    row("V[Y_l]*C_l ", VC);
    row("Consistency ", consistency);
    row("Kurtosis", kurtosis);
    row("NNZ-Sampler", snnz);
    row("NNZ-ForwardSolve", pnnz);
    os << std::string(total_width, '=') << '\n';
    os.precision(old_prec);
}

void MLMC_Manager::InitRun(std::vector<int>& level_nsamples_init) {
    if ((int)level_nsamples_init.size() != nlevels) throw std::invalid_argument("InitRun: wrong number of levels");
    if (!log_path_.empty() && !logger.is_open()) {
        // a log that has just been replayed into this manager is continued, not truncated
        logger.open(shard_name(log_path_, rank_), append_log_ ? std::ios::app : std::ios::trunc);
        logger.precision(17);    // the per-sample log doubles as a checkpoint (ReplayLog), so it must round-trip
    }
    if (logger.is_open() && *std::max_element(level_nsamples.begin(), level_nsamples.end()) == 0)
        logger << "%" << std::setw(13) << "level " << std::setw(14) << "Y(xi) " << std::setw(14) << "Q(xi)"
               << std::setw(14) << "Q_c(xi)" << std::setw(14) << "c \n";
    pending_.assign((size_t)nlevels * (NVAR + 1), 0.0);
    if (lanes_.size() > 1) {
        run_round_overlapped(level_nsamples_init);
    } else {
        // coarsest level first, then the level pairs from coarse to fine (:110-173)
        for (int ilevel = nlevels - 1; ilevel >= 0; --ilevel) {
            const int ns = level_nsamples_init[ilevel];
            if (ns < 0) throw std::invalid_argument("InitRun: negative sample count");
            if (ns > 0) run_level(ilevel, ns);
        }
    }
    if (nranks_ > 1) {
        const double t0 = now_s();
        reduce_(pending_.data(), (int)pending_.size());
        reduce_seconds_ += now_s() - t0;
        ++reduce_count_;
    }
    for (size_t i = 0; i < (size_t)nlevels * NVAR; ++i) sums[i] += pending_[i];
    for (int l = 0; l < nlevels; ++l) {
        level_seconds[l] += pending_[(size_t)nlevels * NVAR + l];
        level_nsamples[l] += level_nsamples_init[l];
    }
    if (logger.is_open()) logger << std::flush;
    computeNSamplesMSE();
}

// Rebuild sums / counters from a per-sample log written by an earlier run (same columns as the reference's
// MLMC.dat, src/MLMC_Manager.cpp:106-108,133-135,170-172: level, Y, Q, Q_c, cost).  The reference never reads its log
// back; this is the resume path SURVEY.md 5 asks for.  Returns the number of realizations read.
int64_t MLMC_Manager::ReplayLog(const std::string& path) {
    // a farm's log is sharded by rank (shard_name): every rank replays ALL shards, so the rebuilt sums and counters
    // are the global ones on every rank, exactly as after the all-reduce of a live run
    int64_t nread = 0;
    if (path == log_path_) append_log_ = true;
    for (int r = 0; r < nranks_; ++r) {
    std::ifstream in(shard_name(path, r));
    if (!in) throw std::runtime_error("ReplayLog: cannot open " + shard_name(path, r) +
                                      (nranks_ > 1 ? " (a farm of " + std::to_string(nranks_) + " ranks needs every rank's shard)" : ""));
    std::string line;
    while (std::getline(in, line)) {
        if (line.empty() || line[0] == '%') continue;
        std::istringstream ss(line);
        int lvl;
        double y, q, qc, c;
        if (!(ss >> lvl >> y >> q >> qc >> c)) continue;       // truncated last line of an interrupted run
        if (lvl < 0 || lvl >= nlevels) throw std::runtime_error("ReplayLog: level out of range");
        S(lvl, Y3) += y * y * y;
        S(lvl, Y4) += y * y * y * y;
        S(lvl, Y2) += y * y;
        S(lvl, Y) += y;
        S(lvl, ABSY) += std::fabs(y);
        S(lvl, Q2) += q * q;
        S(lvl, Q) += q;
        S(lvl, ABSQ) += std::fabs(q);
        S(lvl, C) += c;
        level_nsamples[lvl] += 1;
        ++nread;
    }
    }
    bool all = true;
    for (int l = 0; l < nlevels; ++l) all = all && level_nsamples[l] > 1;
    if (all) computeNSamplesMSE();
    return nread;
}

void MLMC_Manager::Run() {
    Reset();
    InitRun(v_init_nsamples);
    std::vector<int> grain(nlevels, 0);
    int rounds = 0;
    while (ml_estimator_variance > ratio * eps2) {
        if (++rounds > max_rounds_) throw std::runtime_error("MLMC_Manager::Run: round limit reached");
        for (int i = 0; i < nlevels; ++i) {
            const int64_t miss = level_nsamples_missing[i];
            grain[i] = (int)std::min<int64_t>(miss, (int64_t)v_init_nsamples[i] + grain[i] + miss / 10);
        }
        InitRun(grain);
    }
}

void MLMC_Manager::computeNSamplesMSE() {
    for (int l = 0; l < nlevels; ++l) {
        const double n = (double)level_nsamples[l];
        eY[l] = S(l, Y) / n;
        eABSY[l] = S(l, ABSY) / n;
        eQ[l] = S(l, Q) / n;
        eABSQ[l] = S(l, ABSQ) / n;
        eC[l] = S(l, C) / n;
        varY[l] = S(l, Y2) / n;
        varQ[l] = S(l, Q2) / n;
        kurtosis[l] = S(l, Y4) / n;
    }
    for (int l = 0; l < nlevels; ++l) {
        const double n = (double)level_nsamples[l];
        kurtosis[l] /= varY[l] * varY[l];
        varY[l] -= eY[l] * eY[l];
        varY[l] *= n / (n - 1.0);
        varQ[l] -= eQ[l] * eQ[l];
        varQ[l] *= n / (n - 1.0);
    }
    for (int l = 0; l < nlevels - 1; ++l)
        consistency[l] = std::abs(eQ[l] - eQ[l + 1] + eY[l]) /
                         (3 * (std::sqrt(varQ[l]) + std::sqrt(varQ[l + 1]) + std::sqrt(varY[l])));
    alpha = expWRegression(eY, M, 1);
    alphaABS = expWRegression(eABSY, M, 1);
    beta = expWRegression(varY, M, 1);
    if (nlevels == 1) {
        expected_discretization_error2 = 0.;
    } else {
        const double m = M[0] / M[1];
        if (nlevels > 3)
            expected_discretization_error2 = std::max(std::pow(m, 2. * alphaABS) * eABSY[1] * eABSY[1], eABSY[0] * eABSY[0]) /
                                             (std::pow(std::pow(m, -2. * alphaABS) - 1., 2));
        else if (nlevels == 3)
            expected_discretization_error2 = (eABSY[0] * eABSY[0]) / (std::pow(std::pow(m, -alphaABS) - 1., 2));
        else
            expected_discretization_error2 = (eABSY[0] * eABSY[0]);
    }
    if (auto_eps2) eps2 = expected_discretization_error2 / (1. - ratio);
    ml_estimator_variance = 0.;
    for (int l = 0; l < nlevels; ++l) ml_estimator_variance += varY[l] / (double)level_nsamples[l];
    actualMSE = expected_discretization_error2 + ml_estimator_variance;
    if (wallTime)
        for (int l = 0; l < nlevels; ++l) cost[l] = level_seconds[l] / (double)level_nsamples[l];
    else
        cost = eC;
    gamma = expWRegression(cost, M, 0);
    double prop = 0.;
    for (int i = 0; i < nlevels; ++i) prop += std::sqrt(varY[i] * cost[i]);
    prop /= ratio * eps2;
    for (int i = 0; i < nlevels; ++i) {
        double missings = prop * std::sqrt(varY[i] / cost[i]);
        missings -= (double)level_nsamples[i];
        const double cm = std::ceil(missings);
        level_nsamples_missing[i] = (cm > 0.0 && std::isfinite(cm)) ? (int64_t)std::min(cm, 2.0e9) : 0;
        VC[i] = varY[i] * cost[i];
    }
}


// ---- ML_BayesRatio_Manager ------------------------------------------------------------------------------
namespace {
// bias model shared by R and Z (ML_BayesRatio_Manager.hpp:649-680, same as MLMC_Manager.cpp:339-355)
double bias2_of(int nlevels, const std::vector<double>& M, const std::vector<double>& eABS, double alphaABS) {
    if (nlevels == 1) return 0.;
    const double m = M[0] / M[1];
    if (nlevels > 3)
        return std::max(std::pow(m, 2. * alphaABS) * eABS[1] * eABS[1], eABS[0] * eABS[0]) /
               (std::pow(std::pow(m, -2. * alphaABS) - 1., 2));
    if (nlevels == 3) return (eABS[0] * eABS[0]) / (std::pow(std::pow(m, -alphaABS) - 1., 2));
    return eABS[0] * eABS[0];
}
}  // namespace

ML_BayesRatio_Manager::ML_BayesRatio_Manager(pmc_ctx* ctx, int memspace, int nlevels_, BayesRatioProblem& problem_,
                                             const pmc_mlmc_params& p)
    : wallTime(p.wall_time != 0),
      nlevels(nlevels_),
      eps2(p.eps2),
      ratio(p.ratio),
      problem(problem_),
      auto_eps2(p.eps2 < 0 ? 1 : 0),
      init_nsamples_(p.init_nsamples),
      batch_(p.batch),
      max_rounds_(p.max_rounds),
      zxi(ctx, memspace),
      xi(ctx, memspace),
      zparam(ctx, memspace),
      sparam(ctx, memspace) {
    if (nlevels < 1) throw std::invalid_argument("ML_BayesRatio_Manager: nlevels < 1");
    if (batch_ < 1 || batch_ > 256) throw std::invalid_argument("ML_BayesRatio_Manager: batch must be in 1..256");
    if (!(ratio > 0.0 && ratio < 1.0)) throw std::invalid_argument("ML_BayesRatio_Manager: ratio must be in (0,1)");
    M.resize(nlevels);
    for (int i = 0; i < nlevels; ++i) M[i] = problem.GetGlobalNumberOfDofs(i);
    Reset();
}

void ML_BayesRatio_Manager::SetFarm(int nranks, int rank, std::function<void(double*, int)> reduce) {
    if (nranks < 1 || rank < 0 || rank >= nranks) throw std::invalid_argument("SetFarm: bad rank");
    if (nranks > 1 && !reduce) throw std::invalid_argument("SetFarm: reduction missing");
    nranks_ = nranks;
    rank_ = rank;
    reduce_ = std::move(reduce);
}

void ML_BayesRatio_Manager::Reset() {
    auto z = [&](std::vector<double>& v) { v.assign(nlevels, 0.0); };
    sums.assign((size_t)nlevels * NVAR, 0.0);
    z(eR); z(varR); z(eYR); z(varYR); z(eABS_YR); z(eZ); z(varZ); z(eYZ); z(varYZ); z(eABS_YZ); z(eC); z(cost);
    z(eRatio); z(varRatio); z(eYRatio); z(varYRatio); z(eABS_YRatio);
    z(level_seconds);
    level_nsamples.assign(nlevels, 0);
    level_nsamples_missing.assign(nlevels, 0);
    ml_estimator_variance = ml_estimator_variance_R = ml_estimator_variance_Z = std::numeric_limits<double>::infinity();
    expected_discretization_error2 = expected_discretization_error2_R = expected_discretization_error2_Z = actualMSE =
        std::numeric_limits<double>::infinity();
}

// One level of InitRun (ML_BayesRatio_Manager.hpp:323-372 coarsest, :375-430 level pairs).  Realization i of the level
// uses prior draw 2i for Z and 2i+1 for R (two independent samples, :334-341).
int ML_BayesRatio_Manager::level_batch(int ilevel, int nsamples) const {
    // a level pair evaluates both levels of the pair in one plugin call each: the finer level's width decides
    int b = batch_;
    const int pref = problem.PreferredBatch(ilevel);
    if (pref > 0) b = std::min(b, pref);
    const int share = (nsamples + nranks_ - 1) / nranks_;
    return std::max(1, std::min(b, share));
}

void ML_BayesRatio_Manager::run_level(int ilevel, int nsamples) {
    const uint64_t base = (uint64_t)level_nsamples[ilevel];
    double* psum = pending_.data() + (size_t)ilevel * NVAR;
    const int lb = level_batch(ilevel, nsamples);
    std::vector<double> z(lb), r(lb), zc(lb), rc(lb), c(lb), ctot(lb), tmp(lb);
    const int nblocks = (nsamples + lb - 1) / lb;
    const bool coarsest = (ilevel == nlevels - 1);
    const double t0 = now_s();
    for (int blk = rank_; blk < nblocks; blk += nranks_) {
        const int first = blk * lb;
        const int m = std::min(lb, nsamples - first);
        std::fill(ctot.begin(), ctot.end(), 0.0);
        std::fill(zc.begin(), zc.end(), 0.0);
        std::fill(rc.begin(), rc.end(), 0.0);
        const uint64_t id0 = base + (uint64_t)first;
        // two independent prior draws per realization (:334-341): Z draws use the id range [2^62 + id0, ...), R draws
        // [id0, ...) - disjoint counter ranges of the generator, hence independent streams
        problem.SamplePrior(ilevel, zxi, ((uint64_t)1 << 62) + id0, m);
        problem.EvalPrior(ilevel, zxi, zparam);
        problem.ComputeLikelihoodAndR(ilevel, zparam, z.data(), tmp.data(), c.data());
        for (int b = 0; b < m; ++b) ctot[b] += c[b];
        problem.SamplePrior(ilevel, xi, id0, m);
        problem.EvalPrior(ilevel, xi, sparam);
        problem.ComputeLikelihoodAndR(ilevel, sparam, tmp.data(), r.data(), c.data());
        for (int b = 0; b < m; ++b) ctot[b] += c[b];
        if (!coarsest) {
            problem.EvalPrior(ilevel + 1, zxi, zparam);
            problem.ComputeLikelihoodAndR(ilevel + 1, zparam, zc.data(), tmp.data(), c.data());
            for (int b = 0; b < m; ++b) ctot[b] += c[b];
            problem.EvalPrior(ilevel + 1, xi, sparam);
            problem.ComputeLikelihoodAndR(ilevel + 1, sparam, tmp.data(), rc.data(), c.data());
            for (int b = 0; b < m; ++b) ctot[b] += c[b];
        }
        for (int b = 0; b < m; ++b) {
            const double y_r = coarsest ? r[b] : r[b] - rc[b];
            const double y_z = coarsest ? z[b] : z[b] - zc[b];
            psum[R] += r[b];
            psum[ABS_R] += std::fabs(r[b]);
            psum[R2] += r[b] * r[b];
            psum[YR] += y_r;
            psum[ABS_YR] += std::fabs(y_r);
            psum[YR2] += y_r * y_r;
            psum[Z] += z[b];
            psum[ABS_Z] += std::fabs(z[b]);
            psum[Z2] += z[b] * z[b];
            psum[YZ] += y_z;
            psum[ABS_YZ] += std::fabs(y_z);
            psum[YZ2] += y_z * y_z;
            if (splitting) {
                // "divide, then subtract" (ML_BayesRatio_Splitting_Manager.hpp:329,386-397); the plain ratio manager
                // leaves these columns at zero
                const double q = r[b] / z[b];
                const double y = coarsest ? q : q - rc[b] / zc[b];
                psum[Ratio] += q;
                psum[ABS_Ratio] += std::fabs(q);
                psum[Ratio2] += q * q;
                psum[YRatio] += y;
                psum[ABS_YRatio] += std::fabs(y);
                psum[YRatio2] += y * y;
            }
            psum[C] += ctot[b];
        }
    }
    pending_[(size_t)nlevels * NVAR + ilevel] += now_s() - t0;
}

void ML_BayesRatio_Manager::InitRun(std::vector<int>& level_nsamples_init) {
    if ((int)level_nsamples_init.size() != nlevels) throw std::invalid_argument("InitRun: wrong number of levels");
    pending_.assign((size_t)nlevels * (NVAR + 1), 0.0);
    for (int ilevel = nlevels - 1; ilevel >= 0; --ilevel) {
        const int ns = level_nsamples_init[ilevel];
        if (ns < 0) throw std::invalid_argument("InitRun: negative sample count");
        if (ns > 0) run_level(ilevel, ns);
    }
    if (nranks_ > 1) reduce_(pending_.data(), (int)pending_.size());
    for (size_t i = 0; i < (size_t)nlevels * NVAR; ++i) sums[i] += pending_[i];
    for (int l = 0; l < nlevels; ++l) {
        level_seconds[l] += pending_[(size_t)nlevels * NVAR + l];
        level_nsamples[l] += level_nsamples_init[l];
    }
    computeNSamplesMSE();
}

void ML_BayesRatio_Manager::Run() {
    Reset();
    std::vector<int> v_init(nlevels, init_nsamples_);
    InitRun(v_init);
    std::vector<int> grain(nlevels, 0);
    int rounds = 0;
    while (ml_estimator_variance > ratio * eps2) {
        if (++rounds > max_rounds_) throw std::runtime_error("ML_BayesRatio_Manager::Run: round limit reached");
        for (int i = 0; i < nlevels; ++i) {
            const int64_t miss = level_nsamples_missing[i];
            grain[i] = (int)std::min<int64_t>(miss, (int64_t)init_nsamples_ + grain[i] + miss / 10);
        }
        InitRun(grain);
    }
}

void ML_BayesRatio_Manager::computeNSamplesMSE() {
    for (int l = 0; l < nlevels; ++l) {
        const double n = (double)level_nsamples[l];
        eR[l] = S(l, R) / n;  varR[l] = S(l, R2) / n;  eYR[l] = S(l, YR) / n;  varYR[l] = S(l, YR2) / n;
        eABS_YR[l] = S(l, ABS_YR) / n;
        eZ[l] = S(l, Z) / n;  varZ[l] = S(l, Z2) / n;  eYZ[l] = S(l, YZ) / n;  varYZ[l] = S(l, YZ2) / n;
        eABS_YZ[l] = S(l, ABS_YZ) / n;
        eC[l] = S(l, C) / n;
        const double f = n / (n - 1.0);
        varR[l] = (varR[l] - eR[l] * eR[l]) * f;
        varYR[l] = (varYR[l] - eYR[l] * eYR[l]) * f;
        varZ[l] = (varZ[l] - eZ[l] * eZ[l]) * f;
        varYZ[l] = (varYZ[l] - eYZ[l] * eYZ[l]) * f;
        eRatio[l] = S(l, Ratio) / n;  eYRatio[l] = S(l, YRatio) / n;  eABS_YRatio[l] = S(l, ABS_YRatio) / n;
        varRatio[l] = (S(l, Ratio2) / n - eRatio[l] * eRatio[l]) * f;
        varYRatio[l] = (S(l, YRatio2) / n - eYRatio[l] * eYRatio[l]) * f;
    }
    if (wallTime)
        for (int l = 0; l < nlevels; ++l) cost[l] = level_seconds[l] / (double)level_nsamples[l];
    else
        cost = eC;
    alpha_R = expWRegression(eYR, M, 1);
    alphaABS_R = expWRegression(eABS_YR, M, 1);
    beta_R = expWRegression(varYR, M, 1);
    alpha_Z = expWRegression(eYZ, M, 1);
    alphaABS_Z = expWRegression(eABS_YZ, M, 1);
    beta_Z = expWRegression(varYZ, M, 1);
    gamma = expWRegression(cost, M, 0);
    alpha = expWRegression(eYRatio, M, 1);
    alphaABS = expWRegression(eABS_YRatio, M, 1);
    beta = expWRegression(varYRatio, M, 1);
    if (splitting) {
        // ML_BayesRatio_Splitting_Manager.hpp:668-716: everything is driven by the Ratio columns
        expected_discretization_error2_R = bias2_of(nlevels, M, eABS_YR, alphaABS_R);
        expected_discretization_error2_Z = bias2_of(nlevels, M, eABS_YZ, alphaABS_Z);
        expected_discretization_error2 = bias2_of(nlevels, M, eABS_YRatio, alphaABS);
        if (auto_eps2) eps2 = expected_discretization_error2 / (1. - ratio);
        ml_estimator_variance = ml_estimator_variance_Z = ml_estimator_variance_R = 0.;
        for (int l = 0; l < nlevels; ++l) {
            ml_estimator_variance += varYRatio[l] / (double)level_nsamples[l];
            ml_estimator_variance_Z += varYZ[l] / (double)level_nsamples[l];
            ml_estimator_variance_R += varYR[l] / (double)level_nsamples[l];
        }
        actualMSE = expected_discretization_error2 + ml_estimator_variance;
        double prop = 0.;
        for (int i = 0; i < nlevels; ++i) prop += std::sqrt(varYRatio[i] * cost[i]);
        prop /= ratio * eps2;
        for (int i = 0; i < nlevels; ++i) {
            const double mm = std::ceil(prop * std::sqrt(varYRatio[i] / cost[i]) - (double)level_nsamples[i]);
            level_nsamples_missing[i] = (std::isfinite(mm) && mm > 0.0) ? (int64_t)std::min(mm, 2.0e9) : 0;
        }
        return;
    }
    expected_discretization_error2_R = bias2_of(nlevels, M, eABS_YR, alphaABS_R);
    expected_discretization_error2_Z = bias2_of(nlevels, M, eABS_YZ, alphaABS_Z);
    expected_discretization_error2 = std::max(expected_discretization_error2_R, expected_discretization_error2_Z);
    if (auto_eps2) eps2 = expected_discretization_error2 / (1. - ratio);
    ml_estimator_variance_Z = ml_estimator_variance_R = 0.;
    for (int l = 0; l < nlevels; ++l) {
        ml_estimator_variance_Z += varYZ[l] / (double)level_nsamples[l];
        ml_estimator_variance_R += varYR[l] / (double)level_nsamples[l];
    }
    ml_estimator_variance = std::max(ml_estimator_variance_Z, ml_estimator_variance_R);
    actualMSE = expected_discretization_error2 + ml_estimator_variance;
    double prop_R = 0., prop_Z = 0.;
    for (int i = 0; i < nlevels; ++i) {
        prop_R += std::sqrt(varYR[i] * cost[i]);
        prop_Z += std::sqrt(varYZ[i] * cost[i]);
    }
    prop_R /= ratio * eps2;
    prop_Z /= ratio * eps2;
    for (int i = 0; i < nlevels; ++i) {
        const double mr = std::ceil(prop_R * std::sqrt(varYR[i] / cost[i]) - (double)level_nsamples[i]);
        const double mz = std::ceil(prop_Z * std::sqrt(varYZ[i] / cost[i]) - (double)level_nsamples[i]);
        const double mm = std::max(std::isfinite(mr) ? mr : 0.0, std::isfinite(mz) ? mz : 0.0);
        level_nsamples_missing[i] = mm > 0.0 ? (int64_t)std::min(mm, 2.0e9) : 0;
    }
}

// device-backed problem: prior = PDESampler, forward problem + observations = DarcySolver handle
class DeviceBayesRatioProblem : public BayesRatioProblem {
  public:
    DeviceBayesRatioProblem(pmc_ctx* ctx, pmc_sampler* smp, pmc_darcy* solver, double noise, std::vector<double> G_obs)
        : sampler_(ctx, smp), solver_(solver), bip_(solver, noise, std::move(G_obs)) {}
    void SamplePrior(int level, Vector& xi, uint64_t first_id, int nbatch) override { sampler_.Sample(level, xi, first_id, nbatch); }
    void EvalPrior(int level, const Vector& xi, Vector& s) override { sampler_.Eval(level, xi, s); }
    void ComputeLikelihoodAndR(int level, Vector& s, double* like, double* R, double* C) override {
        std::vector<double> q(s.Batch());
        bip_.ComputeLikelihoodAndQ(level, s, like, C, q.data());
        for (int b = 0; b < s.Batch(); ++b) R[b] = q[b] * like[b];
    }
    int GetGlobalNumberOfDofs(int level) const override { return pmc_darcy_num_dofs(solver_, level); }
    int PreferredBatch(int level) const override {
        return std::min(sampler_.PreferredBatch(level), pmc_darcy_batch_width(solver_, level));
    }

  private:
    PDESampler sampler_;
    pmc_darcy* solver_;
    BayesianInverseProblem bip_;
};

class CallbackBayesRatioProblem : public BayesRatioProblem {
  public:
    CallbackBayesRatioProblem(int nlevels, const pmc_plugin_callbacks& cb, pmc_cb_likelihood like)
        : prior_(nlevels, cb), cb_(cb), like_(like) {
        if (!like || !cb.ndofs) throw std::invalid_argument("likelihood callback / ndofs missing");
        ndofs_.assign(cb.ndofs, cb.ndofs + nlevels);
    }
    void SamplePrior(int level, Vector& xi, uint64_t first_id, int nbatch) override { prior_.Sample(level, xi, first_id, nbatch); }
    void EvalPrior(int level, const Vector& xi, Vector& s) override { prior_.Eval(level, xi, s); }
    void ComputeLikelihoodAndR(int level, Vector& s, double* like, double* R, double* C) override {
        if (like_(cb_.user, level, s.Batch(), s.GetData(), like, R, C) != 0) throw std::runtime_error("likelihood callback failed");
    }
    int GetGlobalNumberOfDofs(int level) const override { return ndofs_.at(level); }

  private:
    CallbackSampler prior_;
    pmc_plugin_callbacks cb_;
    pmc_cb_likelihood like_;
    std::vector<int> ndofs_;
};

}  // namespace parelagmc

// =================================================================================================
// C surface
using namespace parelagmc;

struct pmc_mlmc {
    std::unique_ptr<MLSampler> sampler;
    std::unique_ptr<PhysicalMLSolver> solver;
    std::vector<std::unique_ptr<MLSampler>> lane_samplers;
    std::vector<std::unique_ptr<PhysicalMLSolver>> lane_solvers;
    std::unique_ptr<MLMC_Manager> mgr;
    pmc_ctx* ctx = nullptr;
    std::vector<int64_t> ns, miss;
};

static thread_local std::string g_host_err;

template <class F>
static int hguard(F&& f) {
    try {
        f();
        return PMC_OK;
    } catch (const std::invalid_argument& e) {
        g_host_err = e.what();
        return PMC_ERR_INVALID;
    } catch (const std::exception& e) {
        g_host_err = e.what();
        return PMC_ERR_INTERNAL;
    } catch (...) {
        g_host_err = "unknown exception";
        return PMC_ERR_INTERNAL;
    }
}

extern "C" {

const char* pmc_host_last_error(void) { return g_host_err.c_str(); }

void pmc_mlmc_params_default(pmc_mlmc_params* p) {
    if (!p) return;
    p->eps2 = 0.001;
    p->ratio = 0.5;
    p->init_nsamples = 10;
    p->array_nsamples = nullptr;
    p->wall_time = 1;
    p->batch = 256;
    p->max_rounds = 1000;
    p->log_file = nullptr;
}

int pmc_mlmc_create(pmc_ctx* ctx, pmc_sampler* sampler, pmc_darcy* solver, int nlevels, const pmc_mlmc_params* params,
                    pmc_mlmc** out) {
    return hguard([&] {
        if (!ctx || !sampler || !solver || !out) throw std::invalid_argument("pmc_mlmc_create: NULL argument");
        pmc_mlmc_params p;
        pmc_mlmc_params_default(&p);
        if (params) p = *params;
        std::unique_ptr<pmc_mlmc> m(new pmc_mlmc());
        m->ctx = ctx;
        m->sampler.reset(new PDESampler(ctx, sampler));
        m->solver.reset(new DarcySolver(ctx, solver));
        m->mgr.reset(new MLMC_Manager(ctx, PMC_MEM_DEVICE, nlevels, *m->solver, *m->sampler, p));
        *out = m.release();
    });
}

int pmc_mlmc_create_callbacks(int nlevels, const pmc_plugin_callbacks* cb, const pmc_mlmc_params* params, pmc_mlmc** out) {
    return hguard([&] {
        if (!cb || !out) throw std::invalid_argument("pmc_mlmc_create_callbacks: NULL argument");
        pmc_mlmc_params p;
        pmc_mlmc_params_default(&p);
        if (params) p = *params;
        std::unique_ptr<pmc_mlmc> m(new pmc_mlmc());
        m->sampler.reset(new CallbackSampler(nlevels, *cb));
        m->solver.reset(new CallbackSolver(nlevels, *cb));
        m->mgr.reset(new MLMC_Manager(nullptr, PMC_MEM_HOST, nlevels, *m->solver, *m->sampler, p));
        *out = m.release();
    });
}

int pmc_mlmc_add_lane(pmc_mlmc* m, pmc_ctx* ctx, pmc_sampler* sampler, pmc_darcy* solver) {
    return hguard([&] {
        if (!m || !ctx || !sampler || !solver) throw std::invalid_argument("pmc_mlmc_add_lane: NULL argument");
        if (!m->ctx) throw std::invalid_argument("pmc_mlmc_add_lane: callback managers have a single lane");
        m->lane_samplers.emplace_back(new PDESampler(ctx, sampler));
        m->lane_solvers.emplace_back(new DarcySolver(ctx, solver));
        m->mgr->AddLane(ctx, *m->lane_solvers.back(), *m->lane_samplers.back());
    });
}

void pmc_mlmc_destroy(pmc_mlmc* m) { delete m; }

int pmc_mlmc_set_farm(pmc_mlmc* m, int nranks, int rank, pmc_reduce_fn reduce, void* user) {
    return hguard([&] {
        if (!m) throw std::invalid_argument("manager is NULL");
        std::function<void(double*, int)> fn;
        if (reduce) {
            fn = [reduce, user](double* b, int n) {
                if (reduce(b, n, user) != 0) throw std::runtime_error("reduce callback failed");
            };
        } else if (nranks > 1) {
            pmc_ctx* ctx = m->ctx;
            if (!ctx) throw std::invalid_argument("pmc_mlmc_set_farm: no reduction and no device context");
            fn = [ctx](double* b, int n) {
                if (pmc_allreduce_sum_f64(ctx, b, n) != PMC_OK) throw std::runtime_error(pmc_last_error());
            };
        }
        m->mgr->SetFarm(nranks, rank, fn);
    });
}

int pmc_mlmc_run(pmc_mlmc* m) {
    return hguard([&] {
        if (!m) throw std::invalid_argument("manager is NULL");
        m->mgr->Run();
    });
}
int pmc_mlmc_replay_log(pmc_mlmc* m, const char* path, int64_t* nread) {
    return hguard([&] {
        if (!m || !path) throw std::invalid_argument("pmc_mlmc_replay_log: NULL argument");
        const int64_t n = m->mgr->ReplayLog(path);
        if (nread) *nread = n;
    });
}
int pmc_mlmc_reset(pmc_mlmc* m) {
    return hguard([&] {
        if (!m) throw std::invalid_argument("manager is NULL");
        m->mgr->Reset();
    });
}
int pmc_mlmc_init_run(pmc_mlmc* m, const int32_t* nsamples) {
    return hguard([&] {
        if (!m || !nsamples) throw std::invalid_argument("pmc_mlmc_init_run: NULL argument");
        std::vector<int> v(nsamples, nsamples + m->mgr->nlevels);
        m->mgr->InitRun(v);
    });
}
int pmc_mlmc_show_me(pmc_mlmc* m, char* buf, size_t cap, size_t* needed) {
    return hguard([&] {
        if (!m) throw std::invalid_argument("manager is NULL");
        std::ostringstream ss;
        m->mgr->ShowMe(ss);
        const std::string t = ss.str();
        if (needed) *needed = t.size() + 1;
        if (buf && cap) {
            const size_t k = std::min(cap - 1, t.size());
            std::memcpy(buf, t.data(), k);
            buf[k] = '\0';
        }
    });
}
int pmc_mlmc_print_timers(pmc_mlmc* m, char* buf, size_t cap, size_t* needed) {
    return hguard([&] {
        if (!m) throw std::invalid_argument("manager is NULL");
        std::ostringstream ss;
        m->mgr->PrintTimers(ss);
        const std::string t = ss.str();
        if (needed) *needed = t.size() + 1;
        if (buf && cap) {
            const size_t k = std::min(cap - 1, t.size());
            std::memcpy(buf, t.data(), k);
            buf[k] = '\0';
        }
    });
}
int pmc_mlmc_phase_times(pmc_mlmc* m, int level, double* sampler_mult_ms, double* darcy_setup_ms, double* darcy_mult_ms,
                         int64_t* sampler_realizations, int64_t* darcy_realizations) {
    return hguard([&] {
        if (!m) throw std::invalid_argument("manager is NULL");
        m->mgr->PhaseTimesOfLevel(level, sampler_mult_ms, darcy_setup_ms, darcy_mult_ms, sampler_realizations,
                                  darcy_realizations);
    });
}
int pmc_mlmc_farm_times(pmc_mlmc* m, double* allreduce_ms, int64_t* reductions) {
    return hguard([&] {
        if (!m) throw std::invalid_argument("manager is NULL");
        double sec = 0.0;
        m->mgr->FarmTimes(&sec, reductions);
        if (allreduce_ms) *allreduce_ms = 1e3 * sec;
    });
}
int pmc_mlmc_result_get(pmc_mlmc* m, pmc_mlmc_result* r) {
    return hguard([&] {
        if (!m || !r) throw std::invalid_argument("pmc_mlmc_result_get: NULL argument");
        MLMC_Manager& g = *m->mgr;
        r->nlevels = g.nlevels;
        double est = 0;
        for (double x : g.eY) est += x;
        r->estimate = est;
        r->eps2 = g.eps2;
        r->actual_mse = g.actualMSE;
        r->estimator_variance = g.ml_estimator_variance;
        r->bias2 = g.expected_discretization_error2;
        r->alpha = g.alpha;
        r->alpha_abs = g.alphaABS;
        r->beta = g.beta;
        r->gamma = g.gamma;
        r->eY = g.eY.data(); r->eABSY = g.eABSY.data(); r->eQ = g.eQ.data(); r->eABSQ = g.eABSQ.data();
        r->eC = g.eC.data(); r->varY = g.varY.data(); r->varQ = g.varQ.data(); r->consistency = g.consistency.data();
        r->kurtosis = g.kurtosis.data(); r->VC = g.VC.data(); r->cost = g.cost.data();
        r->sums = g.sums.data();
        r->nsamples = g.level_nsamples.data();
        r->nsamples_missing = g.level_nsamples_missing.data();
        r->level_seconds = g.level_seconds.data();
    });
}

int pmc_bayes_likelihood(pmc_darcy* solver, int level, int nbatch, const double* k, int memspace, const double* G_obs,
                         int nobs, double noise, double* likelihood, double* C, double* Q, double* R) {
    return hguard([&] {
        if (!solver || !k || !G_obs || !likelihood || nbatch < 1 || nobs < 1 || !(noise > 0.0))
            throw std::invalid_argument("pmc_bayes_likelihood: bad argument");
        // a non-owning batch view of k
        struct View : Vector {
            View(double* p, int n, int nb, int ms) : Vector(nullptr, ms) { Adopt(p, n, nb); }
            ~View() { Release(); }
        };
        BayesianInverseProblem prob(solver, noise, std::vector<double>(G_obs, G_obs + nobs));
        std::vector<double> G, q(nbatch);
        std::vector<double> cdummy(nbatch);
        // ComputeG needs the per-realization size of k = n_p(level): recover it from the solver
        int n_k = pmc_darcy_num_pressure_dofs(solver, level);
        if (n_k <= 0) throw std::invalid_argument("pmc_bayes_likelihood: level out of range");
        View kv(const_cast<double*>(k), n_k, nbatch, memspace);
        prob.ComputeLikelihoodAndQ(level, kv, likelihood, C ? C : cdummy.data(), q.data());
        for (int b = 0; b < nbatch; ++b) {
            if (Q) Q[b] = q[b];
            if (R) R[b] = q[b] * likelihood[b];
        }
    });
}

struct pmc_ratio {
    std::unique_ptr<BayesRatioProblem> problem;
    std::unique_ptr<ML_BayesRatio_Manager> mgr;
    pmc_ctx* ctx = nullptr;
};

int pmc_ratio_create(pmc_ctx* ctx, pmc_sampler* sampler, pmc_darcy* solver, int nlevels, const double* G_obs, int nobs,
                     double noise, const pmc_mlmc_params* params, pmc_ratio** out) {
    return hguard([&] {
        if (!ctx || !sampler || !solver || !G_obs || !out || nobs < 1 || !(noise > 0.0))
            throw std::invalid_argument("pmc_ratio_create: bad argument");
        pmc_mlmc_params p;
        pmc_mlmc_params_default(&p);
        if (params) p = *params;
        std::unique_ptr<pmc_ratio> m(new pmc_ratio());
        m->ctx = ctx;
        m->problem.reset(new DeviceBayesRatioProblem(ctx, sampler, solver, noise, std::vector<double>(G_obs, G_obs + nobs)));
        m->mgr.reset(new ML_BayesRatio_Manager(ctx, PMC_MEM_DEVICE, nlevels, *m->problem, p));
        *out = m.release();
    });
}
int pmc_ratio_create_callbacks(int nlevels, const pmc_plugin_callbacks* cb, pmc_cb_likelihood like,
                               const pmc_mlmc_params* params, pmc_ratio** out) {
    return hguard([&] {
        if (!cb || !like || !out) throw std::invalid_argument("pmc_ratio_create_callbacks: NULL argument");
        pmc_mlmc_params p;
        pmc_mlmc_params_default(&p);
        if (params) p = *params;
        std::unique_ptr<pmc_ratio> m(new pmc_ratio());
        m->problem.reset(new CallbackBayesRatioProblem(nlevels, *cb, like));
        m->mgr.reset(new ML_BayesRatio_Manager(nullptr, PMC_MEM_HOST, nlevels, *m->problem, p));
        *out = m.release();
    });
}
void pmc_ratio_destroy(pmc_ratio* m) { delete m; }
int pmc_ratio_set_farm(pmc_ratio* m, int nranks, int rank, pmc_reduce_fn reduce, void* user) {
    return hguard([&] {
        if (!m) throw std::invalid_argument("manager is NULL");
        std::function<void(double*, int)> fn;
        if (reduce) {
            fn = [reduce, user](double* b, int n) {
                if (reduce(b, n, user) != 0) throw std::runtime_error("reduce callback failed");
            };
        } else if (nranks > 1) {
            pmc_ctx* ctx = m->ctx;
            if (!ctx) throw std::invalid_argument("pmc_ratio_set_farm: no reduction and no device context");
            fn = [ctx](double* b, int n) {
                if (pmc_allreduce_sum_f64(ctx, b, n) != PMC_OK) throw std::runtime_error(pmc_last_error());
            };
        }
        m->mgr->SetFarm(nranks, rank, fn);
    });
}
int pmc_ratio_set_splitting(pmc_ratio* m, int on) {
    return hguard([&] {
        if (!m) throw std::invalid_argument("manager is NULL");
        m->mgr->SetSplitting(on != 0);
    });
}
int pmc_ratio_run(pmc_ratio* m) {
    return hguard([&] {
        if (!m) throw std::invalid_argument("manager is NULL");
        m->mgr->Run();
    });
}
int pmc_ratio_init_run(pmc_ratio* m, const int32_t* nsamples) {
    return hguard([&] {
        if (!m || !nsamples) throw std::invalid_argument("pmc_ratio_init_run: NULL argument");
        std::vector<int> v(nsamples, nsamples + m->mgr->nlevels);
        m->mgr->InitRun(v);
    });
}
int pmc_ratio_result_get(pmc_ratio* m, pmc_ratio_result* r) {
    return hguard([&] {
        if (!m || !r) throw std::invalid_argument("pmc_ratio_result_get: NULL argument");
        ML_BayesRatio_Manager& g = *m->mgr;
        double er = 0, ez = 0;
        for (double x : g.eYR) er += x;
        for (double x : g.eYZ) ez += x;
        r->nlevels = g.nlevels;
        r->R_estimate = er;
        r->Z_estimate = ez;
        double eq = 0;
        for (double x : g.eYRatio) eq += x;
        r->ratio_estimate = g.splitting ? eq : er / ez;   // "Ratio Estimate": eYRatio.Sum() vs R / Z
        r->eps2 = g.eps2;
        r->actual_mse = g.actualMSE;
        r->estimator_variance = g.ml_estimator_variance;
        r->estimator_variance_R = g.ml_estimator_variance_R;
        r->estimator_variance_Z = g.ml_estimator_variance_Z;
        r->bias2 = g.expected_discretization_error2;
        r->bias2_R = g.expected_discretization_error2_R;
        r->bias2_Z = g.expected_discretization_error2_Z;
        r->alpha_R = g.alpha_R; r->alpha_abs_R = g.alphaABS_R; r->beta_R = g.beta_R;
        r->alpha_Z = g.alpha_Z; r->alpha_abs_Z = g.alphaABS_Z; r->beta_Z = g.beta_Z;
        r->gamma = g.gamma;
        r->alpha = g.alpha; r->alpha_abs = g.alphaABS; r->beta = g.beta;
        r->eRatio = g.eRatio.data(); r->varRatio = g.varRatio.data(); r->eYRatio = g.eYRatio.data();
        r->varYRatio = g.varYRatio.data(); r->eABS_YRatio = g.eABS_YRatio.data();
        r->eR = g.eR.data(); r->varR = g.varR.data(); r->eYR = g.eYR.data(); r->varYR = g.varYR.data();
        r->eABS_YR = g.eABS_YR.data(); r->eZ = g.eZ.data(); r->varZ = g.varZ.data(); r->eYZ = g.eYZ.data();
        r->varYZ = g.varYZ.data(); r->eABS_YZ = g.eABS_YZ.data(); r->eC = g.eC.data(); r->cost = g.cost.data();
        r->sums = g.sums.data();
        r->nsamples = g.level_nsamples.data();
        r->nsamples_missing = g.level_nsamples_missing.data();
    });
}

double pmc_exp_w_regression(const double* y, const double* x, int n, int skip_n_last) {
    return expWRegression(std::vector<double>(y, y + n), std::vector<double>(x, x + n), skip_n_last);
}

}  // extern "C"
