// extern "C" surface of libpmc.so (see include/pmc.h).  No exception leaves this file.
#include <dlfcn.h>

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "handles.hpp"

namespace pmc {

static thread_local std::string g_last_error;
void set_last_error(const std::string& m) { g_last_error = m; }

void Ctx::activate() const { PMC_HIP(hipSetDevice(device)); }
void Ctx::phase_mark(int i) const { PMC_HIP(hipEventRecord(ev_phase[i], stream)); }
void Ctx::phase_report(pmc_stats* stats, int nb) const {
    PMC_HIP(hipEventSynchronize(ev_phase[2]));
    float setup = 0.f, solve = 0.f;
    PMC_HIP(hipEventElapsedTime(&setup, ev_phase[0], ev_phase[1]));
    PMC_HIP(hipEventElapsedTime(&solve, ev_phase[1], ev_phase[2]));
    for (int k = 0; k < nb; ++k) {
        stats[k].setup_ms = (double)setup / nb;
        stats[k].solve_ms = (double)solve / nb;
    }
}

static std::atomic<int> g_live_ctx[64];
int Ctx::contexts_on_device(int device) { return g_live_ctx[device & 63].load(); }

Lanes Ctx::lanes(bool split) {
    if (split && !stream2) {
        PMC_HIP(hipStreamCreateWithFlags(&stream2, hipStreamNonBlocking));
        PMC_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
        PMC_HIP(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
    }
    return Lanes{stream, split ? stream2 : stream, ev_fork, ev_join, split};
}

template <class F>
static int guarded(F&& f) {
    try {
        f();
        return PMC_OK;
    } catch (const Error& e) {
        set_last_error(e.what());
        return e.code;
    } catch (const std::bad_alloc&) {
        set_last_error("host allocation failed");
        return PMC_ERR_INTERNAL;
    } catch (const std::exception& e) {
        set_last_error(e.what());
        return PMC_ERR_INTERNAL;
    } catch (...) {
        set_last_error("unknown exception");
        return PMC_ERR_INTERNAL;
    }
}

// ---- RCCL, resolved lazily so that single-GPU use never needs the library ---------------------
struct UniqueId { char internal[128]; };
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(void**, int, UniqueId, int) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
static Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char* n : names) {
            r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (r.lib) break;
        }
        if (!r.lib) return;
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.lib, "ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.lib, "ncclCommInitRank"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(r.lib, "ncclAllReduce"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
    });
    if (!r.lib || !r.GetUniqueId || !r.CommInitRank || !r.AllReduce || !r.CommDestroy)
        throw Error(PMC_ERR_COMM, "RCCL (librccl.so) could not be loaded");
    return r;
}
static void rccl_check(int rc, const char* what) {
    if (rc != 0) {
        Rccl& r = rccl();
        throw Error(PMC_ERR_COMM, std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(rc) : "RCCL error"));
    }
}

}  // namespace pmc

using namespace pmc;

extern "C" {

int pmc_version(void) { return 100; }
int pmc_abi_version(void) { return PMC_ABI_VERSION; }
int pmc_krylov_z_bytes(void) { return 4; }   // of the default options (PMC_STORAGE_FP32)
uint64_t pmc_kernel_launches(void) { return kernel_launch_count(); }
const char* pmc_last_error(void) { return g_last_error.c_str(); }

static void check_abi(const pmc_solver_opts& o) {
    if (o.abi_version != PMC_ABI_VERSION)
        throw Error(PMC_ERR_INVALID, "pmc_solver_opts.abi_version is " + std::to_string(o.abi_version) + ", this library has " +
                                         std::to_string(PMC_ABI_VERSION) +
                                         ": fill the struct with pmc_solver_opts_default() of the header you compile against");
}

void pmc_solver_opts_default(pmc_solver_opts* o) {
    if (!o) return;
    o->max_iter = 300;
    o->rel_tol = 1e-6;
    o->abs_tol = 1e-12;
    o->abi_version = PMC_ABI_VERSION;
    // tuned on MI355X (scripts/sweep.py, cube_tet r=5): degree 2 on M needs no more MINRES iterations than
    // degree 3; smoothing interval [lmax/8, lmax]
    o->cheb_degree_M = 0;
    o->cheb_ratio_M = 0.0;
    o->mg_smooth_degree = 2;
    o->mg_smooth_ratio = 8.0;
    o->mg_coarse_degree = 12;
    o->mg_coarse_ratio = 100.0;
    o->check_every = 2;
    o->schur_scale = 1.0;
    o->mg_coarsening = 2;
    o->mini_max_rows = 6000;
    o->two_streams = 0;
    o->use_graph = 0;   // measured: no gain single-stream (kernels are latency-, not launch-bound), slower with 4 lanes
    o->precond_storage = PMC_STORAGE_FP32;
}

#undef pmc_ctx_create   // the header redirects the name to pmc_ctx_create_abi for C / C++ callers
int pmc_ctx_create(int device_id, pmc_ctx** out);
int pmc_ctx_create_abi(int device_id, int abi_version, pmc_ctx** out) {
    if (abi_version != PMC_ABI_VERSION) {
        if (out) *out = nullptr;
        set_last_error("caller was compiled against PMC_ABI_VERSION " + std::to_string(abi_version) + ", this library has " +
                       std::to_string(PMC_ABI_VERSION) + " (layout of pmc_solver_opts / pmc_stats): rebuild against include/pmc.h");
        return PMC_ERR_INVALID;
    }
    return pmc_ctx_create(device_id, out);
}
int pmc_ctx_create(int device_id, pmc_ctx** out) {
    return guarded([&] {
        PMC_REQUIRE(out != nullptr, "pmc_ctx_create: out is NULL");
        *out = nullptr;
        int ndev = 0;
        hipError_t e = hipGetDeviceCount(&ndev);
        if (e != hipSuccess || ndev == 0)
            throw Error(PMC_ERR_DEVICE, "no HIP device available (libpmc has no CPU fallback)");
        PMC_REQUIRE(device_id >= 0 && device_id < ndev, "pmc_ctx_create: device id out of range");
        std::unique_ptr<pmc_ctx> c(new pmc_ctx());
        c->device = device_id;
        PMC_HIP(hipSetDevice(device_id));
        PMC_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        PMC_HIP(hipEventCreate(&c->ev0));
        PMC_HIP(hipEventCreate(&c->ev1));
        for (hipEvent_t& e : c->ev_phase) PMC_HIP(hipEventCreate(&e));
        PMC_HIP(hipHostMalloc(reinterpret_cast<void**>(&c->h_flag), sizeof(int) * 16, hipHostMallocDefault));
        PMC_HIP(hipHostMalloc(reinterpret_cast<void**>(&c->h_scal), sizeof(double) * pmc::Ctx::kHostScratch, hipHostMallocDefault));
        g_live_ctx[device_id & 63].fetch_add(1);
        *out = c.release();
    });
}

void pmc_ctx_destroy(pmc_ctx* c) {
    if (!c) return;
    g_live_ctx[c->device & 63].fetch_sub(1);
    (void)hipSetDevice(c->device);
    if (c->nccl) {
        try { rccl().CommDestroy(c->nccl); } catch (...) {}
        c->nccl = nullptr;
    }
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->stream2) (void)hipStreamSynchronize(c->stream2);
    if (c->h_flag) (void)hipHostFree(c->h_flag);
    if (c->h_scal) (void)hipHostFree(c->h_scal);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    for (hipEvent_t e : c->ev_phase)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : {c->ev_fork, c->ev_join})
        if (e) (void)hipEventDestroy(e);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int pmc_ctx_synchronize(pmc_ctx* c) {
    return guarded([&] {
        PMC_REQUIRE(c != nullptr, "ctx is NULL");
        c->activate();
        PMC_HIP(hipStreamSynchronize(c->stream));
    });
}

void* pmc_ctx_stream(pmc_ctx* c) { return c ? (void*)c->stream : nullptr; }

int pmc_timer_start(pmc_ctx* c) {
    return guarded([&] {
        PMC_REQUIRE(c != nullptr, "ctx is NULL");
        c->activate();
        PMC_HIP(hipEventRecord(c->ev0, c->stream));
    });
}
int pmc_timer_stop(pmc_ctx* c, double* ms) {
    return guarded([&] {
        PMC_REQUIRE(c != nullptr && ms != nullptr, "ctx/ms is NULL");
        c->activate();
        PMC_HIP(hipEventRecord(c->ev1, c->stream));
        PMC_HIP(hipEventSynchronize(c->ev1));
        float f = 0.f;
        PMC_HIP(hipEventElapsedTime(&f, c->ev0, c->ev1));
        *ms = (double)f;
    });
}

int pmc_malloc(pmc_ctx* c, size_t bytes, void** dptr) {
    return guarded([&] {
        PMC_REQUIRE(c != nullptr && dptr != nullptr, "pmc_malloc: NULL argument");
        c->activate();
        *dptr = nullptr;
        if (bytes) PMC_HIP(hipMalloc(dptr, bytes));
    });
}
int pmc_free(pmc_ctx* c, void* dptr) {
    return guarded([&] {
        PMC_REQUIRE(c != nullptr, "ctx is NULL");
        c->activate();
        if (dptr) PMC_HIP(hipFree(dptr));
    });
}
int pmc_memcpy_h2d(pmc_ctx* c, void* dst, const void* src, size_t bytes) {
    return guarded([&] {
        PMC_REQUIRE(c != nullptr, "ctx is NULL");
        c->activate();
        if (bytes) {
            PMC_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
            PMC_HIP(hipStreamSynchronize(c->stream));
        }
    });
}
int pmc_memcpy_d2h(pmc_ctx* c, void* dst, const void* src, size_t bytes) {
    return guarded([&] {
        PMC_REQUIRE(c != nullptr, "ctx is NULL");
        c->activate();
        if (bytes) {
            PMC_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
            PMC_HIP(hipStreamSynchronize(c->stream));
        }
    });
}

int pmc_rng_seed(pmc_ctx* c, uint64_t seed, int nparts, int mypart) {
    return guarded([&] {
        PMC_REQUIRE(c != nullptr, "ctx is NULL");
        PMC_REQUIRE(nparts >= 1 && mypart >= 0 && mypart < nparts, "pmc_rng_seed: need 0 <= mypart < nparts");
        c->seed = seed;
        c->nparts = nparts;
        c->mypart = mypart;
    });
}

int pmc_normal_fill(pmc_ctx* c, double mean, double sigma2, uint64_t first_id, uint32_t stream, int nbatch, int n,
                    double* out, int memspace) {
    return guarded([&] {
        PMC_REQUIRE(c != nullptr && out != nullptr, "pmc_normal_fill: NULL argument");
        PMC_REQUIRE(n >= 0 && nbatch >= 0 && sigma2 >= 0.0, "pmc_normal_fill: bad size / variance");
        c->activate();
        const double sigma = std::sqrt(sigma2);   // NormalDistributionSampler ctor, cpp:17-19
        if (memspace == PMC_MEM_DEVICE) {
            k::normal_fill(c->stream, n, nbatch, c->seed, c->stream_id(first_id), stream, mean, sigma, out, (uint64_t)c->nparts);
        } else {
            DevBuf<double> tmp((size_t)n * nbatch);
            k::normal_fill(c->stream, n, nbatch, c->seed, c->stream_id(first_id), stream, mean, sigma, tmp.p, (uint64_t)c->nparts);
            PMC_HIP(hipMemcpyAsync(out, tmp.p, sizeof(double) * n * nbatch, hipMemcpyDeviceToHost, c->stream));
            PMC_HIP(hipStreamSynchronize(c->stream));
        }
    });
}

// ---- sampler -----------------------------------------------------------------------------------
int pmc_sampler_create(pmc_ctx* c, int nlevels, int n_mc_levels, const pmc_sampler_level* levels, double alpha,
                       double matern_g, int lognormal, const pmc_solver_opts* opts, pmc_sampler** out) {
    return guarded([&] {
        PMC_REQUIRE(c != nullptr && out != nullptr, "pmc_sampler_create: NULL argument");
        *out = nullptr;
        pmc_solver_opts o;
        pmc_solver_opts_default(&o);
        if (opts) {
            check_abi(*opts);
            o = *opts;
        }
        PMC_REQUIRE(o.max_iter >= 1 && o.cheb_degree_M >= 0 && o.mg_smooth_degree >= 1 && o.mg_coarse_degree >= 1 &&
                        (o.cheb_ratio_M <= 0.0 || o.cheb_ratio_M > 1.0) && o.mg_smooth_ratio > 1.0 && o.mg_coarse_ratio > 1.0 &&
                        (o.precond_storage == PMC_STORAGE_FP32 || o.precond_storage == PMC_STORAGE_FP64),
                    "solver options out of range");
        *out = new pmc_sampler(*c, nlevels, n_mc_levels, levels, alpha, matern_g, lognormal != 0, o);
        for (int l = 0; l < (*out)->impl.nlevels; ++l) (*out)->impl.lv[l].out_size = (*out)->impl.lv[l].n_s;
    });
}
int pmc_sampler_create_hybrid(pmc_ctx* c, int nlevels, const pmc_hybrid_level* levels, double alpha, double matern_g,
                              int lognormal, const pmc_solver_opts* opts, pmc_sampler** out) {
    return guarded([&] {
        PMC_REQUIRE(c != nullptr && out != nullptr, "pmc_sampler_create_hybrid: NULL argument");
        *out = nullptr;
        pmc_solver_opts o;
        pmc_solver_opts_default(&o);
        if (opts) {
            check_abi(*opts);
            o = *opts;
        }
        PMC_REQUIRE(o.max_iter >= 1 && o.mg_smooth_degree >= 1 && o.mg_coarse_degree >= 1 && o.mg_smooth_ratio > 1.0 &&
                        o.mg_coarse_ratio > 1.0 && (o.precond_storage == PMC_STORAGE_FP32 || o.precond_storage == PMC_STORAGE_FP64),
                    "solver options out of range");
        *out = new pmc_sampler(*c, nlevels, levels, alpha, matern_g, lognormal != 0, o);
        for (int l = 0; l < (*out)->impl.nlevels; ++l) (*out)->impl.lv[l].out_size = (*out)->impl.lv[l].n_s;
    });
}
int pmc_sampler_is_hybrid(const pmc_sampler* s) { return s ? (s->impl.hybrid ? 1 : 0) : PMC_ERR_INVALID; }
void pmc_sampler_destroy(pmc_sampler* s) {
    if (!s) return;
    (void)hipSetDevice(s->impl.ctx.device);
    (void)hipStreamSynchronize(s->impl.ctx.stream);
    delete s;
}
int pmc_sampler_set_projection(pmc_sampler* s, int level, int kind, const pmc_csr* Gt, const int32_t* idx,
                               const double* inv_w, int orig_size) {
    return guarded([&] {
        PMC_REQUIRE(s != nullptr, "sampler is NULL");
        s->impl.set_projection(level, kind, Gt, idx, inv_w, orig_size);
    });
}
int pmc_sampler_num_levels(const pmc_sampler* s) { return s ? s->impl.n_mc : PMC_ERR_INVALID; }
int pmc_sampler_xi_size(const pmc_sampler* s, int level) {
    if (!s || level < 0 || level >= s->impl.nlevels) return PMC_ERR_INVALID;
    return s->impl.lv[level].n_s;
}
int pmc_sampler_sample_size(const pmc_sampler* s, int level) {
    if (!s || level < 0 || level >= s->impl.nlevels) return PMC_ERR_INVALID;
    return s->impl.lv[level].out_size;
}
int pmc_sampler_krylov_z_bytes(const pmc_sampler* s) {
    return s ? (s->impl.opts.precond_storage == PMC_STORAGE_FP64 ? 8 : 4) : PMC_ERR_INVALID;
}
int pmc_darcy_krylov_z_bytes(const pmc_darcy* d) {
    return d ? (d->impl.opts.precond_storage == PMC_STORAGE_FP64 ? 8 : 4) : PMC_ERR_INVALID;
}
int pmc_sampler_mult(pmc_sampler* s, int level, int nbatch, const double* rhs, double* sol, int use_sol_as_guess,
                     int memspace, pmc_stats* stats) {
    return guarded([&] {
        PMC_REQUIRE(s != nullptr, "sampler is NULL");
        PMC_REQUIRE(memspace == PMC_MEM_HOST || memspace == PMC_MEM_DEVICE, "bad memspace");
        s->impl.mult(level, nbatch, rhs, sol, use_sol_as_guess != 0, memspace, stats);
    });
}
int pmc_sampler_apply_preconditioner(pmc_sampler* s, int level, int nbatch, const double* r, double* z, int memspace) {
    return guarded([&] {
        PMC_REQUIRE(s != nullptr, "sampler is NULL");
        PMC_REQUIRE(memspace == PMC_MEM_HOST || memspace == PMC_MEM_DEVICE, "bad memspace");
        s->impl.apply_preconditioner(level, nbatch, r, z, memspace);
    });
}
int pmc_sampler_batch_width(const pmc_sampler* s, int level) {
    if (!s || level < 0 || level >= s->impl.nlevels) return PMC_ERR_INVALID;
    return batch_width((size_t)s->impl.lv[level].n_u + s->impl.lv[level].n_s, false, s->impl.ctx.device);
}
int pmc_sampler_true_p(const pmc_sampler* s, int level, pmc_csr* out) {
    return guarded([&] {
        PMC_REQUIRE(s != nullptr && out != nullptr, "pmc_sampler_true_p: NULL argument");
        PMC_REQUIRE(level >= 0 && level + 1 < s->impl.nlevels, "pmc_sampler_true_p: level has no coarser level");
        const HostCsr& P = s->impl.lv[level].P_host;
        *out = pmc_csr{P.nrows, P.ncols, P.rowptr.data(), P.colind.data(), P.vals.data()};
    });
}

int64_t pmc_sampler_nnz(const pmc_sampler* s, int level) {
    if (!s || level < 0 || level >= s->impl.nlevels) return PMC_ERR_INVALID;
    return s->impl.lv[level].nnz;
}
int pmc_sampler_sample(pmc_sampler* s, int level, uint64_t first_id, int nbatch, double* xi, int memspace) {
    return guarded([&] {
        PMC_REQUIRE(s != nullptr, "sampler is NULL");
        s->impl.sample(level, first_id, nbatch, xi, memspace);
    });
}
int pmc_sampler_eval(pmc_sampler* s, int level, int xi_level, int nbatch, const double* xi, double* s_out,
                     const double* init_s, int init_level, int use_init, double* embed_s_out, int memspace,
                     pmc_stats* stats) {
    return guarded([&] {
        PMC_REQUIRE(s != nullptr, "sampler is NULL");
        s->impl.eval(level, xi_level, nbatch, xi, s_out, init_s, init_level, use_init != 0, embed_s_out, memspace, stats);
    });
}

int pmc_sampler_apply_operator(pmc_sampler* s, int level, int nbatch, const double* x, double* y, int memspace,
                               int repeat, double* avg_ms, double* bytes) {
    return guarded([&] {
        PMC_REQUIRE(s != nullptr, "sampler is NULL");
        s->impl.apply_operator(level, nbatch, x, y, memspace, repeat, avg_ms, bytes);
    });
}

int pmc_sampler_set_operator_timing(pmc_sampler* s, int on) {
    return guarded([&] {
        PMC_REQUIRE(s != nullptr, "sampler is NULL");
        s->impl.work.op_timer.on = on != 0;
    });
}

int pmc_sampler_operator_time(pmc_sampler* s, double* total_ms, int64_t* launches) {
    return guarded([&] {
        PMC_REQUIRE(s != nullptr, "sampler is NULL");
        if (total_ms) *total_ms = s->impl.work.op_timer.ms;
        if (launches) *launches = s->impl.work.op_timer.launches;
        s->impl.work.op_timer.ms = 0.0;
        s->impl.work.op_timer.launches = 0;
    });
}

int pmc_sampler_smoother_time(pmc_sampler* s, double* total_ms, int64_t* launches, double* event_overhead_ms) {
    return guarded([&] {
        PMC_REQUIRE(s != nullptr, "sampler is NULL");
        if (total_ms) *total_ms = s->impl.vc_timer.ms;
        if (launches) *launches = s->impl.vc_timer.launches;
        if (event_overhead_ms) *event_overhead_ms = s->impl.vc_timer.gap_ms;
        s->impl.vc_timer.clear();
    });
}
int pmc_sampler_smoother_bytes(const pmc_sampler* s, int level, int nbatch, double* bytes) {
    return guarded([&] {
        PMC_REQUIRE(s != nullptr && bytes != nullptr, "pmc_sampler_smoother_bytes: NULL argument");
        *bytes = s->impl.smoother_bytes(level, nbatch);
    });
}

int pmc_sampler_vcycle_info(const pmc_sampler* s, int level, int vlevel, int* nvlevels, int64_t info[7]) {
    return guarded([&] {
        PMC_REQUIRE(s != nullptr && nvlevels != nullptr && info != nullptr, "pmc_sampler_vcycle_info: NULL argument");
        PMC_REQUIRE(level >= 0 && level < s->impl.n_mc, "pmc_sampler_vcycle_info: level out of range");
        const bool own = level < (int)s->impl.amg.size() && s->impl.amg[level];
        const Multigrid& mg = own ? *s->impl.amg[level] : s->impl.mg;
        const int first = own ? 0 : level;
        *nvlevels = (int)mg.L.size() - first;
        PMC_REQUIRE(vlevel >= 0 && vlevel < *nvlevels, "pmc_sampler_vcycle_info: vlevel out of range");
        const MgLevel& m = mg.L[(size_t)(first + vlevel)];
        bool in_tail = false;
        for (int l = first; l <= first + vlevel; ++l) in_tail = in_tail || (l < (int)mg.tail.size() && mg.tail[l].p != nullptr);
        info[0] = m.n;
        info[1] = m.S.nnz;
        info[2] = m.S.nslots;
        info[3] = m.has_sp ? m.SP.nnz : 0;
        info[4] = m.has_sp ? m.SP.nslots : 0;
        info[5] = (in_tail ? 1 : 0) | (m.dense_inv.p ? 2 : 0) | ((int64_t)m.split_log2 << 4);
        info[6] = (m.p_agg || m.p_oct) ? 1 : 0;
    });
}

int pmc_sampler_operator_event_overhead(pmc_sampler* s, double* total_ms) {
    return guarded([&] {
        PMC_REQUIRE(s != nullptr && total_ms != nullptr, "operator_event_overhead: bad arguments");
        *total_ms = s->impl.work.op_timer.gap_ms;
        s->impl.work.op_timer.gap_ms = 0.0;
    });
}

// ---- Darcy ------------------------------------------------------------------------------------
static int darcy_create(pmc_ctx* c, int nlevels, int n_mc_levels, const pmc_darcy_level* levels, int k_divides,
                        const pmc_solver_opts* opts, pmc_darcy** out, bool hybrid) {
    return guarded([&] {
        PMC_REQUIRE(c != nullptr && out != nullptr, "pmc_darcy_create: NULL argument");
        *out = nullptr;
        pmc_solver_opts o;
        pmc_solver_opts_default(&o);
        if (opts) {
            check_abi(*opts);
            o = *opts;
        }
        PMC_REQUIRE(o.max_iter >= 1 && o.cheb_degree_M >= 0 && o.mg_smooth_degree >= 1 && o.mg_coarse_degree >= 1 &&
                        (o.cheb_ratio_M <= 0.0 || o.cheb_ratio_M > 1.0) && o.mg_smooth_ratio > 1.0 && o.mg_coarse_ratio > 1.0 &&
                        (o.precond_storage == PMC_STORAGE_FP32 || o.precond_storage == PMC_STORAGE_FP64),
                    "solver options out of range");
        *out = new pmc_darcy(*c, nlevels, n_mc_levels, levels, k_divides != 0, o, hybrid);
    });
}
int pmc_darcy_create(pmc_ctx* c, int nlevels, int n_mc_levels, const pmc_darcy_level* levels, int k_divides,
                     const pmc_solver_opts* opts, pmc_darcy** out) {
    return darcy_create(c, nlevels, n_mc_levels, levels, k_divides, opts, out, false);
}
int pmc_darcy_create_hybrid(pmc_ctx* c, int nlevels, int n_mc_levels, const pmc_darcy_level* levels, int k_divides,
                            const pmc_solver_opts* opts, pmc_darcy** out) {
    return darcy_create(c, nlevels, n_mc_levels, levels, k_divides, opts, out, true);
}
void pmc_darcy_destroy(pmc_darcy* d) {
    if (!d) return;
    (void)hipSetDevice(d->impl.ctx.device);
    (void)hipStreamSynchronize(d->impl.ctx.stream);
    delete d;
}
int pmc_darcy_num_dofs(const pmc_darcy* d, int level) {
    if (!d || level < 0 || level >= d->impl.nlevels) return PMC_ERR_INVALID;
    return d->impl.lv[level].n_u + d->impl.lv[level].n_p;
}
int pmc_darcy_num_pressure_dofs(const pmc_darcy* d, int level) {
    if (!d || level < 0 || level >= d->impl.nlevels) return PMC_ERR_INVALID;
    return d->impl.lv[level].n_p;
}
int pmc_darcy_set_operator_timing(pmc_darcy* d, int on) {
    return guarded([&] {
        PMC_REQUIRE(d != nullptr, "darcy handle is NULL");
        d->impl.op_timer.on = on != 0;
        d->impl.poly_timer.on = on != 0;
    });
}
int pmc_darcy_operator_time(pmc_darcy* d, double* total_ms, int64_t* launches, double* event_overhead_ms) {
    return guarded([&] {
        PMC_REQUIRE(d != nullptr, "darcy handle is NULL");
        if (total_ms) *total_ms = d->impl.op_timer.ms;
        if (launches) *launches = d->impl.op_timer.launches;
        if (event_overhead_ms) *event_overhead_ms = d->impl.op_timer.gap_ms;
        d->impl.op_timer.clear();
    });
}
int pmc_darcy_operator_bytes(const pmc_darcy* d, int level, int nbatch, double* bytes) {
    return guarded([&] {
        PMC_REQUIRE(d != nullptr && bytes != nullptr && level >= 0 && level < d->impl.nlevels && valid_batch(nbatch),
                    "pmc_darcy_operator_bytes: bad arguments");
        *bytes = d->impl.operator_bytes(level, nbatch);
    });
}
int pmc_darcy_poly_time(pmc_darcy* d, double* total_ms, int64_t* launches, double* event_overhead_ms) {
    return guarded([&] {
        PMC_REQUIRE(d != nullptr, "darcy handle is NULL");
        if (total_ms) *total_ms = d->impl.poly_timer.ms;
        if (launches) *launches = d->impl.poly_timer.launches;
        if (event_overhead_ms) *event_overhead_ms = d->impl.poly_timer.gap_ms;
        d->impl.poly_timer.clear();
    });
}
int pmc_darcy_poly_bytes(const pmc_darcy* d, int level, int nbatch, double* bytes) {
    return guarded([&] {
        PMC_REQUIRE(d != nullptr && bytes != nullptr && level >= 0 && level < d->impl.nlevels && valid_batch(nbatch),
                    "pmc_darcy_poly_bytes: bad arguments");
        *bytes = d->impl.poly_bytes(level, nbatch);
    });
}
int pmc_darcy_batch_width(const pmc_darcy* d, int level) {
    if (!d || level < 0 || level >= d->impl.nlevels) return PMC_ERR_INVALID;
    return batch_width((size_t)d->impl.lv[level].n_u + d->impl.lv[level].n_p, true, d->impl.ctx.device);
}
int64_t pmc_darcy_nnz(const pmc_darcy* d, int level) {
    if (!d || level < 0 || level >= d->impl.nlevels) return PMC_ERR_INVALID;
    return d->impl.lv[level].nnz;
}
int pmc_darcy_solve_fwd(pmc_darcy* d, int level, int nbatch, const double* kf, double* Q, double* C, double* sol_out,
                        int memspace, pmc_stats* stats) {
    return guarded([&] {
        PMC_REQUIRE(d != nullptr, "darcy is NULL");
        d->impl.solve_fwd(level, nbatch, kf, Q, C, sol_out, memspace, stats);
    });
}

int pmc_darcy_solve_fwd_pressure(pmc_darcy* d, int level, int nbatch, const double* kf, double* p_out, double* C,
                                 double* Q, int compute_Q, int memspace, pmc_stats* stats) {
    return guarded([&] {
        PMC_REQUIRE(d != nullptr && p_out != nullptr, "SolveFwd_RtnPressure: NULL argument");
        PMC_REQUIRE(nbatch >= 1, "SolveFwd_RtnPressure: bad batch");
        std::vector<double> q(nbatch);
        d->impl.solve_fwd(level, nbatch, kf, q.data(), C, p_out, memspace, stats, 2);
        if (compute_Q && Q)
            for (int b = 0; b < nbatch; ++b) Q[b] = q[b];
    });
}

int pmc_darcy_set_observations(pmc_darcy* d, int level, const pmc_csr* Gobs) {
    return guarded([&] {
        PMC_REQUIRE(d != nullptr, "darcy is NULL");
        d->impl.set_observations(level, Gobs);
    });
}
int pmc_darcy_num_observations(const pmc_darcy* d, int level) {
    if (!d || level < 0 || level >= d->impl.nlevels) return PMC_ERR_INVALID;
    return d->impl.lv[level].n_gobs;
}
int pmc_darcy_compute_G(pmc_darcy* d, int level, int nbatch, const double* kf, double* G, double* C, double* Q,
                        int memspace, pmc_stats* stats) {
    return guarded([&] {
        PMC_REQUIRE(d != nullptr, "darcy is NULL");
        d->impl.compute_G(level, nbatch, kf, G, C, Q, memspace, stats);
    });
}

// ---- communicator -----------------------------------------------------------------------------
int pmc_comm_unique_id(void* id128) {
    return guarded([&] {
        PMC_REQUIRE(id128 != nullptr, "id buffer is NULL");
        UniqueId id;
        rccl_check(rccl().GetUniqueId(&id), "ncclGetUniqueId");
        std::memcpy(id128, &id, sizeof(id));
    });
}
int pmc_comm_init(pmc_ctx* c, const void* id128, int nranks, int rank) {
    return guarded([&] {
        PMC_REQUIRE(c != nullptr && id128 != nullptr, "pmc_comm_init: NULL argument");
        PMC_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "pmc_comm_init: bad rank");
        c->activate();
        UniqueId id;
        std::memcpy(&id, id128, sizeof(id));
        rccl_check(rccl().CommInitRank(&c->nccl, nranks, id, rank), "ncclCommInitRank");
        c->nranks = nranks;
        c->rank = rank;
    });
}
int pmc_comm_destroy(pmc_ctx* c) {
    return guarded([&] {
        PMC_REQUIRE(c != nullptr, "ctx is NULL");
        if (c->nccl) {
            c->activate();
            rccl_check(rccl().CommDestroy(c->nccl), "ncclCommDestroy");
            c->nccl = nullptr;
        }
        c->nranks = 1;
        c->rank = 0;
    });
}
int pmc_allreduce_sum_f64(pmc_ctx* c, double* host_buf, int n) {
    return guarded([&] {
        PMC_REQUIRE(c != nullptr && (host_buf != nullptr || n == 0) && n >= 0, "pmc_allreduce_sum_f64: bad argument");
        if (n == 0) return;
        if (c->nccl == nullptr) {
            PMC_REQUIRE(c->nranks == 1, "pmc_allreduce_sum_f64: communicator not initialised");
            return;   // single rank without a communicator: the sum is the buffer itself
        }
        c->activate();
        c->comm_buf.ensure((size_t)n);
        PMC_HIP(hipMemcpyAsync(c->comm_buf.p, host_buf, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
        rccl_check(rccl().AllReduce(c->comm_buf.p, c->comm_buf.p, (size_t)n, /*ncclDouble*/ 8, /*ncclSum*/ 0, c->nccl,
                                    c->stream),
                   "ncclAllReduce");
        PMC_HIP(hipMemcpyAsync(host_buf, c->comm_buf.p, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
        PMC_HIP(hipStreamSynchronize(c->stream));
    });
}

}  // extern "C"
