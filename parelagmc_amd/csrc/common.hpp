// Internal C++ declarations shared by the translation units of libpmc.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/pmc.h"

namespace pmc {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define PMC_HIP(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            throw ::pmc::Error(PMC_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_) + \
                                                   " (" + __FILE__ + ":" + std::to_string(__LINE__) + ")"); \
    } while (0)

#define PMC_REQUIRE(cond, msg)                                                                 \
    do {                                                                                       \
        if (!(cond)) throw ::pmc::Error(PMC_ERR_INVALID, std::string(msg) + " [" #cond "]");   \
    } while (0)

void set_last_error(const std::string& msg);

// ---- device memory ------------------------------------------------------------------------
template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    explicit DevBuf(size_t n_) { alloc(n_); }
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept {
        if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; }
        return *this;
    }
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    void alloc(size_t n_) {
        release();
        n = n_;
        if (n) PMC_HIP(hipMalloc(reinterpret_cast<void**>(&p), n * sizeof(T)));
    }
    void ensure(size_t n_) {
        if (n_ > n) alloc(n_);
    }
    void upload(const T* h, size_t cnt, hipStream_t st) {
        ensure(cnt);
        if (cnt) PMC_HIP(hipMemcpyAsync(p, h, cnt * sizeof(T), hipMemcpyHostToDevice, st));
    }
    void upload(const std::vector<T>& h, hipStream_t st) { upload(h.data(), h.size(), st); }
    void zero(hipStream_t st) {
        if (n) PMC_HIP(hipMemsetAsync(p, 0, n * sizeof(T), st));
    }
};

// ---- host CSR helper ----------------------------------------------------------------------
struct HostCsr {
    int nrows = 0, ncols = 0;
    std::vector<int> rowptr, colind;
    std::vector<double> vals;
    int64_t nnz() const { return (int64_t)colind.size(); }
};
HostCsr csr_from_c(const pmc_csr& c, bool need_vals, const char* what);
HostCsr csr_transpose(const HostCsr& a);
void csr_sort_rows(HostCsr& a);
std::vector<double> csr_diag(const HostCsr& a);
// vertical/horizontal assembly of the saddle-point operator [M Bt; B D] (D diagonal or absent)
HostCsr csr_block2x2(const HostCsr& M, const HostCsr& Bt, const HostCsr& B, const double* d11);

// algebraic coarsening helpers (setup): piecewise-constant aggregation
HostCsr csr_galerkin_agg(const HostCsr& A, const std::vector<int>& agg, int nc);
int pairwise_match(const HostCsr& A, double theta, std::vector<int>& agg, bool allow_weak = false, bool join_singletons = false);
int aggregate_rows(const HostCsr& K, int passes, double theta, std::vector<int>& agg, bool allow_weak = false,
                   bool join_singletons = false);
HostCsr prolongator_from_agg(const std::vector<int>& agg, int nc);
// renumbering that makes every aggregate of the indicator prolongator P a run of consecutive rows inside one 64-row slice
// (new2old; empty = not possible) + per slice the segments [seg_ptr[s], seg_ptr[s + 1]): coarse id, (first row in slice << 8) | rows
std::vector<int> agg_pack_rows(const HostCsr& P, std::vector<int>& seg_ptr, std::vector<int>& seg_cid, std::vector<int>& seg_pos);
HostCsr csr_permute(const HostCsr& A, const std::vector<int>& new2old, bool rows, bool cols);
HostCsr csr_split_rows(const HostCsr& A, int sl);

// Element-local inverses behind both hybridized systems (hybrid_build.hip): per element (row of B, which lists its faces with
// the sign of the outward normal against the face's global one) [[X, y], [y^T, z]] = [[M_e, b_e^T], [b_e, corner_e]]^-1 with
// M_e recovered from the contribution lists of the u-mass matrix (pmc_darcy_level's format); X and y carry the signs
// C_e = sign(B[e, f]): X = C X C^T (m x m per element, m = most faces of an element), Y = C y.  corner == nullptr: zeros.
struct ElementInverses {
    int m = 0;
    std::vector<double> X, Y, z;
};
ElementInverses element_inverses(const HostCsr& Mp, const HostCsr& B, const int32_t* c_ptr, const int32_t* c_elem,
                                 const double* c_val, const double* corner, const char* who);

HostCsr csr_spgemm(const HostCsr& A, const HostCsr& B);
// P(i, i / 8) == 1 is the only entry of row i, for every row, and P has 8 rows per column
bool csr_is_oct_injection(const HostCsr& P);
double csr_anisotropy(const HostCsr& K);
// lambda_min(D^-1 M) estimate (host Lanczos) and the M-block Chebyshev ratio derived from it
double lanczos_lambda_min_scaled(const HostCsr& M, const std::vector<double>& dinv, int steps);
double mass_block_ratio(const HostCsr& M, const std::vector<double>& l1inv);

struct AmgLevelHost {
    HostCsr S;   // operator of this level
    HostCsr P;   // prolongator from the next coarser level (empty on the last level)
};
std::vector<AmgLevelHost> sa_hierarchy(const HostCsr& K0, const std::vector<double>& w0, int passes, double theta,
                                       int min_size, int max_levels);
// plain aggregation (indicator prolongators), matching by coupling magnitude: SPD operators with couplings of either sign
// (passes0 matching passes on the finest level, `passes` below it)
std::vector<AmgLevelHost> agg_hierarchy(const HostCsr& A0, int passes0, int passes, double theta, int min_size, int max_levels);

// ---- SELL-64 device matrix ---------------------------------------------------------------
// Rows are grouped in slices of 64 (one wavefront); inside a slice entries are stored
// column-major: slot(s, j, lane) = slice_off[s] + j*64 + lane, so a wavefront's loads of
// values / column indices are fully coalesced 512 B / 256 B segments.  Padding slots carry the
// row's own index with value 0.  `src` (optional) maps a slot to the CSR nonzero it came from.
struct Sell {
    int nrows = 0, ncols = 0, nslices = 0;
    int64_t nnz = 0;       // algorithmic (CSR) nonzeros
    int64_t nslots = 0;    // stored incl. padding
    // built with every row's diagonal entry stored LAST (and padded with zero-weight copies of it): the last slice column
    // of every row then gathers x[row] - see sell_row_range's xlast
    bool diag_last = false;
    DevBuf<int> slice_off; // nslices+1
    DevBuf<int> cols;      // nslots
    DevBuf<double> vals;   // nslots (shared values) - empty for batched-value matrices
    // Optional processing order of the slices (a permutation of 0..nslices-1).  For a block operator
    // [u-rows; s-rows] it interleaves the u- and s-slices of the same mesh region, so the x entries both
    // kinds of rows gather are fetched once while they are still in the XCD's L2.
    DevBuf<int> sched;
    std::vector<int> h_slice_off, h_cols, h_src;  // host mirrors (h_src: slot -> csr nnz or -1)
    std::vector<int> h_sched;                     // host mirror of sched (empty = natural order)
    // algorithmic bytes of one SpMV with this matrix: 12 nnz + 12 nrows + 8 ncols
    double spmv_bytes() const { return 12.0 * nnz + 12.0 * nrows + 8.0 * ncols; }
};
// diag_last: request the diagonal-last order (granted when A is square and stores its whole diagonal)
void sell_build(Sell& S, const HostCsr& A, bool upload_vals, bool keep_src, hipStream_t st, bool diag_last = false);
// values of A*diag(colscale) laid out on the SELL pattern S was built with (S must keep its host mirrors)
std::vector<double> sell_scaled_values(const Sell& S, const HostCsr& A, const std::vector<double>& colscale);
// schedule that merges the slices of row block [0, n0) with those of [n0, nrows): every second-block slice right behind
// the first-block slices its rows reference (A given) or by relative position (A == nullptr)
void sell_schedule_two_blocks(Sell& S, int n0, hipStream_t st, const HostCsr* A = nullptr);

// ---- context ------------------------------------------------------------------------------
// The two HIP streams one solve runs on.  Independent kernel chains (the two diagonal blocks of the preconditioner) are
// issued on `main` and `aux` between fork() and join(): events only, no host synchronisation, capturable in a hipGraph.
// split == false: aux == main and fork / join do nothing (small levels, where the extra event calls cost more than the
// overlap returns).
struct Lanes {
    hipStream_t main = nullptr, aux = nullptr;
    hipEvent_t ef = nullptr, ej = nullptr;
    bool split = false;
    void fork() const {
        if (!split) return;
        PMC_HIP(hipEventRecord(ef, main));
        PMC_HIP(hipStreamWaitEvent(aux, ef, 0));
    }
    void join() const {
        if (!split) return;
        PMC_HIP(hipEventRecord(ej, aux));
        PMC_HIP(hipStreamWaitEvent(main, ej, 0));
    }
    hipStream_t side() const { return split ? aux : main; }
};

struct Ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;   // second stream of the same handle (see Lanes)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // brackets of one launch group of a solve: [0] start, [1] right-hand side / per-realization operators done, [2] Krylov
    // solve done (pmc_stats.setup_ms / solve_ms)
    hipEvent_t ev_phase[3] = {nullptr, nullptr, nullptr};
    void phase_mark(int i) const;
    // fills setup_ms / solve_ms of the nb stats entries from the three marks (synchronises on the last one)
    void phase_report(pmc_stats* stats, int nb) const;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // split: spread a solve over two streams.  The second stream is created on first use, and only a handle that is the
    // only one on its device ever asks for it (contexts_on_device): every additional HIP stream changes how the runtime
    // maps streams to hardware queues, which cost multi-lane runs up to 30 % when each lane carried an idle second stream.
    Lanes lanes(bool split);
    static int contexts_on_device(int device);
    uint64_t seed = 0;
    // NormalDistributionSampler::Split: part `mypart` of `nparts` owns the realizations mypart, mypart + nparts, ...;
    // realization id i of this handle is the generator's realization i * nparts + mypart (see stream_id)
    int nparts = 1, mypart = 0;
    uint64_t stream_id(uint64_t local_id) const { return local_id * (uint64_t)nparts + (uint64_t)mypart; }
    void* nccl = nullptr;          // ncclComm_t
    int nranks = 1, rank = 0;
    DevBuf<double> comm_buf;
    int* h_flag = nullptr;         // pinned host word for convergence polls
    double* h_scal = nullptr;      // pinned host scratch (kHostScratch doubles)
    static constexpr size_t kHostScratch = 16384;
    void activate() const;
};

// Tuning overrides from the environment exist only in LABORATORY builds of the library (make lab-lib: -DPMC_LAB,
// scripts/lab/): the product library reads no PMC_* variable except PMC_VERBOSE (setup report on stderr), and every default
// below is what the product runs with.  What a caller may legitimately choose is in pmc_solver_opts.
inline const char* lab_env(const char* name) {
#ifdef PMC_LAB
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

// Interleaved batch width of a level with `rows` unknowns (saddle-point system).  Large levels are bandwidth-bound at 16
// realizations per launch; sampler levels up to 5 M rows (PMC_S_WIDE_ROWS) and Darcy levels small enough to be bound by
// launch latency take 32 (PMC_WIDE_ROWS: limit, default 300 000 rows; set, it applies to both kinds, 0 = always 16), and the smaller the level the more column groups of 32 one launch carries - 64, 128 or 256
// realizations (PMC_W64_ROWS / PMC_W128_ROWS / PMC_W256_ROWS: limits, 0 = never) - until a launch fills the chip.
// device: the handle's device (the 32-wide limit of sampler levels comes from ITS memory)
int batch_width(size_t rows, bool darcy, int device);
static constexpr int kGroup = 32;      // widest compile-time interleave (Lay<32>); wider batches are column groups of it
static constexpr int kMaxBatch = 256;  // widest batch of one launch: 8 column groups
inline bool valid_batch(int nb) {
    return nb == 1 || nb == 2 || nb == 4 || nb == 8 || nb == 16 || nb == 32 || nb == 64 || nb == 128 || nb == 256;
}

}  // namespace pmc

struct pmc_ctx : pmc::Ctx {};
