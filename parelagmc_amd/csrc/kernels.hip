// CDNA4 (gfx950) kernels of the ParELAGMC hot path.  fp64 values, int32 indices, HBM-bound:
// no MFMA (<= 0.17 flop/byte), 64-wide wavefronts, SELL-64 operators so that every matrix load
// of a wavefront is one contiguous 512 B (values) / 256 B (columns) segment.
//
// Kernel <-> reference operation:
//   sell_spmm / sell_residual / sell_cheb_step : K5 block saddle-point SpMV, K7 smoother on M,
//        K8 V-cycle smoothing/residual/transfer, K3/K4 restriction/prolongation, K11 projection
//   minres_* / lincomb3 / dot                  : K6 MINRES vector operations
//   normal_fill                                : K1 NormalDistributionSampler
//   interleave / deinterleave                  : K2 white-noise RHS scaling, K9 exp, K10 gather (fused with the layout change)
//   darcy_*                                    : K12-K15 per-sample M(k), BC elimination, Schur refresh, QoI
#include "kernels.hpp"

#include <cstddef>
#include <cstdlib>

#include <algorithm>
#include <atomic>

namespace pmc {

#ifndef PMC_KBLOCK
#define PMC_KBLOCK 256
#endif
static constexpr int kBlock = PMC_KBLOCK;   // workgroup size of the streaming / SpMM kernels (tuning builds may override)
static constexpr int kWave = 64;
#ifndef PMC_LEAN_GATHER
#define PMC_LEAN_GATHER 1
#endif
static constexpr bool kLeanGather = PMC_LEAN_GATHER != 0;   // see sell_row_part: fp32 gathers without column scaling
#ifndef PMC_LEAN_CS
#define PMC_LEAN_CS 1
#endif
static constexpr bool kLeanCs = PMC_LEAN_CS != 0;           // ... fp64 gathers / column scaling (eg_poly2_kernel)
#ifndef PMC_EG_LATE_COEF
#define PMC_EG_LATE_COEF 1
#endif
static constexpr bool kEgLateCoef = PMC_EG_LATE_COEF != 0;  // see eg_row_product
#ifndef PMC_LEAN_RANGE_MIN_NB
#define PMC_LEAN_RANGE_MIN_NB 32
#endif
// sell_row_range takes the lean gather loop (see sell_row_part) from this batch width on: at NB = 32 the one-column loop
// needs 170-184 registers (two waves per SIMD), the lean one fits three; at NB = 16 (four waves either way) it changed nothing
static constexpr int kLeanRangeMinNb = PMC_LEAN_RANGE_MIN_NB;

static unsigned dot_grid_bound();
int dot_capacity(int nrows, int nb) {
    // upper bound on the partial blocks (of nb doubles each) any fused dot over nrows rows of a batch of nb writes: slice
    // kernels one block per 4 slices, flat kernels one per kBlock threads, both bounded by dot_grid_bound() - except the
    // block operator's two-stage reduction, which keeps every slice block behind kCompressBlocks compressed ones (k::spmm)
    const size_t slice_blocks = ((size_t)nrows + 63) / 64 / (kBlock / kWave) + 1;
    const size_t flat_blocks = ((size_t)nrows * nb / (nb >= 32 ? 4 : (nb >= 2 ? 2 : 1)) + kBlock - 1) / kBlock;
    const size_t bounded = std::min<size_t>(std::max(slice_blocks, flat_blocks), dot_grid_bound());
    return (int)std::max<size_t>(bounded, 256 + slice_blocks) + 2;
}

// Batch layout helper.  A row of NB interleaved values is handled by T lanes, C = 2 doubles (one 16 B
// access) each, so a group of T lanes touches NB*8 contiguous bytes and a wavefront G = 64/T rows.
template <int NB>
struct Lay {
    static constexpr int C = NB >= 32 ? 4 : (NB >= 2 ? 2 : 1);   // NB = 32: two 16 B accesses per lane, still 8 lanes per row
    static constexpr int T = NB / C;
    static constexpr int G = kWave / T;
};

// Column groups (super-batches).  A batch wider than kGroup realizations is ONE interleaved vector with row stride
// ld = nb doubles, worked on as nb / kGroup groups of kGroup columns: group g owns the columns [g kGroup, (g + 1) kGroup) of
// every row, blockIdx.y names the group, and every launch carries all groups - a level too small to fill the chip with 32
// realizations is solved 64 ... 256 at a time.  Only the widest instantiation (NB == kGroup) is group-capable; the
// narrower ones keep the compile-time row stride NB (ld is ignored, gridDim.y == 1).
template <int NB>
__device__ __forceinline__ int row_ld(int ld) {
    if constexpr (NB == kGroup) return ld;
    else return NB;
}
template <int NB>
__device__ __forceinline__ int col0() {
    if constexpr (NB == kGroup) return (int)blockIdx.y * NB;
    else return 0;
}

template <int C>
__device__ __forceinline__ void load_c(const double* __restrict__ p, double (&v)[C]) {
    if constexpr (C == 1) {
        v[0] = p[0];
    } else {
#pragma unroll
        for (int i = 0; i < C / 2; ++i) {
            const double2 t = reinterpret_cast<const double2*>(p)[i];
            v[2 * i] = t.x;
            v[2 * i + 1] = t.y;
        }
    }
}
template <int C>
__device__ __forceinline__ void store_c(double* __restrict__ p, const double (&v)[C]) {
    if constexpr (C == 1) {
        p[0] = v[0];
    } else {
#pragma unroll
        for (int i = 0; i < C / 2; ++i) reinterpret_cast<double2*>(p)[i] = make_double2(v[2 * i], v[2 * i + 1]);
    }
}

// Per-realization matrix values of the PRECONDITIONER (Darcy: the Schur-complement hierarchy S(k)) may be stored in fp32
// (BV == 2; BV == 1: fp64): the preconditioner stays a fixed symmetric linear operator - MINRES converges to the same
// solution at the same tolerance - while the dominant stream of its kernels halves.  Arithmetic stays fp64.
template <int C>
__device__ __forceinline__ void load_cf(const float* __restrict__ p, double (&v)[C]) {
    if constexpr (C == 1) {
        v[0] = (double)p[0];
    } else if constexpr (C == 2) {
        const float2 t = *reinterpret_cast<const float2*>(p);
        v[0] = (double)t.x;
        v[1] = (double)t.y;
    } else {
#pragma unroll
        for (int i = 0; i < C / 4; ++i) {
            const float4 t = reinterpret_cast<const float4*>(p)[i];
            v[4 * i] = (double)t.x;
            v[4 * i + 1] = (double)t.y;
            v[4 * i + 2] = (double)t.z;
            v[4 * i + 3] = (double)t.w;
        }
    }
}
// values of batched matrix entry `idx` (in units of one value): fp64 or fp32 storage
template <int BV, int C>
__device__ __forceinline__ void load_bv(const double* __restrict__ vals, size_t idx, double (&v)[C]) {
    if constexpr (BV == 2) load_cf<C>(reinterpret_cast<const float*>(vals) + idx, v);
    else load_c<C>(vals + idx, v);
}
// advance a batched-value pointer by c columns
template <int BV>
__device__ __forceinline__ const double* shift_bv(const double* vals, int c) {
    if constexpr (BV == 2) return reinterpret_cast<const double*>(reinterpret_cast<const float*>(vals) + c);
    else return vals + c;
}

// typed vector accesses: fp64 or fp32 storage, fp64 in registers
template <typename T>
struct ident { using type = T; };   // keeps a parameter out of template argument deduction (nullptr arguments)
template <int C>
__device__ __forceinline__ void load_v(const double* __restrict__ p, double (&v)[C]) { load_c<C>(p, v); }
template <int C>
__device__ __forceinline__ void load_v(const float* __restrict__ p, double (&v)[C]) { load_cf<C>(p, v); }
template <int C>
__device__ __forceinline__ void store_v(double* __restrict__ p, const double (&v)[C]) { store_c<C>(p, v); }
template <int C>
__device__ __forceinline__ void store_v(float* __restrict__ p, const double (&v)[C]) {
    if constexpr (C == 1) {
        p[0] = (float)v[0];
    } else if constexpr (C == 2) {
        *reinterpret_cast<float2*>(p) = make_float2((float)v[0], (float)v[1]);
    } else {
#pragma unroll
        for (int i = 0; i < C / 4; ++i)
            reinterpret_cast<float4*>(p)[i] = make_float4((float)v[4 * i], (float)v[4 * i + 1], (float)v[4 * i + 2], (float)v[4 * i + 3]);
    }
}
// values an fp32 store will keep: rounding BEFORE a fused dot keeps <., out> consistent with what is stored
template <typename T, int C>
__device__ __forceinline__ void round_to(double (&v)[C]) {
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int i = 0; i < C; ++i) v[i] = (double)(float)v[i];
    }
}

// What a gather leaves in registers until its FMA: fp32-stored vectors stay fp32 (half the registers per gather in flight)
// and are widened only when they are consumed.
template <typename XT, int C>
struct RawVec {
    XT v[C];
};
template <int C>
__device__ __forceinline__ void load_raw(const double* __restrict__ p, RawVec<double, C>& r) { load_c<C>(p, r.v); }
template <int C>
__device__ __forceinline__ void load_raw(const float* __restrict__ p, RawVec<float, C>& r) {
    if constexpr (C == 1) {
        r.v[0] = p[0];
    } else if constexpr (C == 2) {
        const float2 t = *reinterpret_cast<const float2*>(p);
        r.v[0] = t.x;
        r.v[1] = t.y;
    } else {
#pragma unroll
        for (int i = 0; i < C / 4; ++i) {
            const float4 t = reinterpret_cast<const float4*>(p)[i];
            r.v[4 * i] = t.x;
            r.v[4 * i + 1] = t.y;
            r.v[4 * i + 2] = t.z;
            r.v[4 * i + 3] = t.w;
        }
    }
}

// The T = 8 gathers of one slice column (fp32 rows, four columns per lane) as ONE point of use.  hipcc sinks loads from
// __restrict__ pointers past __builtin_amdgcn_sched_barrier (they carry no ordering against it) down to their first use; in
// the kernels whose gathered fp32 values die in their own FMA group that turned "eight gathers in flight" into load - wait -
// convert - fma, eight times per slice column (ISA of vc_residual_kernel<32, float, ...> and vc_poly2_kernel<32, float, ...>,
// round 4: a level of 44 k rows took 67-74 us per launch where the fp64-gather kernel on the same matrix took 30).  An empty
// asm that takes all eight results as operands cannot be split: every load is issued before it.
typedef float pmc_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void pin_gathers(RawVec<float, 4> (&xr)[8]) {
    pmc_f4 q0 = {xr[0].v[0], xr[0].v[1], xr[0].v[2], xr[0].v[3]}, q1 = {xr[1].v[0], xr[1].v[1], xr[1].v[2], xr[1].v[3]};
    pmc_f4 q2 = {xr[2].v[0], xr[2].v[1], xr[2].v[2], xr[2].v[3]}, q3 = {xr[3].v[0], xr[3].v[1], xr[3].v[2], xr[3].v[3]};
    pmc_f4 q4 = {xr[4].v[0], xr[4].v[1], xr[4].v[2], xr[4].v[3]}, q5 = {xr[5].v[0], xr[5].v[1], xr[5].v[2], xr[5].v[3]};
    pmc_f4 q6 = {xr[6].v[0], xr[6].v[1], xr[6].v[2], xr[6].v[3]}, q7 = {xr[7].v[0], xr[7].v[1], xr[7].v[2], xr[7].v[3]};
    asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7));
    const pmc_f4 q[8] = {q0, q1, q2, q3, q4, q5, q6, q7};
#pragma unroll
    for (int rs = 0; rs < 8; ++rs)
#pragma unroll
        for (int c = 0; c < 4; ++c) xr[rs].v[c] = q[rs][c];
}
template <typename XT, int C, int T>
__device__ __forceinline__ void pin_gathers(RawVec<XT, C> (&)[T]) {}
// the same for sixteen widened values (8 row steps x 2 columns or 4 x 4): the own-row reads of an epilogue, issued together
template <int R, int C>
__device__ __forceinline__ void pin_block(double (&a)[R][C]) {
    if constexpr (R == 8 && C == 2)
        asm volatile("" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[2][0]), "+v"(a[2][1]), "+v"(a[3][0]),
                     "+v"(a[3][1]), "+v"(a[4][0]), "+v"(a[4][1]), "+v"(a[5][0]), "+v"(a[5][1]), "+v"(a[6][0]), "+v"(a[6][1]),
                     "+v"(a[7][0]), "+v"(a[7][1]));
    else if constexpr (R == 4 && C == 4)
        asm volatile("" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[0][2]), "+v"(a[0][3]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[1][2]),
                     "+v"(a[1][3]), "+v"(a[2][0]), "+v"(a[2][1]), "+v"(a[2][2]), "+v"(a[2][3]), "+v"(a[3][0]), "+v"(a[3][1]),
                     "+v"(a[3][2]), "+v"(a[3][3]));
    else if constexpr (R == 2 && C == 4)
        asm volatile("" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[0][2]), "+v"(a[0][3]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[1][2]),
                     "+v"(a[1][3]));
}

// Vector streams without reuse inside the iteration (MINRES w / x updates of large levels): non-temporal variants, so that a
// flat kernel running beside a gather kernel (second stream, other lanes) does not sweep that kernel's rows out of L2.
// Measured at 0.6 M rows x 16: one lane 1096 -> 1112, four lanes 1446 -> 1454 samples/s; small levels keep the cached
// accesses (their vectors live in the caches from one iteration to the next).
template <bool NT, int C>
__device__ __forceinline__ void load_c_nt(const double* __restrict__ p, double (&v)[C]) {
    if constexpr (NT) {
#pragma unroll
        for (int i = 0; i < C; ++i) v[i] = __builtin_nontemporal_load(p + i);
    } else {
        load_c<C>(p, v);
    }
}
template <bool NT, int C>
__device__ __forceinline__ void load_v_nt(const double* __restrict__ p, double (&v)[C]) { load_c_nt<NT, C>(p, v); }
template <bool NT, int C>
__device__ __forceinline__ void load_v_nt(const float* __restrict__ p, double (&v)[C]) {
    if constexpr (NT) {
#pragma unroll
        for (int i = 0; i < C; ++i) v[i] = (double)__builtin_nontemporal_load(p + i);
    } else {
        load_cf<C>(p, v);
    }
}
template <bool NT, int C>
__device__ __forceinline__ void store_c_nt(double* __restrict__ p, const double (&v)[C]) {
    if constexpr (NT) {
#pragma unroll
        for (int i = 0; i < C; ++i) __builtin_nontemporal_store(v[i], p + i);
    } else {
        store_c<C>(p, v);
    }
}

// Streaming accesses (matrix values / indices read once, result rows written once) with NT = true are non-temporal, so
// that they do not displace the gathered x rows - the only data with reuse - from the XCD's L2.  Worth it only when the
// operands exceed the 256 MiB Infinity Cache, which non-temporal accesses bypass: measured on the block operator at
// 4.7 M rows (1.65 GB per launch) 460-475 -> 435-448 us inside the solver loop; at 0.6 M rows (206 MB, cache-resident when
// launched back to back) 43 -> 55 us, and no change inside the loop.  The launcher picks NT by operand size.
template <bool NT, int C>
__device__ __forceinline__ void store_c_stream(double* __restrict__ p, const double (&v)[C]) {
    if constexpr (NT) {
#pragma unroll
        for (int i = 0; i < C; ++i) __builtin_nontemporal_store(v[i], p + i);
    } else {
        store_c<C>(p, v);
    }
}
template <bool NT>
__device__ __forceinline__ int load_stream(const int* __restrict__ p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
template <bool NT>
__device__ __forceinline__ double load_stream(const double* __restrict__ p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}

template <bool NT, int C>
__device__ __forceinline__ void store_v_stream(double* __restrict__ p, const double (&v)[C]) { store_c_stream<NT, C>(p, v); }
template <bool NT, int C>
__device__ __forceinline__ void store_v_stream(float* __restrict__ p, const double (&v)[C]) {
    if constexpr (NT) {
#pragma unroll
        for (int i = 0; i < C; ++i) __builtin_nontemporal_store((float)v[i], p + i);
    } else {
        store_v<C>(p, v);
    }
}

template <int NB>
__device__ __forceinline__ void load_row(const double* __restrict__ p, double (&v)[NB]) {
    if constexpr (NB == 1) {
        v[0] = p[0];
    } else {
        const double2* q = reinterpret_cast<const double2*>(p);
#pragma unroll
        for (int i = 0; i < NB / 2; ++i) {
            double2 t = q[i];
            v[2 * i] = t.x;
            v[2 * i + 1] = t.y;
        }
    }
}
template <int NB>
__device__ __forceinline__ void store_row(double* __restrict__ p, const double (&v)[NB]) {
    if constexpr (NB == 1) {
        p[0] = v[0];
    } else {
        double2* q = reinterpret_cast<double2*>(p);
#pragma unroll
        for (int i = 0; i < NB / 2; ++i) q[i] = make_double2(v[2 * i], v[2 * i + 1]);
    }
}

// Column-wise block reduction.  Every lane holds partial sums p[0..C) for columns (lane % T)*C + c.
// Deterministic: fixed xor tree over the lanes that share a column, fixed order over the 4 wavefronts.
// Virtual block index / grid extent of the slice kernels.  A launch of several column groups (nb > kGroup) is laid out as
// dim3(8, groups, chunks) (groups_xcd): the hardware deals workgroups to the 8 XCDs by their linear id x + 8 y + 8 groups z, so
// the blocks (x, 0, z) and (x, 1, z) - the SAME slices for column group 0 and 1 - run on the same XCD right after each other and
// the second one finds the slices' (index, value) pairs in that XCD's L2 instead of reading the matrix from HBM once more.
// Ordinary launches are dim3(blocks, groups, 1): vblock() == blockIdx.x.
__device__ __forceinline__ int vblock() { return (int)(blockIdx.z * gridDim.x + blockIdx.x); }
__device__ __forceinline__ int vgrid() { return (int)(gridDim.z * gridDim.x); }

// Writes partial[vblock()*LD + k] (partial already points at the group's first column).
template <int NB>
__device__ __forceinline__ void reduce_cols_store(double (&p)[Lay<NB>::C], double* __restrict__ partial, int LD = NB) {
    constexpr int C = Lay<NB>::C, T = Lay<NB>::T;
    __shared__ double lds[kBlock / kWave][NB];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        double v = p[c];
#pragma unroll
        for (int off = kWave / 2; off >= T; off >>= 1) v += __shfl_xor(v, off, kWave);
        p[c] = v;
    }
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    if (lane < T) {
#pragma unroll
        for (int c = 0; c < C; ++c) lds[wave][lane * C + c] = p[c];
    }
    __syncthreads();
    if (threadIdx.x < NB) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < kBlock / kWave; ++w) s += lds[w][threadIdx.x];
        partial[(size_t)vblock() * LD + threadIdx.x] = s;
    }
}

// The same for the flat kernels on a batch of W = ld > kGroup columns (runtime width, W divides 4 * kBlock): thread t owns the
// columns (4 t) % W + c of the rows it visits; LDS holds every thread's four sums, thread j < W adds the 4 kBlock / W
// entries of column j in index order (deterministic).  Writes partial[blockIdx.x * W + j].
__device__ __forceinline__ void reduce_cols_store_wide(const double (&p)[4], double* __restrict__ partial, int W) {
    __shared__ double lds[kBlock * 4];
#pragma unroll
    for (int c = 0; c < 4; ++c) lds[threadIdx.x * 4 + c] = p[c];
    __syncthreads();
    if ((int)threadIdx.x < W) {
        double s = 0.0;
        for (int m = threadIdx.x; m < kBlock * 4; m += W) s += lds[m];
        partial[(size_t)blockIdx.x * W + threadIdx.x] = s;
    }
}
template <int NB>
__device__ __forceinline__ void reduce_flat_store(double (&p)[Lay<NB>::C], double* __restrict__ partial, int W) {
    if constexpr (NB == kGroup) {
        if (W > NB) {
            reduce_cols_store_wide(p, partial, W);
            return;
        }
    }
    reduce_cols_store<NB>(p, partial);
}

// ------------------------------------------------------------------------------------------
// SELL-64 sparse matrix times interleaved multi-vector.  One wavefront per 64-row slice.  Every lane
// loads the value / column of "its" row for slice column j (one fully coalesced 512 B + 256 B access
// per wavefront), then the wavefront sweeps the slice in T steps of G rows: lane (g, t) takes row
// rs*G+g and the 16 B column pair t, fetching that row's value / column index with a cross-lane
// shuffle, so each x gather and each y store is one contiguous NB*8-byte segment per row.
// sell_row_range works on `width` slice columns starting at slot `off`.  CS: every gathered x[col] is multiplied by a
// second gathered per-realization vector cs[col] (column scaling A D^-1 without stored scaled values).  ZERO: acc is
// cleared first, otherwise accumulated into.
// xlast (optional): receives the x rows gathered by the LAST slice column.  A matrix built diagonal-last (Sell::diag_last:
// every row ends with its diagonal entry and is padded with zero-weight copies of it) gathers x[row] there, so a fused
// <x, Ax> needs no second read of x - which by the end of a slice has long left the L2 (measured at 0.6 M rows: 25 MB of
// 280 MB per launch).
template <int NB>
__device__ __forceinline__ constexpr bool lean_range() {
    return Lay<NB>::T > 1 && NB >= kLeanRangeMinNb;
}
template <int NB, int BV, bool CS, bool ZERO, bool NT = false, typename XT = double>
__device__ __forceinline__ void sell_row_range(const int* __restrict__ cols, const double* __restrict__ vals,
                                               const XT* __restrict__ x, const double* __restrict__ cs, int off,
                                               int width, int lane, int LD, double (&acc)[Lay<NB>::T][Lay<NB>::C],
                                               double (*xlast)[Lay<NB>::C] = nullptr, double* pdot = nullptr) {
    constexpr int C = Lay<NB>::C, T = Lay<NB>::T, G = Lay<NB>::G;
    const int g = lane / T, t = lane % T;
    if constexpr (ZERO) {
#pragma unroll
        for (int rs = 0; rs < T; ++rs)
#pragma unroll
            for (int c = 0; c < C; ++c) acc[rs][c] = 0.0;
    }
    int slot = off + lane;

    if constexpr (T == 1) {
        // one lane per row (NB = 1, 2): take JU slice columns at a time so that JU (index, value) pairs and then
        // JU gathers are in flight together instead of one dependent chain per column
        constexpr int JU = 4;
        for (int j = 0; j < width; j += JU, slot += JU * kWave) {
            int cc[JU];
            double aa[JU];
#pragma unroll
            for (int u = 0; u < JU; ++u) {
                const bool ok = j + u < width;
                const int at = ok ? slot + u * kWave : slot;   // out-of-range columns re-read column j, weight 0
                cc[u] = cols[at];
                if constexpr (BV) aa[u] = ok ? 1.0 : 0.0;
                else aa[u] = ok ? vals[at] : 0.0;
            }
            __builtin_amdgcn_sched_barrier(0);
            double xv[JU][C], av[JU][C], sv[JU][C];
#pragma unroll
            for (int u = 0; u < JU; ++u) {
                load_v<C>(x + (size_t)cc[u] * LD, xv[u]);
                if constexpr (CS) load_c<C>(cs + (size_t)cc[u] * LD, sv[u]);
                if constexpr (BV) load_bv<BV, C>(vals, (size_t)(j + u < width ? slot + u * kWave : slot) * LD, av[u]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < JU; ++u)
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    if constexpr (CS) xv[u][c] *= sv[u][c];
                    if constexpr (BV) acc[0][c] = fma(aa[u] * av[u][c], xv[u][c], acc[0][c]);
                    else acc[0][c] = fma(aa[u], xv[u][c], acc[0][c]);
                }
        }
        return;
    }

    int cj = 0;
    double vj = 0.0;
    if (width > 0) {
        cj = load_stream<NT>(cols + slot);
        if constexpr (!BV) vj = load_stream<NT>(vals + slot);
    }
    if constexpr (lean_range<NB>()) {
        // lean loop (as sell_row_part): 32-bit element offsets, gathered rows and fp32 per-realization values stay in their
        // storage type until the FMA, shared values are fetched across lanes after the gathers have been issued; with pdot
        // the fused <x, A x> of a diagonal-last matrix is taken right at the row's last column (rows past the end carry
        // zero values)
        for (int j = 0; j < width; ++j, slot += kWave) {
            int cn = cj;
            double vn = vj;
            if (j + 1 < width) {
                cn = load_stream<NT>(cols + slot + kWave);
                if constexpr (!BV) vn = load_stream<NT>(vals + slot + kWave);
            }
            unsigned at[T];
#pragma unroll
            for (int rs = 0; rs < T; ++rs) at[rs] = (unsigned)__shfl(cj, rs * G + g, kWave) * (unsigned)LD + (unsigned)(t * C);
            RawVec<XT, C> xr[T];
            RawVec<float, C> avf[BV == 2 ? T : 1];
            double avd[BV == 1 ? T : 1][C];
            double sv[CS ? T : 1][C];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int rs = 0; rs < T; ++rs) {
                load_raw<C>(x + at[rs], xr[rs]);
                if constexpr (CS) load_c<C>(cs + at[rs], sv[rs]);
                if constexpr (BV == 2)
                    load_raw<C>(reinterpret_cast<const float*>(vals) + ((size_t)(slot - lane + rs * G + g) * LD + t * C), avf[rs]);
                if constexpr (BV == 1) load_c<C>(vals + ((size_t)(slot - lane + rs * G + g) * LD + t * C), avd[rs]);
            }
            if (!pdot && !xlast) pin_gathers(xr);   // (with pdot / xlast the values live past the FMAs and stay grouped)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int rs = 0; rs < T; ++rs) {
                double a = 0.0;
                if constexpr (!BV) a = __shfl(vj, rs * G + g, kWave);
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    double xv = (double)xr[rs].v[c];
                    if constexpr (CS) xv *= sv[rs][c];
                    if constexpr (BV == 2) acc[rs][c] = fma((double)avf[rs].v[c], xv, acc[rs][c]);
                    else if constexpr (BV == 1) acc[rs][c] = fma(avd[rs][c], xv, acc[rs][c]);
                    else acc[rs][c] = fma(a, xv, acc[rs][c]);
                }
            }
            if (j + 1 == width) {
                if (pdot) {
#pragma unroll
                    for (int rs = 0; rs < T; ++rs)
#pragma unroll
                        for (int c = 0; c < C; ++c) pdot[c] = fma((double)xr[rs].v[c], acc[rs][c], pdot[c]);
                } else if (xlast) {
#pragma unroll
                    for (int rs = 0; rs < T; ++rs)
#pragma unroll
                        for (int c = 0; c < C; ++c) xlast[rs][c] = (double)xr[rs].v[c];
                }
            }
            cj = cn;
            vj = vn;
        }
        return;
    }
    for (int j = 0; j < width; ++j, slot += kWave) {
        // software pipeline: the next slice column's (value, index) pair is requested before this
        // column's gathers, so its latency overlaps them
        int cn = cj;
        double vn = vj;
        if (j + 1 < width) {
            cn = load_stream<NT>(cols + slot + kWave);
            if constexpr (!BV) vn = load_stream<NT>(vals + slot + kWave);
        }
        // phase 1: all cross-lane fetches, phase 2: all gathers (independent registers, so the T loads of a
        // slice column are in flight together), phase 3: FMAs
        int cc[T];
        double aa[T];
#pragma unroll
        for (int rs = 0; rs < T; ++rs) {
            const int src = rs * G + g;
            cc[rs] = (T == 1) ? cj : __shfl(cj, src, kWave);
            if constexpr (!BV) aa[rs] = (T == 1) ? vj : __shfl(vj, src, kWave);
        }
        double xv[T][C];
        double av[T][C];
        double sv[T][C];
        if constexpr (T > 1) __builtin_amdgcn_sched_barrier(0);   // hipcc otherwise re-serialises load -> wait -> fma
#pragma unroll
        for (int rs = 0; rs < T; ++rs) {
            load_v<C>(x + (size_t)cc[rs] * LD + t * C, xv[rs]);
            if constexpr (CS) load_c<C>(cs + (size_t)cc[rs] * LD + t * C, sv[rs]);
            if constexpr (BV) load_bv<BV, C>(vals, (size_t)(slot - lane + rs * G + g) * LD + t * C, av[rs]);
        }
        if constexpr (T > 1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rs = 0; rs < T; ++rs) {
#pragma unroll
            for (int c = 0; c < C; ++c) {
                if constexpr (CS) xv[rs][c] *= sv[rs][c];
                if constexpr (BV) acc[rs][c] = fma(av[rs][c], xv[rs][c], acc[rs][c]);
                else acc[rs][c] = fma(aa[rs], xv[rs][c], acc[rs][c]);
            }
        }
        if (xlast && j + 1 == width) {
#pragma unroll
            for (int rs = 0; rs < T; ++rs)
#pragma unroll
                for (int c = 0; c < C; ++c) xlast[rs][c] = xv[rs][c];
        }
        cj = cn;
        vj = vn;
    }
}

template <int NB, int BV, typename XT = double>
__device__ __forceinline__ void sell_row_product(const int* __restrict__ slice_off, const int* __restrict__ cols,
                                                 const double* __restrict__ vals, const XT* __restrict__ x,
                                                 int slice, int lane, int LD, double (&acc)[Lay<NB>::T][Lay<NB>::C]) {
    const int off = slice_off[slice];
    const int width = (slice_off[slice + 1] - off) >> 6;
    sell_row_range<NB, BV, false, true, false, XT>(cols, vals, x, nullptr, off, width, lane, LD, acc);
}

// XCD-aware slice assignment.  The dispatcher deals workgroups round-robin over the 8 XCDs (block b runs on XCD b % 8),
// and each XCD has its own 4 MiB L2.  XCD x owns one CONTIGUOUS eighth of the slices (in processing order), so the x
// entries its gathers touch (mesh neighbours = nearby indices) stay in that XCD's L2 instead of being fetched by all
// eight.  Inside the eighth the slices are dealt CYCLICALLY over the XCD's workgroups (block i of the XCD takes the
// slices 4 i .. 4 i + 3, then those one full round of workgroups further on, ...): whatever the grid size, the slices
// in flight on an XCD at any time form one compact window of the rows.  (With one contiguous chunk per workgroup - the
// round-1 layout - a bounded grid of 4096 workgroups at 4.7 M rows had concurrently running workgroups 19 slices apart:
// the window of x rows in flight was 5x wider than the L2 and x was fetched 3 times, 2.87 GB per launch against
// 1.65 GB algorithmic.)  Placement only affects speed, never results.
struct SliceWalk {
    int begin, end, stride;
};
__device__ __forceinline__ SliceWalk slice_walk(int nslices) {
    constexpr int WPB = kBlock / kWave;                     // wavefronts = slices per workgroup and round
    const int nblk = vgrid(), bid = vblock(), wave = threadIdx.x / kWave;
    if (nblk < 8) return SliceWalk{bid * WPB + wave, nslices, nblk * WPB};
    const int xcd = bid % 8, idx = bid / 8;
    const int nb_x = nblk / 8 + (xcd < nblk % 8 ? 1 : 0);   // workgroups of this XCD
    const int per = (nslices + 7) / 8;                      // slices of an XCD (the last one may get fewer)
    const int lo = min(xcd * per, nslices), hi = min(lo + per, nslices);
    return SliceWalk{lo + idx * WPB + wave, hi, nb_x * WPB};
}

// MODE 0: y = Ax   1: y += Ax   2: y = r - Ax ; DOT: partial sums of <dot_with, result>.
// Wavefronts stride over the slices (grid may be smaller than the slice count: bounded partial-sum count).
// TAG only names the instantiation: 1 = the block saddle-point operator (K5) inside the solver, 2 = the same operator
// launched by pmc_sampler_apply_operator (the isolated roofline measurement), so that profiles show the
// hot operator's launches on their own row; 0 = every other matrix (transfers, residuals, ...).
// R8 (with MODE 2, no DOT): the rows of the result are also summed in groups of 8 consecutive rows into
// partial[(row / 8) * NB + k] - the restriction P^T res of a prolongator whose parent i has exactly the children
// 8 i .. 8 i + 7 with unit weights (uniformly refined tetrahedra / hexahedra): a slice holds 8 whole groups, the sum is
// a fixed xor tree over the lanes of a row step, and the separate restriction kernel (13 us of dependent latency for a
// few MB) disappears from the V-cycle.
// XT: storage type of x and dot_with (fp32 or fp64 for the preconditioned Krylov vectors, zvec)
template <int NB, int BV, int MODE, bool DOT, int TAG, bool NT = false, bool R8 = false, bool DL = false, typename XT = double>
__global__ __launch_bounds__(kBlock, (NB >= 32 && BV == 0 && sizeof(XT) == 4 ? 3 : 1)) void sell_spmm_kernel(int nrows, int nslices, const int* __restrict__ slice_off,
                                                           const int* __restrict__ sched,
                                                           const int* __restrict__ cols,
                                                           const double* __restrict__ vals,
                                                           const XT* __restrict__ x, double* __restrict__ y,
                                                           const double* __restrict__ r,
                                                           const typename ident<XT>::type* __restrict__ dot_with,
                                                           double* __restrict__ partial, int ld) {
    static_assert(!R8 || (MODE == 2 && !DOT), "fused restriction goes with the residual");
    const int LD = row_ld<NB>(ld);
    {
        const int c0 = col0<NB>();   // this group's columns of every interleaved operand
        x += c0;
        y += c0;
        if constexpr (BV) vals = shift_bv<BV>(vals, c0);
        if constexpr (MODE == 2) r += c0;
        if constexpr (DOT && !DL) dot_with += c0;
        if constexpr (DOT || R8) partial += c0;
    }
    static_assert(!DL || (DOT && !BV && Lay<NB>::T > 1), "diagonal-last serves the fused <x, Ax> of shared-value operators");
    constexpr int C = Lay<NB>::C, T = Lay<NB>::T, G = Lay<NB>::G;
    const int lane = threadIdx.x & (kWave - 1);
    const int g = lane / T, t = lane % T;
    double p[C];
#pragma unroll
    for (int c = 0; c < C; ++c) p[c] = 0.0;
    const SliceWalk sw = slice_walk(nslices);
    for (int si = sw.begin; si < sw.end; si += sw.stride) {
        const int slice = sched ? sched[si] : si;   // optional processing order (locality), see Sell::sched
        double acc[T][C];
        constexpr bool LEAN_DL = DL && lean_range<NB>();   // the dot is taken inside the gather loop
        double xd[DL && !LEAN_DL ? T : 1][C];
        if constexpr (LEAN_DL) {
            const int off = slice_off[slice];
            sell_row_range<NB, false, false, true, NT, XT>(cols, vals, x, nullptr, off, (slice_off[slice + 1] - off) >> 6,
                                                             lane, LD, acc, nullptr, p);
        } else if constexpr (DL) {
            const int off = slice_off[slice];
            sell_row_range<NB, false, false, true, NT, XT>(cols, vals, x, nullptr, off, (slice_off[slice + 1] - off) >> 6,
                                                             lane, LD, acc, xd);
        } else if constexpr (NT) {
            const int off = slice_off[slice];
            sell_row_range<NB, BV, false, true, true, XT>(cols, vals, x, nullptr, off, (slice_off[slice + 1] - off) >> 6, lane, LD, acc);
        } else {
            sell_row_product<NB, BV, XT>(slice_off, cols, vals, x, slice, lane, LD, acc);
        }
#pragma unroll
        for (int rs = 0; rs < T; ++rs) {
            const int row = slice * kWave + rs * G + g;
            if (row < nrows) {
                const size_t at = (size_t)row * LD + t * C;
                if constexpr (MODE == 1) {
                    double old[C];
                    load_c<C>(y + at, old);
#pragma unroll
                    for (int c = 0; c < C; ++c) acc[rs][c] += old[c];
                } else if constexpr (MODE == 2) {
                    double rv[C];
                    load_c<C>(r + at, rv);
#pragma unroll
                    for (int c = 0; c < C; ++c) acc[rs][c] = rv[c] - acc[rs][c];
                }
                store_c_stream<NT, C>(y + at, acc[rs]);
                if constexpr (LEAN_DL) {
                } else if constexpr (DL) {
#pragma unroll
                    for (int c = 0; c < C; ++c) p[c] = fma(xd[rs][c], acc[rs][c], p[c]);
                } else if constexpr (DOT) {
                    double w[C];
                    load_v<C>(dot_with + at, w);
#pragma unroll
                    for (int c = 0; c < C; ++c) p[c] = fma(w[c], acc[rs][c], p[c]);
                }
            } else if constexpr (R8) {
#pragma unroll
                for (int c = 0; c < C; ++c) acc[rs][c] = 0.0;     // rows past the end add nothing to their group
            }
            if constexpr (R8) {
                // lanes (g, t): the 8 rows of a group differ in the low three bits of g = lane / T
                double s[C];
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    double v = acc[rs][c];
                    v += __shfl_xor(v, T, kWave);
                    v += __shfl_xor(v, 2 * T, kWave);
                    v += __shfl_xor(v, 4 * T, kWave);
                    s[c] = v;
                }
                if ((g & 7) == 0 && row < nrows) store_c<C>(partial + (size_t)(row >> 3) * LD + t * C, s);
            }
        }
    }
    if constexpr (DOT) reduce_cols_store<NB>(p, partial, LD);
}

// Chebyshev / Jacobi step: d = a d + b dinv (r - A xin); xout = xin + d ; DOT: partials of <r, xout>
// OT != double: the LAST step of a polynomial whose result is a preconditioned Krylov vector (zvec storage): d is not
// written back, the iterate is rounded to its storage before the fused dot
template <int NB, int BV, bool DOT, typename OT = double>
__global__ __launch_bounds__(kBlock) void sell_cheb_kernel(int nrows, int nslices, const int* __restrict__ slice_off,
                                                           const int* __restrict__ sched,
                                                           const int* __restrict__ cols,
                                                           const double* __restrict__ vals,
                                                           const double* __restrict__ dinv,
                                                           const double* __restrict__ r,
                                                           const double* __restrict__ xin, double* __restrict__ d,
                                                           OT* __restrict__ xout, double a, double b,
                                                           double* __restrict__ partial, int ld) {
    constexpr int C = Lay<NB>::C, T = Lay<NB>::T, G = Lay<NB>::G;
    const int LD = row_ld<NB>(ld);
    {
        const int c0 = col0<NB>();
        r += c0; xin += c0; d += c0; xout += c0;
        if constexpr (BV) { vals = shift_bv<BV>(vals, c0); dinv += c0; }
        if constexpr (DOT) partial += c0;
    }
    const int lane = threadIdx.x & (kWave - 1);
    const int g = lane / T, t = lane % T;
    double p[C];
#pragma unroll
    for (int c = 0; c < C; ++c) p[c] = 0.0;
    const SliceWalk sw = slice_walk(nslices);
    for (int si = sw.begin; si < sw.end; si += sw.stride) {
        const int slice = sched ? sched[si] : si;
        double acc[T][C];
        sell_row_product<NB, BV>(slice_off, cols, vals, xin, slice, lane, LD, acc);
#pragma unroll
        for (int rs = 0; rs < T; ++rs) {
            const int row = slice * kWave + rs * G + g;
            if (row >= nrows) continue;
            const size_t at = (size_t)row * LD + t * C;
            double rv[C], dv[C], xv[C], di[C];
            load_c<C>(r + at, rv);
            load_c<C>(xin + at, xv);
            if (a != 0.0) {
                load_c<C>(d + at, dv);
            } else {
#pragma unroll
                for (int c = 0; c < C; ++c) dv[c] = 0.0;
            }
            if constexpr (BV) {
                load_c<C>(dinv + at, di);
            } else {
                const double s = dinv[row];
#pragma unroll
                for (int c = 0; c < C; ++c) di[c] = s;
            }
#pragma unroll
            for (int c = 0; c < C; ++c) {
                dv[c] = a * dv[c] + b * di[c] * (rv[c] - acc[rs][c]);
                xv[c] += dv[c];
            }
            round_to<OT>(xv);
            if constexpr (DOT) {
#pragma unroll
                for (int c = 0; c < C; ++c) p[c] = fma(rv[c], xv[c], p[c]);
            }
            if constexpr (std::is_same<OT, double>::value) store_c<C>(d + at, dv);
            store_v<C>(xout + at, xv);
        }
    }
    if constexpr (DOT) reduce_cols_store<NB>(p, partial, LD);
}

// Degree-2 Chebyshev polynomial from a ZERO initial guess in ONE pass.  With t = D^-1 r the two steps
//   x1 = t/theta ;  x2 = x1 + rho1 rho0 x1 + (2 rho1/delta) D^-1 (r - A x1)
// collapse to  x2_i = dinv_i (c0 r_i - c1 (A D^-1 r)_i),  so with the column-scaled values As = A D^-1
// (precomputed at create time) a single SpMM over r gives the result: 1 gather pass instead of the
// 3 + 5 vector passes of cheb_first + cheb_step.  DOT: partials of <r, x2>.
// From a NONZERO guess x0 the same polynomial acts on the residual: x2 = x0 + p2(r - A x0); then r is that residual,
// xadd = x0 (may alias xout: no gathers on it) and the dot is taken with dot_with (the right-hand side).
template <int NB, int BV, bool DOT, bool NT = false, typename OT = double>
__global__ __launch_bounds__(kBlock, (NB >= 32 && BV == 0 ? 3 : 1)) void sell_poly2_kernel(int nrows, int nslices, const int* __restrict__ slice_off,
                                                            const int* __restrict__ sched,
                                                            const int* __restrict__ cols,
                                                            const double* __restrict__ vals_scaled,
                                                            const double* __restrict__ dinv,
                                                            const double* __restrict__ r, OT* xout,
                                                            double c0, double c1, double* __restrict__ partial,
                                                            const double* xadd, const double* __restrict__ dot_with,
                                                            const int* __restrict__ padd_idx,
                                                            const double* __restrict__ padd_x, int ld) {
    constexpr int C = Lay<NB>::C, T = Lay<NB>::T, G = Lay<NB>::G;
    const int LD = row_ld<NB>(ld);
    {
        const int c0 = col0<NB>();
        r += c0; xout += c0;
        if constexpr (BV) { vals_scaled = shift_bv<BV>(vals_scaled, c0); dinv += c0; }
        if (xadd) xadd += c0;
        if (dot_with) dot_with += c0;
        if (padd_x) padd_x += c0;
        if constexpr (DOT) partial += c0;
    }
    const int lane = threadIdx.x & (kWave - 1);
    const int g = lane / T, t = lane % T;
    double p[C];
#pragma unroll
    for (int c = 0; c < C; ++c) p[c] = 0.0;
    const SliceWalk sw = slice_walk(nslices);
    for (int si = sw.begin; si < sw.end; si += sw.stride) {
        const int slice = sched ? sched[si] : si;
        double acc[T][C];
        if constexpr (NT) {
            const int off = slice_off[slice];
            sell_row_range<NB, BV, false, true, true>(cols, vals_scaled, r, nullptr, off, (slice_off[slice + 1] - off) >> 6,
                                                        lane, LD, acc);
        } else {
            sell_row_product<NB, BV>(slice_off, cols, vals_scaled, r, slice, lane, LD, acc);
        }
        // own-row reads in batches of H row steps (shared values only: the per-realization instantiations have no registers
        // to spare); rows past the end re-read the last row and store nothing
        constexpr int H = (BV == 0 && T >= 4) ? 2 : 1;   // (4 spills in the fp64-output instantiations: 168 registers are the cap)
        double rvb[H][C], dib[H];
#pragma unroll
        for (int rs = 0; rs < T; ++rs) {
            const int row = slice * kWave + rs * G + g;
            if constexpr (H > 1) {
                if (rs % H == 0) {
#pragma unroll
                    for (int u = 0; u < H; ++u) {
                        const int rc = min(row + u * G, nrows - 1);
                        load_c<C>(r + (size_t)rc * LD + t * C, rvb[u]);
                        dib[u] = dinv[rc];
                    }
                    pin_block(rvb);
                }
            }
            if (row >= nrows) continue;
            const size_t at = (size_t)row * LD + t * C;
            double rv[C], xv[C], di[C];
            if constexpr (H > 1) {
#pragma unroll
                for (int c = 0; c < C; ++c) { rv[c] = rvb[rs % H][c]; di[c] = dib[rs % H]; }
            } else {
                load_c<C>(r + at, rv);
                if constexpr (BV) {
                    load_c<C>(dinv + at, di);
                } else {
                    const double s = dinv[row];
#pragma unroll
                    for (int c = 0; c < C; ++c) di[c] = s;
                }
            }
#pragma unroll
            for (int c = 0; c < C; ++c) xv[c] = di[c] * (c0 * rv[c] - c1 * acc[rs][c]);
            if (xadd) {
                double x0[C];
                load_c<C>(xadd + at, x0);
#pragma unroll
                for (int c = 0; c < C; ++c) xv[c] += x0[c];
            }
            if (padd_idx) {   // + (P xc)_row for an injection-type prolongator: xc[parent[row]]
                double pc[C];
                load_c<C>(padd_x + (size_t)padd_idx[row] * LD + t * C, pc);
#pragma unroll
                for (int c = 0; c < C; ++c) xv[c] += pc[c];
            }
            round_to<OT>(xv);
            if constexpr (DOT) {
                if (dot_with) load_c<C>(dot_with + at, rv);
#pragma unroll
                for (int c = 0; c < C; ++c) p[c] = fma(rv[c], xv[c], p[c]);
            }
            store_v_stream<NT, C>(xout + at, xv);
        }
    }
    if constexpr (DOT) reduce_cols_store<NB>(p, partial, LD);
}

// ------------------------------------------------------------------------------------------
// V-cycle kernels with fp32 INTERMEDIATES (shared-value hierarchies: the sampler).  The vectors that live only inside one
// application of the preconditioner - the pre-smoothed iterate and the residuals of a level - are stored in fp32; the
// preconditioner's input and output, every coarse right-hand side / correction and all arithmetic stay fp64.  A
// preconditioner whose result carries fp32-sized rounding does not limit what MINRES attains: its search directions are
// the preconditioned vectors themselves, q = A z is formed from the z actually delivered, so the residual recurrence stays
// consistent - measured (z rounded to fp32 after every application, cube_tet r = 4): identical iteration counts at 1e-6 ...
// 1e-12 and fields equal to the unrounded run's to 7e-16.  What it saves is 87 MB of the 1 089 MB an iteration moves at r = 5.

// DEEP gather loop for levels too small to fill the chip (NB = 32, shared values): J slice columns - J x 8 gathers per lane -
// are in flight together.  A launch of a few hundred to a few thousand wavefronts (one per 64-row slice) runs less than one
// wavefront per SIMD; each walks its slice's columns as a chain of dependent round trips to L2, so the launch lasts
// `slice width` x latency whatever its size (round 5: the five V-cycle kernels of the 4 964-row level of the hybridized
// hierarchy - 78 wavefronts, 17-27 columns - took ~38 us each, those of the 43 622-row level ~30 us).  With J columns per
// trip the chain is J times shorter; registers (J x 32 for the raw fp32 rows) are no concern at that occupancy.  Same FMA
// order per accumulator as the one-column loop (column after column), out-of-range columns re-read the last column with
// weight zero: bit-identical results.
typedef float pmc_f4x __attribute__((ext_vector_type(4)));
template <int J>
__device__ __forceinline__ void pin_deep(pmc_f4x (&q)[J][8]) {
    if constexpr (J == 2)
        asm volatile("" : "+v"(q[0][0]), "+v"(q[0][1]), "+v"(q[0][2]), "+v"(q[0][3]), "+v"(q[0][4]), "+v"(q[0][5]), "+v"(q[0][6]),
                     "+v"(q[0][7]), "+v"(q[1][0]), "+v"(q[1][1]), "+v"(q[1][2]), "+v"(q[1][3]), "+v"(q[1][4]), "+v"(q[1][5]),
                     "+v"(q[1][6]), "+v"(q[1][7]));
    else if constexpr (J == 4)
        asm volatile("" : "+v"(q[0][0]), "+v"(q[0][1]), "+v"(q[0][2]), "+v"(q[0][3]), "+v"(q[0][4]), "+v"(q[0][5]), "+v"(q[0][6]),
                     "+v"(q[0][7]), "+v"(q[1][0]), "+v"(q[1][1]), "+v"(q[1][2]), "+v"(q[1][3]), "+v"(q[1][4]), "+v"(q[1][5]),
                     "+v"(q[1][6]), "+v"(q[1][7]), "+v"(q[2][0]), "+v"(q[2][1]), "+v"(q[2][2]), "+v"(q[2][3]), "+v"(q[2][4]),
                     "+v"(q[2][5]), "+v"(q[2][6]), "+v"(q[2][7]), "+v"(q[3][0]), "+v"(q[3][1]), "+v"(q[3][2]), "+v"(q[3][3]),
                     "+v"(q[3][4]), "+v"(q[3][5]), "+v"(q[3][6]), "+v"(q[3][7]));
}
typedef double pmc_d2x __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void pin_deep_d(pmc_d2x (&q)[2][8][2]) {
    asm volatile("" : "+v"(q[0][0][0]), "+v"(q[0][0][1]), "+v"(q[0][1][0]), "+v"(q[0][1][1]), "+v"(q[0][2][0]), "+v"(q[0][2][1]),
                 "+v"(q[0][3][0]), "+v"(q[0][3][1]), "+v"(q[0][4][0]), "+v"(q[0][4][1]), "+v"(q[0][5][0]), "+v"(q[0][5][1]),
                 "+v"(q[0][6][0]), "+v"(q[0][6][1]), "+v"(q[0][7][0]), "+v"(q[0][7][1]), "+v"(q[1][0][0]), "+v"(q[1][0][1]),
                 "+v"(q[1][1][0]), "+v"(q[1][1][1]), "+v"(q[1][2][0]), "+v"(q[1][2][1]), "+v"(q[1][3][0]), "+v"(q[1][3][1]),
                 "+v"(q[1][4][0]), "+v"(q[1][4][1]), "+v"(q[1][5][0]), "+v"(q[1][5][1]), "+v"(q[1][6][0]), "+v"(q[1][6][1]),
                 "+v"(q[1][7][0]), "+v"(q[1][7][1]));
}
template <int NB, typename XT, int J>
__device__ __forceinline__ void sell_row_range_deep(const int* __restrict__ cols, const double* __restrict__ vals,
                                                    const XT* __restrict__ x, int off, int width, int lane, int LD,
                                                    double (&acc)[Lay<NB>::T][Lay<NB>::C]) {
    constexpr int C = Lay<NB>::C, T = Lay<NB>::T, G = Lay<NB>::G;
    static_assert(C == 4 && T == 8, "the deep loop is written for the 32-wide layout");
    static_assert((sizeof(XT) == 4 && (J == 2 || J == 4)) || (sizeof(XT) == 8 && J == 2), "columns per trip");
    const int g = lane / T, t = lane % T;
#pragma unroll
    for (int rs = 0; rs < T; ++rs)
#pragma unroll
        for (int c = 0; c < C; ++c) acc[rs][c] = 0.0;
    int slot = off + lane;
    int cj[J];
    double vj[J];
#pragma unroll
    for (int u = 0; u < J; ++u) {
        const bool ok = u < width;
        const int at = ok ? slot + u * kWave : slot;
        cj[u] = width > 0 ? cols[at] : 0;
        vj[u] = (ok && width > 0) ? vals[at] : 0.0;
    }
    for (int j = 0; j < width; j += J, slot += J * kWave) {
        int cn[J];
        double vn[J];
#pragma unroll
        for (int u = 0; u < J; ++u) {   // the next trip's (index, value) pairs: requested before this trip's gathers
            const bool ok = j + J + u < width;
            const int at = ok ? slot + (J + u) * kWave : slot;
            cn[u] = cols[at];
            vn[u] = ok ? vals[at] : 0.0;
        }
        unsigned at[J][T];
#pragma unroll
        for (int u = 0; u < J; ++u)
#pragma unroll
            for (int rs = 0; rs < T; ++rs) at[u][rs] = (unsigned)__shfl(cj[u], rs * G + g, kWave) * (unsigned)LD + (unsigned)(t * C);
        if constexpr (sizeof(XT) == 4) {
            pmc_f4x q[J][8];
#pragma unroll
            for (int u = 0; u < J; ++u)
#pragma unroll
                for (int rs = 0; rs < T; ++rs) q[u][rs] = *reinterpret_cast<const pmc_f4x*>(x + at[u][rs]);
            pin_deep<J>(q);
#pragma unroll
            for (int u = 0; u < J; ++u)
#pragma unroll
                for (int rs = 0; rs < T; ++rs) {
                    const double a = __shfl(vj[u], rs * G + g, kWave);
#pragma unroll
                    for (int c = 0; c < C; ++c) acc[rs][c] = fma(a, (double)q[u][rs][c], acc[rs][c]);
                }
        } else {
            pmc_d2x q[2][8][2];
#pragma unroll
            for (int u = 0; u < J; ++u)
#pragma unroll
                for (int rs = 0; rs < T; ++rs) {
                    q[u][rs][0] = reinterpret_cast<const pmc_d2x*>(x + at[u][rs])[0];
                    q[u][rs][1] = reinterpret_cast<const pmc_d2x*>(x + at[u][rs])[1];
                }
            pin_deep_d(q);
#pragma unroll
            for (int u = 0; u < J; ++u)
#pragma unroll
                for (int rs = 0; rs < T; ++rs) {
                    const double a = __shfl(vj[u], rs * G + g, kWave);
#pragma unroll
                    for (int c = 0; c < C; ++c) acc[rs][c] = fma(a, q[u][rs][c >> 1][c & 1], acc[rs][c]);
                }
        }
#pragma unroll
        for (int u = 0; u < J; ++u) {
            cj[u] = cn[u];
            vj[u] = vn[u];
        }
    }
}

// acc = A x for one slice, shared fp64 values, gathered vector of type XT (the T > 1 schedule of sell_row_range; T == 1
// walks the slice columns one by one)
// DEEP > 1 (NB = 32, shared values): sell_row_range_deep with that many slice columns per trip
template <int NB, typename XT, bool NT = false, int BV = 0, int DEEP = 1>
__device__ __forceinline__ void sell_row_range_t(const int* __restrict__ cols, const double* __restrict__ vals,
                                                 const XT* __restrict__ x, int off, int width, int lane, int LD,
                                                 double (&acc)[Lay<NB>::T][Lay<NB>::C]) {
    if constexpr (DEEP > 1 && NB == 32 && BV == 0)
        sell_row_range_deep<NB, XT, DEEP>(cols, vals, x, off, width, lane, LD, acc);
    else
        sell_row_range<NB, BV, false, true, NT, XT>(cols, vals, x, nullptr, off, width, lane, LD, acc);
}

// out = dinv (c0 r - c1 As r) (+ xadd) (+ padd_x[padd_idx]) with r of type XT (gathered and read at the own row), out of
// type OT, xadd of type AT; DOT: partials of <dot_with, out> (dot_with fp64).  See sell_poly2_kernel.
#ifndef PMC_VC_MIN_WAVES
#define PMC_VC_MIN_WAVES 3   // post-smoothing of the 400 k-row multiplier level: 172 -> 168 registers, 83.6 -> 79.7 us (LAB_NOTES 10.3)
#endif
#ifndef PMC_VC_MIN_WAVES_D
#define PMC_VC_MIN_WAVES_D 1
#endif
// wavefronts per SIMD the fp32-gather V-cycle kernels of the 32-wide layout are compiled for (laboratory macro)
template <int NB, typename XT, int BV, int DEEP>
constexpr int vc_min_waves() {
    return (NB >= 32 && BV == 0 && DEEP == 1) ? (sizeof(XT) == 4 ? PMC_VC_MIN_WAVES : PMC_VC_MIN_WAVES_D) : 1;
}
// SPL (launches of at most 8 realizations on the small levels of an aggregation hierarchy, whose rows hold 20-40 entries): the
// matrix stores every row as 2^sl consecutive pieces (csr_split_rows), nrows counts the ROWS; the pieces of a row sit in
// neighbouring lane groups and are added with a shuffle tree, the first piece's lanes finish the row.  A 5 k-row level then
// runs 2^sl times the wavefronts over slices 2^sl times shorter - these launches are one chain of dependent gathers per slice.
template <int NB, int C, int T>
__device__ __forceinline__ void split_row_sums(double (&acc)[T][C], int sl) {
#pragma unroll
    for (int rs = 0; rs < T; ++rs)
#pragma unroll
        for (int c = 0; c < C; ++c) {
            double v = acc[rs][c];
            for (int o = 0; o < sl; ++o) v += __shfl_xor(v, T << o, kWave);
            acc[rs][c] = v;
        }
}
// GIB (launches of 64 realizations = two column groups): ONE workgroup sweeps a slice for both groups back to back - the slice's
// (index, value) pairs come from L1 / L2 the second time instead of being fetched again by another workgroup at another time
// (what `traffic` showed as 1.34 x the algorithmic bytes) - gridDim.y is 1 and the partial sums keep their layout.
template <int NB, typename XT, typename OT, typename AT, bool DOT, bool NT = false, int BV = 0, int DEEP = 1, bool SPL = false,
          bool GIB = false>
__global__ __launch_bounds__(kBlock, (vc_min_waves<NB, XT, BV, DEEP>())) void vc_poly2_kernel(int nrows, int nslices, const int* __restrict__ slice_off,
                                                          const int* __restrict__ cols, const double* __restrict__ vals_scaled,
                                                          const double* __restrict__ dinv, const XT* __restrict__ r, OT* xout,
                                                          double c0, double c1, double* __restrict__ partial, const AT* xadd,
                                                          const double* __restrict__ dot_with,
                                                          const int* __restrict__ padd_idx, const double* __restrict__ padd_x,
                                                          int ld, int sl = 0) {
    static_assert(!SPL || (NB <= 8 && BV == 0 && DEEP == 1), "row-split instantiations: narrow launches, shared values");
    static_assert(!GIB || (NB == kGroup && BV == 0 && DEEP == 1 && !SPL), "both column groups in one workgroup: 64 wide, shared values");
    constexpr int C = Lay<NB>::C, T = Lay<NB>::T, G = Lay<NB>::G;
    constexpr int NG = GIB ? 2 : 1;
    const int LD = row_ld<NB>(ld);
    if constexpr (!GIB) {
        const int g0 = col0<NB>();
        r += g0; xout += g0;
        if constexpr (BV != 0) { vals_scaled = shift_bv<BV>(vals_scaled, g0); dinv += g0; }
        if (xadd) xadd += g0;
        if (dot_with) dot_with += g0;
        if (padd_x) padd_x += g0;
        if constexpr (DOT) partial += g0;
    }
    const int lane = threadIdx.x & (kWave - 1);
    const int g = lane / T, t = lane % T;
    const SliceWalk sw = slice_walk(nslices);
    // (the group loop is the OUTER one and not unrolled: inside the slice loop the two groups' gathers interleave and 84-188
    // registers spill; a workgroup walks one or two slices per wavefront, so the second sweep still finds them in L1 / L2)
#pragma unroll 1
    for (int grp = 0; grp < NG; ++grp) {
      const int go = grp * NB;                         // 0 unless GIB
      const XT* rg = r + go;
      OT* xg = xout + go;
      const AT* xa = xadd ? xadd + go : nullptr;
      const double* dw = dot_with ? dot_with + go : nullptr;
      const double* px = padd_x ? padd_x + go : nullptr;
      double p[C];
#pragma unroll
      for (int c = 0; c < C; ++c) p[c] = 0.0;
      for (int slice = sw.begin; slice < sw.end; slice += sw.stride) {
        double acc[T][C];
        const int off = slice_off[slice];
        sell_row_range_t<NB, XT, NT, BV, DEEP>(cols, vals_scaled, rg, off, (slice_off[slice + 1] - off) >> 6, lane, LD, acc);
        if constexpr (SPL) split_row_sums<NB, C, T>(acc, sl);
        // row steps in pairs: the own-row reads of both (and the parent indices of the coarse correction) are issued before
        // either is consumed - rows past the end re-read the last row and store nothing
        // (pairs only where the gathers above leave the registers for it - the fp32-gather instantiations; the fp64-gather
        // ones, with sixteen 16-byte gathers in flight, would drop from three to two waves per SIMD)
        constexpr int H = (T >= 2 && sizeof(XT) == 4) ? 2 : 1;
#pragma unroll
        for (int h0 = 0; h0 < T; h0 += H) {
            double rv[H][C], di[H][C], x0[H][C], pc[H][C], wv[H][C];
            size_t at[H];
            int par[H];
            bool ok[H];
#pragma unroll
            for (int u = 0; u < H; ++u) {
                const int piece = slice * kWave + (h0 + u) * G + g;
                const int row = SPL ? piece >> sl : piece;
                ok[u] = row < nrows && (!SPL || (piece & ((1 << sl) - 1)) == 0);
                const int rowc = row < nrows ? row : nrows - 1;
                at[u] = (size_t)rowc * LD + t * C;
                load_v<C>(rg + at[u], rv[u]);
                if constexpr (BV != 0) {
                    load_c<C>(dinv + at[u], di[u]);
                } else {
                    const double sdi = dinv[rowc];
#pragma unroll
                    for (int c = 0; c < C; ++c) di[u][c] = sdi;
                }
                if (xa) load_v<C>(xa + at[u], x0[u]);
                if (padd_idx) par[u] = padd_idx[rowc];
                if constexpr (DOT) load_c<C>(dw + at[u], wv[u]);
            }
            if (padd_idx) {
#pragma unroll
                for (int u = 0; u < H; ++u) load_c<C>(px + (size_t)par[u] * LD + t * C, pc[u]);
            }
#pragma unroll
            for (int u = 0; u < H; ++u) {
                double xv[C];
#pragma unroll
                for (int c = 0; c < C; ++c) xv[c] = di[u][c] * (c0 * rv[u][c] - c1 * acc[h0 + u][c]);
                if (xa) {
#pragma unroll
                    for (int c = 0; c < C; ++c) xv[c] += x0[u][c];
                }
                if (padd_idx) {
#pragma unroll
                    for (int c = 0; c < C; ++c) xv[c] += pc[u][c];
                }
                round_to<OT>(xv);
                if (ok[u]) {
                    if constexpr (DOT) {
#pragma unroll
                        for (int c = 0; c < C; ++c) p[c] = fma(wv[u][c], xv[c], p[c]);
                    }
                    store_v_stream<NT, C>(xg + at[u], xv);
                }
            }
        }
      }
      if constexpr (DOT) {
          reduce_cols_store<NB>(p, partial + go, LD);
          if (grp + 1 < NG) __syncthreads();            // the reduction's LDS scratch is reused by the next group
      }
    }
}

// y = r - A x with x of type XT (gathered), r of type RT, y of type YT; R8: rows also summed in groups of 8 into `coarse`
// (fp64), see sell_spmm_kernel
// BV: 0 shared fp64 values, 2 per-realization fp32 values; STORE = false: only the restricted sums are wanted (y unused)
// RAGG (aggregation levels renumbered by agg_pack_rows: every aggregate a run of consecutive rows inside one slice): the
// wavefront keeps its 64 x NB residual tile - the fp32 values it has just stored - in LDS and sums its own aggregates from it
// in increasing row order: coarse[cid] = sum of the rows of segment (cid, first row, rows).  No other wavefront touches those
// coarse rows: deterministic, no atomics, and the separate product with P^T (one more pass over the residual) is gone.
// SPL: rows stored in 2^sl pieces, see vc_poly2_kernel
template <int NB, typename XT, typename RT, typename YT, bool R8, int BV = 0, bool STORE = true, int DEEP = 1, bool RAGG = false,
          bool SPL = false>
__global__ __launch_bounds__(kBlock) void vc_residual_kernel(int nrows, int nslices, const int* __restrict__ slice_off,
                                                             const int* __restrict__ cols, const double* __restrict__ vals,
                                                             const XT* __restrict__ x, const RT* r, YT* y,
                                                             double* __restrict__ coarse, int ld,
                                                             const int* __restrict__ seg_ptr = nullptr,
                                                             const int* __restrict__ seg_cid = nullptr,
                                                             const int* __restrict__ seg_pos = nullptr, int sl = 0) {
    static_assert(!SPL || (NB <= 8 && BV == 0 && DEEP == 1 && !R8 && !RAGG && STORE), "row-split instantiations: narrow launches, plain residual");
    static_assert(STORE || R8, "a residual that is neither stored nor restricted");
    static_assert(!RAGG || (!R8 && STORE && BV == 0 && sizeof(YT) == 4), "fused aggregate restriction: shared values, fp32 residual");
    constexpr int C = Lay<NB>::C, T = Lay<NB>::T, G = Lay<NB>::G;
    const int LD = row_ld<NB>(ld);
    {
        const int g0 = col0<NB>();
        x += g0; r += g0;
        if constexpr (STORE) y += g0;
        if constexpr (BV != 0) vals = shift_bv<BV>(vals, g0);
        if constexpr (R8 || RAGG) coarse += g0;
    }
    const int lane = threadIdx.x & (kWave - 1);
    const int g = lane / T, t = lane % T;
    __shared__ float tile_all[RAGG ? (kBlock / kWave) * kWave * NB : 1];
    float* tile = tile_all + (RAGG ? (threadIdx.x / kWave) * kWave * NB : 0);
    const SliceWalk sw = slice_walk(nslices);
    for (int slice = sw.begin; slice < sw.end; slice += sw.stride) {
        double acc[T][C];
        const int off = slice_off[slice];
        sell_row_range_t<NB, XT, false, BV, DEEP>(cols, vals, x, off, (slice_off[slice + 1] - off) >> 6, lane, LD, acc);
        if constexpr (SPL) split_row_sums<NB, C, T>(acc, sl);
        // the own-row reads of H row steps are issued together (rows past the end re-read the last row): one latency per
        // batch instead of one per row step - these launches are single occupancy rounds of dependent loads
        constexpr int H = T >= 4 ? 4 : T;
        double rvb[H][C];
#pragma unroll
        for (int rs = 0; rs < T; ++rs) {
            const int piece = slice * kWave + rs * G + g;
            const int row = SPL ? piece >> sl : piece;
            if (rs % H == 0) {
#pragma unroll
                for (int u = 0; u < H; ++u) {
                    const int ru = SPL ? (piece + u * G) >> sl : row + u * G;
                    load_v<C>(r + (size_t)(ru < nrows ? ru : nrows - 1) * LD + t * C, rvb[u]);
                }
            }
            if (row < nrows && (!SPL || (piece & ((1 << sl) - 1)) == 0)) {
                const size_t at = (size_t)row * LD + t * C;
#pragma unroll
                for (int c = 0; c < C; ++c) acc[rs][c] = rvb[rs % H][c] - acc[rs][c];
                if constexpr (STORE) store_v<C>(y + at, acc[rs]);
                if constexpr (RAGG) {
#pragma unroll
                    for (int c = 0; c < C; ++c) tile[(rs * G + g) * NB + t * C + c] = (float)acc[rs][c];
                }
            } else if constexpr (R8) {
#pragma unroll
                for (int c = 0; c < C; ++c) acc[rs][c] = 0.0;
            }
            if constexpr (R8) {
                double s[C];
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    double v = acc[rs][c];
                    v += __shfl_xor(v, T, kWave);
                    v += __shfl_xor(v, 2 * T, kWave);
                    v += __shfl_xor(v, 4 * T, kWave);
                    s[c] = v;
                }
                if ((g & 7) == 0 && row < nrows) store_c<C>(coarse + (size_t)(row >> 3) * LD + t * C, s);
            }
        }
        if constexpr (RAGG) {
            // the tile is private to this wavefront: its own LDS writes are complete once the wait below has passed
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int s0 = seg_ptr[slice], s1 = seg_ptr[slice + 1];
            for (int sg = s0 + g; sg < s1; sg += G) {
                const int pos = seg_pos[sg], first = pos >> 8, len = pos & 255;
                double sum[C];
#pragma unroll
                for (int c = 0; c < C; ++c) sum[c] = 0.0;
                for (int q = 0; q < len; ++q)
#pragma unroll
                    for (int c = 0; c < C; ++c) sum[c] += (double)tile[(first + q) * NB + t * C + c];
                store_c<C>(coarse + (size_t)seg_cid[sg] * LD + t * C, sum);
            }
            __builtin_amdgcn_wave_barrier();    // the next slice overwrites the tile
        }
    }
}

// x (fp32) += xc[row >> 3] (fp64): the coarse correction of a prolongator over groups of 8 consecutive rows
template <int NB>
__global__ __launch_bounds__(kBlock) void vc_prolong8_kernel(size_t nflat, float* __restrict__ x, const double* __restrict__ xc,
                                                             int ld) {
    constexpr int C = Lay<NB>::C;
    const int W = row_ld<NB>(ld);
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= nflat) return;
    const size_t e = i * C;
    const size_t row = e / W;
    const int k0 = (int)(e % W);
    double xv[C], cv[C];
    load_cf<C>(x + e, xv);
    load_c<C>(xc + (row >> 3) * W + k0, cv);
#pragma unroll
    for (int c = 0; c < C; ++c) xv[c] += cv[c];
    store_v<C>(x + e, xv);
}

// y = A1 x1 + A2 x2 over the SAME rows: A1 with per-realization values, A2 with shared values (the u-rows
// [M(k) | B^T] of the Darcy operator in one pass); DOT: partials of <dot_with, y>.
template <int NB, bool DOT, typename XT = double>
__global__ __launch_bounds__(kBlock) void sell_pair_spmm_kernel(
    int nrows, int nslices, const int* __restrict__ off1, const int* __restrict__ cols1, const double* __restrict__ vals1,
    const int* __restrict__ off2, const int* __restrict__ cols2, const double* __restrict__ vals2,
    const XT* __restrict__ x1, const typename ident<XT>::type* __restrict__ x2, double* __restrict__ y,
    const typename ident<XT>::type* __restrict__ dot_with, double* __restrict__ partial, int ld) {
    constexpr int C = Lay<NB>::C, T = Lay<NB>::T, G = Lay<NB>::G;
    const int LD = row_ld<NB>(ld);
    {
        const int c0 = col0<NB>();
        vals1 += c0; x1 += c0; x2 += c0; y += c0;
        if constexpr (DOT) { dot_with += c0; partial += c0; }
    }
    const int lane = threadIdx.x & (kWave - 1);
    const int g = lane / T, t = lane % T;
    double p[C];
#pragma unroll
    for (int c = 0; c < C; ++c) p[c] = 0.0;
    const SliceWalk sw = slice_walk(nslices);
    for (int slice = sw.begin; slice < sw.end; slice += sw.stride) {
        double acc[T][C], acc2[T][C];
        sell_row_product<NB, true, XT>(off1, cols1, vals1, x1, slice, lane, LD, acc);
        sell_row_product<NB, false, XT>(off2, cols2, vals2, x2, slice, lane, LD, acc2);
#pragma unroll
        for (int rs = 0; rs < T; ++rs) {
            const int row = slice * kWave + rs * G + g;
            if (row >= nrows) continue;
            const size_t at = (size_t)row * LD + t * C;
#pragma unroll
            for (int c = 0; c < C; ++c) acc[rs][c] += acc2[rs][c];
            store_c<C>(y + at, acc[rs]);
            if constexpr (DOT) {
                double w[C];
                load_v<C>(dot_with + at, w);
#pragma unroll
                for (int c = 0; c < C; ++c) p[c] = fma(w[c], acc[rs][c], p[c]);
            }
        }
    }
    if constexpr (DOT) reduce_cols_store<NB>(p, partial, LD);
}

// sell_row_range for the TH row steps rs0 .. rs0 + TH - 1 of a slice only (shared values).  A kernel that sweeps a slice in
// T / TH such passes keeps TH instead of T rows' accumulators, gathers and shuffled slot data alive - the element-grouped
// kernels below, which carry two accumulators and up to three gathered vectors per row, drop from 206-246 VGPRs (two waves per
// SIMD) to four waves per SIMD; the (index, value) pairs of the later passes come from L1.
template <int NB, bool CS, typename XT>
__device__ __forceinline__ constexpr bool lean_part() {
    return Lay<NB>::T > 1 && ((!CS && sizeof(XT) == 4) ? kLeanGather : kLeanCs);
}
template <int NB, bool CS, bool ZERO, bool NT, int TH, typename XT = double>
__device__ __forceinline__ void sell_row_part(const int* __restrict__ cols, const double* __restrict__ vals,
                                              const XT* __restrict__ x, const double* __restrict__ cs, int off, int width,
                                              int lane, int LD, int rs0, double (&acc)[TH][Lay<NB>::C]) {
    constexpr int C = Lay<NB>::C, T = Lay<NB>::T, G = Lay<NB>::G;
    const int g = lane / T, t = lane % T;
    if constexpr (ZERO) {
#pragma unroll
        for (int q = 0; q < TH; ++q)
#pragma unroll
            for (int c = 0; c < C; ++c) acc[q][c] = 0.0;
    }
    int slot = off + lane;
    int cj = 0;
    double vj = 0.0;
    if (width > 0) {
        cj = load_stream<NT>(cols + slot);
        vj = load_stream<NT>(vals + slot);
    }
    if constexpr (lean_part<NB, CS, XT>()) {
        // Lean loop: 32-bit element offsets address the gathers, gathered rows stay in their storage type until their FMA
        // (fp32 rows of the preconditioned Krylov vectors: half the registers), and the matrix values are fetched across
        // lanes only AFTER the gathers have been issued - fewer registers live while the loads are in flight, and the
        // cross-lane traffic overlaps the gather latency.  Hex 64^3 x 16, rocprofv3 averages: Darcy operator u-rows
        // 96.9 -> 85.1 us, M-block polynomial 119.5 -> 114.8 us (compiled for three waves per SIMD instead the operator
        // spills and takes 90.5 us; the same loop in the block operator K5 - 92 instead of 114 registers, five waves - changed
        // nothing measurable).
        for (int j = 0; j < width; ++j, slot += kWave) {
            int cn = cj;
            double vn = vj;
            if (j + 1 < width) {
                cn = load_stream<NT>(cols + slot + kWave);
                vn = load_stream<NT>(vals + slot + kWave);
            }
            unsigned at[TH];
#pragma unroll
            for (int q = 0; q < TH; ++q) at[q] = (unsigned)__shfl(cj, (rs0 + q) * G + g, kWave) * (unsigned)LD + (unsigned)(t * C);
            RawVec<XT, C> xr[TH];
            double sv[CS ? TH : 1][C];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < TH; ++q) {
                load_raw<C>(x + at[q], xr[q]);
                if constexpr (CS) load_c<C>(cs + at[q], sv[q]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < TH; ++q) {
                const double a = __shfl(vj, (rs0 + q) * G + g, kWave);
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    double xv = (double)xr[q].v[c];
                    if constexpr (CS) xv *= sv[q][c];
                    acc[q][c] = fma(a, xv, acc[q][c]);
                }
            }
            cj = cn;
            vj = vn;
        }
        return;
    }
    for (int j = 0; j < width; ++j, slot += kWave) {
        int cn = cj;
        double vn = vj;
        if (j + 1 < width) {
            cn = load_stream<NT>(cols + slot + kWave);
            vn = load_stream<NT>(vals + slot + kWave);
        }
        int cc[TH];
        double aa[TH];
#pragma unroll
        for (int q = 0; q < TH; ++q) {
            const int src = (rs0 + q) * G + g;
            cc[q] = (T == 1) ? cj : __shfl(cj, src, kWave);
            aa[q] = (T == 1) ? vj : __shfl(vj, src, kWave);
        }
        double xv[TH][C], sv[TH][C];
        if constexpr (T > 1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < TH; ++q) {
            load_v<C>(x + (size_t)cc[q] * LD + t * C, xv[q]);
            if constexpr (CS) load_c<C>(cs + (size_t)cc[q] * LD + t * C, sv[q]);
        }
        if constexpr (T > 1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < TH; ++q)
#pragma unroll
            for (int c = 0; c < C; ++c) {
                if constexpr (CS) xv[q][c] *= sv[q][c];
                acc[q][c] = fma(aa[q], xv[q][c], acc[q][c]);
            }
        cj = cn;
        vj = vn;
    }
}

// row steps per pass of the element-grouped kernels
template <int NB>
struct EgPass {
    static constexpr int TH = NB >= 32 ? 4 : Lay<NB>::T;   // only the widest instantiation (256 VGPRs, one wave per SIMD otherwise)
};

// Element-grouped per-realization mass matrix (EgView): acc = c1 * (group 1 row sums) + c2 * (group 2 row sums), for the row
// steps rs0 .. rs0 + TH - 1 of the slice.  The two coefficient rows are requested before the sweeps they scale.
// kEgNt: non-temporal matrix / result streams on large levels (hex 64^3, one lane: 28.2 -> 27.3 ms per 16 Darcy solves).
template <int NB, bool CS, bool kEgNt, int TH, typename XT = double>
__device__ __forceinline__ void eg_row_product(const int* __restrict__ cols, const double* __restrict__ w,
                                               const int* __restrict__ e12, const double* __restrict__ coef, int gw,
                                               const XT* __restrict__ x, const double* __restrict__ cs, int nrows,
                                               int slice, int lane, int LD, int rs0, double (&y)[TH][Lay<NB>::C]) {
    constexpr int C = Lay<NB>::C, T = Lay<NB>::T, G = Lay<NB>::G;
    const int g = lane / T, t = lane % T;
    const int off = slice * 2 * gw * kWave;
    double a[TH][C], c1[TH][C];
    // lean variant (fp32 x, three waves per SIMD instead of two): only the coefficient-row INDICES are fetched ahead of a
    // sweep, the rows themselves after it - 2 TH C registers fewer live while the gathers are in flight
    constexpr bool LATE = lean_part<NB, CS, XT>() && kEgLateCoef;
    int e1[TH];
#pragma unroll
    for (int q = 0; q < TH; ++q) {
        const int row = min(slice * kWave + (rs0 + q) * G + g, nrows - 1);
        if constexpr (LATE) e1[q] = e12[2 * row];
        else load_c<C>(coef + (size_t)e12[2 * row] * LD + t * C, c1[q]);
    }
    sell_row_part<NB, CS, true, kEgNt, TH, XT>(cols, w, x, cs, off, gw, lane, LD, rs0, a);
    if constexpr (LATE) {
#pragma unroll
        for (int q = 0; q < TH; ++q) load_c<C>(coef + (size_t)e1[q] * LD + t * C, c1[q]);
    }
#pragma unroll
    for (int q = 0; q < TH; ++q)
#pragma unroll
        for (int c = 0; c < C; ++c) y[q][c] = c1[q][c] * a[q][c];
#pragma unroll
    for (int q = 0; q < TH; ++q) {
        const int row = min(slice * kWave + (rs0 + q) * G + g, nrows - 1);
        if constexpr (LATE) e1[q] = e12[2 * row + 1];
        else load_c<C>(coef + (size_t)e12[2 * row + 1] * LD + t * C, c1[q]);
    }
    sell_row_part<NB, CS, true, kEgNt, TH, XT>(cols, w, x, cs, off + gw * kWave, gw, lane, LD, rs0, a);
    if constexpr (LATE) {
#pragma unroll
        for (int q = 0; q < TH; ++q) load_c<C>(coef + (size_t)e1[q] * LD + t * C, c1[q]);
    }
#pragma unroll
    for (int q = 0; q < TH; ++q)
#pragma unroll
        for (int c = 0; c < C; ++c) y[q][c] = fma(c1[q][c], a[q][c], y[q][c]);
}

template <int NB, bool DOT, bool kEgNt = false, typename XT = double>
__global__ __launch_bounds__(kBlock) void eg_pair_spmm_kernel(
    int nrows, int nslices, int gw, const int* __restrict__ cols1, const double* __restrict__ w1,
    const int* __restrict__ e12, const double* __restrict__ coef, const int* __restrict__ off2,
    const int* __restrict__ cols2, const double* __restrict__ vals2, const XT* __restrict__ x1,
    const typename ident<XT>::type* __restrict__ x2, double* __restrict__ y,
    const typename ident<XT>::type* __restrict__ dot_with, double* __restrict__ partial, int ld) {
    constexpr int C = Lay<NB>::C, T = Lay<NB>::T, G = Lay<NB>::G;
    const int LD = row_ld<NB>(ld);
    {
        const int c0 = col0<NB>();
        coef += c0; x1 += c0; x2 += c0; y += c0;
        if constexpr (DOT) { dot_with += c0; partial += c0; }
    }
    const int lane = threadIdx.x & (kWave - 1);
    const int g = lane / T, t = lane % T;
    double p[C];
#pragma unroll
    for (int c = 0; c < C; ++c) p[c] = 0.0;
    constexpr int TH = EgPass<NB>::TH;
    const SliceWalk sw = slice_walk(nslices);
    for (int slice = sw.begin; slice < sw.end; slice += sw.stride) {
        const int o2 = off2[slice];
        const int w2 = (off2[slice + 1] - o2) >> 6;
#pragma unroll 1
        for (int rs0 = 0; rs0 < T; rs0 += TH) {
            double acc[TH][C];
            eg_row_product<NB, false, kEgNt, TH, XT>(cols1, w1, e12, coef, gw, x1, nullptr, nrows, slice, lane, LD, rs0, acc);
            sell_row_part<NB, false, false, kEgNt, TH, XT>(cols2, vals2, x2, nullptr, o2, w2, lane, LD, rs0, acc);
            // the dot operand of all TH row steps is requested at once (rows past the end re-read the last row): the guarded
            // per-row-step form left the compiler one load - wait - fma chain per row step
            double wv[DOT ? TH : 1][C];
            if constexpr (DOT) {
#pragma unroll
                for (int q = 0; q < TH; ++q) {
                    const int rowc = min(slice * kWave + (rs0 + q) * G + g, nrows - 1);
                    load_v<C>(dot_with + (size_t)rowc * LD + t * C, wv[q]);
                }
                pin_block(wv);
            }
#pragma unroll
            for (int q = 0; q < TH; ++q) {
                const int row = slice * kWave + (rs0 + q) * G + g;
                if (row >= nrows) continue;
                const size_t at = (size_t)row * LD + t * C;
                store_c_stream<kEgNt, C>(y + at, acc[q]);
                if constexpr (DOT) {
#pragma unroll
                    for (int c = 0; c < C; ++c) p[c] = fma(wv[q][c], acc[q][c], p[c]);
                }
            }
        }
    }
    if constexpr (DOT) reduce_cols_store<NB>(p, partial, LD);
}

template <int NB, bool DOT, bool kEgNt = false, typename OT = double>
__global__ __launch_bounds__(kBlock) void eg_poly2_kernel(int nrows, int nslices, int gw, const int* __restrict__ cols,
                                                          const double* __restrict__ w, const int* __restrict__ e12,
                                                          const double* __restrict__ coef,
                                                          const double* __restrict__ dinv, const double* __restrict__ r,
                                                          OT* __restrict__ xout, double c0, double c1,
                                                          double* __restrict__ partial, int ld) {
    constexpr int C = Lay<NB>::C, T = Lay<NB>::T, G = Lay<NB>::G;
    const int LD = row_ld<NB>(ld);
    {
        const int g0 = col0<NB>();
        coef += g0; dinv += g0; r += g0; xout += g0;
        if constexpr (DOT) partial += g0;
    }
    const int lane = threadIdx.x & (kWave - 1);
    const int g = lane / T, t = lane % T;
    double p[C];
#pragma unroll
    for (int c = 0; c < C; ++c) p[c] = 0.0;
    constexpr int TH = EgPass<NB>::TH;
    const SliceWalk sw = slice_walk(nslices);
    for (int slice = sw.begin; slice < sw.end; slice += sw.stride) {
#pragma unroll 1
        for (int rs0 = 0; rs0 < T; rs0 += TH) {
            double acc[TH][C];
            eg_row_product<NB, true, kEgNt, TH>(cols, w, e12, coef, gw, r, dinv, nrows, slice, lane, LD, rs0, acc);
            // own-row reads of all TH row steps at once (see eg_pair_spmm_kernel)
            double rv[TH][C], di[TH][C];
#pragma unroll
            for (int q = 0; q < TH; ++q) {
                const size_t atc = (size_t)min(slice * kWave + (rs0 + q) * G + g, nrows - 1) * LD + t * C;
                load_c<C>(r + atc, rv[q]);
                load_c<C>(dinv + atc, di[q]);
            }
            pin_block(rv);
            pin_block(di);
#pragma unroll
            for (int q = 0; q < TH; ++q) {
                const int row = slice * kWave + (rs0 + q) * G + g;
                if (row >= nrows) continue;
                const size_t at = (size_t)row * LD + t * C;
                double xv[C];
#pragma unroll
                for (int c = 0; c < C; ++c) xv[c] = di[q][c] * (c0 * rv[q][c] - c1 * acc[q][c]);
                round_to<OT>(xv);
                if constexpr (DOT) {
#pragma unroll
                    for (int c = 0; c < C; ++c) p[c] = fma(rv[q][c], xv[c], p[c]);
                }
                store_v_stream<kEgNt, C>(xout + at, xv);
            }
        }
    }
    if constexpr (DOT) reduce_cols_store<NB>(p, partial, LD);
}

// out[slot][k] = vals[slot][k] * colscale[cols[slot]][k]   (per-realization column scaling A(k) D(k)^-1)
template <int NB>
__global__ __launch_bounds__(kBlock) void scale_cols_bv_kernel(size_t nflat, const int* __restrict__ cols,
                                                               const double* __restrict__ vals,
                                                               const double* __restrict__ colscale,
                                                               double* __restrict__ out, int ld) {
    constexpr int C = Lay<NB>::C;
    const int W = row_ld<NB>(ld);   // flat kernel: a wide batch is simply a wider row
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < nflat; i += (size_t)gridDim.x * kBlock) {
        const size_t e = i * C;
        const size_t slot = e / W;
        const int k0 = (int)(e % W);
        double v[C], sc[C];
        load_c<C>(vals + e, v);
        load_c<C>(colscale + (size_t)cols[slot] * W + k0, sc);
#pragma unroll
        for (int c = 0; c < C; ++c) v[c] *= sc[c];
        store_c<C>(out + e, v);
    }
}

// fp32 copies for the preconditioner kernels: out_scaled[slot][k] = (float)(vals[slot][k] * colscale[cols[slot]][k]),
// out_vals[slot][k] = (float)vals[slot][k]
template <int NB>
__global__ __launch_bounds__(kBlock) void scale_cols_bv32_kernel(size_t nflat, const int* __restrict__ cols,
                                                                 const double* __restrict__ vals,
                                                                 const double* __restrict__ colscale,
                                                                 float* __restrict__ out_scaled, float* __restrict__ out_vals,
                                                                 int ld) {
    constexpr int C = Lay<NB>::C;
    const int W = row_ld<NB>(ld);
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < nflat; i += (size_t)gridDim.x * kBlock) {
        const size_t e = i * C;
        const size_t slot = e / W;
        const int k0 = (int)(e % W);
        double v[C], sc[C];
        load_c<C>(vals + e, v);
        load_c<C>(colscale + (size_t)cols[slot] * W + k0, sc);
#pragma unroll
        for (int c = 0; c < C; ++c) {
            out_vals[e + c] = (float)v[c];
            out_scaled[e + c] = (float)(v[c] * sc[c]);
        }
    }
}

// MINRES w / x update restricted to an index list of rows: w, x are compact [nsel][NB], u is full
template <int NB, typename UT>
__global__ __launch_bounds__(kBlock) void minres_wx_idx_kernel(size_t nflat, const int* __restrict__ rows,
                                                               const double* __restrict__ c0, const UT* __restrict__ u,
                                                               const double* __restrict__ c1, double* __restrict__ w0,
                                                               const double* __restrict__ c2, const double* __restrict__ w1,
                                                               const double* __restrict__ c3, double* __restrict__ x, int ld) {
    constexpr int C = Lay<NB>::C;
    const int W = row_ld<NB>(ld);
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= nflat) return;
    const size_t e = i * C;
    const size_t sel = e / W;
    const int k0 = (int)(e % W);
    double uv[C], w0v[C], w1v[C], xv[C];
    load_v<C>(u + (size_t)rows[sel] * W + k0, uv);
    load_c<C>(w0 + e, w0v);
    load_c<C>(w1 + e, w1v);
    load_c<C>(x + e, xv);
#pragma unroll
    for (int c = 0; c < C; ++c) {
        w0v[c] = c0[k0 + c] * uv[c] + c1[k0 + c] * w0v[c] + c2[k0 + c] * w1v[c];
        xv[c] += c3[k0 + c] * w0v[c];
    }
    store_c<C>(w0 + e, w0v);
    store_c<C>(x + e, xv);
}

// ---- flat element-wise kernels: thread i owns the C doubles at flat index i*C, i.e. row (i*C)/NB and
// columns ((i*C) % NB) + c; consecutive lanes touch consecutive 16 B -> fully coalesced.  Grid-stride:
// the stride gridDim*256 is a multiple of T, so a thread keeps its column pair.
template <int NB, bool BV, bool DOT>
__global__ __launch_bounds__(kBlock) void cheb_first_kernel(size_t nflat, const double* __restrict__ dinv,
                                                            const double* __restrict__ r, double* __restrict__ d,
                                                            double* __restrict__ x, double b,
                                                            double* __restrict__ partial, int ld) {
    constexpr int C = Lay<NB>::C;
    const int W = row_ld<NB>(ld);
    double p[C];
#pragma unroll
    for (int c = 0; c < C; ++c) p[c] = 0.0;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < nflat; i += (size_t)gridDim.x * kBlock) {
        const size_t e = i * C;
        double rv[C], di[C], xv[C];
        load_c<C>(r + e, rv);
        if constexpr (BV) {
            load_c<C>(dinv + e, di);
        } else {
            const double s = dinv[e / W];
#pragma unroll
            for (int c = 0; c < C; ++c) di[c] = s;
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            xv[c] = b * di[c] * rv[c];
            if constexpr (DOT) p[c] = fma(rv[c], xv[c], p[c]);
        }
        store_c<C>(d + e, xv);
        store_c<C>(x + e, xv);
    }
    if constexpr (DOT) reduce_flat_store<NB>(p, partial, W);
}

template <int NB, typename BT = double>
__global__ __launch_bounds__(kBlock) void dot_kernel(size_t nflat, const double* __restrict__ a,
                                                     const BT* __restrict__ b, double* __restrict__ partial, int ld) {
    constexpr int C = Lay<NB>::C;
    double p[C];
#pragma unroll
    for (int c = 0; c < C; ++c) p[c] = 0.0;
    // grid-stride: the stride gridDim*256 is a multiple of T, so a thread keeps its column pair
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < nflat; i += (size_t)gridDim.x * kBlock) {
        double av[C], bv[C];
        load_c<C>(a + i * C, av);
        load_v<C>(b + i * C, bv);
#pragma unroll
        for (int c = 0; c < C; ++c) p[c] = fma(av[c], bv[c], p[c]);
    }
    reduce_flat_store<NB>(p, partial, row_ld<NB>(ld));
}

// z = storage-rounded copy of a preconditioner result that a kernel without a typed output left in fp64, with the fused
// <r, z> of the rounded values (the paths off the hot configurations: higher-degree smoothers, algebraic transfers)
template <int NB, bool DOT, typename OT>
__global__ __launch_bounds__(kBlock) void convert_dot_kernel(size_t nflat, const double* __restrict__ in,
                                                             OT* __restrict__ out, const double* __restrict__ r,
                                                             double* __restrict__ partial, int ld) {
    constexpr int C = Lay<NB>::C;
    double p[C];
#pragma unroll
    for (int c = 0; c < C; ++c) p[c] = 0.0;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < nflat; i += (size_t)gridDim.x * kBlock) {
        double v[C];
        load_c<C>(in + i * C, v);
        round_to<OT>(v);
        if constexpr (DOT) {
            double rv[C];
            load_c<C>(r + i * C, rv);
#pragma unroll
            for (int c = 0; c < C; ++c) p[c] = fma(rv[c], v[c], p[c]);
        }
        store_v<C>(out + i * C, v);
    }
    if constexpr (DOT) reduce_flat_store<NB>(p, partial, row_ld<NB>(ld));
}

// y32 (optional): the result is also written in fp32 - the copy the V-cycle's first two kernels gather and read (k::vc_*_r32)
template <int NB, bool NT = false>
__global__ __launch_bounds__(kBlock) void lincomb3_kernel(size_t nflat, const double* __restrict__ c0,
                                                          const double* __restrict__ a, const double* __restrict__ c1,
                                                          const double* __restrict__ b, const double* __restrict__ c2,
                                                          double* __restrict__ y, int ld, float* __restrict__ y32 = nullptr) {
    constexpr int C = Lay<NB>::C;
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= nflat) return;
    const size_t e = i * C;
    const int k0 = (int)(e % row_ld<NB>(ld));
    double av[C], bv[C], yv[C];
    load_c_nt<NT, C>(a + e, av);   // the three inputs are read once; the result is gathered by the next kernels
    load_c_nt<NT, C>(b + e, bv);
    load_c_nt<NT, C>(y + e, yv);
#pragma unroll
    for (int c = 0; c < C; ++c) yv[c] = c0[k0 + c] * av[c] + c1[k0 + c] * bv[c] + c2[k0 + c] * yv[c];
    store_c<C>(y + e, yv);
    if (y32) store_v<C>(y32 + e, yv);
}

template <int NB, bool NT, typename UT>
__global__ __launch_bounds__(kBlock) void minres_wx_kernel(size_t nflat, const double* __restrict__ c0,
                                                           const UT* __restrict__ u, const double* __restrict__ c1,
                                                           double* __restrict__ w0, const double* __restrict__ c2,
                                                           const double* __restrict__ w1, const double* __restrict__ c3,
                                                           double* __restrict__ x, int ld) {
    constexpr int C = Lay<NB>::C;
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= nflat) return;
    const size_t e = i * C;
    const int k0 = (int)(e % row_ld<NB>(ld));
    double uv[C], w0v[C], w1v[C], xv[C];
    load_v_nt<NT, C>(u + e, uv);
    load_c_nt<NT, C>(w0 + e, w0v);
    load_c_nt<NT, C>(w1 + e, w1v);
    load_c_nt<NT, C>(x + e, xv);
#pragma unroll
    for (int c = 0; c < C; ++c) {
        w0v[c] = c0[k0 + c] * uv[c] + c1[k0 + c] * w0v[c] + c2[k0 + c] * w1v[c];
        xv[c] += c3[k0 + c] * w0v[c];
    }
    store_c_nt<NT, C>(w0 + e, w0v);
    store_c_nt<NT, C>(x + e, xv);
}

// The w / x updates of B.cnt <= kWxDefer iterations in one pass (see kWxDefer): the u vectors of all pending iterations are
// requested first, the recurrences then run in registers in iteration order - the same operations in the same order as
// cnt successive minres_wx launches.
template <int NB, bool NT, typename UT>
__global__ __launch_bounds__(kBlock) void minres_wx_deferred_kernel(size_t nflat, k::WxDeferred B,
                                                                    const double* __restrict__ cW, double* __restrict__ w0,
                                                                    double* __restrict__ w1, double* __restrict__ x, int ld) {
    constexpr int C = Lay<NB>::C;
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= nflat) return;
    const size_t e = i * C;
    const int k0 = (int)(e % row_ld<NB>(ld));
    double uv[k::kWxDefer][C], a[C], b[C], xv[C];
#pragma unroll
    for (int j = 0; j < k::kWxDefer; ++j)
        if (j < B.cnt) load_v_nt<NT, C>(static_cast<const UT*>(B.u[j]) + e, uv[j]);
    load_c_nt<NT, C>(w0 + e, a);
    load_c_nt<NT, C>(w1 + e, b);
    load_c_nt<NT, C>(x + e, xv);
#pragma unroll
    for (int j = 0; j < k::kWxDefer; ++j) {
        if (j < B.cnt) {
            const double* cj = cW + (size_t)B.slot[j] * 4 * kMaxBatch + k0;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const double wn = cj[c] * uv[j][c] + cj[kMaxBatch + c] * a[c] + cj[2 * kMaxBatch + c] * b[c];
                xv[c] += cj[3 * kMaxBatch + c] * wn;
                a[c] = b[c];
                b[c] = wn;
            }
        }
    }
    store_c_nt<NT, C>(w0 + e, a);
    store_c_nt<NT, C>(w1 + e, b);
    store_c_nt<NT, C>(x + e, xv);
}

// partial sums of <w, x[:,k]> with a shared (non-batched) weight vector w   (K15 QoI)
template <int NB>
__global__ __launch_bounds__(kBlock) void wdot_kernel(size_t nflat, const double* __restrict__ w,
                                                      const double* __restrict__ x, double* __restrict__ partial, int ld) {
    constexpr int C = Lay<NB>::C;
    const int W = row_ld<NB>(ld);
    double p[C];
#pragma unroll
    for (int c = 0; c < C; ++c) p[c] = 0.0;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < nflat; i += (size_t)gridDim.x * kBlock) {
        double xv[C];
        load_c<C>(x + i * C, xv);
        const double ww = w[(i * C) / W];
#pragma unroll
        for (int c = 0; c < C; ++c) p[c] = fma(ww, xv[c], p[c]);
    }
    reduce_flat_store<NB>(p, partial, W);
}

__global__ void fill_kernel(size_t n, double* __restrict__ x, double v) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) x[i] = v;
}

// ------------------------------------------------------------------------------------------
// MINRES scalar recurrences.  One block of 256 threads; thread k < nb owns column k.  Restates the
// preconditioned MINRES of Paige & Saunders in the form MFEM's MINRESSolver uses (normalised Lanczos
// vectors); the vectors are kept UNnormalised here and the 1/beta factors are folded into the
// update coefficients.
//
// Column sums of the per-block partials: thread t reads column t % nb of blocks t / nb, t / nb + 256/nb, ...
// (coalesced), then thread k adds the 256/nb group sums of its column in a fixed order (deterministic).
static constexpr int kScalBlock = 1024;
// Two segments (partial: nblocks blocks, partial2: nblocks2 blocks) are summed as one concatenated list: kernels that
// run side by side on two streams each write their own segment.
__device__ __forceinline__ double reduce_partials(const double* __restrict__ partial, int nblocks, int nb,
                                                  const double* __restrict__ partial2 = nullptr, int nblocks2 = 0) {
    __shared__ double lds[kScalBlock];
    const int k = threadIdx.x % nb, q = threadIdx.x / nb, nq = kScalBlock / nb;
    // eight independent chains keep the loads in flight (the partials of a 2 000-block launch are 36 values per thread)
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, s4 = 0.0, s5 = 0.0, s6 = 0.0, s7 = 0.0;
    int b = q;
    for (; b + 7 * nq < nblocks; b += 8 * nq) {
        s0 += partial[(size_t)b * nb + k];
        s1 += partial[(size_t)(b + nq) * nb + k];
        s2 += partial[(size_t)(b + 2 * nq) * nb + k];
        s3 += partial[(size_t)(b + 3 * nq) * nb + k];
        s4 += partial[(size_t)(b + 4 * nq) * nb + k];
        s5 += partial[(size_t)(b + 5 * nq) * nb + k];
        s6 += partial[(size_t)(b + 6 * nq) * nb + k];
        s7 += partial[(size_t)(b + 7 * nq) * nb + k];
    }
    for (; b < nblocks; b += nq) s0 += partial[(size_t)b * nb + k];
    for (b = q; b + 3 * nq < nblocks2; b += 4 * nq) {
        s4 += partial2[(size_t)b * nb + k];
        s5 += partial2[(size_t)(b + nq) * nb + k];
        s6 += partial2[(size_t)(b + 2 * nq) * nb + k];
        s7 += partial2[(size_t)(b + 3 * nq) * nb + k];
    }
    for (; b < nblocks2; b += nq) s1 += partial2[(size_t)b * nb + k];
    s0 += s4;
    s1 += s5;
    s2 += s6;
    s3 += s7;
    const double s = (s0 + s1) + (s2 + s3);
    __syncthreads();
    lds[threadIdx.x] = s;
    __syncthreads();
    // narrow batches (nb < 32: the one-realization-per-call path hands over ONE column) leave 64 ... 1 024 groups per
    // column: fold them pairwise down to 32 before the serial sum - at nb = 1 thread 0 otherwise walks 1 024 dependent LDS
    // reads, 26 us per MINRES iteration of a 215 us iteration (round 5).  nb >= 32 takes the loop below unchanged.
    int groups = nq;
    for (int stride = nq >> 1; stride >= 32; stride >>= 1) {
        if (q < stride) lds[threadIdx.x] += lds[threadIdx.x + stride * nb];
        __syncthreads();
        groups = stride;
    }
    double t = 0.0;
    if ((int)threadIdx.x < nb)
        for (int g = 0; g < groups; ++g) t += lds[g * nb + threadIdx.x];
    return t;   // valid for threadIdx.x < nb
}

__device__ __forceinline__ void count_active(k::MinresState* st, int nb, bool bump) {
    __syncthreads();
    const int na = __syncthreads_count((int)threadIdx.x < nb && st->active[threadIdx.x] != 0);
    if (threadIdx.x == 0) {
        st->n_active = na;
        st->it = bump ? st->it + 1 : 0;
    }
}

__global__ __launch_bounds__(kScalBlock) void minres_init_kernel(k::MinresState* st, const double* __restrict__ partial,
                                                             int nblocks, int nb, double rel_tol, double abs_tol,
                                                             const double* __restrict__ partial2, int nblocks2, int ring) {
    if (threadIdx.x == 0) st->ring = ring;
    const double d = reduce_partials(partial, nblocks, nb, partial2, nblocks2);
    const int k = threadIdx.x;
    if (k < nb) {
        const double beta = d > 0.0 ? sqrt(d) : 0.0;
        st->beta[k] = beta;
        st->beta_old[k] = 1.0;
        st->eta[k] = beta;
        st->eta0[k] = beta;
        st->gamma0[k] = st->gamma1[k] = 1.0;
        st->sigma0[k] = st->sigma1[k] = 0.0;
        st->goal[k] = fmax(rel_tol * beta, abs_tol);
        st->iters[k] = 0;
        st->flag[k] = (d < 0.0 || d != d) ? -1 : 0;   // preconditioner not SPD / NaN
        st->active[k] = (beta > st->goal[k] && st->flag[k] == 0) ? 1 : 0;
    }
    count_active(st, nb, false);
}

// after q = A u1 and d1 = <u1, q>   (thread k < nb owns column k)
__device__ __forceinline__ void scal1_body(k::MinresState* st, int k, double d1) {
    if (st->active[k]) {
        const double beta = st->beta[k];
        const double ib = 1.0 / beta;
        const double alpha = d1 * ib * ib;
        st->alpha[k] = alpha;
        st->cV[0][k] = ib;                       // q / beta
        st->cV[1][k] = -alpha * ib;              // - alpha v1
        st->cV[2][k] = -beta / st->beta_old[k];  // - beta v0
        st->delta[k] = st->gamma1[k] * alpha - st->gamma0[k] * st->sigma1[k] * beta;
        st->rho3[k] = st->sigma0[k] * beta;
        st->rho2[k] = st->sigma1[k] * alpha + st->gamma0[k] * st->gamma1[k] * beta;
    } else {
        st->cV[0][k] = st->cV[1][k] = st->cV[2][k] = 0.0;
    }
}
// after z_new = prec(v_new) and d2 = <v_new, z_new>
__device__ __forceinline__ void scal2_body(k::MinresState* st, int k, double d2) {
    if (st->active[k]) {
        if (d2 < 0.0 || d2 != d2) st->flag[k] = -1;
        const double beta_new = d2 > 0.0 ? sqrt(d2) : 0.0;
        const double delta = st->delta[k];
        const double rho1 = hypot(delta, beta_new);
        const double ir = rho1 > 0.0 ? 1.0 / rho1 : 0.0;
        double (*cW)[kMaxBatch] = st->cW[st->it % st->ring];   // this iteration's coefficient set
        cW[0][k] = ir / st->beta[k];
        cW[1][k] = -st->rho3[k] * ir;
        cW[2][k] = -st->rho2[k] * ir;
        st->gamma0[k] = st->gamma1[k];
        st->gamma1[k] = delta * ir;
        cW[3][k] = st->gamma1[k] * st->eta[k];
        st->sigma0[k] = st->sigma1[k];
        st->sigma1[k] = beta_new * ir;
        st->eta[k] = -st->sigma1[k] * st->eta[k];
        st->beta_old[k] = st->beta[k];
        st->beta[k] = beta_new;
        st->iters[k] = st->it + 1;
        if (fabs(st->eta[k]) <= st->goal[k] || beta_new == 0.0 || st->flag[k] != 0) st->active[k] = 0;
    } else {
        double (*cW)[kMaxBatch] = st->cW[st->it % st->ring];
        cW[0][k] = cW[1][k] = cW[2][k] = cW[3][k] = 0.0;
    }
}

__global__ __launch_bounds__(kScalBlock) void minres_scal1_kernel(k::MinresState* st, const double* __restrict__ partial,
                                                              int nblocks, int nb, const double* __restrict__ partial2,
                                                              int nblocks2) {
    const double d1 = reduce_partials(partial, nblocks, nb, partial2, nblocks2);
    if ((int)threadIdx.x < nb) scal1_body(st, threadIdx.x, d1);
}

__global__ __launch_bounds__(kScalBlock) void minres_scal2_kernel(k::MinresState* st, const double* __restrict__ partial,
                                                              int nblocks, int nb, const double* __restrict__ partial2,
                                                              int nblocks2) {
    const double d2 = reduce_partials(partial, nblocks, nb, partial2, nblocks2);
    if ((int)threadIdx.x < nb) scal2_body(st, threadIdx.x, d2);
    count_active(st, nb, true);
}

// Both scalar steps in one launch: the recurrences of iteration i (from <v_new, z_new>) and, with the operator product
// of iteration i + 1 already done, the first half of iteration i + 1 (from <z_new, A z_new>).  One single-block launch
// per iteration instead of two.
__global__ __launch_bounds__(kScalBlock) void minres_scal21_kernel(k::MinresState* st, const double* __restrict__ pa,
                                                               int na, const double* __restrict__ pa2, int na2,
                                                               const double* __restrict__ pb, int nbk,
                                                               const double* __restrict__ pb2, int nbk2, int nb) {
    const double d2 = reduce_partials(pa, na, nb, pa2, na2);
    const double d1 = reduce_partials(pb, nbk, nb, pb2, nbk2);
    if ((int)threadIdx.x < nb) {
        scal2_body(st, threadIdx.x, d2);
        scal1_body(st, threadIdx.x, d1);          // same thread, same column: sees the state scal2 has just written
    }
    count_active(st, nb, true);
}

// First stage for long partial lists (a fine-level iteration hands ~4 600 blocks x nb values to the single-block scalar
// kernel, which then spends ~19 us on load latency alone): kStageBlocks workgroups sum contiguous chunks of the two dot
// products' lists into stage[list][g][nb]; the scalar kernel adds the kStageBlocks chunk sums in index order.  Fixed chunking
// and fixed order: deterministic.
static constexpr int kStageBlocks = 64;
__global__ __launch_bounds__(256) void stage_partials_kernel(const double* __restrict__ pa, int na,
                                                             const double* __restrict__ pa2, int na2,
                                                             const double* __restrict__ pb, int nbk,
                                                             const double* __restrict__ pb2, int nbk2, int nb,
                                                             double* __restrict__ stage) {
    __shared__ double lds[2][256];
    const int g = blockIdx.x, k = threadIdx.x % nb, q = threadIdx.x / nb, nq = 256 / nb;
    auto chunk = [&](const double* __restrict__ p, int n) {
        const int per = (n + kStageBlocks - 1) / kStageBlocks;
        const int lo = min(n, g * per), hi = min(n, lo + per);
        double a = 0.0, b = 0.0;
        int i = lo + q;
        for (; i + nq < hi; i += 2 * nq) {
            a += p[(size_t)i * nb + k];
            b += p[(size_t)(i + nq) * nb + k];
        }
        if (i < hi) a += p[(size_t)i * nb + k];
        return a + b;
    };
    lds[0][threadIdx.x] = chunk(pa, na) + chunk(pa2, na2);
    lds[1][threadIdx.x] = chunk(pb, nbk) + chunk(pb2, nbk2);
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * nb; e += 256) {   // nb up to 256: two entries per thread
        const int list = e / nb, c = e % nb;
        double t = 0.0;
        for (int j = 0; j < nq; ++j) t += lds[list][j * nb + c];
        stage[((size_t)list * kStageBlocks + g) * nb + c] = t;
    }
}

// ------------------------------------------------------------------------------------------
// Philox4x32-10 + AS241 inverse normal CDF (bit-level twin: oracle/rng_oracle.py)
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    c[0] = hi1 ^ c[1] ^ k0;
    c[1] = lo1;
    c[2] = hi0 ^ c[3] ^ k1;
    c[3] = lo0;
}

__device__ __forceinline__ double u01_open(uint32_t hi, uint32_t lo) {
    // 52 bits: (m + 1/2) / 2^52 is exact, strictly inside (0,1)
    const uint64_t m = ((uint64_t)(hi >> 6) << 26) + (uint64_t)(lo >> 6);
    return ((double)m + 0.5) * (1.0 / 4503599627370496.0);
}

#pragma clang fp contract(off)
__device__ double inv_normal_cdf(double p) {
    const double q = p - 0.5;
    if (fabs(q) <= 0.425) {
        const double r = 0.180625 - q * q;
        const double num = (((((((2.5090809287301226727e+3 * r + 3.3430575583588128105e+4) * r +
                                 6.7265770927008700853e+4) * r + 4.5921953931549871457e+4) * r +
                               1.3731693765509461125e+4) * r + 1.9715909503065514427e+3) * r +
                             1.3314166789178437745e+2) * r + 3.3871328727963666080e0);
        const double den = (((((((5.2264952788528545610e+3 * r + 2.8729085735721942674e+4) * r +
                                 3.9307895800092710610e+4) * r + 2.1213794301586595867e+4) * r +
                               5.3941960214247511077e+3) * r + 6.8718700749205790830e+2) * r +
                             4.2313330701600911252e+1) * r + 1.0);
        return q * num / den;
    }
    double r = q < 0.0 ? p : 1.0 - p;
    r = sqrt(-log(r));
    double val;
    if (r <= 5.0) {
        r -= 1.6;
        const double num = (((((((7.74545014278341407640e-4 * r + 2.27238449892691845833e-2) * r +
                                 2.41780725177450611770e-1) * r + 1.27045825245236838258e0) * r +
                               3.64784832476320460504e0) * r + 5.76949722146069140550e0) * r +
                             4.63033784615654529590e0) * r + 1.42343711074968357734e0);
        const double den = (((((((1.05075007164441684324e-9 * r + 5.47593808499534494600e-4) * r +
                                 1.51986665636164571966e-2) * r + 1.48103976427480074590e-1) * r +
                               6.89767334985100004550e-1) * r + 1.67638483018380384940e0) * r +
                             2.05319162663775882187e0) * r + 1.0);
        val = num / den;
    } else {
        r -= 5.0;
        const double num = (((((((2.01033439929228813265e-7 * r + 2.71155556874348757815e-5) * r +
                                 1.24266094738807843860e-3) * r + 2.65321895265761230930e-2) * r +
                               2.96560571828504891230e-1) * r + 1.78482653991729133580e0) * r +
                             5.46378491116411436990e0) * r + 6.65790464350110377720e0);
        const double den = (((((((2.04426310338993978564e-15 * r + 1.42151175831644588870e-7) * r +
                                 1.84631831751005468180e-5) * r + 7.86869131145613259100e-4) * r +
                               1.48753612908506148525e-2) * r + 1.36929880922735805310e-1) * r +
                             5.99832206555887937690e-1) * r + 1.0);
        val = num / den;
    }
    return q < 0.0 ? -val : val;
}
#pragma clang fp contract(fast)

// out[b*n + i], sample-major.  One thread per (pair of elements, realization).
// Realization b of the launch is the generator's realization first_id + b * id_stride (id_stride = nparts of a split
// generator: part p owns the ids p, p + nparts, ...).
__global__ __launch_bounds__(kBlock) void normal_fill_kernel(int n, int nbatch, uint64_t seed, uint64_t first_id,
                                                             uint64_t id_stride, uint32_t stream, double mean, double sigma,
                                                             double* __restrict__ out) {
    const int npair = (n + 1) >> 1;
    const int j = blockIdx.x * kBlock + threadIdx.x;
    const int b = blockIdx.y;
    if (j >= npair || b >= nbatch) return;
    const uint64_t sid = first_id + (uint64_t)b * id_stride;
    uint32_t c[4] = {(uint32_t)j, (uint32_t)sid, (uint32_t)(sid >> 32), stream};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    double* o = out + (size_t)b * n + 2 * (size_t)j;
    o[0] = mean + sigma * inv_normal_cdf(u01_open(c[0], c[1]));
    if (2 * j + 1 < n) o[1] = mean + sigma * inv_normal_cdf(u01_open(c[2], c[3]));
}

// ------------------------------------------------------------------------------------------
// layout changes fused with the sampler's pointwise maps
// out[i*NB + k] = scale * in[k*n + i] * (w ? w[i] : 1)        (K2: rhs_s = -g W^{1/2} xi)
// src (optional): row i of the result takes row src[i] of the input (a renumbering of the rows)
template <int NB>
__global__ __launch_bounds__(kBlock) void interleave_kernel(int n, const double* __restrict__ in,
                                                            const double* __restrict__ w, double scale,
                                                            double* __restrict__ out, int ld, const int* __restrict__ src) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int LD = row_ld<NB>(ld), c0 = col0<NB>();
    const double f = w ? scale * w[i] : scale;
    const int is = src ? src[i] : i;
    double v[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) v[k] = f * in[(size_t)(c0 + k) * n + is];
    store_row<NB>(out + (size_t)i * LD + c0, v);
}

// out[k*m + i] = post( rowscale[i] * in[idx ? idx[i] : i][k] ),  post = exp if do_exp   (K9, K10)
template <int NB>
__global__ __launch_bounds__(kBlock) void deinterleave_kernel(int m, const double* __restrict__ in,
                                                              const int* __restrict__ idx,
                                                              const double* __restrict__ rowscale, int do_exp,
                                                              double* __restrict__ out, int ld) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= m) return;
    const int LD = row_ld<NB>(ld), c0 = col0<NB>();
    const int src = idx ? idx[i] : i;
    double v[NB];
    load_row<NB>(in + (size_t)src * LD + c0, v);
    const double f = rowscale ? rowscale[i] : 1.0;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        double t = f * v[k];
        if (do_exp) t = exp(t);
        out[(size_t)(c0 + k) * m + i] = t;
    }
}

// ------------------------------------------------------------------------------------------
// Darcy per-sample numeric refresh (K12-K14), batched values: arrays are [slot][NB].
// coef[e*NB+k] = 1/k or k.
template <int NB>
__global__ __launch_bounds__(kBlock) void darcy_coef_kernel(int n, const double* __restrict__ kfield, int k_divides,
                                                            double* __restrict__ coef, int ld) {
    // kfield sample-major [nb][n] -> interleaved coefficient
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int LD = row_ld<NB>(ld), c0 = col0<NB>();
    double v[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const double t = kfield[(size_t)(c0 + k) * n + i];
        v[k] = k_divides ? 1.0 / t : t;
    }
    store_row<NB>(coef + (size_t)i * LD + c0, v);
}

// One lane per row of the SELL-stored M: raw values from element contributions, essential
// row/col elimination (DarcySolver.cpp:487-498), rhs fix-up, diagonal and l1 row sums.
template <int NB>
__global__ __launch_bounds__(kBlock) void darcy_assemble_kernel(
    int nrows, int nslices, const int* __restrict__ slice_off, const int* __restrict__ cols,
    const int* __restrict__ slot_src, const int* __restrict__ c_ptr, const int* __restrict__ c_elem,
    const double* __restrict__ c_val, const double* __restrict__ coef, const unsigned char* __restrict__ ess,
    const double* __restrict__ ess_data, const double* __restrict__ rhs0, double* __restrict__ mvals,
    double* __restrict__ diag, double* __restrict__ l1inv, double* __restrict__ rhs_bc, int ld) {
    const int LD = row_ld<NB>(ld);
    {
        const int c0 = col0<NB>();
        coef += c0; diag += c0; l1inv += c0; rhs_bc += c0;
        if (mvals) mvals += c0;
    }
    const int row = blockIdx.x * kBlock + threadIdx.x;
    const int slice = row >> 6, lane = row & 63;
    if (slice >= nslices) return;
    const int off = slice_off[slice];
    const int width = (slice_off[slice + 1] - off) >> 6;
    const bool live = row < nrows;
    const bool row_ess = live && ess[row];
    double dg[NB], l1[NB], fix[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) dg[k] = l1[k] = fix[k] = 0.0;
    int slot = off + lane;
    for (int j = 0; j < width; ++j, slot += kWave) {
        const int p = slot_src[slot];
        double v[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) v[k] = 0.0;
        const int c = cols[slot];
        if (p >= 0) {
            for (int t = c_ptr[p]; t < c_ptr[p + 1]; ++t) {
                const double cv = c_val[t];
                double ce[NB];
                load_row<NB>(coef + (size_t)c_elem[t] * LD, ce);
#pragma unroll
                for (int k = 0; k < NB; ++k) v[k] = fma(ce[k], cv, v[k]);
            }
            const bool col_ess = ess[c];
            if (row_ess || col_ess) {
                if (col_ess && !row_ess) {
                    const double dval = ess_data[c];
#pragma unroll
                    for (int k = 0; k < NB; ++k) fix[k] = fma(v[k], dval, fix[k]);
                }
                const double e = (c == row) ? 1.0 : 0.0;
#pragma unroll
                for (int k = 0; k < NB; ++k) v[k] = e;
            }
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                l1[k] += fabs(v[k]);
                if (c == row) dg[k] = v[k];
            }
        }
        if (mvals) store_row<NB>(mvals + (size_t)slot * LD, v);
    }
    if (!live) return;
    double rb[NB];
    const double r0 = row_ess ? ess_data[row] : rhs0[row];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        rb[k] = row_ess ? r0 : r0 - fix[k];
        l1[k] = 1.0 / l1[k];
    }
    store_row<NB>(diag + (size_t)row * LD, dg);
    store_row<NB>(l1inv + (size_t)row * LD, l1);
    store_row<NB>(rhs_bc + (size_t)row * LD, rb);
}

// Generic numeric refresh of a derived matrix on a fixed pattern:
//   out[slot][k] = sum_{t in ptr[slot]..ptr[slot+1]} w[t] * f(src[idx[t]][k]),  f = 1/x if recip else x
// used for S = B diag(M)^-1 B^T (recip, src = diag(M)) and for coarse S_c = 1/2 P^T S P.
// Also produces 1/diag of the derived matrix when dinv != nullptr (is_diag[slot] marks diagonal slots).
template <int NB>
__global__ __launch_bounds__(kBlock) void refresh_kernel(int64_t nslots, const int* __restrict__ ptr,
                                                         const int* __restrict__ idx, const double* __restrict__ w,
                                                         const double* __restrict__ src, int recip,
                                                         double* __restrict__ out, int ld) {
    const int64_t slot = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (slot >= nslots) return;
    const int LD = row_ld<NB>(ld);
    src += col0<NB>();
    out += col0<NB>();
    double acc[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) acc[k] = 0.0;
    for (int t = ptr[slot]; t < ptr[slot + 1]; ++t) {
        double s[NB];
        load_row<NB>(src + (size_t)idx[t] * LD, s);
        const double wt = w[t];
#pragma unroll
        for (int k = 0; k < NB; ++k) acc[k] = fma(wt, recip ? 1.0 / s[k] : s[k], acc[k]);
    }
    store_row<NB>(out + (size_t)slot * LD, acc);
}

// dinv[row][k] = 1 / vals[diag_slot[row]][k]
template <int NB>
__global__ __launch_bounds__(kBlock) void diag_inv_kernel(int n, const int* __restrict__ diag_slot,
                                                          const double* __restrict__ vals, double* __restrict__ dinv, int ld) {
    const int row = blockIdx.x * kBlock + threadIdx.x;
    if (row >= n) return;
    const int LD = row_ld<NB>(ld), c0 = col0<NB>();
    double v[NB];
    load_row<NB>(vals + (size_t)diag_slot[row] * LD + c0, v);
#pragma unroll
    for (int k = 0; k < NB; ++k) v[k] = 1.0 / v[k];
    store_row<NB>(dinv + (size_t)row * LD + c0, v);
}

// Per-realization Gershgorin bound of D^-1 S on batched values:  g[k] = max_i dinv[i][k] * sum_j |S_ij(k)|  (atomic max
// on the bit pattern of the non-negative doubles; g zeroed by the launcher), then dinv[i][k] /= 1.0001 g[k] so that the
// Chebyshev smoothers run on (0, 1] for every realization of the batch.
template <int NB>
__global__ __launch_bounds__(kBlock) void gersh_bv_kernel(int nrows, const int* __restrict__ slice_off,
                                                          const double* __restrict__ vals, const double* __restrict__ dinv,
                                                          unsigned long long* __restrict__ g, int ld) {
    const int row = blockIdx.x * kBlock + threadIdx.x;
    const int LD = row_ld<NB>(ld);
    {
        const int c0 = col0<NB>();
        vals += c0; dinv += c0; g += c0;
    }
    double acc[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) acc[k] = 0.0;
    if (row < nrows) {
        const int slice = row >> 6, lane = row & 63;
        const int off = slice_off[slice];
        const int width = (slice_off[slice + 1] - off) >> 6;
        for (int j = 0; j < width; ++j) {
            double v[NB];
            load_row<NB>(vals + ((size_t)off + (size_t)j * 64 + lane) * LD, v);
#pragma unroll
            for (int k = 0; k < NB; ++k) acc[k] += fabs(v[k]);
        }
        double d[NB];
        load_row<NB>(dinv + (size_t)row * LD, d);
#pragma unroll
        for (int k = 0; k < NB; ++k) acc[k] *= fabs(d[k]);
    }
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        double v = acc[k];
        for (int o = kWave / 2; o >= 1; o >>= 1) v = fmax(v, __shfl_xor(v, o, kWave));
        if ((threadIdx.x & (kWave - 1)) == 0) atomicMax(g + k, (unsigned long long)__double_as_longlong(v));
    }
}

template <int NB>
__global__ __launch_bounds__(kBlock) void gersh_scale_kernel(int nrows, const unsigned long long* __restrict__ g,
                                                             double* __restrict__ dinv, int ld) {
    const int row = blockIdx.x * kBlock + threadIdx.x;
    if (row >= nrows) return;
    const int LD = row_ld<NB>(ld), c0 = col0<NB>();
    double d[NB];
    load_row<NB>(dinv + (size_t)row * LD + c0, d);
#pragma unroll
    for (int k = 0; k < NB; ++k) d[k] /= 1.0001 * __longlong_as_double((long long)g[c0 + k]);
    store_row<NB>(dinv + (size_t)row * LD + c0, d);
}

// First stage of a two-stage reduction for launches with many partial blocks: block j of kCompressBlocks sums the input
// blocks j, j + kCompressBlocks, ... (fixed order: deterministic) into out[j][k].
static constexpr int kCompressBlocks = 256;   // dot_capacity() reserves this many blocks ahead of the uncompressed ones
__global__ __launch_bounds__(256) void compress_partials_kernel(const double* __restrict__ in, int nblocks, int nb,
                                                                double* __restrict__ out) {
    __shared__ double lds[256];
    const int k = threadIdx.x % nb, q = threadIdx.x / nb, nq = 256 / nb;
    double s0 = 0.0, s1 = 0.0;
    int b = blockIdx.x + kCompressBlocks * q;
    for (; b + kCompressBlocks * nq < nblocks; b += 2 * kCompressBlocks * nq) {
        s0 += in[(size_t)b * nb + k];
        s1 += in[(size_t)(b + kCompressBlocks * nq) * nb + k];
    }
    if (b < nblocks) s0 += in[(size_t)b * nb + k];
    lds[threadIdx.x] = s0 + s1;
    __syncthreads();
    if ((int)threadIdx.x < nb) {
        double t = 0.0;
        for (int g = 0; g < nq; ++g) t += lds[g * nb + threadIdx.x];
        out[(size_t)blockIdx.x * nb + threadIdx.x] = t;
    }
}

// out[k] = sum_b partial[b*nb+k]   (single block)
__global__ __launch_bounds__(kScalBlock) void reduce_final_kernel(const double* __restrict__ partial, int nblocks, int nb,
                                                              double* __restrict__ out) {
    const double s = reduce_partials(partial, nblocks, nb);
    if ((int)threadIdx.x < nb) out[threadIdx.x] = s;
}

// out[i*NB+k] = a[i] (broadcast a shared vector into an interleaved batch)
template <int NB>
__global__ __launch_bounds__(kBlock) void broadcast_kernel(int n, const double* __restrict__ a, double* __restrict__ out,
                                                           int ld) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    double v[NB];
    const double t = a[i];
#pragma unroll
    for (int k = 0; k < NB; ++k) v[k] = t;
    store_row<NB>(out + (size_t)i * row_ld<NB>(ld) + col0<NB>(), v);
}

// ------------------------------------------------------------------------------------------
// V-cycle tail in LDS.  Small levels are launch-latency bound as separate kernels (a few us each, ~20
// launches per V-cycle); here one workgroup per realization sweeps all of them with __syncthreads()
// between phases.  Vectors live in LDS ([r | x | d] per level), matrices are read from global memory (L2).
static constexpr int kTailThreads = 1024;
// The tail's device functions are inlined into their two kernels: as real calls they cost ~50 callee-saved registers spilled
// at every entry and a register allocation split at the call boundary (Darcy iteration on a 16^3 level 168 -> 147 us,
// config 3 +5 %).
#define PMC_TAIL_INLINE __device__ __forceinline__
static constexpr size_t kTailLdsBytes = 160 * 1024 - 1024;   // dynamic LDS budget (static reduction scratch on top)

#ifndef PMC_TAIL_WIDE
#define PMC_TAIL_WIDE 0
#endif
__device__ __forceinline__ double tail_row_dot(const int* __restrict__ off, const int* __restrict__ cols,
                                               const double* __restrict__ vals, int vstride, size_t vk, int row,
                                               const double* xl) {
    const int slice = row >> 6, lane = row & 63;
    const int o = off[slice];
    const int width = (off[slice + 1] - o) >> 6;
    double acc = 0.0;
    int slot = o + lane;
#if PMC_TAIL_WIDE
    // wide slices (aggregation hierarchies of the hybridized sampler: 17-27 entries per row): 16 pairs per trip - a sweep over
    // such a level is a chain of trips to L2 (one workgroup per realization, ~1 us each), and the chain is what a pass
    // costs (LAB_NOTES 10.4).  Same summation order as the 8-wide loop: bit-identical.
    if (width > 12) {
        for (int j0 = 0; j0 < width; j0 += 16, slot += 16 * kWave) {
            int c[16];
            double v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const bool ok = j0 + u < width;
                const int at = ok ? slot + u * kWave : slot;
                c[u] = cols[at];
                v[u] = ok ? vals[(size_t)at * vstride + vk] : 0.0;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 16; ++u) acc = fma(v[u], xl[c[u]], acc);
        }
        return acc;
    }
#endif
    // 8 (index, value) pairs are requested together, then the 8 LDS gathers: two memory latencies per 8 entries
    // instead of one dependent chain per entry (rows have 1..8 entries on these levels)
    for (int j0 = 0; j0 < width; j0 += 8, slot += 8 * kWave) {
        int c[8];
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool ok = j0 + u < width;
            const int at = ok ? slot + u * kWave : slot;
            c[u] = cols[at];
            v[u] = ok ? vals[(size_t)at * vstride + vk] : 0.0;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = fma(v[u], xl[c[u]], acc);
    }
    return acc;
}

// Many-step Chebyshev solve on the LAST tail level with the thread's matrix rows held in registers.  In tail_cheb every
// step re-reads the (index, value) pairs of its rows from L2 - two dependent round trips per step, ~3 us, while the
// arithmetic of a step on a few thousand rows takes a fraction of that; a degree-14 solve costs ~80 us that way.  Here a
// thread loads the pairs of its R rows (at most W entries each) once and all steps run on registers + LDS.
// Same recurrences and summation order as tail_cheb: bit-identical results.  Returns false (nothing done) when the
// level does not fit R rows per thread x W entries per row.
template <int R, int W>
PMC_TAIL_INLINE bool tail_cheb_cached(const TailLevelDev& L, int bv, int nb, int k, int degree, double ratio, const double* r,
                                 double* x, double* d) {
    const int n = L.n;
    if (n > R * kTailThreads) return false;
    const int vstride = bv == 1 ? nb : 1;
    const size_t vk = bv == 1 ? (size_t)k : bv == 2 ? (size_t)k * L.nslots : 0;
    const size_t dk = bv == 1 ? (size_t)k : bv == 2 ? (size_t)k * L.n : 0;
    // widths are uniform per slice; reject the level if any slice is wider than W (uniform decision: every thread scans
    // the same few slice offsets)
    for (int s = 0; s < L.nslices; ++s)
        if (((L.slice_off[s + 1] - L.slice_off[s]) >> 6) > W) return false;
    int c[R][W];
    double v[R][W], di[R], rr[R];
#pragma unroll
    for (int q = 0; q < R; ++q) {
        const int i = threadIdx.x + q * kTailThreads;
        const bool live = i < n;
        const int row = live ? i : 0;
        const int slice = row >> 6, lane = row & 63;
        const int o = L.slice_off[slice];
        const int width = (L.slice_off[slice + 1] - o) >> 6;
#pragma unroll
        for (int j = 0; j < W; ++j) {
            const bool ok = live && j < width;
            const int at = o + (ok ? j : 0) * kWave + lane;
            c[q][j] = L.cols[at];
            const double val = L.vals[(size_t)at * vstride + vk];
            v[q][j] = ok ? val : 0.0;
        }
        di[q] = L.dinv[(size_t)row * vstride + dk];
        rr[q] = live ? r[row] : 0.0;
    }
    const double lmax = L.lmax, lmin = lmax / ratio;
    const double theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin), sigma = theta / delta;
    double rho_old = 1.0 / sigma;
    // step 0 from a zero guess: d = x = dinv r / theta
#pragma unroll
    for (int q = 0; q < R; ++q) {
        const int i = threadIdx.x + q * kTailThreads;
        if (i < n) {
            const double t = di[q] * rr[q] / theta;
            d[i] = t;
            x[i] = t;
        }
    }
    __syncthreads();
    for (int step = 1; step < degree; ++step) {
        const double rho = 1.0 / (2.0 * sigma - rho_old);
        const double a = rho * rho_old, b = 2.0 * rho / delta;
        rho_old = rho;
        double dn[R];
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const int i = threadIdx.x + q * kTailThreads;
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < W; ++j) acc = fma(v[q][j], x[c[q][j]], acc);
            dn[q] = i < n ? a * d[i] + b * di[q] * (rr[q] - acc) : 0.0;
        }
        __syncthreads();       // every gather of x is done before anyone updates it
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const int i = threadIdx.x + q * kTailThreads;
            if (i < n) {
                d[i] = dn[q];
                x[i] += dn[q];
            }
        }
        __syncthreads();
    }
    return true;
}

// Chebyshev iteration on one tail level, in place: x (zero or given) -> x.  All threads participate.  The matrix is re-read
// from L2 every step; the many-step solve of the LAST tail level runs on register-cached rows instead (tail_cheb_cached).
PMC_TAIL_INLINE void tail_cheb(const TailLevelDev& L, int bv, int nb, int k, int degree, double ratio, bool zero_guess,
                          const double* r, double* x, double* d) {
    const int n = L.n;
    // value / diagonal addressing: shared, interleaved per realization, or transposed per realization
    const int vstride = bv == 1 ? nb : 1;
    const size_t vk = bv == 1 ? (size_t)k : bv == 2 ? (size_t)k * L.nslots : 0;
    const size_t dk = bv == 1 ? (size_t)k : bv == 2 ? (size_t)k * L.n : 0;
    const double lmax = L.lmax, lmin = lmax / ratio;
    const double theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin), sigma = theta / delta;
    double rho_old = 1.0 / sigma;
    int step = 0;
    if (zero_guess && degree == 2 && L.vals_scaled) {
        const double rho1 = 1.0 / (2.0 * sigma - rho_old);
        const double c0 = (1.0 + rho1 * rho_old) / theta + 2.0 * rho1 / delta, c1 = 2.0 * rho1 / (delta * theta);
        for (int i = threadIdx.x; i < n; i += kTailThreads) {
            const double acc = tail_row_dot(L.slice_off, L.cols, L.vals_scaled, vstride, vk, i, r);
            x[i] = L.dinv[(size_t)i * vstride + dk] * (c0 * r[i] - c1 * acc);
        }
        __syncthreads();
        return;
    }
    if (zero_guess) {
        for (int i = threadIdx.x; i < n; i += kTailThreads) {
            const double v = L.dinv[(size_t)i * vstride + dk] * r[i] / theta;
            d[i] = v;
            x[i] = v;
        }
        __syncthreads();
        step = 1;
    }
    for (; step < degree; ++step) {
        double a, b;
        if (step == 0) {
            a = 0.0;
            b = 1.0 / theta;
        } else {
            const double rho = 1.0 / (2.0 * sigma - rho_old);
            a = rho * rho_old;
            b = 2.0 * rho / delta;
            rho_old = rho;
        }
        for (int i = threadIdx.x; i < n; i += kTailThreads) {
            const double acc = tail_row_dot(L.slice_off, L.cols, L.vals, vstride, vk, i, x);
            const double dold = (a != 0.0) ? d[i] : 0.0;
            d[i] = a * dold + b * L.dinv[(size_t)i * vstride + dk] * (r[i] - acc);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += kTailThreads) x[i] += d[i];
        __syncthreads();
    }
}

// V-cycle over the tail levels for realization k: the right-hand side is in LDS at lev[0]'s r block on entry, the
// result in its x block on return.  All kTailThreads threads of the workgroup participate.
PMC_TAIL_INLINE void tail_vcycle_lds(const TailParams& P, int nb, int k, double* lds) {
    const int nlev = P.nlev;
    // down sweep
    int l = 0;
    for (;; ++l) {
        const TailLevelDev& L = P.lev[l];
        double* r = lds + L.lds_off;
        double* x = r + L.n;
        double* d = x + L.n;
        if (l == nlev - 1 && L.ainv) {   // exact coarse solve with the precomputed dense inverse (symmetric: read column-wise)
            const int n = L.n;
            for (int i = threadIdx.x; i < n; i += kTailThreads) {
                double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
                int j = 0;
                for (; j + 3 < n; j += 4) {
                    s0 = fma(L.ainv[(size_t)j * n + i], r[j], s0);
                    s1 = fma(L.ainv[(size_t)(j + 1) * n + i], r[j + 1], s1);
                    s2 = fma(L.ainv[(size_t)(j + 2) * n + i], r[j + 2], s2);
                    s3 = fma(L.ainv[(size_t)(j + 3) * n + i], r[j + 3], s3);
                }
                for (; j < n; ++j) s0 = fma(L.ainv[(size_t)j * n + i], r[j], s0);
                x[i] = (s0 + s1) + (s2 + s3);
            }
            __syncthreads();
            break;
        }
        if (l == nlev - 1) {     // host guarantees last_degree > 0 on the final tail level
            // rows in registers: 3 rows x 5 entries (tets: 4 neighbours + diagonal) or 2 rows x 7 (hexahedra) per thread
            if (!(L.last_degree > 2 && (tail_cheb_cached<3, 5>(L, P.bv, nb, k, L.last_degree, L.last_ratio, r, x, d) ||
                                        tail_cheb_cached<2, 7>(L, P.bv, nb, k, L.last_degree, L.last_ratio, r, x, d))))
                tail_cheb(L, P.bv, nb, k, L.last_degree, L.last_ratio, true, r, x, d);
            break;
        }
        tail_cheb(L, P.bv, nb, k, P.smooth_degree, P.smooth_ratio, true, r, x, d);
        const int vstride = P.bv == 1 ? nb : 1;
        const size_t vk = P.bv == 1 ? (size_t)k : P.bv == 2 ? (size_t)k * L.nslots : 0;
        for (int i = threadIdx.x; i < L.n; i += kTailThreads)          // residual into d
            d[i] = r[i] - tail_row_dot(L.slice_off, L.cols, L.vals, vstride, vk, i, x);
        __syncthreads();
        const TailLevelDev& Lc = P.lev[l + 1];
        double* rc = lds + Lc.lds_off;
        for (int i = threadIdx.x; i < Lc.n; i += kTailThreads)         // restriction r_c = P^T res
            rc[i] = tail_row_dot(L.pt_off, L.pt_cols, L.pt_vals, 1, 0, i, d);
        __syncthreads();
    }
    // up sweep
    for (--l; l >= 0; --l) {
        const TailLevelDev& L = P.lev[l];
        double* r = lds + L.lds_off;
        double* x = r + L.n;
        double* d = x + L.n;
        const double* xc = lds + P.lev[l + 1].lds_off + P.lev[l + 1].n;
        for (int i = threadIdx.x; i < L.n; i += kTailThreads) x[i] += tail_row_dot(L.p_off, L.p_cols, L.p_vals, 1, 0, i, xc);
        __syncthreads();
        tail_cheb(L, P.bv, nb, k, P.smooth_degree, P.smooth_ratio, false, r, x, d);
    }
}

// out32: xout points at fp32 storage (the preconditioned Krylov vectors in fp32 storage)
__global__ __launch_bounds__(kTailThreads) void mg_tail_kernel(const TailParams* __restrict__ pp, int nb,
                                                               const double* __restrict__ rin, double* __restrict__ xout,
                                                               double* __restrict__ partial, int out32) {
    extern __shared__ __align__(16) double lds[];
    __shared__ double red[kTailThreads / kWave];
    const TailParams& P = *pp;
    const int k = blockIdx.x;
    const TailLevelDev& L0 = P.lev[0];
    {
        double* r0 = lds + L0.lds_off;
        for (int i = threadIdx.x; i < L0.n; i += kTailThreads) r0[i] = rin[(size_t)i * nb + k];
    }
    __syncthreads();
    tail_vcycle_lds(P, nb, k, lds);
    const double* r0 = lds + L0.lds_off;
    const double* x0 = r0 + L0.n;
    double p = 0.0;
    for (int i = threadIdx.x; i < L0.n; i += kTailThreads) {
        double xi = x0[i];
        if (out32) {
            const float xf = (float)xi;
            reinterpret_cast<float*>(xout)[(size_t)i * nb + k] = xf;
            xi = (double)xf;
        } else {
            xout[(size_t)i * nb + k] = xi;
        }
        p = fma(r0[i], xi, p);
    }
    if (partial) {
#pragma unroll
        for (int off = kWave / 2; off > 0; off >>= 1) p += __shfl_down(p, off, kWave);
        if ((threadIdx.x & (kWave - 1)) == 0) red[threadIdx.x / kWave] = p;
        __syncthreads();
        if (threadIdx.x == 0) {
            double sum = 0.0;
            for (int w = 0; w < kTailThreads / kWave; ++w) sum += red[w];
            partial[k] = sum;     // one partial block: partial[0*nb + k]
        }
    }
}

// ------------------------------------------------------------------------------------------
// Persistent per-realization solver for SMALL levels (shared matrix values: the SPDE sampler).  As separate kernels a
// MINRES iteration on a ~17 k-row level is 7 launches of a few microseconds each - the GPU's dispatch rate, not its
// bandwidth, bounds the throughput.  Here ONE workgroup runs the whole preconditioned MINRES solve of ONE realization:
// operator, M-block polynomial, S-block V-cycle (the LDS tail above), dots and scalar recurrences, separated only by
// __syncthreads(); vectors live in a per-realization scratch area that stays in L2, no host round trip until the solve
// has finished.  Same recurrences, stopping rule and per-realization results as the batched kernels.
__device__ __forceinline__ double block_sum(double v, double* red) {
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
    __syncthreads();                                   // red may still be read from the previous call
    if ((threadIdx.x & (kWave - 1)) == 0) red[threadIdx.x / kWave] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < kTailThreads / kWave; ++w) s += red[w];   // same order in every thread: deterministic, no broadcast
    return s;
}

__global__ __launch_bounds__(kTailThreads) void mini_sampler_kernel(MiniSamplerParams P, int nb, const double* __restrict__ b,
                                                                     double* __restrict__ x, int zero_guess,
                                                                     double* __restrict__ scratch, pmc_stats* __restrict__ stats) {
    extern __shared__ __align__(16) double lds[];
    __shared__ double red[kTailThreads / kWave];
    const int k = blockIdx.x;
    const int n_u = P.n_u, n_s = P.n_s, n = n_u + n_s;
    const int tid = threadIdx.x;
    double* v0 = scratch + (size_t)k * P.scratch_per_col;
    double* v1 = v0 + n;
    double* u0 = v1 + n;
    double* u1 = u0 + n;
    double* q = u1 + n;
    double* w0 = q + n;
    double* w1 = w0 + P.x_nrows;
    double* xs = w1 + P.x_nrows;
    const TailParams& T = *P.tail;
    const TailLevelDev& L0 = T.lev[0];
    double* r0 = lds + L0.lds_off;
    const double* x0 = r0 + L0.n;

    // z = B^-1 r: u-block one-pass degree-2 polynomial in D^-1 M, s-block V-cycle in LDS; returns <r, z>
    auto prec = [&](const double* r, double* z) {
        double p = 0.0;
        for (int i = tid; i < n_u; i += kTailThreads) {
            const double acc = tail_row_dot(P.m_off, P.m_cols, P.m_scaled, 1, 0, i, r);
            const double zi = P.m_dinv[i] * (P.mc0 * r[i] - P.mc1 * acc);
            z[i] = zi;
            p = fma(r[i], zi, p);
        }
        for (int i = tid; i < n_s; i += kTailThreads) r0[i] = r[n_u + i];
        __syncthreads();
        tail_vcycle_lds(T, nb, k, lds);
        for (int i = tid; i < n_s; i += kTailThreads) {
            const double zi = x0[i];
            z[n_u + i] = zi;
            p = fma(r0[i], zi, p);
        }
        return block_sum(p, red);     // its barriers also order the z writes before the next phase reads them
    };

    // v1 = b - A x0
    if (zero_guess) {
        for (int i = tid; i < n; i += kTailThreads) v1[i] = b[(size_t)i * nb + k];
        for (int i = tid; i < P.x_nrows; i += kTailThreads) xs[i] = 0.0;
    } else {
        for (int i = tid; i < n; i += kTailThreads) u1[i] = x[(size_t)i * nb + k];
        __syncthreads();
        for (int i = tid; i < n; i += kTailThreads)
            v1[i] = b[(size_t)i * nb + k] - tail_row_dot(P.a_off, P.a_cols, P.a_vals, 1, 0, i, u1);
        for (int i = tid; i < P.x_nrows; i += kTailThreads) xs[i] = u1[P.x_row0 + i];
    }
    for (int i = tid; i < n; i += kTailThreads) v0[i] = 0.0;
    for (int i = tid; i < P.x_nrows; i += kTailThreads) { w0[i] = 0.0; w1[i] = 0.0; }
    __syncthreads();
    const double d0 = prec(v1, u1);
    double beta = d0 > 0.0 ? sqrt(d0) : 0.0, beta_old = 1.0, eta = beta;
    double gamma0 = 1.0, gamma1 = 1.0, sigma0 = 0.0, sigma1 = 0.0;
    const double eta0 = beta;
    const double goal = fmax(P.rel_tol * beta, P.abs_tol);
    int flag = (d0 < 0.0 || d0 != d0) ? -1 : 0;
    bool active = beta > goal && flag == 0;
    int it = 0;
    while (active && it < P.max_iter) {
        // q = A u1, <u1, q>
        double p = 0.0;
        for (int i = tid; i < n; i += kTailThreads) {
            const double qi = tail_row_dot(P.a_off, P.a_cols, P.a_vals, 1, 0, i, u1);
            q[i] = qi;
            p = fma(u1[i], qi, p);
        }
        const double d1 = block_sum(p, red);
        const double ib = 1.0 / beta;
        const double alpha = d1 * ib * ib;
        const double cV0 = ib, cV1 = -alpha * ib, cV2 = -beta / beta_old;
        const double delta = gamma1 * alpha - gamma0 * sigma1 * beta;
        const double rho3 = sigma0 * beta;
        const double rho2 = sigma1 * alpha + gamma0 * gamma1 * beta;
        for (int i = tid; i < n; i += kTailThreads) v0[i] = cV0 * q[i] + cV1 * v1[i] + cV2 * v0[i];
        __syncthreads();
        const double d2 = prec(v0, u0);
        if (d2 < 0.0 || d2 != d2) flag = -1;
        const double beta_new = d2 > 0.0 ? sqrt(d2) : 0.0;
        const double rho1 = hypot(delta, beta_new);
        const double ir = rho1 > 0.0 ? 1.0 / rho1 : 0.0;
        const double cW0 = ir / beta, cW1 = -rho3 * ir, cW2 = -rho2 * ir;
        gamma0 = gamma1;
        gamma1 = delta * ir;
        const double cW3 = gamma1 * eta;
        sigma0 = sigma1;
        sigma1 = beta_new * ir;
        eta = -sigma1 * eta;
        beta_old = beta;
        beta = beta_new;
        for (int i = tid; i < P.x_nrows; i += kTailThreads) {
            const double w = cW0 * u1[P.x_row0 + i] + cW1 * w0[i] + cW2 * w1[i];
            w0[i] = w;
            xs[i] += cW3 * w;
        }
        ++it;
        if (fabs(eta) <= goal || beta_new == 0.0 || flag != 0) active = false;
        // role swap (every thread holds the same pointers)
        double* t;
        t = u0; u0 = u1; u1 = t;
        t = v0; v0 = v1; v1 = t;
        t = w0; w0 = w1; w1 = t;
        __syncthreads();
    }
    for (int i = tid; i < P.x_nrows; i += kTailThreads) x[(size_t)(P.x_row0 + i) * nb + k] = xs[i];
    if (tid == 0) {
        stats[k].iterations = it;
        stats[k].converged = flag != 0 ? -1 : (fabs(eta) <= goal ? 1 : 0);   // -1: indefinite preconditioner / NaN
        stats[k].initial_norm = eta0;
        stats[k].final_norm = fabs(eta);
        stats[k].solve_ms = 0.0;      // filled on the host from the launch's events
        stats[k].setup_ms = 0.0;
    }
}

// ==========================================================================================
// launchers
#define PMC_DISPATCH_NB(nb, ...)                                          \
    switch (nb) {                                                         \
        case 1: { constexpr int NB = 1; __VA_ARGS__; } break;             \
        case 2: { constexpr int NB = 2; __VA_ARGS__; } break;             \
        case 4: { constexpr int NB = 4; __VA_ARGS__; } break;             \
        case 8: { constexpr int NB = 8; __VA_ARGS__; } break;             \
        case 16: { constexpr int NB = 16; __VA_ARGS__; } break;           \
        case 32: case 64: case 128: case 256: { constexpr int NB = 32; __VA_ARGS__; } break;   /* column groups of 32 */ \
        default: throw Error(PMC_ERR_INTERNAL, "unsupported batch width"); \
    }

// the row-split instantiations (SellView::split_log2) exist for launches of at most 8 realizations
#define PMC_DISPATCH_NARROW(nb, ...)                                      \
    switch (nb) {                                                         \
        case 1: { constexpr int NB = 1; __VA_ARGS__; } break;             \
        case 2: { constexpr int NB = 2; __VA_ARGS__; } break;             \
        case 4: { constexpr int NB = 4; __VA_ARGS__; } break;             \
        case 8: { constexpr int NB = 8; __VA_ARGS__; } break;             \
        default: throw Error(PMC_ERR_INTERNAL, "row-split level kernels serve launches of 1, 2, 4 or 8 realizations"); \
    }

// grid of a group-capable kernel: y = number of column groups of a batch of nb realizations (1 up to kGroup)
static inline dim3 groups(dim3 g, int nb) { return dim3(g.x, nb > kGroup ? (unsigned)(nb / kGroup) : 1u); }
// the slice kernels' grid (see vblock()): column groups of the same slices adjacent on the same XCD
static inline bool xcd_layout(dim3 g, int nb) {
    // off in the product: one lane gains 2.4 % from it, four lanes lose 1 % (LAB_NOTES 10.11); laboratory switch PMC_XCD_GROUPS=1
    static const bool on = [] { const char* e = lab_env("PMC_XCD_GROUPS"); return e && atoi(e) != 0; }();
    return on && nb > kGroup && g.x >= 16;
}
static inline dim3 groups_xcd(dim3 g, int nb) {
    if (!xcd_layout(g, nb)) return groups(g, nb);
    return dim3(8u, (unsigned)(nb / kGroup), (g.x + 7u) / 8u);
}
// number of per-block dot partials such a launch writes (its virtual grid extent)
static inline int dot_blocks(dim3 g, int nb) { return xcd_layout(g, nb) ? (int)((g.x + 7u) / 8u * 8u) : (int)g.x; }
static inline dim3 groups(unsigned g, int nb) { return groups(dim3(g), nb); }
static inline dim3 groups(int g, int nb) { return groups(dim3((unsigned)g), nb); }
static inline dim3 grid_rows(int n) { return dim3((unsigned)((n + kBlock - 1) / kBlock)); }
static inline dim3 grid_slices(int nslices) { return dim3((unsigned)((nslices + kBlock / kWave - 1) / (kBlock / kWave))); }
static inline int lay_c(int nb) { return nb >= 32 ? 4 : (nb >= 2 ? 2 : 1); }   // = Lay<nb>::C
static inline size_t flat_count(int n, int nb) { return (size_t)n * nb / lay_c(nb); }
static inline dim3 grid_flat(int n, int nb) { return dim3((unsigned)((flat_count(n, nb) + kBlock - 1) / kBlock)); }
static std::atomic<uint64_t> g_kernel_launches{0};
uint64_t kernel_launch_count() { return g_kernel_launches.load(std::memory_order_relaxed); }
void count_kernel_launches(int n) { g_kernel_launches.fetch_add((uint64_t)n, std::memory_order_relaxed); }
static inline void check_launch(int n = 1) {
    PMC_HIP(hipGetLastError());
    count_kernel_launches(n);
}
// the lean gather loops address rows with 32-bit element offsets: rows x row stride of every gathered vector must fit
static inline void check_offsets32(const SellView& A, int nb) {
    if ((uint64_t)std::max(A.nrows, A.ncols_hint) * (uint64_t)nb >= (1ull << 32))
        throw Error(PMC_ERR_INVALID, "operand exceeds the 32-bit gather offsets of the SpMM kernels (rows x batch width >= 2^32)");
}
// flat vector kernels stream non-temporally once one vector exceeds PMC_NT_FLAT_MB MiB (default 8; 0 = never)
static inline bool nt_flat(size_t doubles) {
    static const double limit = [] {
        const char* e = lab_env("PMC_NT_FLAT_MB");
        return (e ? atof(e) : 8.0) * 1024.0 * 1024.0;
    }();
    return limit > 0.0 && (double)doubles * 8.0 > limit;
}

// kernels with a fused dot write one partial per block; the bound trades occupancy of the SpMM against the length of
// the single-block final reduction (PMC_DOT_GRID overrides it for tuning runs)
static unsigned dot_grid_bound() {
    static const unsigned v = [] {
        const char* e = lab_env("PMC_DOT_GRID");
        const long x = e ? atol(e) : 0;
        return x >= 8 ? (unsigned)x : 4096u;
    }();
    return v;
}

namespace k {

static inline dim3 grid_bounded(dim3 g, bool bounded) { return bounded ? dim3(std::min(g.x, dot_grid_bound())) : g; }

// non-temporal matrix / result streams for the block operator once its operands no longer fit the Infinity Cache.  Inside
// the MINRES loop (in_loop: the launch with the fused dot) they never are cache-resident from half that size on - the rest
// of the iteration moves ~5 x the operator's bytes in between - and the hints pay earlier (0.6 M rows, 195 MB: one lane
// 1057 -> 1063, four lanes 1417 -> 1434 samples/s); PMC_NT_MIN_MB overrides the in-loop threshold.
static inline bool nt_streams(const SellView& A, int nb, bool in_loop) {
    const double bytes = 12.0 * (double)A.nslices * 64.0 * 6.0 + 16.0 * nb * (double)A.nrows;   // ~6 entries per row
    static const double loop_limit = [] {
        const char* e = lab_env("PMC_NT_MIN_MB");
        return (e ? atof(e) : 128.0) * 1024.0 * 1024.0;
    }();
    return bytes > (in_loop ? loop_limit : 256.0 * 1024.0 * 1024.0);
}

// The same hints for the one-pass polynomial kernels (smoothers) of large levels: their matrix and result streams no longer
// displace the gathered rows from L2 (0.6 M rows: one lane 1048 -> 1081 samples/s, four lanes 1367 -> 1380).  Not for the
// residual kernels: their result is read again two launches later (one lane 1103 -> 1084 with the hints).
// PMC_NT_POLY_MB: threshold in MiB of operands, 0 = never.
static inline bool nt_poly(const SellView& A, int nb) {
    static const double limit = [] {
        const char* e = lab_env("PMC_NT_POLY_MB");
        return (e ? atof(e) : 32.0) * 1024.0 * 1024.0;
    }();
    const double bytes = 12.0 * (double)A.nslices * 64.0 * 6.0 + 16.0 * nb * (double)A.nrows;
    return limit > 0.0 && bytes > limit;
}

template <int NB, int TAG, typename XT>
static void spmm_launch(hipStream_t st, int nb, dim3 g, const SellView& A, const XT* x, double* y, bool accumulate,
                        double* dot_partial, const XT* dot_with) {
    // <x, Ax> with a diagonal-last matrix: x_i is what the row's last slice column gathers
    const bool dl = TAG == 1 && Lay<NB>::T > 1 && A.diag_last && dot_with == x;
    if constexpr (!std::is_same<XT, double>::value) {
        // fp32-stored input (the preconditioned Krylov vectors): the operator products of the solver loop only
        if (A.bv || accumulate) throw Error(PMC_ERR_INTERNAL, "spmm: fp32 input with per-realization values / accumulation");
        const bool nt = TAG != 0 && nt_streams(A, NB, dot_partial != nullptr);
        if (nt) {
            if (dot_partial && dl)
                sell_spmm_kernel<NB, false, 0, true, TAG, true, false, (Lay<NB>::T > 1), XT><<<groups_xcd(g, nb), kBlock, 0, st>>>(A.nrows, A.nslices, A.slice_off, A.sched, A.cols, A.vals, x, y, nullptr, dot_with, dot_partial, nb);
            else if (dot_partial)
                sell_spmm_kernel<NB, false, 0, true, TAG, true, false, false, XT><<<groups_xcd(g, nb), kBlock, 0, st>>>(A.nrows, A.nslices, A.slice_off, A.sched, A.cols, A.vals, x, y, nullptr, dot_with, dot_partial, nb);
            else
                sell_spmm_kernel<NB, false, 0, false, TAG, true, false, false, XT><<<groups_xcd(g, nb), kBlock, 0, st>>>(A.nrows, A.nslices, A.slice_off, A.sched, A.cols, A.vals, x, y, nullptr, nullptr, nullptr, nb);
        } else {
            if (dot_partial && dl)
                sell_spmm_kernel<NB, false, 0, true, TAG, false, false, (Lay<NB>::T > 1), XT><<<groups_xcd(g, nb), kBlock, 0, st>>>(A.nrows, A.nslices, A.slice_off, A.sched, A.cols, A.vals, x, y, nullptr, dot_with, dot_partial, nb);
            else if (dot_partial)
                sell_spmm_kernel<NB, false, 0, true, TAG, false, false, false, XT><<<groups_xcd(g, nb), kBlock, 0, st>>>(A.nrows, A.nslices, A.slice_off, A.sched, A.cols, A.vals, x, y, nullptr, dot_with, dot_partial, nb);
            else
                sell_spmm_kernel<NB, false, 0, false, TAG, false, false, false, XT><<<groups_xcd(g, nb), kBlock, 0, st>>>(A.nrows, A.nslices, A.slice_off, A.sched, A.cols, A.vals, x, y, nullptr, nullptr, nullptr, nb);
        }
        return;
    } else
    if (A.bv && A.f32) {
        if (dot_partial)
            sell_spmm_kernel<NB, 2, 0, true, TAG><<<groups_xcd(g, nb), kBlock, 0, st>>>(A.nrows, A.nslices, A.slice_off, A.sched, A.cols, A.vals, x, y, nullptr, dot_with, dot_partial, nb);
        else if (accumulate)
            sell_spmm_kernel<NB, 2, 1, false, TAG><<<groups_xcd(g, nb), kBlock, 0, st>>>(A.nrows, A.nslices, A.slice_off, A.sched, A.cols, A.vals, x, y, nullptr, nullptr, nullptr, nb);
        else
            sell_spmm_kernel<NB, 2, 0, false, TAG><<<groups_xcd(g, nb), kBlock, 0, st>>>(A.nrows, A.nslices, A.slice_off, A.sched, A.cols, A.vals, x, y, nullptr, nullptr, nullptr, nb);
    } else if (A.bv) {
        if (dot_partial)
            sell_spmm_kernel<NB, true, 0, true, TAG><<<groups_xcd(g, nb), kBlock, 0, st>>>(A.nrows, A.nslices, A.slice_off, A.sched, A.cols, A.vals, x, y, nullptr, dot_with, dot_partial, nb);
        else if (accumulate)
            sell_spmm_kernel<NB, true, 1, false, TAG><<<groups_xcd(g, nb), kBlock, 0, st>>>(A.nrows, A.nslices, A.slice_off, A.sched, A.cols, A.vals, x, y, nullptr, nullptr, nullptr, nb);
        else
            sell_spmm_kernel<NB, true, 0, false, TAG><<<groups_xcd(g, nb), kBlock, 0, st>>>(A.nrows, A.nslices, A.slice_off, A.sched, A.cols, A.vals, x, y, nullptr, nullptr, nullptr, nb);
    } else if (TAG != 0 && nt_streams(A, NB, dot_partial != nullptr)) {
        if (dot_partial && dl)
            sell_spmm_kernel<NB, false, 0, true, TAG, true, false, (Lay<NB>::T > 1)><<<groups_xcd(g, nb), kBlock, 0, st>>>(A.nrows, A.nslices, A.slice_off, A.sched, A.cols, A.vals, x, y, nullptr, dot_with, dot_partial, nb);
        else if (dot_partial)
            sell_spmm_kernel<NB, false, 0, true, TAG, true><<<groups_xcd(g, nb), kBlock, 0, st>>>(A.nrows, A.nslices, A.slice_off, A.sched, A.cols, A.vals, x, y, nullptr, dot_with, dot_partial, nb);
        else if (accumulate)
            sell_spmm_kernel<NB, false, 1, false, TAG, true><<<groups_xcd(g, nb), kBlock, 0, st>>>(A.nrows, A.nslices, A.slice_off, A.sched, A.cols, A.vals, x, y, nullptr, nullptr, nullptr, nb);
        else
            sell_spmm_kernel<NB, false, 0, false, TAG, true><<<groups_xcd(g, nb), kBlock, 0, st>>>(A.nrows, A.nslices, A.slice_off, A.sched, A.cols, A.vals, x, y, nullptr, nullptr, nullptr, nb);
    } else {
        if (dot_partial && dl)
            sell_spmm_kernel<NB, false, 0, true, TAG, false, false, (Lay<NB>::T > 1)><<<groups_xcd(g, nb), kBlock, 0, st>>>(A.nrows, A.nslices, A.slice_off, A.sched, A.cols, A.vals, x, y, nullptr, dot_with, dot_partial, nb);
        else if (dot_partial)
            sell_spmm_kernel<NB, false, 0, true, TAG><<<groups_xcd(g, nb), kBlock, 0, st>>>(A.nrows, A.nslices, A.slice_off, A.sched, A.cols, A.vals, x, y, nullptr, dot_with, dot_partial, nb);
        else if (accumulate)
            sell_spmm_kernel<NB, false, 1, false, TAG><<<groups_xcd(g, nb), kBlock, 0, st>>>(A.nrows, A.nslices, A.slice_off, A.sched, A.cols, A.vals, x, y, nullptr, nullptr, nullptr, nb);
        else
            sell_spmm_kernel<NB, false, 0, false, TAG><<<groups_xcd(g, nb), kBlock, 0, st>>>(A.nrows, A.nslices, A.slice_off, A.sched, A.cols, A.vals, x, y, nullptr, nullptr, nullptr, nb);
    }
}

template <typename XT>
static int spmm_t(hipStream_t st, int nb, const SellView& A, const XT* x, double* y, bool accumulate, double* dot_partial,
                  const XT* dot_with) {
    check_offsets32(A, nb);
    if (A.nrows == 0) return 0;
    if (dot_partial && !dot_with) throw Error(PMC_ERR_INTERNAL, "spmm: fused dot without its second vector");
    dim3 g = grid_bounded(grid_slices(A.nslices), dot_partial != nullptr);
    // The block operator keeps one slice per wavefront also with a fused dot (a bounded grid of looping workgroups cost
    // 22 us of 450 at 4.7 M rows); its partial blocks - more than the single-block consumers should read - are first
    // compressed to kCompressBlocks.  They are written behind the compressed ones: dot_capacity() leaves the room.
    const bool two_stage = dot_partial && A.tag != 0 && !A.bv && grid_slices(A.nslices).x > g.x;
    double* kernel_partial = dot_partial;
    if (two_stage) {
        g = grid_slices(A.nslices);
        kernel_partial = dot_partial + (size_t)kCompressBlocks * nb;
    }
    PMC_DISPATCH_NB(nb, {
        if (A.tag == 1) spmm_launch<NB, 1, XT>(st, nb, g, A, x, y, accumulate, kernel_partial, dot_with);
        else if (A.tag == 2) spmm_launch<NB, 2, XT>(st, nb, g, A, x, y, accumulate, kernel_partial, dot_with);
        else spmm_launch<NB, 0, XT>(st, nb, g, A, x, y, accumulate, kernel_partial, dot_with);
    });
    check_launch();
    if (two_stage) {
        compress_partials_kernel<<<kCompressBlocks, 256, 0, st>>>(kernel_partial, dot_blocks(g, nb), nb, dot_partial);
        check_launch();
        return kCompressBlocks;
    }
    return dot_partial ? dot_blocks(g, nb) : 0;
}
int spmm(hipStream_t st, int nb, const SellView& A, const double* x, double* y, bool accumulate, double* dot_partial,
         const double* dot_with) {
    return spmm_t<double>(st, nb, A, x, y, accumulate, dot_partial, dot_with);
}
int spmm_z(hipStream_t st, int nb, const SellView& A, zvec x, double* y, double* dot_partial, zvec dot_with) {
    if (x.f32) return spmm_t<float>(st, nb, A, x.as<float>(), y, false, dot_partial, dot_with.as<float>());
    return spmm_t<double>(st, nb, A, x.as<double>(), y, false, dot_partial, dot_with.as<double>());
}

void residual_restrict8(hipStream_t st, int nb, const SellView& A, const double* r, const double* x, double* out,
                        double* coarse) {
    check_offsets32(A, nb);
    if (A.nrows == 0) return;
    if (A.nrows % 8 != 0) throw Error(PMC_ERR_INTERNAL, "residual_restrict8: rows are not groups of 8");
    const dim3 g = grid_slices(A.nslices);
    PMC_DISPATCH_NB(nb, {
        if (A.bv && A.f32)
            sell_spmm_kernel<NB, 2, 2, false, 0, false, true><<<groups_xcd(g, nb), kBlock, 0, st>>>(A.nrows, A.nslices, A.slice_off, A.sched, A.cols, A.vals, x, out, r, nullptr, coarse, nb);
        else if (A.bv)
            sell_spmm_kernel<NB, true, 2, false, 0, false, true><<<groups_xcd(g, nb), kBlock, 0, st>>>(A.nrows, A.nslices, A.slice_off, A.sched, A.cols, A.vals, x, out, r, nullptr, coarse, nb);
        else
            sell_spmm_kernel<NB, false, 2, false, 0, false, true><<<groups_xcd(g, nb), kBlock, 0, st>>>(A.nrows, A.nslices, A.slice_off, A.sched, A.cols, A.vals, x, out, r, nullptr, coarse, nb);
    });
    check_launch();
}

void residual(hipStream_t st, int nb, const SellView& A, const double* r, const double* x, double* out) {
    check_offsets32(A, nb);
    if (A.nrows == 0) return;
    const dim3 g = grid_slices(A.nslices);
    PMC_DISPATCH_NB(nb, {
        if (A.bv && A.f32)
            sell_spmm_kernel<NB, 2, 2, false, 0><<<groups_xcd(g, nb), kBlock, 0, st>>>(A.nrows, A.nslices, A.slice_off, A.sched, A.cols, A.vals, x, out, r, nullptr, nullptr, nb);
        else if (A.bv)
            sell_spmm_kernel<NB, true, 2, false, 0><<<groups_xcd(g, nb), kBlock, 0, st>>>(A.nrows, A.nslices, A.slice_off, A.sched, A.cols, A.vals, x, out, r, nullptr, nullptr, nb);
        else
            sell_spmm_kernel<NB, false, 2, false, 0><<<groups_xcd(g, nb), kBlock, 0, st>>>(A.nrows, A.nslices, A.slice_off, A.sched, A.cols, A.vals, x, out, r, nullptr, nullptr, nb);
    });
    check_launch();
}

template <typename OT>
static int cheb_step_t(hipStream_t st, int nb, const SellView& A, const double* dinv, bool dinv_bv, const double* r,
                       const double* xin, double* d, OT* xout, double a, double b, double* dot_partial) {
    check_offsets32(A, nb);
    if (A.nrows == 0) return 0;
    if (A.bv != dinv_bv) throw Error(PMC_ERR_INTERNAL, "cheb_step: value/diagonal batching mismatch");
    const dim3 g = grid_bounded(grid_slices(A.nslices), dot_partial != nullptr);
    PMC_DISPATCH_NB(nb, {
        if (A.bv && A.f32) {
            if (dot_partial)
                sell_cheb_kernel<NB, 2, true, OT><<<groups_xcd(g, nb), kBlock, 0, st>>>(A.nrows, A.nslices, A.slice_off, A.sched, A.cols, A.vals, dinv, r, xin, d, xout, a, b, dot_partial, nb);
            else
                sell_cheb_kernel<NB, 2, false, OT><<<groups_xcd(g, nb), kBlock, 0, st>>>(A.nrows, A.nslices, A.slice_off, A.sched, A.cols, A.vals, dinv, r, xin, d, xout, a, b, nullptr, nb);
        } else if (A.bv) {
            if (dot_partial)
                sell_cheb_kernel<NB, 1, true, OT><<<groups_xcd(g, nb), kBlock, 0, st>>>(A.nrows, A.nslices, A.slice_off, A.sched, A.cols, A.vals, dinv, r, xin, d, xout, a, b, dot_partial, nb);
            else
                sell_cheb_kernel<NB, 1, false, OT><<<groups_xcd(g, nb), kBlock, 0, st>>>(A.nrows, A.nslices, A.slice_off, A.sched, A.cols, A.vals, dinv, r, xin, d, xout, a, b, nullptr, nb);
        } else {
            if (dot_partial)
                sell_cheb_kernel<NB, 0, true, OT><<<groups_xcd(g, nb), kBlock, 0, st>>>(A.nrows, A.nslices, A.slice_off, A.sched, A.cols, A.vals, dinv, r, xin, d, xout, a, b, dot_partial, nb);
            else
                sell_cheb_kernel<NB, 0, false, OT><<<groups_xcd(g, nb), kBlock, 0, st>>>(A.nrows, A.nslices, A.slice_off, A.sched, A.cols, A.vals, dinv, r, xin, d, xout, a, b, nullptr, nb);
        }
    });
    check_launch();
    return dot_partial ? dot_blocks(g, nb) : 0;
}

int cheb_step(hipStream_t st, int nb, const SellView& A, const double* dinv, bool dinv_bv, const double* r,
              const double* xin, double* d, double* xout, double a, double b, double* dot_partial) {
    return cheb_step_t<double>(st, nb, A, dinv, dinv_bv, r, xin, d, xout, a, b, dot_partial);
}

// fp64 storage: the iterate of the last step goes straight to zout, d is left as it was (as the typed kernel does)
int cheb_step_z(hipStream_t st, int nb, const SellView& A, const double* dinv, bool dinv_bv, const double* r,
                const double* xin, double* d, zvec zout, double a, double b, double* dot_partial) {
    if (zout.f32) return cheb_step_t<float>(st, nb, A, dinv, dinv_bv, r, xin, d, zout.as<float>(), a, b, dot_partial);
    return cheb_step_t<double>(st, nb, A, dinv, dinv_bv, r, xin, d, zout.as<double>(), a, b, dot_partial);
}

template <typename OT>
static int poly2_t(hipStream_t st, int nb, const SellView& As, const double* dinv, bool dinv_bv, const double* r, OT* xout,
                   double c0, double c1, double* dot_partial, const double* xadd, const double* dot_with, const int* padd_idx,
                   const double* padd_x) {
    check_offsets32(As, nb);
    if (As.nrows == 0) return 0;
    if (As.bv != dinv_bv) throw Error(PMC_ERR_INTERNAL, "poly2: value/diagonal batching mismatch");
    const dim3 g = grid_bounded(grid_slices(As.nslices), dot_partial != nullptr);
    PMC_DISPATCH_NB(nb, {
        if (As.bv && As.f32) {
            if (dot_partial)
                sell_poly2_kernel<NB, 2, true, false, OT><<<groups_xcd(g, nb), kBlock, 0, st>>>(As.nrows, As.nslices, As.slice_off, As.sched, As.cols, As.vals, dinv, r, xout, c0, c1, dot_partial, xadd, dot_with, padd_idx, padd_x, nb);
            else
                sell_poly2_kernel<NB, 2, false, false, OT><<<groups_xcd(g, nb), kBlock, 0, st>>>(As.nrows, As.nslices, As.slice_off, As.sched, As.cols, As.vals, dinv, r, xout, c0, c1, nullptr, xadd, dot_with, padd_idx, padd_x, nb);
        } else if (As.bv) {
            if (dot_partial)
                sell_poly2_kernel<NB, 1, true, false, OT><<<groups_xcd(g, nb), kBlock, 0, st>>>(As.nrows, As.nslices, As.slice_off, As.sched, As.cols, As.vals, dinv, r, xout, c0, c1, dot_partial, xadd, dot_with, padd_idx, padd_x, nb);
            else
                sell_poly2_kernel<NB, 1, false, false, OT><<<groups_xcd(g, nb), kBlock, 0, st>>>(As.nrows, As.nslices, As.slice_off, As.sched, As.cols, As.vals, dinv, r, xout, c0, c1, nullptr, xadd, dot_with, padd_idx, padd_x, nb);
        } else if (nt_poly(As, NB)) {
            if (dot_partial)
                sell_poly2_kernel<NB, 0, true, true, OT><<<groups_xcd(g, nb), kBlock, 0, st>>>(As.nrows, As.nslices, As.slice_off, As.sched, As.cols, As.vals, dinv, r, xout, c0, c1, dot_partial, xadd, dot_with, padd_idx, padd_x, nb);
            else
                sell_poly2_kernel<NB, 0, false, true, OT><<<groups_xcd(g, nb), kBlock, 0, st>>>(As.nrows, As.nslices, As.slice_off, As.sched, As.cols, As.vals, dinv, r, xout, c0, c1, nullptr, xadd, dot_with, padd_idx, padd_x, nb);
        } else {
            if (dot_partial)
                sell_poly2_kernel<NB, 0, true, false, OT><<<groups_xcd(g, nb), kBlock, 0, st>>>(As.nrows, As.nslices, As.slice_off, As.sched, As.cols, As.vals, dinv, r, xout, c0, c1, dot_partial, xadd, dot_with, padd_idx, padd_x, nb);
            else
                sell_poly2_kernel<NB, 0, false, false, OT><<<groups_xcd(g, nb), kBlock, 0, st>>>(As.nrows, As.nslices, As.slice_off, As.sched, As.cols, As.vals, dinv, r, xout, c0, c1, nullptr, xadd, dot_with, padd_idx, padd_x, nb);
        }
    });
    check_launch();
    return dot_partial ? dot_blocks(g, nb) : 0;
}

int poly2(hipStream_t st, int nb, const SellView& As, const double* dinv, bool dinv_bv, const double* r, double* xout,
          double c0, double c1, double* dot_partial, const double* xadd, const double* dot_with, const int* padd_idx,
          const double* padd_x) {
    return poly2_t<double>(st, nb, As, dinv, dinv_bv, r, xout, c0, c1, dot_partial, xadd, dot_with, padd_idx, padd_x);
}

int poly2_z(hipStream_t st, int nb, const SellView& As, const double* dinv, bool dinv_bv, const double* r, zvec xout,
            double c0, double c1, double* dot_partial, const double* xadd, const double* dot_with, const int* padd_idx,
            const double* padd_x) {
    if (xout.f32)
        return poly2_t<float>(st, nb, As, dinv, dinv_bv, r, xout.as<float>(), c0, c1, dot_partial, xadd, dot_with, padd_idx, padd_x);
    return poly2_t<double>(st, nb, As, dinv, dinv_bv, r, xout.as<double>(), c0, c1, dot_partial, xadd, dot_with, padd_idx, padd_x);
}

// a level runs the deep gather loop (sell_row_range_deep) when its launch has at most this many wavefronts - about two per
// SIMD of the chip (PMC_DEEP_WAVES in laboratory builds; 0 = never)
static inline bool deep_level(const SellView& A, int nb) {
    static const long limit = [] {
        const char* e = lab_env("PMC_DEEP_WAVES");
        return e ? atol(e) : 0L;   // off in the product: measured neutral on config 2 (LAB_NOTES 10), kept for laboratory runs
    }();
    return nb >= kGroup && !A.bv && (long)A.nslices * (nb / kGroup) <= limit;
}

void vc_presmooth32(hipStream_t st, int nb, const SellView& As, const double* dinv, const double* r, float* xout, double c0,
                    double c1) {
    check_offsets32(As, nb);
    if (As.nrows == 0) return;
    if (As.bv) throw Error(PMC_ERR_INTERNAL, "vc_presmooth32: shared values expected");
    const dim3 g = grid_slices(As.nslices);
    if (As.split_log2) {
        PMC_DISPATCH_NARROW(nb, {
            vc_poly2_kernel<NB, double, float, float, false, false, 0, 1, true><<<g, kBlock, 0, st>>>(As.nrows >> As.split_log2, As.nslices, As.slice_off, As.cols, As.vals, dinv, r, xout, c0, c1, nullptr, nullptr, nullptr, nullptr, nullptr, nb, As.split_log2);
        });
        check_launch();
        return;
    }
    if (deep_level(As, nb)) {
        vc_poly2_kernel<kGroup, double, float, float, false, false, 0, 2><<<groups_xcd(g, nb), kBlock, 0, st>>>(As.nrows, As.nslices, As.slice_off, As.cols, As.vals, dinv, r, xout, c0, c1, nullptr, nullptr, nullptr, nullptr, nullptr, nb);
        check_launch();
        return;
    }
    PMC_DISPATCH_NB(nb, {
        if (nt_poly(As, NB))
            vc_poly2_kernel<NB, double, float, float, false, true><<<groups_xcd(g, nb), kBlock, 0, st>>>(As.nrows, As.nslices, As.slice_off, As.cols, As.vals, dinv, r, xout, c0, c1, nullptr, nullptr, nullptr, nullptr, nullptr, nb);
        else
            vc_poly2_kernel<NB, double, float, float, false><<<groups_xcd(g, nb), kBlock, 0, st>>>(As.nrows, As.nslices, As.slice_off, As.cols, As.vals, dinv, r, xout, c0, c1, nullptr, nullptr, nullptr, nullptr, nullptr, nb);
    });
    check_launch();
}

// the same from the fp32 copy of the right-hand side (the top level of a cycle inside the MINRES loop: the Lanczos update
// writes that copy, k::lincomb3): r is gathered - and read at the own row - as 128-byte fp32 rows instead of 256-byte fp64 ones
void vc_presmooth32_r32(hipStream_t st, int nb, const SellView& As, const double* dinv, const float* r32, float* xout,
                        double c0, double c1) {
    check_offsets32(As, nb);
    if (As.nrows == 0) return;
    if (As.bv) throw Error(PMC_ERR_INTERNAL, "vc_presmooth32_r32: shared values expected");
    const dim3 g = grid_slices(As.nslices);
    PMC_DISPATCH_NB(nb, {
        if (nt_poly(As, NB))
            vc_poly2_kernel<NB, float, float, float, false, true><<<groups_xcd(g, nb), kBlock, 0, st>>>(As.nrows, As.nslices, As.slice_off, As.cols, As.vals, dinv, r32, xout, c0, c1, nullptr, nullptr, nullptr, nullptr, nullptr, nb);
        else
            vc_poly2_kernel<NB, float, float, float, false><<<groups_xcd(g, nb), kBlock, 0, st>>>(As.nrows, As.nslices, As.slice_off, As.cols, As.vals, dinv, r32, xout, c0, c1, nullptr, nullptr, nullptr, nullptr, nullptr, nb);
    });
    check_launch();
}

void vc_residual_restrict8_32(hipStream_t st, int nb, const SellView& A, const double* r, const float* x, float* out,
                              double* coarse) {
    check_offsets32(A, nb);
    if (A.nrows == 0) return;
    if (A.bv || A.nrows % 8 != 0) throw Error(PMC_ERR_INTERNAL, "vc_residual_restrict8_32: shared values and groups of 8 rows expected");
    const dim3 g = grid_slices(A.nslices);
    PMC_DISPATCH_NB(nb, {
        vc_residual_kernel<NB, float, double, float, true><<<groups_xcd(g, nb), kBlock, 0, st>>>(A.nrows, A.nslices, A.slice_off, A.cols, A.vals, x, r, out, coarse, nb);
    });
    check_launch();
}

void vc_residual32(hipStream_t st, int nb, const SellView& A, const double* r, const float* x, float* out) {
    check_offsets32(A, nb);
    if (A.nrows == 0) return;
    if (A.bv) throw Error(PMC_ERR_INTERNAL, "vc_residual32: shared values expected");
    const dim3 g = grid_slices(A.nslices);
    if (A.split_log2) {
        PMC_DISPATCH_NARROW(nb, {
            vc_residual_kernel<NB, float, double, float, false, 0, true, 1, false, true><<<g, kBlock, 0, st>>>(A.nrows >> A.split_log2, A.nslices, A.slice_off, A.cols, A.vals, x, r, out, nullptr, nb, nullptr, nullptr, nullptr, A.split_log2);
        });
        check_launch();
        return;
    }
    if (deep_level(A, nb)) {
        vc_residual_kernel<kGroup, float, double, float, false, 0, true, 4><<<groups_xcd(g, nb), kBlock, 0, st>>>(A.nrows, A.nslices, A.slice_off, A.cols, A.vals, x, r, out, nullptr, nb);
        check_launch();
        return;
    }
    PMC_DISPATCH_NB(nb, {
        vc_residual_kernel<NB, float, double, float, false><<<groups_xcd(g, nb), kBlock, 0, st>>>(A.nrows, A.nslices, A.slice_off, A.cols, A.vals, x, r, out, nullptr, nb);
    });
    check_launch();
}

void vc_residual_restrict_agg32(hipStream_t st, int nb, const SellView& A, const double* r, const float* x, float* out,
                                double* coarse, const int* seg_ptr, const int* seg_cid, const int* seg_pos) {
    check_offsets32(A, nb);
    if (A.nrows == 0) return;
    if (A.bv) throw Error(PMC_ERR_INTERNAL, "vc_residual_restrict_agg32: shared values expected");
    const dim3 g = grid_slices(A.nslices);
    PMC_DISPATCH_NB(nb, {
        vc_residual_kernel<NB, float, double, float, false, 0, true, 1, true><<<groups_xcd(g, nb), kBlock, 0, st>>>(A.nrows, A.nslices, A.slice_off, A.cols, A.vals, x, r, out, coarse, nb, seg_ptr, seg_cid, seg_pos);
    });
    check_launch();
}

void vc_residual_restrict_agg32_r32(hipStream_t st, int nb, const SellView& A, const float* r32, const float* x, float* out,
                                    double* coarse, const int* seg_ptr, const int* seg_cid, const int* seg_pos) {
    check_offsets32(A, nb);
    if (A.nrows == 0) return;
    if (A.bv) throw Error(PMC_ERR_INTERNAL, "vc_residual_restrict_agg32_r32: shared values expected");
    const dim3 g = grid_slices(A.nslices);
    PMC_DISPATCH_NB(nb, {
        vc_residual_kernel<NB, float, float, float, false, 0, true, 1, true><<<groups_xcd(g, nb), kBlock, 0, st>>>(A.nrows, A.nslices, A.slice_off, A.cols, A.vals, x, r32, out, coarse, nb, seg_ptr, seg_cid, seg_pos);
    });
    check_launch();
}

void vc_residual32_r32(hipStream_t st, int nb, const SellView& A, const float* r32, const float* x, float* out) {
    check_offsets32(A, nb);
    if (A.nrows == 0) return;
    if (A.bv) throw Error(PMC_ERR_INTERNAL, "vc_residual32_r32: shared values expected");
    const dim3 g = grid_slices(A.nslices);
    PMC_DISPATCH_NB(nb, {
        vc_residual_kernel<NB, float, float, float, false><<<groups_xcd(g, nb), kBlock, 0, st>>>(A.nrows, A.nslices, A.slice_off, A.cols, A.vals, x, r32, out, nullptr, nb);
    });
    check_launch();
}

void vc_residual_coarse32(hipStream_t st, int nb, const SellView& SP, float* res, const double* xc) {
    check_offsets32(SP, nb);
    if (SP.nrows == 0) return;
    if (SP.bv) throw Error(PMC_ERR_INTERNAL, "vc_residual_coarse32: shared values expected");
    const dim3 g = grid_slices(SP.nslices);
    if (SP.split_log2) {
        PMC_DISPATCH_NARROW(nb, {
            vc_residual_kernel<NB, double, float, float, false, 0, true, 1, false, true><<<g, kBlock, 0, st>>>(SP.nrows >> SP.split_log2, SP.nslices, SP.slice_off, SP.cols, SP.vals, xc, res, res, nullptr, nb, nullptr, nullptr, nullptr, SP.split_log2);
        });
        check_launch();
        return;
    }
    if (deep_level(SP, nb)) {
        vc_residual_kernel<kGroup, double, float, float, false, 0, true, 2><<<groups_xcd(g, nb), kBlock, 0, st>>>(SP.nrows, SP.nslices, SP.slice_off, SP.cols, SP.vals, xc, res, res, nullptr, nb);
        check_launch();
        return;
    }
    PMC_DISPATCH_NB(nb, {
        vc_residual_kernel<NB, double, float, float, false><<<groups_xcd(g, nb), kBlock, 0, st>>>(SP.nrows, SP.nslices, SP.slice_off, SP.cols, SP.vals, xc, res, res, nullptr, nb);
    });
    check_launch();
}

template <typename OT>
static int vc_postsmooth32_t(hipStream_t st, int nb, const SellView& As, const double* dinv, const float* res, const float* x,
                             OT* xout, double c0, double c1, const double* r, const int* parent, const double* xc,
                             double* dot_partial) {
    check_offsets32(As, nb);
    if (As.nrows == 0) return 0;
    if (As.bv) throw Error(PMC_ERR_INTERNAL, "vc_postsmooth32: shared values expected");
    const dim3 g = grid_bounded(grid_slices(As.nslices), dot_partial != nullptr);
    if (As.split_log2) {
        if (dot_partial) throw Error(PMC_ERR_INTERNAL, "vc_postsmooth32: the row-split form serves inner levels (no fused dot)");
        PMC_DISPATCH_NARROW(nb, {
            vc_poly2_kernel<NB, float, OT, float, false, false, 0, 1, true><<<g, kBlock, 0, st>>>(As.nrows >> As.split_log2, As.nslices, As.slice_off, As.cols, As.vals, dinv, res, xout, c0, c1, nullptr, x, nullptr, parent, xc, nb, As.split_log2);
        });
        check_launch();
        return 0;
    }
    if (deep_level(As, nb)) {
        if (dot_partial)
            vc_poly2_kernel<kGroup, float, OT, float, true, false, 0, 4><<<groups_xcd(g, nb), kBlock, 0, st>>>(As.nrows, As.nslices, As.slice_off, As.cols, As.vals, dinv, res, xout, c0, c1, dot_partial, x, r, parent, xc, nb);
        else
            vc_poly2_kernel<kGroup, float, OT, float, false, false, 0, 4><<<groups_xcd(g, nb), kBlock, 0, st>>>(As.nrows, As.nslices, As.slice_off, As.cols, As.vals, dinv, res, xout, c0, c1, nullptr, x, nullptr, parent, xc, nb);
        check_launch();
        return dot_partial ? dot_blocks(g, nb) : 0;
    }
    // 64 per launch on a large level: both column groups of a slice in one workgroup (vc_poly2_kernel, GIB).  Off in the
    // product: one lane 2 870 -> 2 815 samples/s, four lanes 3 745 -> 3 757 (LAB_NOTES 10.19); laboratory switch PMC_GIB=1
    static const bool gib = [] { const char* e = lab_env("PMC_GIB"); return e && atoi(e) != 0; }();
    if (gib && nb == 2 * kGroup && nt_poly(As, kGroup) && !xcd_layout(g, nb)) {
        if (dot_partial)
            vc_poly2_kernel<kGroup, float, OT, float, true, true, 0, 1, false, true><<<g, kBlock, 0, st>>>(As.nrows, As.nslices, As.slice_off, As.cols, As.vals, dinv, res, xout, c0, c1, dot_partial, x, r, parent, xc, nb);
        else
            vc_poly2_kernel<kGroup, float, OT, float, false, true, 0, 1, false, true><<<g, kBlock, 0, st>>>(As.nrows, As.nslices, As.slice_off, As.cols, As.vals, dinv, res, xout, c0, c1, nullptr, x, nullptr, parent, xc, nb);
        check_launch();
        return dot_partial ? (int)g.x : 0;
    }
    PMC_DISPATCH_NB(nb, {
        if (nt_poly(As, NB)) {
            if (dot_partial)
                vc_poly2_kernel<NB, float, OT, float, true, true><<<groups_xcd(g, nb), kBlock, 0, st>>>(As.nrows, As.nslices, As.slice_off, As.cols, As.vals, dinv, res, xout, c0, c1, dot_partial, x, r, parent, xc, nb);
            else
                vc_poly2_kernel<NB, float, OT, float, false, true><<<groups_xcd(g, nb), kBlock, 0, st>>>(As.nrows, As.nslices, As.slice_off, As.cols, As.vals, dinv, res, xout, c0, c1, nullptr, x, nullptr, parent, xc, nb);
        } else if (dot_partial)
            vc_poly2_kernel<NB, float, OT, float, true><<<groups_xcd(g, nb), kBlock, 0, st>>>(As.nrows, As.nslices, As.slice_off, As.cols, As.vals, dinv, res, xout, c0, c1, dot_partial, x, r, parent, xc, nb);
        else
            vc_poly2_kernel<NB, float, OT, float, false><<<groups_xcd(g, nb), kBlock, 0, st>>>(As.nrows, As.nslices, As.slice_off, As.cols, As.vals, dinv, res, xout, c0, c1, nullptr, x, nullptr, parent, xc, nb);
    });
    check_launch();
    return dot_partial ? dot_blocks(g, nb) : 0;
}

int vc_postsmooth32(hipStream_t st, int nb, const SellView& As, const double* dinv, const float* res, const float* x,
                    double* xout, double c0, double c1, const double* r, const int* parent, const double* xc, double* dot_partial) {
    return vc_postsmooth32_t<double>(st, nb, As, dinv, res, x, xout, c0, c1, r, parent, xc, dot_partial);
}
int vc_postsmooth32_z(hipStream_t st, int nb, const SellView& As, const double* dinv, const float* res, const float* x,
                      zvec xout, double c0, double c1, const double* r, const int* parent, const double* xc, double* dot_partial) {
    if (xout.f32) return vc_postsmooth32_t<float>(st, nb, As, dinv, res, x, xout.as<float>(), c0, c1, r, parent, xc, dot_partial);
    return vc_postsmooth32_t<double>(st, nb, As, dinv, res, x, xout.as<double>(), c0, c1, r, parent, xc, dot_partial);
}

// ---- the same level with per-realization fp32 values (Darcy; SellView::f32) and per-realization diagonals
void vc_presmooth32_bv(hipStream_t st, int nb, const SellView& As, const double* dinv, const double* r, float* xout, double c0,
                       double c1) {
    check_offsets32(As, nb);
    if (As.nrows == 0) return;
    if (!(As.bv && As.f32)) throw Error(PMC_ERR_INTERNAL, "vc_presmooth32_bv: per-realization fp32 values expected");
    const dim3 g = grid_slices(As.nslices);
    PMC_DISPATCH_NB(nb, {
        vc_poly2_kernel<NB, double, float, float, false, false, 2><<<groups_xcd(g, nb), kBlock, 0, st>>>(As.nrows, As.nslices, As.slice_off, As.cols, As.vals, dinv, r, xout, c0, c1, nullptr, nullptr, nullptr, nullptr, nullptr, nb);
    });
    check_launch();
}

void vc_restrict8_32_bv(hipStream_t st, int nb, const SellView& A, const double* r, const float* x, double* coarse) {
    check_offsets32(A, nb);
    if (A.nrows == 0) return;
    if (!(A.bv && A.f32) || A.nrows % 8 != 0) throw Error(PMC_ERR_INTERNAL, "vc_restrict8_32_bv: operand mismatch");
    const dim3 g = grid_slices(A.nslices);
    PMC_DISPATCH_NB(nb, {
        vc_residual_kernel<NB, float, double, float, true, 2, false><<<groups_xcd(g, nb), kBlock, 0, st>>>(A.nrows, A.nslices, A.slice_off, A.cols, A.vals, x, r, nullptr, coarse, nb);
    });
    check_launch();
}

void vc_prolong8_32(hipStream_t st, int nb, int n, float* x, const double* xc) {
    if (n == 0) return;
    PMC_DISPATCH_NB(nb, { vc_prolong8_kernel<NB><<<grid_flat(n, nb), kBlock, 0, st>>>(flat_count(n, nb), x, xc, nb); });
    check_launch();
}

void vc_residual32_bv(hipStream_t st, int nb, const SellView& A, const double* r, const float* x, float* out) {
    check_offsets32(A, nb);
    if (A.nrows == 0) return;
    if (!(A.bv && A.f32)) throw Error(PMC_ERR_INTERNAL, "vc_residual32_bv: per-realization fp32 values expected");
    const dim3 g = grid_slices(A.nslices);
    PMC_DISPATCH_NB(nb, {
        vc_residual_kernel<NB, float, double, float, false, 2><<<groups_xcd(g, nb), kBlock, 0, st>>>(A.nrows, A.nslices, A.slice_off, A.cols, A.vals, x, r, out, nullptr, nb);
    });
    check_launch();
}

template <typename OT>
static int vc_postsmooth32_bv_t(hipStream_t st, int nb, const SellView& As, const double* dinv, const float* res, const float* x,
                                OT* xout, double c0, double c1, const double* r, double* dot_partial) {
    check_offsets32(As, nb);
    if (As.nrows == 0) return 0;
    if (!(As.bv && As.f32)) throw Error(PMC_ERR_INTERNAL, "vc_postsmooth32_bv: per-realization fp32 values expected");
    const dim3 g = grid_bounded(grid_slices(As.nslices), dot_partial != nullptr);
    PMC_DISPATCH_NB(nb, {
        if (dot_partial)
            vc_poly2_kernel<NB, float, OT, float, true, false, 2><<<groups_xcd(g, nb), kBlock, 0, st>>>(As.nrows, As.nslices, As.slice_off, As.cols, As.vals, dinv, res, xout, c0, c1, dot_partial, x, r, nullptr, nullptr, nb);
        else
            vc_poly2_kernel<NB, float, OT, float, false, false, 2><<<groups_xcd(g, nb), kBlock, 0, st>>>(As.nrows, As.nslices, As.slice_off, As.cols, As.vals, dinv, res, xout, c0, c1, nullptr, x, nullptr, nullptr, nullptr, nb);
    });
    check_launch();
    return dot_partial ? dot_blocks(g, nb) : 0;
}

int vc_postsmooth32_bv(hipStream_t st, int nb, const SellView& As, const double* dinv, const float* res, const float* x,
                       double* xout, double c0, double c1, const double* r, double* dot_partial) {
    return vc_postsmooth32_bv_t<double>(st, nb, As, dinv, res, x, xout, c0, c1, r, dot_partial);
}
int vc_postsmooth32_bv_z(hipStream_t st, int nb, const SellView& As, const double* dinv, const float* res, const float* x,
                         zvec xout, double c0, double c1, const double* r, double* dot_partial) {
    if (xout.f32) return vc_postsmooth32_bv_t<float>(st, nb, As, dinv, res, x, xout.as<float>(), c0, c1, r, dot_partial);
    return vc_postsmooth32_bv_t<double>(st, nb, As, dinv, res, x, xout.as<double>(), c0, c1, r, dot_partial);
}

template <typename XT>
static int eg_pair_spmm_t(hipStream_t st, int nb, const EgView& M, const double* coef, const XT* x1, const SellView& A2,
                          const XT* x2, double* y, double* dot_partial, const XT* dot_with) {
    if (M.nrows == 0) return 0;
    if (A2.bv || A2.nrows != M.nrows || A2.nslices != M.nslices)
        throw Error(PMC_ERR_INTERNAL, "eg_pair_spmm: second operator must share the rows and carry shared values");
    // the kernels address their gathers with 32-bit element offsets (sell_row_part)
    if ((uint64_t)std::max(M.nrows, A2.ncols_hint) * (uint64_t)nb >= (1ull << 32))
        throw Error(PMC_ERR_INVALID, "eg_pair_spmm: rows x batch width exceed 32-bit gather offsets");
    const dim3 g = grid_bounded(grid_slices(M.nslices), dot_partial != nullptr);
    PMC_DISPATCH_NB(nb, {
        if (nt_flat((size_t)M.nrows * NB * 2)) {   // from 4 MiB per vector on
            if (dot_partial)
                eg_pair_spmm_kernel<NB, true, true, XT><<<groups_xcd(g, nb), kBlock, 0, st>>>(M.nrows, M.nslices, M.gw, M.cols, M.w, M.e12, coef, A2.slice_off, A2.cols, A2.vals, x1, x2, y, dot_with, dot_partial, nb);
            else
                eg_pair_spmm_kernel<NB, false, true, XT><<<groups_xcd(g, nb), kBlock, 0, st>>>(M.nrows, M.nslices, M.gw, M.cols, M.w, M.e12, coef, A2.slice_off, A2.cols, A2.vals, x1, x2, y, nullptr, nullptr, nb);
        } else if (dot_partial)
            eg_pair_spmm_kernel<NB, true, false, XT><<<groups_xcd(g, nb), kBlock, 0, st>>>(M.nrows, M.nslices, M.gw, M.cols, M.w, M.e12, coef, A2.slice_off, A2.cols, A2.vals, x1, x2, y, dot_with, dot_partial, nb);
        else
            eg_pair_spmm_kernel<NB, false, false, XT><<<groups_xcd(g, nb), kBlock, 0, st>>>(M.nrows, M.nslices, M.gw, M.cols, M.w, M.e12, coef, A2.slice_off, A2.cols, A2.vals, x1, x2, y, nullptr, nullptr, nb);
    });
    check_launch();
    return dot_partial ? dot_blocks(g, nb) : 0;
}

int eg_pair_spmm(hipStream_t st, int nb, const EgView& M, const double* coef, const double* x1, const SellView& A2,
                 const double* x2, double* y, double* dot_partial, const double* dot_with) {
    return eg_pair_spmm_t<double>(st, nb, M, coef, x1, A2, x2, y, dot_partial, dot_with);
}
int eg_pair_spmm_z(hipStream_t st, int nb, const EgView& M, const double* coef, zvec x1, const SellView& A2,
                   zvec x2, double* y, double* dot_partial, zvec dot_with) {
    if (x1.f32)
        return eg_pair_spmm_t<float>(st, nb, M, coef, x1.as<float>(), A2, x2.as<float>(), y, dot_partial, dot_with.as<float>());
    return eg_pair_spmm_t<double>(st, nb, M, coef, x1.as<double>(), A2, x2.as<double>(), y, dot_partial, dot_with.as<double>());
}

template <typename OT>
static int eg_poly2_t(hipStream_t st, int nb, const EgView& M, const double* coef, const double* dinv, const double* r, OT* xout,
                      double c0, double c1, double* dot_partial) {
    if (M.nrows == 0) return 0;
    if ((uint64_t)M.nrows * (uint64_t)nb >= (1ull << 32))
        throw Error(PMC_ERR_INVALID, "eg_poly2: rows x batch width exceed 32-bit gather offsets");
    const dim3 g = grid_bounded(grid_slices(M.nslices), dot_partial != nullptr);
    PMC_DISPATCH_NB(nb, {
        if (nt_flat((size_t)M.nrows * NB * 2)) {
            if (dot_partial)
                eg_poly2_kernel<NB, true, true, OT><<<groups_xcd(g, nb), kBlock, 0, st>>>(M.nrows, M.nslices, M.gw, M.cols, M.w, M.e12, coef, dinv, r, xout, c0, c1, dot_partial, nb);
            else
                eg_poly2_kernel<NB, false, true, OT><<<groups_xcd(g, nb), kBlock, 0, st>>>(M.nrows, M.nslices, M.gw, M.cols, M.w, M.e12, coef, dinv, r, xout, c0, c1, nullptr, nb);
        } else if (dot_partial)
            eg_poly2_kernel<NB, true, false, OT><<<groups_xcd(g, nb), kBlock, 0, st>>>(M.nrows, M.nslices, M.gw, M.cols, M.w, M.e12, coef, dinv, r, xout, c0, c1, dot_partial, nb);
        else
            eg_poly2_kernel<NB, false, false, OT><<<groups_xcd(g, nb), kBlock, 0, st>>>(M.nrows, M.nslices, M.gw, M.cols, M.w, M.e12, coef, dinv, r, xout, c0, c1, nullptr, nb);
    });
    check_launch();
    return dot_partial ? dot_blocks(g, nb) : 0;
}

int eg_poly2(hipStream_t st, int nb, const EgView& M, const double* coef, const double* dinv, const double* r, double* xout,
             double c0, double c1, double* dot_partial) {
    return eg_poly2_t<double>(st, nb, M, coef, dinv, r, xout, c0, c1, dot_partial);
}
int eg_poly2_z(hipStream_t st, int nb, const EgView& M, const double* coef, const double* dinv, const double* r, zvec xout,
               double c0, double c1, double* dot_partial) {
    if (xout.f32) return eg_poly2_t<float>(st, nb, M, coef, dinv, r, xout.as<float>(), c0, c1, dot_partial);
    return eg_poly2_t<double>(st, nb, M, coef, dinv, r, xout.as<double>(), c0, c1, dot_partial);
}

template <typename XT>
static int pair_spmm_t(hipStream_t st, int nb, const SellView& A1, const XT* x1, const SellView& A2, const XT* x2, double* y,
                       double* dot_partial, const XT* dot_with) {
    check_offsets32(A1, nb);
    if (A1.nrows == 0) return 0;
    if (!A1.bv || A2.bv || A1.nrows != A2.nrows || A1.nslices != A2.nslices)
        throw Error(PMC_ERR_INTERNAL, "pair_spmm: operand mismatch");
    const dim3 g = grid_bounded(grid_slices(A1.nslices), dot_partial != nullptr);
    PMC_DISPATCH_NB(nb, {
        if (dot_partial)
            sell_pair_spmm_kernel<NB, true, XT><<<groups_xcd(g, nb), kBlock, 0, st>>>(A1.nrows, A1.nslices, A1.slice_off, A1.cols, A1.vals, A2.slice_off, A2.cols, A2.vals, x1, x2, y, dot_with, dot_partial, nb);
        else
            sell_pair_spmm_kernel<NB, false, XT><<<groups_xcd(g, nb), kBlock, 0, st>>>(A1.nrows, A1.nslices, A1.slice_off, A1.cols, A1.vals, A2.slice_off, A2.cols, A2.vals, x1, x2, y, nullptr, nullptr, nb);
    });
    check_launch();
    return dot_partial ? dot_blocks(g, nb) : 0;
}

int pair_spmm(hipStream_t st, int nb, const SellView& A1, const double* x1, const SellView& A2, const double* x2, double* y,
              double* dot_partial, const double* dot_with) {
    return pair_spmm_t<double>(st, nb, A1, x1, A2, x2, y, dot_partial, dot_with);
}
int pair_spmm_z(hipStream_t st, int nb, const SellView& A1, zvec x1, const SellView& A2, zvec x2, double* y,
                double* dot_partial, zvec dot_with) {
    if (x1.f32) return pair_spmm_t<float>(st, nb, A1, x1.as<float>(), A2, x2.as<float>(), y, dot_partial, dot_with.as<float>());
    return pair_spmm_t<double>(st, nb, A1, x1.as<double>(), A2, x2.as<double>(), y, dot_partial, dot_with.as<double>());
}

void scale_cols_bv(hipStream_t st, int nb, int64_t nslots, const int* cols, const double* vals, const double* colscale,
                   double* out) {
    if (nslots == 0) return;
    const size_t nf = (size_t)nslots * nb / lay_c(nb);
    const unsigned g = (unsigned)std::min<size_t>((nf + kBlock - 1) / kBlock, 8192);
    PMC_DISPATCH_NB(nb, { scale_cols_bv_kernel<NB><<<g, kBlock, 0, st>>>(nf, cols, vals, colscale, out, nb); });
    check_launch();
}

void scale_cols_bv32(hipStream_t st, int nb, int64_t nslots, const int* cols, const double* vals, const double* colscale,
                     float* out_scaled, float* out_vals) {
    if (nslots == 0) return;
    const size_t nf = (size_t)nslots * nb / lay_c(nb);
    const unsigned g = (unsigned)std::min<size_t>((nf + kBlock - 1) / kBlock, 8192);
    PMC_DISPATCH_NB(nb, { scale_cols_bv32_kernel<NB><<<g, kBlock, 0, st>>>(nf, cols, vals, colscale, out_scaled, out_vals, nb); });
    check_launch();
}

void minres_wx_idx(hipStream_t st, int nb, int nsel, const int* rows, const double* c0, zvec u, const double* c1,
                   double* w0, const double* c2, const double* w1, const double* c3, double* x) {
    if (nsel == 0) return;
    PMC_DISPATCH_NB(nb, {
        if (u.f32) minres_wx_idx_kernel<NB, float><<<grid_flat(nsel, nb), kBlock, 0, st>>>(flat_count(nsel, nb), rows, c0, u.as<float>(), c1, w0, c2, w1, c3, x, nb);
        else minres_wx_idx_kernel<NB, double><<<grid_flat(nsel, nb), kBlock, 0, st>>>(flat_count(nsel, nb), rows, c0, u.as<double>(), c1, w0, c2, w1, c3, x, nb);
    });
    check_launch();
}

int cheb_first(hipStream_t st, int nb, int n, const double* dinv, bool dinv_bv, const double* r, double* d, double* x,
               double b, double* dot_partial) {
    if (n == 0) return 0;
    const size_t nf = flat_count(n, nb);
    const dim3 g = grid_bounded(grid_flat(n, nb), dot_partial != nullptr);
    PMC_DISPATCH_NB(nb, {
        if (dinv_bv) {
            if (dot_partial) cheb_first_kernel<NB, true, true><<<g, kBlock, 0, st>>>(nf, dinv, r, d, x, b, dot_partial, nb);
            else cheb_first_kernel<NB, true, false><<<g, kBlock, 0, st>>>(nf, dinv, r, d, x, b, nullptr, nb);
        } else {
            if (dot_partial) cheb_first_kernel<NB, false, true><<<g, kBlock, 0, st>>>(nf, dinv, r, d, x, b, dot_partial, nb);
            else cheb_first_kernel<NB, false, false><<<g, kBlock, 0, st>>>(nf, dinv, r, d, x, b, nullptr, nb);
        }
    });
    check_launch();
    return dot_partial ? (int)g.x : 0;
}

static inline dim3 grid_dot(int n, int nb) { return dim3(std::min(grid_flat(n, nb).x, 1024u)); }

int dot(hipStream_t st, int nb, int n, const double* a, const double* b, double* partial) {
    const dim3 g = grid_dot(n, nb);
    PMC_DISPATCH_NB(nb, { dot_kernel<NB><<<g, kBlock, 0, st>>>(flat_count(n, nb), a, b, partial, nb); });
    check_launch();
    return (int)g.x;
}

int dot_z(hipStream_t st, int nb, int n, const double* a, zvec b, double* partial) {
    const dim3 g = grid_dot(n, nb);
    PMC_DISPATCH_NB(nb, {
        if (b.f32) dot_kernel<NB, float><<<g, kBlock, 0, st>>>(flat_count(n, nb), a, b.as<float>(), partial, nb);
        else dot_kernel<NB, double><<<g, kBlock, 0, st>>>(flat_count(n, nb), a, b.as<double>(), partial, nb);
    });
    check_launch();
    return (int)g.x;
}

int convert_z(hipStream_t st, int nb, int n, const double* in, zvec out, const double* r, double* dot_partial) {
    if (n == 0) return 0;
    const dim3 g = grid_dot(n, nb);
    PMC_DISPATCH_NB(nb, {
        if (out.f32) {
            if (dot_partial) convert_dot_kernel<NB, true, float><<<g, kBlock, 0, st>>>(flat_count(n, nb), in, out.as<float>(), r, dot_partial, nb);
            else convert_dot_kernel<NB, false, float><<<g, kBlock, 0, st>>>(flat_count(n, nb), in, out.as<float>(), nullptr, nullptr, nb);
        } else {
            if (dot_partial) convert_dot_kernel<NB, true, double><<<g, kBlock, 0, st>>>(flat_count(n, nb), in, out.as<double>(), r, dot_partial, nb);
            else convert_dot_kernel<NB, false, double><<<g, kBlock, 0, st>>>(flat_count(n, nb), in, out.as<double>(), nullptr, nullptr, nb);
        }
    });
    check_launch();
    return dot_partial ? (int)g.x : 0;
}

int wdot(hipStream_t st, int nb, int n, const double* w, const double* x, double* partial) {
    const dim3 g = grid_dot(n, nb);
    PMC_DISPATCH_NB(nb, { wdot_kernel<NB><<<g, kBlock, 0, st>>>(flat_count(n, nb), w, x, partial, nb); });
    check_launch();
    return (int)g.x;
}

void reduce_final(hipStream_t st, int nb, int nblocks, const double* partial, double* out) {
    reduce_final_kernel<<<1, kScalBlock, 0, st>>>(partial, nblocks, nb, out);
    check_launch();
}

void lincomb3(hipStream_t st, int nb, int n, const double* c0, const double* a, const double* c1, const double* b,
              const double* c2, double* y, float* y32) {
    // non-temporal loads on large levels (one lane 1104 -> 1129, four lanes 1400 -> 1412 samples/s); PMC_NT_LINCOMB=0: off
    static const bool nt_on = [] { const char* e = lab_env("PMC_NT_LINCOMB"); return !e || atoi(e) != 0; }();
    const bool nt = nt_on && nt_flat((size_t)n * nb);
    PMC_DISPATCH_NB(nb, {
        if (nt) lincomb3_kernel<NB, true><<<grid_flat(n, nb), kBlock, 0, st>>>(flat_count(n, nb), c0, a, c1, b, c2, y, nb, y32);
        else lincomb3_kernel<NB><<<grid_flat(n, nb), kBlock, 0, st>>>(flat_count(n, nb), c0, a, c1, b, c2, y, nb, y32);
    });
    check_launch();
}

template <typename UT>
static void minres_wx_t(hipStream_t st, int nb, int n, const double* c0, const UT* u, const double* c1, double* w0,
                        const double* c2, const double* w1, const double* c3, double* x) {
    const bool nt = nt_flat((size_t)n * nb);
    PMC_DISPATCH_NB(nb, {
        if (nt) minres_wx_kernel<NB, true, UT><<<grid_flat(n, nb), kBlock, 0, st>>>(flat_count(n, nb), c0, u, c1, w0, c2, w1, c3, x, nb);
        else minres_wx_kernel<NB, false, UT><<<grid_flat(n, nb), kBlock, 0, st>>>(flat_count(n, nb), c0, u, c1, w0, c2, w1, c3, x, nb);
    });
}
void minres_wx(hipStream_t st, int nb, int n, const double* c0, zvec u, const double* c1, double* w0,
               const double* c2, const double* w1, const double* c3, double* x) {
    if (u.f32) minres_wx_t<float>(st, nb, n, c0, u.as<float>(), c1, w0, c2, w1, c3, x);
    else minres_wx_t<double>(st, nb, n, c0, u.as<double>(), c1, w0, c2, w1, c3, x);
    check_launch();
}

void minres_wx_deferred(hipStream_t st, int nb, int n, const MinresState* s, const WxDeferred& B, double* w0, double* w1,
                        double* x) {
    if (n == 0 || B.cnt == 0) return;
    if (B.cnt < 0 || B.cnt > kWxDefer) throw Error(PMC_ERR_INTERNAL, "minres_wx_deferred: bad count");
    const double* cW = reinterpret_cast<const double*>(reinterpret_cast<const char*>(s) + offsetof(MinresState, cW));
    const bool nt = nt_flat((size_t)n * nb);
    PMC_DISPATCH_NB(nb, {
        if (B.f32) {
            if (nt) minres_wx_deferred_kernel<NB, true, float><<<grid_flat(n, nb), kBlock, 0, st>>>(flat_count(n, nb), B, cW, w0, w1, x, nb);
            else minres_wx_deferred_kernel<NB, false, float><<<grid_flat(n, nb), kBlock, 0, st>>>(flat_count(n, nb), B, cW, w0, w1, x, nb);
        } else {
            if (nt) minres_wx_deferred_kernel<NB, true, double><<<grid_flat(n, nb), kBlock, 0, st>>>(flat_count(n, nb), B, cW, w0, w1, x, nb);
            else minres_wx_deferred_kernel<NB, false, double><<<grid_flat(n, nb), kBlock, 0, st>>>(flat_count(n, nb), B, cW, w0, w1, x, nb);
        }
    });
    check_launch();
}

void fill(hipStream_t st, size_t n, double* x, double v) {
    if (n == 0) return;
    if (v == 0.0) {
        PMC_HIP(hipMemsetAsync(x, 0, n * sizeof(double), st));
        return;
    }
    const unsigned g = (unsigned)std::min<size_t>((n + kBlock - 1) / kBlock, 2048);
    fill_kernel<<<g, kBlock, 0, st>>>(n, x, v);
    check_launch();
}

__global__ __launch_bounds__(kBlock) void scale_kernel(size_t n, const double* __restrict__ in, double a, double* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) out[i] = a * in[i];
}
void scale(hipStream_t st, size_t n, const double* in, double a, double* out) {
    if (n == 0) return;
    const unsigned g = (unsigned)std::min<size_t>((n + kBlock - 1) / kBlock, 2048);
    scale_kernel<<<g, kBlock, 0, st>>>(n, in, a, out);
    check_launch();
}

void copy(hipStream_t st, size_t n, const double* src, double* dst) {
    if (n && src != dst) PMC_HIP(hipMemcpyAsync(dst, src, n * sizeof(double), hipMemcpyDeviceToDevice, st));
}

void minres_init(hipStream_t st, int nb, MinresState* s, const DotParts& d, double rel_tol, double abs_tol, int ring) {
    minres_init_kernel<<<1, kScalBlock, 0, st>>>(s, d.p1, d.n1, nb, rel_tol, abs_tol, d.p2, d.n2, ring);
    check_launch();
}
void minres_scal1(hipStream_t st, int nb, MinresState* s, const DotParts& d) {
    minres_scal1_kernel<<<1, kScalBlock, 0, st>>>(s, d.p1, d.n1, nb, d.p2, d.n2);
    check_launch();
}
void minres_scal2(hipStream_t st, int nb, MinresState* s, const DotParts& d) {
    minres_scal2_kernel<<<1, kScalBlock, 0, st>>>(s, d.p1, d.n1, nb, d.p2, d.n2);
    check_launch();
}
size_t scal_stage_doubles() { return (size_t)2 * kStageBlocks * kMaxBatch; }
void minres_scal21(hipStream_t st, int nb, MinresState* s, const DotParts& d2, const DotParts& d1, double* stage) {
    if (stage && d2.total() + d1.total() >= 8 * kStageBlocks) {
        stage_partials_kernel<<<kStageBlocks, 256, 0, st>>>(d2.p1, d2.n1, d2.p2, d2.n2, d1.p1, d1.n1, d1.p2, d1.n2, nb, stage);
        check_launch();
        minres_scal21_kernel<<<1, kScalBlock, 0, st>>>(s, stage, kStageBlocks, nullptr, 0,
                                                        stage + (size_t)kStageBlocks * nb, kStageBlocks, nullptr, 0, nb);
    } else {
        minres_scal21_kernel<<<1, kScalBlock, 0, st>>>(s, d2.p1, d2.n1, d2.p2, d2.n2, d1.p1, d1.n1, d1.p2, d1.n2, nb);
    }
    check_launch();
}

void normal_fill(hipStream_t st, int n, int nbatch, uint64_t seed, uint64_t first_id, uint32_t stream, double mean,
                 double sigma, double* out, uint64_t id_stride) {
    if (n == 0 || nbatch == 0) return;
    const int npair = (n + 1) / 2;
    dim3 g((unsigned)((npair + kBlock - 1) / kBlock), (unsigned)nbatch);
    normal_fill_kernel<<<g, kBlock, 0, st>>>(n, nbatch, seed, first_id, id_stride, stream, mean, sigma, out);
    check_launch();
}

void interleave(hipStream_t st, int nb, int n, const double* in, const double* w, double scale, double* out, const int* src) {
    PMC_DISPATCH_NB(nb, { interleave_kernel<NB><<<groups(grid_rows(n), nb), kBlock, 0, st>>>(n, in, w, scale, out, nb, src); });
    check_launch();
}

void deinterleave(hipStream_t st, int nb, int m, const double* in, const int* idx, const double* rowscale, bool do_exp,
                  double* out) {
    if (m == 0) return;
    PMC_DISPATCH_NB(nb, { deinterleave_kernel<NB><<<groups(grid_rows(m), nb), kBlock, 0, st>>>(m, in, idx, rowscale, do_exp ? 1 : 0, out, nb); });
    check_launch();
}

void broadcast(hipStream_t st, int nb, int n, const double* a, double* out) {
    PMC_DISPATCH_NB(nb, { broadcast_kernel<NB><<<groups(grid_rows(n), nb), kBlock, 0, st>>>(n, a, out, nb); });
    check_launch();
}

void darcy_coef(hipStream_t st, int nb, int n, const double* kfield, bool k_divides, double* coef) {
    PMC_DISPATCH_NB(nb, { darcy_coef_kernel<NB><<<groups(grid_rows(n), nb), kBlock, 0, st>>>(n, kfield, k_divides ? 1 : 0, coef, nb); });
    check_launch();
}

void darcy_assemble(hipStream_t st, int nb, const SellView& Mp, const int* slot_src, const int* c_ptr, const int* c_elem,
                    const double* c_val, const double* coef, const unsigned char* ess, const double* ess_data,
                    const double* rhs0, double* mvals, double* diag, double* l1inv, double* rhs_bc) {
    const dim3 g = grid_rows(Mp.nslices * kWave);
    PMC_DISPATCH_NB(nb, {
        darcy_assemble_kernel<NB><<<groups(g, nb), kBlock, 0, st>>>(Mp.nrows, Mp.nslices, Mp.slice_off, Mp.cols, slot_src, c_ptr,
                                                                   c_elem, c_val, coef, ess, ess_data, rhs0, mvals, diag, l1inv,
                                                                   rhs_bc, nb);
    });
    check_launch();
}

void gersh_scale_bv(hipStream_t st, int nb, const SellView& S, double* dinv, double* gwork) {
    unsigned long long* g = reinterpret_cast<unsigned long long*>(gwork);
    PMC_HIP(hipMemsetAsync(g, 0, sizeof(unsigned long long) * kMaxBatch, st));
    const int grid = (S.nrows + kBlock - 1) / kBlock;
    PMC_DISPATCH_NB(nb, {
        gersh_bv_kernel<NB><<<groups(grid, nb), kBlock, 0, st>>>(S.nrows, S.slice_off, S.vals, dinv, g, nb);
        gersh_scale_kernel<NB><<<groups(grid, nb), kBlock, 0, st>>>(S.nrows, g, dinv, nb);
    });
    check_launch(2);
}

void refresh(hipStream_t st, int nb, int64_t nslots, const int* ptr, const int* idx, const double* w, const double* src,
             bool recip, double* out) {
    if (nslots == 0) return;
    const dim3 g((unsigned)((nslots + kBlock - 1) / kBlock));
    PMC_DISPATCH_NB(nb, { refresh_kernel<NB><<<groups(g, nb), kBlock, 0, st>>>(nslots, ptr, idx, w, src, recip ? 1 : 0, out, nb); });
    check_launch();
}

void diag_inv(hipStream_t st, int nb, int n, const int* diag_slot, const double* vals, double* dinv) {
    PMC_DISPATCH_NB(nb, { diag_inv_kernel<NB><<<groups(grid_rows(n), nb), kBlock, 0, st>>>(n, diag_slot, vals, dinv, nb); });
    check_launch();
}

void mini_sampler_solve(hipStream_t st, int nb, const MiniSamplerParams& P, size_t lds_doubles, const double* b, double* x,
                        bool zero_guess, double* scratch, pmc_stats* stats) {
    const size_t bytes = lds_doubles * sizeof(double);
    if (bytes > kTailLdsBytes) throw Error(PMC_ERR_INTERNAL, "mini solver: LDS request too large");
    static std::atomic<unsigned long long> attr_mask{0};
    int dev = 0;
    PMC_HIP(hipGetDevice(&dev));
    if (!(attr_mask.load() & (1ull << (dev & 63)))) {
        PMC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(mini_sampler_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)kTailLdsBytes));
        attr_mask.fetch_or(1ull << (dev & 63));
    }
    mini_sampler_kernel<<<nb, kTailThreads, bytes, st>>>(P, nb, b, x, zero_guess ? 1 : 0, scratch, stats);
    check_launch();
}

__global__ __launch_bounds__(kBlock) void transpose_bv_kernel(size_t count, int nb, const double* __restrict__ in,
                                                              double* __restrict__ out) {
    const size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x;   // coalesced reads; the writes of a small array stay in L2
    if (e >= count * nb) return;
    const size_t i = e / nb;
    const int k = (int)(e % nb);
    out[(size_t)k * count + i] = in[e];
}

__global__ __launch_bounds__(kBlock) void transpose_bv32_kernel(size_t count, int nb, const float* __restrict__ in,
                                                                double* __restrict__ out) {
    const size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= count * nb) return;
    const size_t i = e / nb;
    const int k = (int)(e % nb);
    out[(size_t)k * count + i] = (double)in[e];
}

void transpose_bv32(hipStream_t st, int nb, size_t count, const float* in, double* out) {
    if (count == 0) return;
    const size_t total = count * nb;
    transpose_bv32_kernel<<<(unsigned)((total + kBlock - 1) / kBlock), kBlock, 0, st>>>(count, nb, in, out);
    check_launch();
}

void transpose_bv(hipStream_t st, int nb, size_t count, const double* in, double* out) {
    if (count == 0) return;
    const size_t total = count * nb;
    transpose_bv_kernel<<<(unsigned)((total + kBlock - 1) / kBlock), kBlock, 0, st>>>(count, nb, in, out);
    check_launch();
}

// Back-substitution of the hybridized Darcy system (DarcyHybrid): interleaved [row][nb] vectors, one thread per entry.
//   flux:     out[f][k] = kappa[owner[f]][k] * (U0[f] - t[f][k]) + ug[f]          t = U_L lambda
//   pressure: out[e][k] = P0[e] - t[e][k] - zg[e] / kappa[e][k]                  t = P_L lambda
__global__ __launch_bounds__(kBlock) void darcy_backsub_u_kernel(size_t total, int nb, const int* __restrict__ owner,
                                                                  const double* __restrict__ kappa, const double* __restrict__ U0,
                                                                  const double* __restrict__ ug, const double* __restrict__ t,
                                                                  double* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= total) return;
    const size_t f = i / (size_t)nb;
    const int k = (int)(i % (size_t)nb);
    out[i] = kappa[(size_t)owner[f] * nb + k] * (U0[f] - t[i]) + ug[f];
}
__global__ __launch_bounds__(kBlock) void darcy_backsub_p_kernel(size_t total, int nb, const double* __restrict__ kappa,
                                                                  const double* __restrict__ P0, const double* __restrict__ zg,
                                                                  const double* __restrict__ t, double* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= total) return;
    const size_t e = i / (size_t)nb;
    out[i] = P0[e] - t[i] - zg[e] / kappa[i];
}

void darcy_backsub_u(hipStream_t st, int nb, int n_u, const int* owner, const double* kappa, const double* U0, const double* ug,
                     const double* t, double* out) {
    const size_t total = (size_t)n_u * nb;
    if (total == 0) return;
    darcy_backsub_u_kernel<<<(unsigned)((total + kBlock - 1) / kBlock), kBlock, 0, st>>>(total, nb, owner, kappa, U0, ug, t, out);
    check_launch();
}
void darcy_backsub_p(hipStream_t st, int nb, int n_p, const double* kappa, const double* P0, const double* zg, const double* t,
                     double* out) {
    const size_t total = (size_t)n_p * nb;
    if (total == 0) return;
    darcy_backsub_p_kernel<<<(unsigned)((total + kBlock - 1) / kBlock), kBlock, 0, st>>>(total, nb, kappa, P0, zg, t, out);
    check_launch();
}

// x[i][k] = sum_j ainv[i][j] r[j][k] for a launch of at most 8 realizations: one wavefront per row, lanes over the columns of the
// (symmetric, row-major) dense inverse, so the matrix is read once, coalesced, by n wavefronts spread over the chip - the exact
// solve of a level of a few hundred rows that a narrow launch would otherwise cycle through in ONE workgroup's LDS tail
__global__ __launch_bounds__(kBlock) void dense_apply_kernel(int n, int nb, const double* __restrict__ ainv,
                                                              const double* __restrict__ r, double* __restrict__ x) {
    const int lane = threadIdx.x & (kWave - 1);
    const int i = blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    if (i >= n) return;
    double acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.0;
    const double* row = ainv + (size_t)i * n;
    for (int j = lane; j < n; j += kWave) {
        const double a = row[j];
        const double* rj = r + (size_t)j * nb;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k < nb) acc[k] = fma(a, rj[k], acc[k]);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (k >= nb) break;
        double v = acc[k];
#pragma unroll
        for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
        if (lane == 0) x[(size_t)i * nb + k] = v;
    }
}

void dense_apply(hipStream_t st, int nb, int n, const double* ainv, const double* r, double* x) {
    if (nb < 1 || nb > 8) throw Error(PMC_ERR_INTERNAL, "dense_apply: serves launches of at most 8 realizations");
    if (n <= 0) return;
    const int rows_per_block = kBlock / kWave;
    dense_apply_kernel<<<(n + rows_per_block - 1) / rows_per_block, kBlock, 0, st>>>(n, nb, ainv, r, x);
    check_launch();
}

int mg_tail(hipStream_t st, int nb, const TailParams* dev_params, size_t lds_doubles, const double* r, double* xout,
            double* dot_partial, bool out32) {
    const size_t bytes = lds_doubles * sizeof(double);
    if (bytes > kTailLdsBytes) throw Error(PMC_ERR_INTERNAL, "mg_tail: LDS request too large");
    // the dynamic-LDS limit is a per-device function attribute: raise it once per device (idempotent if two lanes race)
    static std::atomic<unsigned long long> attr_mask{0};
    int dev = 0;
    PMC_HIP(hipGetDevice(&dev));
    if (!(attr_mask.load() & (1ull << (dev & 63)))) {
        PMC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(mg_tail_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)kTailLdsBytes));
        attr_mask.fetch_or(1ull << (dev & 63));
    }
    mg_tail_kernel<<<nb, kTailThreads, bytes, st>>>(dev_params, nb, r, xout, dot_partial, out32 ? 1 : 0);
    check_launch();
    return dot_partial ? 1 : 0;
}

}  // namespace k
}  // namespace pmc
