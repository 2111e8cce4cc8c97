// Kernel laboratory for the block operator K5 (not part of the product): times the gather kernel next to flat copies of
// the same byte count on a matrix exported by k5_lab.py, back to back (operator resident in the Infinity Cache) and with ~1.5 GB of
// streaming traffic between launches (the state in which the MINRES loop finds the operator).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../parelagmc_amd/csrc/kernels.hpp"

using namespace pmc;

__global__ void lab_stream_kernel(size_t n, const double2* __restrict__ a, double2* __restrict__ b) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        double2 v = a[i];
        v.x += 1.0;
        b[i] = v;
    }
}

// probe: the gather kernel's structure (one wave per slice, 8 lanes per row, shuffled (value, index) pairs, 16 FMAs per
// slice column) with every x row loaded ONCE per row instead of once per entry: what the kernel would cost if the
// per-entry gathers through the vector L1 were free.  Only meaningful with LAB_COLS=1 (every entry refers to its own row).
__global__ __launch_bounds__(256) void lab_oneload_kernel(int nrows, int nslices, const int* __restrict__ slice_off,
                                                          const int* __restrict__ sched, const int* __restrict__ cols,
                                                          const double* __restrict__ vals, const double* __restrict__ x,
                                                          double* __restrict__ y, int mode) {
    const int lane = threadIdx.x & 63;
    const int g = lane / 8, t = lane % 8;
    const int si = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (si >= nslices) return;
    const int slice = sched ? sched[si] : si;
    const int off = slice_off[slice];
    const int width = (slice_off[slice + 1] - off) >> 6;
    double acc[8][2], xr[8][2];
    for (int rs = 0; rs < 8; ++rs) {
        const int row = min(slice * 64 + rs * 8 + g, nrows - 1);
        const double2 v = *reinterpret_cast<const double2*>(x + (size_t)row * 16 + t * 2);
        xr[rs][0] = v.x; xr[rs][1] = v.y;
        acc[rs][0] = acc[rs][1] = 0.0;
    }
    int slot = off + lane;
    int cj = cols[slot];
    double vj = vals[slot];
    for (int j = 0; j < width; ++j, slot += 64) {
        int cn = cj;
        double vn = vj;
        if (j + 1 < width) { cn = cols[slot + 64]; vn = vals[slot + 64]; }
#pragma unroll
        for (int rs = 0; rs < 8; ++rs) {
            const int src = rs * 8 + g;
            const int cc = __shfl(cj, src, 64);
            const double aa = __shfl(vj, src, 64);
            double x0 = xr[rs][0], x1 = xr[rs][1];
            if (mode == 1) {   // keep the address arithmetic of a gather alive without the load
                x0 += (double)(cc & 1) * 1e-300;
            }
            acc[rs][0] = fma(aa, x0, acc[rs][0]);
            acc[rs][1] = fma(aa, x1, acc[rs][1]);
        }
        cj = cn;
        vj = vn;
    }
    for (int rs = 0; rs < 8; ++rs) {
        const int row = slice * 64 + rs * 8 + g;
        if (row < nrows) *reinterpret_cast<double2*>(y + (size_t)row * 16 + t * 2) = make_double2(acc[rs][0], acc[rs][1]);
    }
}

// a copy with four independent 16-byte loads in flight per thread (the plain grid-stride copy above has one)
__global__ __launch_bounds__(256) void lab_stream4_kernel(size_t n, const double2* __restrict__ a, double2* __restrict__ b) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        const double2 v0 = a[i], v1 = a[i + stride], v2 = a[i + 2 * stride], v3 = a[i + 3 * stride];
        b[i] = v0; b[i + stride] = v1; b[i + 2 * stride] = v2; b[i + 3 * stride] = v3;
    }
    for (; i < n; i += stride) b[i] = a[i];
}
// read-only streaming (sum), to separate read from write bandwidth
__global__ __launch_bounds__(256) void lab_read_kernel(size_t n, const double2* __restrict__ a, double* __restrict__ out) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    double s = 0.0;
    for (; i + 3 * stride < n; i += 4 * stride) {
        const double2 v0 = a[i], v1 = a[i + stride], v2 = a[i + 2 * stride], v3 = a[i + 3 * stride];
        s += (v0.x + v1.x) + (v2.x + v3.x);
    }
    if (s == 1.2345e300) out[0] = s;
}

template <class T>
static std::vector<T> rd(FILE* f, size_t n) {
    std::vector<T> v(n);
    if (n && fread(v.data(), sizeof(T), n, f) != n) { fprintf(stderr, "short read\n"); exit(2); }
    return v;
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: k5_lab matrix.bin [nb] [repeat]\n"); return 2; }
    const int nb = argc > 2 ? atoi(argv[2]) : 16;
    const int repeat = argc > 3 ? atoi(argv[3]) : 20;
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror("open"); return 2; }
    auto hdr = rd<int>(f, 4);
    const int n_u = hdr[0], n_s = hdr[1];
    HostCsr M, B;
    M.nrows = M.ncols = n_u;
    M.rowptr = rd<int>(f, n_u + 1); M.colind = rd<int>(f, hdr[2]); M.vals = rd<double>(f, hdr[2]);
    B.nrows = n_s; B.ncols = n_u;
    B.rowptr = rd<int>(f, n_s + 1); B.colind = rd<int>(f, hdr[3]); B.vals = rd<double>(f, hdr[3]);
    auto w = rd<double>(f, n_s);
    auto al = rd<double>(f, 1);
    fclose(f);
    std::vector<double> maw(n_s);
    for (int i = 0; i < n_s; ++i) maw[i] = -al[0] * w[i];
    HostCsr Bt = csr_transpose(B);
    HostCsr A = csr_block2x2(M, Bt, B, maw.data());
    if (const char* e = getenv("LAB_COLS")) {
        // access-pattern probes: 1 = every entry gathers its own row (sequential, full reuse), 2 = columns shifted by a
        // constant (same locality, no structure change), 3 = random columns (no locality at all)
        const int mode = atoi(e);
        uint64_t r = 1234567;
        for (int i = 0; i < A.nrows; ++i)
            for (int p = A.rowptr[i]; p < A.rowptr[i + 1]; ++p) {
                if (mode == 1) A.colind[p] = i;
                else if (mode == 2) A.colind[p] = (A.colind[p] + 4096) % A.ncols;
                else if (mode == 3) { r ^= r << 13; r ^= r >> 7; r ^= r << 17; A.colind[p] = (int)(r % (uint64_t)A.ncols); }
            }
        printf("LAB_COLS=%d\n", mode);
    }
    const int n = n_u + n_s;
    hipStream_t st;
    PMC_HIP(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    PMC_HIP(hipEventCreate(&e0));
    PMC_HIP(hipEventCreate(&e1));
    const size_t len = (size_t)n * nb;
    std::vector<double> hx(len);
    uint64_t s = 88172645463325252ull;
    for (size_t i = 0; i < len; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; hx[i] = (double)(s >> 11) / 9007199254740992.0 - 0.5; }
    DevBuf<double> x(len), y0(len), y1(len), part((size_t)dot_capacity(n, nb) * nb);
    PMC_HIP(hipMemcpy(x.p, hx.data(), len * 8, hipMemcpyHostToDevice));
    const size_t fl = (size_t)48 << 20;   // double2 elements: 768 MB read + 768 MB written
    DevBuf<double> fa(fl * 2), fb(fl * 2);
    PMC_HIP(hipMemset(fa.p, 0, fl * 16));
    const double bytes = 12.0 * A.nnz() + 4.0 * n + (double)nb * 16.0 * n;
    printf("n_u %d n_s %d nnz %lld nb %d algorithmic MB %.1f\n", n_u, n_s, (long long)A.nnz(), nb, bytes / 1e6);

    auto timeit = [&](const char* name, bool flush, auto&& launch) {
        double tot = 0.0, mn = 1e30;
        launch();
        PMC_HIP(hipStreamSynchronize(st));
        for (int r = 0; r < repeat; ++r) {
            if (flush) lab_stream_kernel<<<4096, 256, 0, st>>>(fl, (const double2*)fa.p, (double2*)fb.p);
            PMC_HIP(hipEventRecord(e0, st));
            launch();
            PMC_HIP(hipEventRecord(e1, st));
            PMC_HIP(hipStreamSynchronize(st));
            float ms = 0.f;
            PMC_HIP(hipEventElapsedTime(&ms, e0, e1));
            tot += ms;
            mn = std::min(mn, (double)ms);
        }
        const double avg = tot / repeat;
        printf("%-34s %-5s avg %7.1f us  min %7.1f us  -> %5.2f TB/s (%.3f of 8)\n", name, flush ? "cold" : "warm", avg * 1e3,
               mn * 1e3, bytes / (avg * 1e-3) / 1e12, bytes / (avg * 1e-3) / 8e12);
        fflush(stdout);
    };
    [[maybe_unused]] auto check = [&](const char* name, const double* ya, const double* yb) {
        std::vector<double> a(len), b(len);
        PMC_HIP(hipMemcpy(a.data(), ya, len * 8, hipMemcpyDeviceToHost));
        PMC_HIP(hipMemcpy(b.data(), yb, len * 8, hipMemcpyDeviceToHost));
        double md = 0.0, mx = 0.0;
        for (size_t i = 0; i < len; ++i) { md = std::max(md, std::fabs(a[i] - b[i])); mx = std::max(mx, std::fabs(a[i])); }
        printf("  check %-28s max |diff| %.3e (max |y| %.3e)\n", name, md, mx);
    };

    try {
        Sell S;
        sell_build(S, A, true, false, st);
        sell_schedule_two_blocks(S, n_u, st, getenv("LAB_OLD_SCHED") ? nullptr : &A);
        SellView V = view(S);
        V.tag = 2;
        {
            // reference: a flat copy that moves the same 206.5 MB (half read, half written), and one moving 288 MB
            const size_t n1 = (size_t)(bytes / 32.0), n2 = (size_t)(bytes * 1.4 / 32.0);   // double2 elements
            DevBuf<double> ca(4 * n1 + 2 * n2 + 16), cb(2 * n2 + 16);
            PMC_HIP(hipMemset(ca.p, 0, ca.n * 8));
            for (int fl_ = 0; fl_ < 2; ++fl_) {
                timeit("flat copy, same bytes", fl_, [&] { lab_stream_kernel<<<4096, 256, 0, st>>>(n1, (const double2*)ca.p, (double2*)cb.p); });
                for (unsigned gsz : {2048u, 8192u, 32768u}) {
                    char nm[64];
                    snprintf(nm, sizeof nm, "copy x4 unrolled, grid %u", gsz);
                    timeit(nm, fl_, [&] { lab_stream4_kernel<<<gsz, 256, 0, st>>>(n1, (const double2*)ca.p, (double2*)cb.p); });
                }
                timeit("read only, same bytes", fl_, [&] { lab_read_kernel<<<8192, 256, 0, st>>>(2 * n1, (const double2*)ca.p, cb.p); });
                timeit("flat copy, 1.4 x the bytes", fl_, [&] { lab_stream_kernel<<<4096, 256, 0, st>>>(n2, (const double2*)ca.p, (double2*)cb.p); });
            }
        }
        if (nb == 16) {
            DevBuf<int> dcols;
            dcols.upload(S.h_cols.empty() ? std::vector<int>(S.nslots, 0) : S.h_cols, st);
            for (int fl_ = 0; fl_ < 2; ++fl_)
                timeit("probe: one x load per row", fl_, [&] {
                    lab_oneload_kernel<<<(S.nslices + 3) / 4, 256, 0, st>>>(S.nrows, S.nslices, S.slice_off.p, S.sched.p, S.cols.p,
                                                                          S.vals.p, x.p, y1.p, 1);
                });
        }
        for (int fl_ = 0; fl_ < 2; ++fl_) {
            timeit("gather (round 1 kernel)", fl_, [&] { k::spmm(st, nb, V, x.p, y0.p, false, nullptr, nullptr); });
            timeit("gather + fused dot", fl_, [&] { k::spmm(st, nb, V, x.p, y0.p, false, part.p, x.p); });
        }
    } catch (const Error& e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
