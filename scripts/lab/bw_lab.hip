// HBM streaming-rate laboratory (not part of the product): what a plain copy / read / write of N bytes achieves on this
// part, over grid size, loads in flight per thread and cache policy.  Calibrates the "practical ceiling" DESIGN.md quotes
// next to the 8 TB/s peak (guide: ~6.3 TB/s for a 16 B/lane copy).   build: hipcc -O3 --offload-arch=gfx950 bw_lab.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)

typedef double __attribute__((ext_vector_type(2))) d2;

template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_kernel(size_t n, const d2* __restrict__ a, d2* __restrict__ b) {
    // block-contiguous: a workgroup owns U consecutive 4 KiB pieces per trip
    const size_t chunk = (size_t)256 * U;
    for (size_t base = (size_t)blockIdx.x * chunk; base < n; base += (size_t)gridDim.x * chunk) {
        d2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = base + (size_t)u * 256 + threadIdx.x;
            if (i < n) v[u] = NT ? __builtin_nontemporal_load(a + i) : a[i];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = base + (size_t)u * 256 + threadIdx.x;
            if (i < n) {
                if (NT) __builtin_nontemporal_store(v[u], b + i);
                else b[i] = v[u];
            }
        }
    }
}
template <int U, bool NT>
__global__ __launch_bounds__(256) void read_kernel(size_t n, const d2* __restrict__ a, double* __restrict__ out) {
    const size_t chunk = (size_t)256 * U;
    double s = 0.0;
    for (size_t base = (size_t)blockIdx.x * chunk; base < n; base += (size_t)gridDim.x * chunk) {
        d2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = base + (size_t)u * 256 + threadIdx.x;
            v[u] = i < n ? (NT ? __builtin_nontemporal_load(a + i) : a[i]) : d2{0.0, 0.0};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) s += v[u].x;
    }
    if (s == 1.2345e300) out[0] = s;
}
template <int U, bool NT>
__global__ __launch_bounds__(256) void write_kernel(size_t n, d2* __restrict__ b) {
    const size_t chunk = (size_t)256 * U;
    const d2 v = {1.0, 2.0};
    for (size_t base = (size_t)blockIdx.x * chunk; base < n; base += (size_t)gridDim.x * chunk) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = base + (size_t)u * 256 + threadIdx.x;
            if (i < n) {
                if (NT) __builtin_nontemporal_store(v, b + i);
                else b[i] = v;
            }
        }
    }
}

// y = c0 a + c1 b + c2 y on interleaved batches of 16 columns (the Lanczos update of the MINRES loop: three streams in, one
// out, in place): U 16-byte pieces per thread, loads / store optionally non-temporal
template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void lincomb_kernel(size_t n, const double* __restrict__ c, const d2* __restrict__ a,
                                                      const d2* __restrict__ b, d2* __restrict__ y) {
    const size_t base = (size_t)blockIdx.x * 256 * U;
    d2 av[U], bv[U], yv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t i = base + (size_t)u * 256 + threadIdx.x;
        if (i < n) {
            av[u] = NTL ? __builtin_nontemporal_load(a + i) : a[i];
            bv[u] = NTL ? __builtin_nontemporal_load(b + i) : b[i];
            yv[u] = NTL ? __builtin_nontemporal_load(y + i) : y[i];
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t i = base + (size_t)u * 256 + threadIdx.x;
        if (i < n) {
            const int k0 = (int)((2 * i) % 16);
            d2 r;
            r.x = c[k0] * av[u].x + c[256 + k0] * bv[u].x + c[512 + k0] * yv[u].x;
            r.y = c[k0 + 1] * av[u].y + c[256 + k0 + 1] * bv[u].y + c[512 + k0 + 1] * yv[u].y;
            if (NTS) __builtin_nontemporal_store(r, y + i);
            else y[i] = r;
        }
    }
}

template <class F>
static double timeit(hipStream_t st, int rep, F f) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    f();
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < rep; ++r) f();
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
    return ms / rep;
}

template <int U, bool NT>
static void run(hipStream_t st, size_t n, const d2* a, d2* b, double* out) {
    const double bytes = (double)n * 16.0;
    for (unsigned grid : {1024u, 2048u, 4096u, 8192u, 16384u, 0u}) {
        const unsigned g = grid ? grid : (unsigned)((n + 256 * U - 1) / (256 * U));
        const double tc = timeit(st, 10, [&] { copy_kernel<U, NT><<<g, 256, 0, st>>>(n, a, b); });
        const double tr = timeit(st, 10, [&] { read_kernel<U, NT><<<g, 256, 0, st>>>(n, a, out); });
        const double tw = timeit(st, 10, [&] { write_kernel<U, NT><<<g, 256, 0, st>>>(n, b); });
        printf("  U=%d nt=%d grid=%7u : copy %6.2f TB/s (r+w)   read %6.2f TB/s   write %6.2f TB/s\n", U, (int)NT, g,
               2.0 * bytes / (tc * 1e-3) / 1e12, bytes / (tr * 1e-3) / 1e12, bytes / (tw * 1e-3) / 1e12);
    }
}

template <int U, bool NTL, bool NTS>
static void run_lincomb(hipStream_t st, size_t n, const double* c, const d2* a, const d2* b, d2* y) {
    const unsigned g = (unsigned)((n + 256 * U - 1) / (256 * U));
    const double t = timeit(st, 10, [&] { lincomb_kernel<U, NTL, NTS><<<g, 256, 0, st>>>(n, c, a, b, y); });
    printf("  lincomb U=%d nt_load=%d nt_store=%d grid=%8u : %7.1f us  %6.2f TB/s (3 r + 1 w)\n", U, (int)NTL, (int)NTS, g,
           t * 1e3, 4.0 * n * 16.0 / (t * 1e-3) / 1e12);
}

static int lincomb_lab(int argc, char** argv) {
    hipStream_t st;
    CK(hipStreamCreate(&st));
    for (int i = 2; i < argc; ++i) {
        const size_t mb = (size_t)atol(argv[i]);
        const size_t n = mb * 1024 * 1024 / 16;
        d2 *a, *b, *y;
        double* c;
        CK(hipMalloc(&a, n * 16));
        CK(hipMalloc(&b, n * 16));
        CK(hipMalloc(&y, n * 16));
        CK(hipMalloc(&c, 768 * 8));
        CK(hipMemsetAsync(a, 0, n * 16, st));
        CK(hipMemsetAsync(b, 0, n * 16, st));
        CK(hipMemsetAsync(y, 0, n * 16, st));
        CK(hipMemsetAsync(c, 0, 768 * 8, st));
        printf("== lincomb, %zu MiB per vector\n", mb);
        run_lincomb<1, false, false>(st, n, c, a, b, y);
        run_lincomb<1, true, false>(st, n, c, a, b, y);
        run_lincomb<1, true, true>(st, n, c, a, b, y);
        run_lincomb<2, false, false>(st, n, c, a, b, y);
        run_lincomb<2, true, false>(st, n, c, a, b, y);
        run_lincomb<2, true, true>(st, n, c, a, b, y);
        run_lincomb<4, false, false>(st, n, c, a, b, y);
        run_lincomb<4, true, false>(st, n, c, a, b, y);
        run_lincomb<4, true, true>(st, n, c, a, b, y);
        CK(hipFree(a));
        CK(hipFree(b));
        CK(hipFree(y));
        CK(hipFree(c));
    }
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1 && std::string(argv[1]) == "lincomb") return lincomb_lab(argc, argv);
    hipStream_t st;
    CK(hipStreamCreate(&st));
    std::vector<size_t> mbs;
    for (int i = 1; i < argc; ++i) mbs.push_back((size_t)atol(argv[i]));
    if (mbs.empty()) mbs = {64, 512, 2048};
    for (size_t mb : mbs) {
        const size_t n = mb * 1024 * 1024 / 16;
        d2 *a, *b;
        double* out;
        CK(hipMalloc(&a, n * 16));
        CK(hipMalloc(&b, n * 16));
        CK(hipMalloc(&out, 64));
        CK(hipMemsetAsync(a, 0, n * 16, st));
        CK(hipMemsetAsync(b, 0, n * 16, st));
        printf("== %zu MiB per array (copy moves twice that)\n", mb);
        run<1, false>(st, n, a, b, out);
        run<2, false>(st, n, a, b, out);
        run<4, false>(st, n, a, b, out);
        run<8, false>(st, n, a, b, out);
        run<4, true>(st, n, a, b, out);
        CK(hipFree(a));
        CK(hipFree(b));
        CK(hipFree(out));
    }
    return 0;
}
