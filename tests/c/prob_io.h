/* Reader of the little binary problem files tests/test_abi_binaries.py writes for the C / C++ boundary tests (test
 * infrastructure).  Plain C so that abi_smoke.c and adapter_smoke.cpp share it.  Layout: see write_problem_file() in
 * the Python test. */
#ifndef PMC_TEST_PROB_IO_H
#define PMC_TEST_PROB_IO_H
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef struct { int32_t nrows, ncols, nnz; int32_t *rp, *ci; double* v; } t_csr;

static void* t_read(FILE* f, size_t bytes) {
    void* p = malloc(bytes ? bytes : 1);
    if (!p || (bytes && fread(p, 1, bytes, f) != bytes)) { fprintf(stderr, "problem file: short read\n"); exit(2); }
    return p;
}
static int32_t t_i32(FILE* f) { int32_t v; if (fread(&v, 4, 1, f) != 1) { fprintf(stderr, "problem file: short read\n"); exit(2); } return v; }
static double t_f64(FILE* f) { double v; if (fread(&v, 8, 1, f) != 1) { fprintf(stderr, "problem file: short read\n"); exit(2); } return v; }
static t_csr t_read_csr(FILE* f) {
    t_csr a;
    a.nrows = t_i32(f); a.ncols = t_i32(f); a.nnz = t_i32(f);
    a.rp = (int32_t*)t_read(f, sizeof(int32_t) * (size_t)(a.nrows + 1));
    a.ci = (int32_t*)t_read(f, sizeof(int32_t) * (size_t)a.nnz);
    a.v = (double*)t_read(f, sizeof(double) * (size_t)a.nnz);
    return a;
}

typedef struct { int32_t n_u, n_s, has_p; t_csr M, B, P; double* w; } t_slevel;
typedef struct { int32_t n_u, n_p, has_p, ncontrib; t_csr M, B, P; int32_t *c_ptr, *c_elem; double *c_val, *rhs, *ess_data, *obs; uint8_t* ess; } t_dlevel;
typedef struct {
    int32_t s_nlevels, lognormal, nbatch;
    double alpha, g;
    t_slevel* sl;
    double* xi;        /* nbatch x n_s(0) */
    double** s_expect; /* per level: nbatch x n_s(level) */
    int32_t d_nlevels, k_divides;
    t_dlevel* dl;
    double** k;        /* per level: nbatch x n_p(level) */
    double** q_expect; /* per level: nbatch */
} t_problem;

static t_problem t_load(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    t_problem p;
    if (t_i32(f) != 0x504d4332) { fprintf(stderr, "problem file: bad magic\n"); exit(2); }
    p.s_nlevels = t_i32(f); p.alpha = t_f64(f); p.g = t_f64(f); p.lognormal = t_i32(f);
    p.sl = (t_slevel*)calloc((size_t)p.s_nlevels, sizeof(t_slevel));
    for (int l = 0; l < p.s_nlevels; ++l) {
        t_slevel* L = &p.sl[l];
        L->n_u = t_i32(f); L->n_s = t_i32(f);
        L->M = t_read_csr(f); L->B = t_read_csr(f);
        L->w = (double*)t_read(f, 8 * (size_t)L->n_s);
        L->has_p = t_i32(f);
        if (L->has_p) L->P = t_read_csr(f);
    }
    p.nbatch = t_i32(f);
    p.xi = (double*)t_read(f, 8 * (size_t)p.nbatch * p.sl[0].n_s);
    p.s_expect = (double**)calloc((size_t)p.s_nlevels, sizeof(double*));
    for (int l = 0; l < p.s_nlevels; ++l) p.s_expect[l] = (double*)t_read(f, 8 * (size_t)p.nbatch * p.sl[l].n_s);
    p.d_nlevels = t_i32(f); p.k_divides = t_i32(f);
    p.dl = (t_dlevel*)calloc((size_t)p.d_nlevels, sizeof(t_dlevel));
    for (int l = 0; l < p.d_nlevels; ++l) {
        t_dlevel* L = &p.dl[l];
        L->n_u = t_i32(f); L->n_p = t_i32(f);
        L->M = t_read_csr(f);
        L->c_ptr = (int32_t*)t_read(f, 4 * (size_t)(L->M.nnz + 1));
        L->ncontrib = t_i32(f);
        L->c_elem = (int32_t*)t_read(f, 4 * (size_t)L->ncontrib);
        L->c_val = (double*)t_read(f, 8 * (size_t)L->ncontrib);
        L->B = t_read_csr(f);
        L->rhs = (double*)t_read(f, 8 * (size_t)(L->n_u + L->n_p));
        L->ess = (uint8_t*)t_read(f, (size_t)L->n_u);
        L->ess_data = (double*)t_read(f, 8 * (size_t)L->n_u);
        L->obs = (double*)t_read(f, 8 * (size_t)(L->n_u + L->n_p));
        L->has_p = t_i32(f);
        if (L->has_p) L->P = t_read_csr(f);
    }
    p.k = (double**)calloc((size_t)p.d_nlevels, sizeof(double*));
    p.q_expect = (double**)calloc((size_t)p.d_nlevels, sizeof(double*));
    for (int l = 0; l < p.d_nlevels; ++l) {
        p.k[l] = (double*)t_read(f, 8 * (size_t)p.nbatch * p.dl[l].n_p);
        p.q_expect[l] = (double*)t_read(f, 8 * (size_t)p.nbatch);
    }
    fclose(f);
    return p;
}
#endif
