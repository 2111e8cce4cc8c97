/* Boundary test in plain C: everything a C caller needs comes from include/pmc.h alone.  Builds the sampler and the
 * Darcy solver from the CSR arrays of a problem file (written by tests/test_abi_binaries.py together with the oracle's
 * expected fields / QoIs), evaluates them through the C ABI on host buffers and on device buffers, and compares.
 * Usage: abi_smoke problem.bin      exit code 0 and a final line "abi_smoke OK" on success. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pmc.h"
#include "prob_io.h"

#define CHECK(call)                                                                        \
    do {                                                                                   \
        int rc_ = (call);                                                                  \
        if (rc_ != PMC_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, pmc_last_error()); return 1; } \
    } while (0)

static pmc_csr as_csr(const t_csr* a) {
    pmc_csr c;
    c.nrows = a->nrows; c.ncols = a->ncols; c.rowptr = a->rp; c.colind = a->ci; c.vals = a->v;
    return c;
}
static double rel_err(const double* a, const double* b, size_t n) {
    double d = 0.0, s = 0.0;
    for (size_t i = 0; i < n; ++i) { d += (a[i] - b[i]) * (a[i] - b[i]); s += b[i] * b[i]; }
    return sqrt(d / s);
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: abi_smoke problem.bin\n"); return 2; }
    t_problem p = t_load(argv[1]);
    printf("pmc version %d, ABI %d\n", pmc_version(), pmc_abi_version());
    if (pmc_abi_version() != PMC_ABI_VERSION) { fprintf(stderr, "header / library ABI mismatch\n"); return 1; }
    if (pmc_krylov_z_bytes() != 4 && pmc_krylov_z_bytes() != 8) { fprintf(stderr, "bad z storage width\n"); return 1; }
    pmc_ctx* ctx = NULL;
    CHECK(pmc_ctx_create(0, &ctx));
    pmc_solver_opts opts;
    pmc_solver_opts_default(&opts);
    if (opts.abi_version != PMC_ABI_VERSION) { fprintf(stderr, "defaults carry the wrong ABI version\n"); return 1; }
    opts.rel_tol = 1e-12;
    opts.abs_tol = 1e-30;
    opts.max_iter = 400;

    /* ---- sampler */
    pmc_sampler_level* sl = (pmc_sampler_level*)calloc((size_t)p.s_nlevels, sizeof(pmc_sampler_level));
    for (int l = 0; l < p.s_nlevels; ++l) {
        sl[l].n_u = p.sl[l].n_u; sl[l].n_s = p.sl[l].n_s;
        sl[l].M = as_csr(&p.sl[l].M); sl[l].B = as_csr(&p.sl[l].B); sl[l].w_diag = p.sl[l].w;
        if (p.sl[l].has_p) sl[l].P = as_csr(&p.sl[l].P);
    }
    pmc_sampler* smp = NULL;
    CHECK(pmc_sampler_create(ctx, p.s_nlevels, p.s_nlevels, sl, p.alpha, p.g, p.lognormal, &opts, &smp));
    if (pmc_sampler_num_levels(smp) != p.s_nlevels) { fprintf(stderr, "level count\n"); return 1; }
    for (int l = 0; l < p.s_nlevels; ++l) {
        const int ns = pmc_sampler_sample_size(smp, l);
        if (ns != p.sl[l].n_s || pmc_sampler_xi_size(smp, l) != ns) { fprintf(stderr, "sizes on level %d\n", l); return 1; }
        double* s = (double*)malloc(8 * (size_t)p.nbatch * ns);
        pmc_stats* st = (pmc_stats*)calloc((size_t)p.nbatch, sizeof(pmc_stats));
        CHECK(pmc_sampler_eval(smp, l, 0, p.nbatch, p.xi, s, NULL, -1, 0, NULL, PMC_MEM_HOST, st));
        const double e = rel_err(s, p.s_expect[l], (size_t)p.nbatch * ns);
        printf("sampler level %d: rel. error vs oracle %.2e, %d iterations, converged %d\n", l, e, st[0].iterations, st[0].converged);
        if (!(e < 1e-9) || st[0].converged != 1) return 1;
        /* ABI 2: device time of the solve ("Sampler: Mult"), the realizations one launch carries on this level */
        if (!(st[0].solve_ms > 0.0) || !(st[0].setup_ms > 0.0) || pmc_sampler_batch_width(smp, l) < 16) {
            fprintf(stderr, "phase times / batch width on level %d\n", l);
            return 1;
        }
        /* the same through device buffers */
        void *dxi = NULL, *ds = NULL;
        CHECK(pmc_malloc(ctx, 8 * (size_t)p.nbatch * p.sl[0].n_s, &dxi));
        CHECK(pmc_malloc(ctx, 8 * (size_t)p.nbatch * ns, &ds));
        CHECK(pmc_memcpy_h2d(ctx, dxi, p.xi, 8 * (size_t)p.nbatch * p.sl[0].n_s));
        CHECK(pmc_sampler_eval(smp, l, 0, p.nbatch, (const double*)dxi, (double*)ds, NULL, -1, 0, NULL, PMC_MEM_DEVICE, NULL));
        double* s2 = (double*)malloc(8 * (size_t)p.nbatch * ns);
        CHECK(pmc_memcpy_d2h(ctx, s2, ds, 8 * (size_t)p.nbatch * ns));
        if (memcmp(s, s2, 8 * (size_t)p.nbatch * ns) != 0) { fprintf(stderr, "host and device paths differ\n"); return 1; }
        CHECK(pmc_free(ctx, dxi));
        CHECK(pmc_free(ctx, ds));
        free(s); free(s2); free(st);
    }
    {   /* a struct of another ABI version is refused, not misread */
        pmc_solver_opts old = opts;
        pmc_sampler* none = NULL;
        old.abi_version = 1;
        if (pmc_sampler_create(ctx, p.s_nlevels, p.s_nlevels, sl, p.alpha, p.g, p.lognormal, &old, &none) == PMC_OK) {
            fprintf(stderr, "ABI version 1 options accepted\n");
            return 1;
        }
    }
    {   /* ABI 3: (i) a caller built against another header version is refused at pmc_ctx_create already (the macro passes
           PMC_ABI_VERSION); (ii) precond_storage = PMC_STORAGE_FP64 - everything stored fp64, as the reference is - gives the
           same field; (iii) pmc_sampler_mult = invA[level]->Mult on the right-hand side Eval builds gives Eval's field */
        pmc_ctx* none_ctx = NULL;
        if (pmc_ctx_create_abi(0, PMC_ABI_VERSION - 1, &none_ctx) == PMC_OK || none_ctx != NULL) {
            fprintf(stderr, "caller of ABI version %d accepted\n", PMC_ABI_VERSION - 1);
            return 1;
        }
        pmc_solver_opts o64 = opts;
        o64.precond_storage = PMC_STORAGE_FP64;
        pmc_sampler* s64 = NULL;
        CHECK(pmc_sampler_create(ctx, p.s_nlevels, p.s_nlevels, sl, p.alpha, p.g, p.lognormal, &o64, &s64));
        if (pmc_sampler_krylov_z_bytes(s64) != 8 || pmc_sampler_krylov_z_bytes(smp) != 4) { fprintf(stderr, "storage widths\n"); return 1; }
        const int ns = p.sl[0].n_s, nu = p.sl[0].n_u;
        double* s = (double*)malloc(8 * (size_t)p.nbatch * ns);
        CHECK(pmc_sampler_eval(s64, 0, 0, p.nbatch, p.xi, s, NULL, -1, 0, NULL, PMC_MEM_HOST, NULL));
        const double e64 = rel_err(s, p.s_expect[0], (size_t)p.nbatch * ns);
        printf("sampler level 0, fp64 storage: rel. error vs oracle %.2e\n", e64);
        if (!(e64 < 1e-9)) return 1;
        {
            double* rhs = (double*)calloc((size_t)(nu + ns), 8);
            double* sol = (double*)malloc(8 * (size_t)(nu + ns));
            for (int i = 0; i < ns; ++i) rhs[nu + i] = -p.g * sqrt(p.sl[0].w[i]) * p.xi[i];   /* src/PDESampler.cpp:423-428 */
            pmc_stats st1;
            CHECK(pmc_sampler_mult(smp, 0, 1, rhs, sol, 0, PMC_MEM_HOST, &st1));
            if (p.lognormal)
                for (int i = 0; i < ns; ++i) sol[nu + i] = exp(sol[nu + i]);                  /* :529-533 */
            const double em = rel_err(sol + nu, p.s_expect[0], (size_t)ns);
            printf("pmc_sampler_mult: s-block rel. error vs oracle %.2e, %d iterations\n", em, st1.iterations);
            if (!(em < 1e-9) || st1.converged != 1) return 1;
            free(rhs); free(sol);
        }
        free(s);
        pmc_sampler_destroy(s64);
    }
    {   /* The hybridized solver (the reference's "Hybridization" option, src/PDESampler.cpp:302-318) reached from C alone:
           the element-local elimination runs inside the library (pmc_hybrid_build) on what BuildHierarchy holds - the
           element decomposition of the u-mass matrix and B without boundary elimination (the Darcy levels of the problem
           file carry both for the same mesh), diag(W), alpha - and the field must equal the oracle's like the default one */
        if (p.d_nlevels < p.s_nlevels) { fprintf(stderr, "problem file: fewer Darcy than sampler levels\n"); return 1; }
        pmc_hybrid_elements* he = (pmc_hybrid_elements*)calloc((size_t)p.s_nlevels, sizeof(pmc_hybrid_elements));
        for (int l = 0; l < p.s_nlevels; ++l) {
            const t_dlevel* D = &p.dl[l];
            if (D->n_u != p.sl[l].n_u || D->n_p != p.sl[l].n_s) { fprintf(stderr, "level %d: sampler / Darcy sizes differ\n", l); return 1; }
            he[l].n_u = D->n_u; he[l].n_s = D->n_p;
            he[l].M_pattern = as_csr(&D->M);
            he[l].c_ptr = D->c_ptr; he[l].c_elem = D->c_elem; he[l].c_val = D->c_val;
            he[l].B = as_csr(&D->B);
            he[l].w_diag = p.sl[l].w;
            if (p.sl[l].has_p) he[l].P = as_csr(&p.sl[l].P);
        }
        pmc_hybrid_system* hs = NULL;
        pmc_hybrid_level hv;
        CHECK(pmc_hybrid_build(&he[0], p.alpha, &hs));
        CHECK(pmc_hybrid_system_level(hs, &hv));
        if (hv.n_lambda != p.sl[0].n_u || hv.n_s != p.sl[0].n_s || hv.H.nrows != hv.n_lambda || hv.G.ncols != hv.n_s) {
            fprintf(stderr, "hybrid system shapes\n");
            return 1;
        }
        for (int i = 0; i < hv.n_s; ++i)
            if (!(hv.z_diag[i] < 0.0)) { fprintf(stderr, "z_diag[%d] = %g is not negative\n", i, hv.z_diag[i]); return 1; }
        for (int i = 0; i < hv.n_lambda; ++i) {          /* SPD: positive diagonal */
            double d = 0.0;
            for (int q = hv.H.rowptr[i]; q < hv.H.rowptr[i + 1]; ++q)
                if (hv.H.colind[q] == i) d = hv.H.vals[q];
            if (!(d > 0.0)) { fprintf(stderr, "H[%d, %d] = %g\n", i, i, d); return 1; }
        }
        /* the two-step way: the view handed to pmc_sampler_create_hybrid (single level) */
        pmc_sampler* h1 = NULL;
        hv.P.rowptr = NULL; hv.P.colind = NULL; hv.P.vals = NULL; hv.P.nrows = hv.P.ncols = 0;
        CHECK(pmc_sampler_create_hybrid(ctx, 1, &hv, p.alpha, p.g, p.lognormal, &opts, &h1));
        pmc_hybrid_system_destroy(hs);                   /* the sampler keeps its own copy */
        /* the one-call way, every level */
        pmc_sampler* hyb = NULL;
        CHECK(pmc_sampler_create_hybrid_from_elements(ctx, p.s_nlevels, he, p.alpha, p.g, p.lognormal, &opts, &hyb));
        if (pmc_sampler_is_hybrid(hyb) != 1 || pmc_sampler_is_hybrid(smp) != 0) { fprintf(stderr, "pmc_sampler_is_hybrid\n"); return 1; }
        for (int l = 0; l < p.s_nlevels; ++l) {
            const int ns = pmc_sampler_sample_size(hyb, l);
            double* s = (double*)malloc(8 * (size_t)p.nbatch * ns);
            pmc_stats* st = (pmc_stats*)calloc((size_t)p.nbatch, sizeof(pmc_stats));
            CHECK(pmc_sampler_eval(hyb, l, 0, p.nbatch, p.xi, s, NULL, -1, 0, NULL, PMC_MEM_HOST, st));
            const double e = rel_err(s, p.s_expect[l], (size_t)p.nbatch * ns);
            printf("hybridized sampler (built in C) level %d: rel. error vs oracle %.2e, %d iterations\n", l, e, st[0].iterations);
            if (!(e < 1e-9) || st[0].converged != 1) return 1;
            if (l == 0) {
                double* s1 = (double*)malloc(8 * (size_t)p.nbatch * ns);
                CHECK(pmc_sampler_eval(h1, 0, 0, p.nbatch, p.xi, s1, NULL, -1, 0, NULL, PMC_MEM_HOST, NULL));
                if (!(rel_err(s1, s, (size_t)p.nbatch * ns) < 1e-12)) { fprintf(stderr, "two-step and one-call construction differ\n"); return 1; }
                free(s1);
            }
            free(s); free(st);
        }
        {   /* a B with eliminated (zeroed) columns is refused with a message, not misread */
            double* bz = (double*)malloc(8 * (size_t)p.dl[0].B.nnz);
            memcpy(bz, p.dl[0].B.v, 8 * (size_t)p.dl[0].B.nnz);
            bz[0] = 0.0;
            pmc_hybrid_elements badl = he[0];
            badl.B.vals = bz;
            pmc_hybrid_system* none = NULL;
            if (pmc_hybrid_build(&badl, p.alpha, &none) != PMC_ERR_INVALID || none != NULL) { fprintf(stderr, "eliminated B accepted\n"); return 1; }
            free(bz);
        }
        pmc_sampler_destroy(h1);
        pmc_sampler_destroy(hyb);
        free(he);
    }
    /* error paths return codes, never abort */
    if (pmc_sampler_eval(smp, p.s_nlevels, 0, 1, p.xi, p.xi, NULL, -1, 0, NULL, PMC_MEM_HOST, NULL) == PMC_OK) {
        fprintf(stderr, "out-of-range level accepted\n");
        return 1;
    }
    pmc_csr P0;
    if (p.s_nlevels > 1) {
        CHECK(pmc_sampler_true_p(smp, 0, &P0));
        if (P0.nrows != p.sl[0].n_s || P0.ncols != p.sl[1].n_s) { fprintf(stderr, "GetTrueP shape\n"); return 1; }
    }

    /* ---- Darcy */
    pmc_darcy_level* dl = (pmc_darcy_level*)calloc((size_t)p.d_nlevels, sizeof(pmc_darcy_level));
    for (int l = 0; l < p.d_nlevels; ++l) {
        const t_dlevel* L = &p.dl[l];
        dl[l].n_u = L->n_u; dl[l].n_p = L->n_p;
        dl[l].M_pattern = as_csr(&L->M);
        dl[l].c_ptr = L->c_ptr; dl[l].c_elem = L->c_elem; dl[l].c_val = L->c_val;
        dl[l].B = as_csr(&L->B);
        dl[l].rhs = L->rhs; dl[l].ess_mask = L->ess; dl[l].ess_data = L->ess_data; dl[l].obs = L->obs;
        if (L->has_p) dl[l].P = as_csr(&L->P);
    }
    pmc_darcy* dar = NULL;
    CHECK(pmc_darcy_create(ctx, p.d_nlevels, p.d_nlevels, dl, p.k_divides, &opts, &dar));
    for (int l = 0; l < p.d_nlevels; ++l) {
        double* Q = (double*)malloc(8 * (size_t)p.nbatch);
        double* C = (double*)malloc(8 * (size_t)p.nbatch);
        pmc_stats* st = (pmc_stats*)calloc((size_t)p.nbatch, sizeof(pmc_stats));
        const uint64_t launches0 = pmc_kernel_launches();
        CHECK(pmc_darcy_set_operator_timing(dar, 1));
        CHECK(pmc_darcy_solve_fwd(dar, l, p.nbatch, p.k[l], Q, C, NULL, PMC_MEM_HOST, st));
        CHECK(pmc_darcy_set_operator_timing(dar, 0));
        const double e = rel_err(Q, p.q_expect[l], (size_t)p.nbatch);
        printf("darcy level %d: Q[0] = %.12g, rel. error vs oracle %.2e, C = %g\n", l, Q[0], e, C[0]);
        if (!(e < 1e-8) || C[0] != (double)(p.dl[l].n_u + p.dl[l].n_p)) return 1;
        /* ABI 2: "Darcy: Build Solver" / "Darcy: Mult" device times, in-loop operator timing, launch counter */
        double op_ms = 0.0, gap_ms = 0.0, op_bytes = 0.0;
        int64_t op_launches = 0;
        CHECK(pmc_darcy_operator_time(dar, &op_ms, &op_launches, &gap_ms));
        CHECK(pmc_darcy_operator_bytes(dar, l, 16, &op_bytes));
        if (!(st[0].solve_ms > 0.0) || !(st[0].setup_ms > 0.0) || st[0].converged != 1 || op_launches < st[0].iterations - 2 ||
            !(op_ms > 0.0) || !(op_bytes > 0.0) || pmc_kernel_launches() <= launches0 || pmc_darcy_batch_width(dar, l) < 16) {
            fprintf(stderr, "timers / operator timing / launch counter on Darcy level %d\n", l);
            return 1;
        }
        free(Q); free(C); free(st);
    }
    pmc_darcy_destroy(dar);
    /* the same level structs through the hybridized solver (the reference's "Hybridization" option of DarcySolver,
     * src/DarcySolver.cpp:586,619): Q and the pressure block against the oracle / the default solver */
    pmc_darcy* hyb = NULL;
    CHECK(pmc_darcy_create_hybrid(ctx, p.d_nlevels, p.d_nlevels, dl, p.k_divides, &opts, &hyb));
    for (int l = 0; l < p.d_nlevels; ++l) {
        double* Q = (double*)malloc(8 * (size_t)p.nbatch);
        double* C = (double*)malloc(8 * (size_t)p.nbatch);
        double* Q2 = (double*)malloc(8 * (size_t)p.nbatch);
        double* pr = (double*)malloc(8 * (size_t)p.nbatch * (size_t)p.dl[l].n_p);
        pmc_stats* st = (pmc_stats*)calloc((size_t)p.nbatch, sizeof(pmc_stats));
        CHECK(pmc_darcy_solve_fwd(hyb, l, p.nbatch, p.k[l], Q, C, NULL, PMC_MEM_HOST, st));
        CHECK(pmc_darcy_solve_fwd_pressure(hyb, l, p.nbatch, p.k[l], pr, C, Q2, 1, PMC_MEM_HOST, NULL));
        const double e = rel_err(Q, p.q_expect[l], (size_t)p.nbatch);
        printf("hybridized darcy level %d: Q[0] = %.12g, rel. error vs oracle %.2e, iterations %d\n", l, Q[0], e, st[0].iterations);
        if (!(e < 1e-8) || !(rel_err(Q2, Q, (size_t)p.nbatch) < 1e-9) || st[0].converged != 1 ||
            C[0] != (double)(p.dl[l].n_u + p.dl[l].n_p))
            return 1;
        free(Q); free(C); free(Q2); free(pr); free(st);
    }
    pmc_darcy_destroy(hyb);
    pmc_sampler_destroy(smp);
    pmc_ctx_destroy(ctx);
    printf("abi_smoke OK\n");
    return 0;
}
