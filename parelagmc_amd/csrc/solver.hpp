// Krylov solver and preconditioner building blocks (host orchestration of kernels.hip).
#pragma once
#include <functional>

#include "kernels.hpp"

namespace pmc {

// Fixed-degree Chebyshev polynomial in D^-1 A (SPD A, positive diagonal scaling D): a fixed SPD
// linear operator, i.e. a legal MINRES preconditioner block / multigrid smoother.  Replaces the
// reference's sequential hypre l1-Gauss-Seidel (CreateSamplerParameterList.hpp:80-93) - the XML
// schema itself lists Chebyshev / l1-Jacobi as hypre smoother alternatives (example_parameters.xml:789-801).
struct ChebParams {
    int degree = 3;
    double lmax = 1.0;   // upper bound of spec(D^-1 A)
    double ratio = 8.0;  // interval [lmax/ratio, lmax]
};
// Runs `degree` steps.  xa holds the initial guess (ignored when zero_guess); the iterate ping-pongs
// between xa and xb; returns the buffer holding the result.  d is work space.
double* cheb_apply(hipStream_t st, int nb, const SellView& A, const double* dinv, bool dinv_bv, const ChebParams& cp,
                   const double* r, double* xa, double* xb, double* d, bool zero_guess);
// number of buffer flips cheb_apply performs
inline int cheb_flips(int degree, bool zero_guess) { return zero_guess ? degree - 1 : degree; }

// One level of the Schur-complement multigrid hierarchy.
struct MgLevel {
    int n = 0;
    Sell S;                      // pattern (+ shared values for the sampler)
    DevBuf<double> dinv;         // shared: n ; batched: n*kMaxBatch
    DevBuf<double> vals_bv;      // batched values (Darcy): nslots*kMaxBatch
    bool bv = false;
    double lmax = 2.0;
    Sell P, Pt;                  // to/from the next coarser level (absent on the last)
    DevBuf<double> r, xa, xb, d, res;
    void ensure(int nb);
    SellView sview() const { return bv ? view_bv(S, vals_bv.p) : view(S); }
};

struct Multigrid {
    std::vector<MgLevel> L;      // [0] finest
    int smooth_degree = 2;
    double smooth_ratio = 4.0;
    int coarse_degree = 12;
    double coarse_ratio = 100.0;
    // x = V(r) starting at level l0 with zero initial guess; result written to xout (n(l0)*nb)
    void vcycle(hipStream_t st, int nb, int l0, const double* r, double* xout);

  private:
    double* cycle(hipStream_t st, int nb, int l, int l0, const double* r, double* target);
};

// Abstract pieces MINRES needs.
struct LinOp {
    int n = 0;
    // y = A x ; when dot_partial != nullptr also per-block partials of <x, A x>; returns the number of
    // partial blocks written (0 when dot_partial == nullptr)
    std::function<int(hipStream_t, int nb, const double* x, double* y, double* dot_partial)> apply;
};
using PrecFn = std::function<void(hipStream_t, int nb, const double* r, double* z)>;

struct MinresWork {
    DevBuf<double> v0, v1, u0, u1, w0, w1, q, partial;
    DevBuf<k::MinresState> state;
    void ensure(int n, int nb);
};

struct MinresResult {
    pmc_stats col[kMaxBatch];
    int iterations = 0;   // iterations executed (max over columns)
};

// Preconditioned MINRES on nb right-hand sides at once.  x holds the initial guess on entry when
// !zero_guess.  b, x: n*nb interleaved device vectors.
MinresResult minres_solve(Ctx& ctx, int nb, const LinOp& A, const PrecFn& prec, const double* b, double* x,
                          bool zero_guess, const pmc_solver_opts& o, MinresWork& w);

// Gershgorin bound of spec(D^-1 A) for D = diag(A) on the host (setup)
double gershgorin_scaled(const HostCsr& A, const std::vector<double>& diag);

}  // namespace pmc
