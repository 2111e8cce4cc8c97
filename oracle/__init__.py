"""CPU ORACLE - TEST INFRASTRUCTURE ONLY, NEVER PART OF THE PRODUCT PATH.

A numpy/scipy (and, for timing, plain C) restatement of the per-realization hot path of
ParELAGMC (PDESampler / EmbeddedPDESampler / L2ProjectionPDESampler ::Eval,
DarcySolver::SolveFwd, MLMC_Manager statistics).  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it, and
only as the checker.  The product (``parelagmc_amd``) never imports this package and fails
loudly when its HIP library is missing.

Why a restatement: the reference delegates all arithmetic to ParELAG / MFEM / hypre / TRNG,
none of which is vendored in /root/reference or installed in this image, so
the reference cannot be compiled here ("unbuildable": needs external libraries and cmake
find-modules).  Linear systems are solved with a sparse *direct* factorisation
(scipy ``splu``) so the oracle is independent of the GPU path's Krylov/multigrid code.

Pinning status
  * Darcy path: PINNED by the reference's only RNG-free known answer, ctest
    ``DarcyDeterministicTest`` (/root/reference/examples/CMakeLists.txt:62-66): QoI = 2 and
    17152 / 2240 / 304 DoFs on the 16^3 / 8^3 / 4^3 hex levels (tests/test_oracle.py).
  * Matérn scaling g: pinned by the closed formula of src/Utilities.hpp:188-200.
  * Sampler field values: PARITY UNPINNED seed for seed against reference output (STATISTICALLY
    pinned, see (iv)).  The reference's sampler
    goldens (examples/CMakeLists.txt:69-87,105-109) are 5-digit statistics of 10 samples
    drawn from TRNG yarn5 in MFEM element order; neither library is available, so they
    cannot be reproduced.  The sampler restatement is instead checked by (i) the algebraic
    identity with the Legacy reduced system (src/PDESampler_Legacy.cpp:172-176,253-331),
    (ii) the analytic moments the reference drivers test against
    (examples/PDESamplerTest.cpp:205-209: E[s]=0, Var[s]=1; lognormal exp(1/2), e(e-1)),
    (iii) the Embedded == L2Projection invariant the reference's own goldens imply for
    aligned hex-in-hex meshes (examples/CMakeLists.txt:73,109),
    (iv) the reference's RNG-dependent goldens taken as samples of the target distribution:
    sampler + Darcy oracle reproduce DarcyRandomInputTest's 10-sample means of the effective
    permeability (examples/CMakeLists.txt:91-95) within 3 standard errors on the 8^3 / 4^3
    levels (tests/test_oracle.py); the 16^3 level and the MLMC estimate 2.5599 (:76-80) are
    checked the same way on the GPU path (tests/test_gpu_parity.py).
"""
