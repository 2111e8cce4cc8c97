"""Statistical pins of the HIP path against the numbers the reference's own ctest suite holds
(/root/reference/examples/CMakeLists.txt:62-117).  The reference's sampler goldens depend on TRNG yarn5 streams that cannot
be reproduced here, but each of them is ONE DRAW of a statistic whose sampling distribution this implementation defines
completely; the tests below estimate that distribution on the device (R independent repetitions of the reference's
experiment, generator + solver + projection + Darcy all on the GPU through the C ABI) and require the reference's draw to be
a typical one (|golden - mean| <= 3.3 sigma, two-sided 99.9 % for a near-normal statistic; the 10-sample statistics below are
sums over thousands of weakly correlated cells).  Run with -m gpu."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NS = 10            # "Number of samples" of the reference's test parameter lists (CreateSamplerParameterList.hpp:33)


def _typical(golden, draws, what, nsig=3.3):
    m, s = float(np.mean(draws)), float(np.std(draws))
    print(f"[pin] {what}: reference {golden} | here {m:.4f} +- {s:.4f} | z = {(golden - m) / s:+.2f}")
    assert abs(golden - m) <= nsig * s, f"{what}: reference {golden} vs {m:.4f} +- {s:.4f} here"
    return (golden - m) / s


def _mean_field_norms(smp, hier, lvl, reps, first_id):
    """R draws of PDESamplerTest's first output column on `lvl`: L2 norm of the NS-sample mean field (exact expectation
    0, examples/PDESamplerTest.cpp:205-209,262-274; ComputeL2Error prolongates the piecewise constants to the fine mesh,
    src/PDESampler.cpp:614-624, which leaves sum_e |e| c_e^2 unchanged)."""
    vol = hier.spaces[lvl].vol
    s = smp.Eval(lvl, smp.Sample(lvl, first_id=first_id, nbatch=reps * NS))
    mean = s.reshape(reps, NS, -1).mean(axis=1)
    return np.sqrt((mean ** 2) @ vol)


def test_pdesamplertest_goldens_are_typical_draws_of_the_device_sampler(gpu_ctx, hex_hierarchy):
    """PDESamplerTest (PDESampler, 4^3 hexes on [0,2]^3 refined twice, corlen 0.1, Gaussian): goldens 1.2593 / 0.93103 /
    0.63853 = || E_10[s] ||_L2 on the 16^3 / 8^3 / 4^3 levels (examples/CMakeLists.txt:83-87).  E[T^2] = (1/10) int Var[s],
    so one sigma of T (2.7 % on the 16^3 level) is 5.5 % in the field's variance - enough to tell the code's Gamma(nu + d)
    normalisation (variance 3.32 x the textbook one) from Gamma(nu + d/2) by 30 sigma, and a 15 % variance error by 2.7."""
    from parelagmc_amd import capi
    from parelagmc_amd.fe import build_sampler_problem
    smp = capi.PDESampler(gpu_ctx, build_sampler_problem(hex_hierarchy, corlen=0.1))
    z = []
    for lvl, gold in enumerate((1.2593, 9.3103e-01, 6.3853e-01)):
        t = _mean_field_norms(smp, hex_hierarchy, lvl, 300, 10_000 * (lvl + 1))
        z.append(_typical(gold, t, f"PDESamplerTest level {lvl}"))
        assert t.std() / t.mean() < (0.04, 0.06, 0.13)[lvl]
    assert np.sum(np.square(z)) < 16.3          # chi^2_3, 99.9 %: the three levels jointly
    smp.close()


def _embedded_setup():
    from parelagmc_amd.fe import box_mesh, build_hierarchy
    m = box_mesh([6, 6, 6], [3.0, 3.0, 3.0], "hex", origin=[-0.5, -0.5, -0.5])     # Build3DHexEnlargedMesh
    cen = m.verts[m.elems].mean(1)
    m.elem_attr[:] = np.where(np.all((cen > 0) & (cen < 2), axis=1), 1, 2)
    return build_hierarchy(m, 2)


def test_embedded_sampler_goldens_are_typical_draws(gpu_ctx, hex_hierarchy):
    """EmbeddedPDESamplerTest and ProjectionPDESamplerTest share one golden (1.1226 / 0.90325 / 0.51372,
    examples/CMakeLists.txt:69-73,105-109): the field is sampled on the enlarged box [-0.5, 2.5]^3 and returned on the
    original [0, 2]^3 mesh, where the boundary inflation of the plain sampler is gone - the numbers are ~10 % below
    PDESamplerTest's, and the matching (gather) and L2-projected samplers must both reproduce them."""
    from parelagmc_amd import capi
    from parelagmc_amd.fe import build_sampler_problem, l2_projection_hierarchy
    he = _embedded_setup()
    sp_ = build_sampler_problem(he, corlen=0.1, embedded=True)
    ga = capi.PDESampler(gpu_ctx, sp_, projection="gather")
    pr = capi.PDESampler(gpu_ctx, build_sampler_problem(he, corlen=0.1), projection="l2",
                         l2_ops=l2_projection_hierarchy(hex_hierarchy, he))
    for lvl, gold in enumerate((1.1226, 9.0325e-01, 5.1372e-01)):
        for name, smp in (("gather", ga), ("l2", pr)):
            # the embedded mesh orders the original elements differently from hex_hierarchy; the L2 norm does not care,
            # but the volumes must be those of the returned elements
            vol = he.spaces[lvl].vol[sp_.orig_index[lvl]] if name == "gather" else hex_hierarchy.spaces[lvl].vol
            s = smp.Eval(lvl, smp.Sample(lvl, first_id=50_000 * (lvl + 1), nbatch=200 * NS))
            t = np.sqrt((s.reshape(200, NS, -1).mean(axis=1) ** 2) @ vol)
            _typical(gold, t, f"embedded ({name}) level {lvl}")
    ga.close()
    pr.close()


def test_likelihood_and_ratio_goldens_are_typical_draws(gpu_ctx, hex_hierarchy):
    """Row f2 (Bayesian callers).  LikelihoodExample prints L_l = exp(-|G_l(xi_b) - G_obs|^2 / (2 noise)) for ONE prior
    draw xi_b on the three levels, G_obs = G_0(xi_a) + N(0, noise), noise 0.1, one observation point (1,1,1), local
    pressure average over the fine cells around it (eps 0.01), L2-projected sampler on the enlarged box
    (examples/LikelihoodExample.cpp:259-278, src/BayesianInverseProblem.cpp:159-205, CreateBayesianParameterList.hpp:56-60):
    goldens 0.9279 / 0.9578 / 0.9269 (examples/CMakeLists.txt:98-102), i.e. |G_l - G_obs| = 0.122 / 0.093 / 0.123.
    RatioEstimator_MC prints 10-sample moments of R = Q * L and Z = L on level 0 for independent draws: E[R] 1.987,
    Var[R] 0.07749, E[Z] 0.8569, Var[Z] 0.009691, E[R]/E[Z] 2.319 (:112-117, examples/RatioEstimator_MC.cpp:293-345).
    Each is one draw over (xi_a, eta, samples); R repetitions of the whole experiment on the device give the distribution."""
    from parelagmc_amd import capi
    from parelagmc_amd.fe import build_darcy_problem, build_sampler_problem, elements_near_points, l2_projection_hierarchy
    import scipy.sparse as sp
    he = _embedded_setup()
    smp = capi.PDESampler(gpu_ctx, build_sampler_problem(he, corlen=0.1, lognormal=True), projection="l2",
                          l2_ops=l2_projection_hierarchy(hex_hierarchy, he))
    dp = build_darcy_problem(hex_hierarchy, [0, 1, 1, 1, 1, 0], [1, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 1])
    ds = capi.DarcySolver(gpu_ctx, dp)
    s0 = hex_hierarchy.spaces[0]
    mark = elements_near_points(s0.mesh, [[1.0, 1.0, 1.0]], 0.01)[0]
    g = sp.csr_matrix(np.where(mark, s0.vol, 0.0)[None, :])
    for lvl in range(3):
        ds.SetObservations(lvl, g)
        if lvl < 2:
            g = (g @ hex_hierarchy.P[lvl]).tocsr()
    noise, R = 0.1, 160
    rng = np.random.Generator(np.random.PCG64(20261004))

    def G_and_Q(lvl, xi):
        G, _, Q = ds.ComputeG(lvl, smp.Eval(lvl, xi, xi_level=0))
        return G[:, 0], Q
    n0 = smp.xi_size(0)
    xi_a = smp.Sample(0, first_id=900_000, nbatch=R)
    G_obs = G_and_Q(0, xi_a)[0] + np.sqrt(noise) * rng.standard_normal(R)
    # LikelihoodExample: one further draw per repetition, evaluated on every level
    xi_b = smp.Sample(0, first_id=910_000, nbatch=R)
    for lvl, gold in enumerate((0.9279, 0.9578, 0.9269)):
        d = np.abs(G_and_Q(lvl, xi_b)[0] - G_obs)
        like = np.exp(-d ** 2 / (2 * noise))
        dg = np.sqrt(-2 * noise * np.log(gold))
        # a single likelihood value is a coarse pin: it must not be an outlier of the distribution (central 99 %)
        print(f"[pin] LikelihoodExample level {lvl}: reference L {gold} (|G - G_obs| {dg:.4f}) | here median L "
              f"{np.median(like):.4f}, |G - G_obs| 99 % [{np.quantile(d, 0.005):.4f}, {np.quantile(d, 0.995):.4f}]")
        assert np.quantile(d, 0.005) <= dg <= np.quantile(d, 0.995), (lvl, dg, np.quantile(d, [0.005, 0.5, 0.995]))
        assert 0.0 < like.min() and like.max() <= 1.0
    # RatioEstimator_MC on level 0: R repetitions x 10 independent (Z, R) pairs against each repetition's own G_obs
    xz = smp.Sample(0, first_id=920_000, nbatch=R * NS)
    xr = smp.Sample(0, first_id=940_000, nbatch=R * NS)
    Gz, _ = G_and_Q(0, xz)
    Gr, Qr = G_and_Q(0, xr)
    Z = np.exp(-(Gz.reshape(R, NS) - G_obs[:, None]) ** 2 / (2 * noise))
    Rr = Qr.reshape(R, NS) * np.exp(-(Gr.reshape(R, NS) - G_obs[:, None]) ** 2 / (2 * noise))
    assert Z.shape == (R, NS) and n0 == he.spaces[0].n_s
    for gold, stat, what in ((0.8569, Z.mean(axis=1), "E[Z]"), (1.987, Rr.mean(axis=1), "E[R]"),
                             (2.319, Rr.mean(axis=1) / Z.mean(axis=1), "E[R]/E[Z]")):
        lo, hi = np.quantile(stat, [0.005, 0.995])
        print(f"[pin] RatioEstimator_MC {what}: reference {gold} | here median {np.median(stat):.4f}, 99 % [{lo:.4f}, {hi:.4f}]")
        assert lo <= gold <= hi, (what, gold, lo, np.median(stat), hi)
    for gold, stat, what in ((0.009691, Z.var(axis=1, ddof=1), "Var[Z]"), (0.07749, Rr.var(axis=1, ddof=1), "Var[R]")):
        lo, hi = np.quantile(stat, [0.005, 0.995])
        print(f"[pin] RatioEstimator_MC {what}: reference {gold} | here median {np.median(stat):.5f}, 99 % [{lo:.5f}, {hi:.5f}]")
        assert lo <= gold <= hi, (what, gold, lo, np.median(stat), hi)
    ds.close()
    smp.close()
