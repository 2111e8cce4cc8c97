"""Host logic of the MLMC / MC managers (libpmc_host.so) against the plain-Python restatement
of MLMC_Manager (oracle/mlmc_oracle.py).  Plugins are Python callbacks: no GPU involved."""
import os

import numpy as np
import pytest

from oracle import mlmc_oracle as mo
from oracle.rng_oracle import normal_fill
from parelagmc_amd import host_api

SIZES = [32, 16, 8]          # level 0 finest
NDOFS = [1000, 140, 20]


class SyntheticPlugin:
    """Cheap analytic sampler/solver with the same call protocol as the real ones."""

    def __init__(self, nlevels=3, seed=11):
        self.nl = nlevels
        self.seed = seed
        self.calls = []

    def sample(self, level, first_id, nbatch):
        return np.stack([normal_fill(SIZES[level], self.seed, first_id + b, level) for b in range(nbatch)])

    def _restrict(self, x, frm, to):
        while frm < to:
            x = x.reshape(x.shape[0], -1, 2).sum(axis=2) / np.sqrt(2.0)
            frm += 1
        return x

    def eval(self, level, xi_level, xi, init, init_level):
        assert xi_level <= level
        g = 0.5 * self._restrict(xi, xi_level, level) + 0.01 * level
        if init is not None:
            assert init_level >= level and init.shape[1] == SIZES[init_level]
        self.calls.append(("eval", level, xi_level, init is not None))
        return np.exp(g), g

    def solve(self, level, k):
        assert k.shape[1] == SIZES[level]
        q = np.log(k).mean(axis=1) * (1.0 + 0.3 * 2.0 ** (-2 * (self.nl - level))) + 1.0 / (1 + level)
        return q, np.full(k.shape[0], float(NDOFS[level]))

    def callbacks(self):
        return dict(sample=self.sample, eval=self.eval, solve=self.solve, xi_size=SIZES[:self.nl],
                    sample_size=SIZES[:self.nl], ndofs=NDOFS[:self.nl])


def python_init_run(pl, nl, counts, nsamples, sums):
    """MLMC_Manager::InitRun restated with the oracle's accumulate (src/MLMC_Manager.cpp:103-179)."""
    for lvl in range(nl - 1, -1, -1):
        for i in range(nsamples[lvl]):
            sid = counts[lvl] + i
            xi = pl.sample(lvl, sid, 1)
            if lvl == nl - 1:
                s, _ = pl.eval(lvl, lvl, xi, None, None)
                q, c = pl.solve(lvl, s)
                mo.accumulate(sums, lvl, q[0], q[0], c[0])
            else:
                sc, emb = pl.eval(lvl + 1, lvl, xi, None, None)
                qc, cc = pl.solve(lvl + 1, sc)
                s, _ = pl.eval(lvl, lvl, xi, emb, lvl + 1)
                q, c = pl.solve(lvl, s)
                mo.accumulate(sums, lvl, q[0] - qc[0], q[0], c[0] + cc[0])
        counts[lvl] += nsamples[lvl]


def test_exp_w_regression_matches_restatement():
    rng = np.random.default_rng(3)
    for n in (2, 3, 4, 6):
        y = rng.uniform(0.1, 2.0, n) * rng.choice([-1, 1], n)
        x = np.sort(rng.uniform(10, 1e6, n))[::-1].copy()
        for skip in (0, 1):
            assert host_api.exp_w_regression(y, x, skip) == pytest.approx(mo.exp_w_regression(y, x, skip), rel=1e-14, abs=0)
    assert host_api.exp_w_regression([1.0], [2.0], 0) == 0.0


@pytest.mark.parametrize("nl", [1, 2, 3])
@pytest.mark.parametrize("batch", [1, 4, 16])
def test_init_run_sums_and_statistics(nl, batch):
    pl = SyntheticPlugin(nl)
    mgr = host_api.MLMCManager(nl, callbacks=pl.callbacks(), wall_time=False, batch=batch, eps2=1e-3)
    ns = [7, 12, 21][:nl]
    r = mgr.InitRun(ns)
    sums = np.zeros((nl, mo.NVAR))
    counts = [0] * nl
    ref_pl = SyntheticPlugin(nl)
    python_init_run(ref_pl, nl, counts, ns, sums)
    assert np.allclose(r["sums"], sums, rtol=1e-13, atol=1e-13)
    assert list(r["nsamples"]) == ns
    ref = mo.compute_nsamples_mse(sums, ns, NDOFS[:nl], 1e-3, 0.5)
    for key in ("eY", "eABSY", "eQ", "eABSQ", "eC", "varY", "varQ", "kurtosis", "consistency", "VC"):
        assert np.allclose(r[key], ref[key], rtol=1e-11, atol=1e-14), key
    for a, b in (("alpha", "alpha"), ("alpha_abs", "alphaABS"), ("beta", "beta"), ("gamma", "gamma"),
                 ("bias2", "bias2"), ("estimator_variance", "estimator_variance"), ("actual_mse", "actualMSE"),
                 ("estimate", "estimate")):
        assert r[a] == pytest.approx(ref[b], rel=1e-10, abs=1e-14), a
    assert list(r["missing"]) == list(ref["missing"])
    # second round continues the sample-id sequence and accumulates
    r2 = mgr.InitRun([3] * nl)
    python_init_run(ref_pl, nl, counts, [3] * nl, sums)
    assert np.allclose(r2["sums"], sums, rtol=1e-13, atol=1e-13)
    mgr.close()


def test_call_protocol_matches_reference_loop():
    """Coarsest level: Sample, Eval(l), SolveFwd(l).  Pairs: Eval(l+1, init=False), Eval(l, use_init=True)."""
    pl = SyntheticPlugin(3)
    mgr = host_api.MLMCManager(3, callbacks=pl.callbacks(), wall_time=False, batch=16)
    mgr.InitRun([2, 2, 2])
    assert pl.calls == [("eval", 2, 2, False), ("eval", 2, 1, False), ("eval", 1, 1, True),
                        ("eval", 1, 0, False), ("eval", 0, 0, True)]
    mgr.close()


def test_run_reaches_target_variance_and_auto_eps():
    pl = SyntheticPlugin(3)
    mgr = host_api.MLMCManager(3, callbacks=pl.callbacks(), wall_time=False, batch=8, eps2=2e-4, init_nsamples=10)
    r = mgr.Run()
    assert r["estimator_variance"] <= 0.5 * 2e-4
    assert (r["nsamples"] >= 10).all()
    mgr.close()
    mgr = host_api.MLMCManager(3, callbacks=pl.callbacks(), wall_time=False, batch=8, eps2=-1.0, init_nsamples=10)
    r = mgr.Run()
    assert r["eps2"] == pytest.approx(r["bias2"] / 0.5)       # auto_eps2 (src/MLMC_Manager.cpp:357-358)
    assert r["estimator_variance"] <= 0.5 * r["eps2"]
    mgr.close()


def test_array_number_of_samples_and_wall_time_cost():
    pl = SyntheticPlugin(3)
    mgr = host_api.MLMCManager(3, callbacks=pl.callbacks(), wall_time=True, batch=4, array_nsamples=[4, 8, 16],
                               eps2=1e9)
    r = mgr.Run()                      # huge eps2 -> only the initial round
    assert list(r["nsamples"]) == [4, 8, 16]
    assert (r["cost"] > 0).all() and np.allclose(r["cost"], r["level_seconds"] / r["nsamples"])
    mgr.close()


def test_bad_arguments_are_reported():
    pl = SyntheticPlugin(2)
    with pytest.raises(Exception):
        host_api.MLMCManager(2, callbacks=pl.callbacks(), batch=0)
    mgr = host_api.MLMCManager(2, callbacks=pl.callbacks(), wall_time=False)
    with pytest.raises(Exception):
        mgr.InitRun([1, -1])
    mgr.close()


def test_oracle_backed_plugin_small_problem(hex_hierarchy_small):
    """Realistic plugin: the CPU oracle's sampler + Darcy on 8^3/4^3 behind the same managers."""
    from oracle.darcy_oracle import DarcyOracle
    from oracle.sampler_oracle import SamplerOracle
    from parelagmc_amd.fe import build_darcy_problem, build_sampler_problem
    sp = build_sampler_problem(hex_hierarchy_small, corlen=0.1, lognormal=True)
    dp = build_darcy_problem(hex_hierarchy_small, [0, 1, 1, 1, 1, 0], [1, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 1])
    so, do = SamplerOracle(sp), DarcyOracle(dp)
    sizes = [L.n_s for L in sp.levels]

    def sample(level, first_id, nb):
        return np.stack([normal_fill(sizes[level], 5, first_id + b, level) for b in range(nb)])

    def ev(level, xi_level, xi, init, init_level):
        out = [so.eval(level, xi_level, x) for x in xi]
        return np.stack([o[0] for o in out]), np.stack([o[1] for o in out])

    def solve(level, k):
        out = [do.solve_fwd(level, kk) for kk in k]
        return np.array([o[0] for o in out]), np.array([o[1] for o in out])

    mgr = host_api.MLMCManager(2, callbacks=dict(sample=sample, eval=ev, solve=solve, xi_size=sizes, sample_size=sizes,
                                                 ndofs=[L.ndofs for L in dp.levels]), wall_time=False, batch=4)
    r = mgr.InitRun([6, 10])
    assert 1.0 < r["eQ"][0] < 4.0 and 1.0 < r["eQ"][1] < 4.0       # effective permeability around exp(.)*2
    assert abs(r["eY"][0]) < abs(r["eQ"][0])                       # level correction smaller than the QoI
    assert np.allclose(r["eC"], [dp.levels[0].ndofs + dp.levels[1].ndofs, dp.levels[1].ndofs])
    mgr.close()


def test_log_replay_resumes_a_run(tmp_path):
    """The per-sample log (the reference's MLMC.dat columns) is a checkpoint: replaying it into a fresh manager
    restores sums, counters and the derived statistics, and later rounds continue the sample-id sequence."""
    log = str(tmp_path / "MLMC.dat")
    pl = SyntheticPlugin(3)
    a = host_api.MLMCManager(3, callbacks=pl.callbacks(), wall_time=False, batch=4, log_file=log)
    ra = a.InitRun([6, 9, 14])
    a.close()
    lines = [ln for ln in open(log) if not ln.startswith("%")]
    assert len(lines) == 29 and len(lines[0].split()) == 5
    b = host_api.MLMCManager(3, callbacks=SyntheticPlugin(3).callbacks(), wall_time=False, batch=4)
    assert b.ReplayLog(log) == 29
    rb = b.result()
    assert np.allclose(rb["sums"], ra["sums"], rtol=1e-14, atol=1e-15) and list(rb["nsamples"]) == [6, 9, 14]
    assert np.allclose(rb["varY"], ra["varY"], rtol=1e-12) and list(rb["missing"]) == list(ra["missing"])
    # continuing from the checkpoint == continuing the original run
    a2 = host_api.MLMCManager(3, callbacks=SyntheticPlugin(3).callbacks(), wall_time=False, batch=4)
    a2.InitRun([6, 9, 14])
    assert np.allclose(b.InitRun([2, 3, 4])["sums"], a2.InitRun([2, 3, 4])["sums"], rtol=1e-13)
    b.close()
    a2.close()
    # resuming INTO the same log continues it: a second resume still sees every realization
    c = host_api.MLMCManager(3, callbacks=SyntheticPlugin(3).callbacks(), wall_time=False, batch=4, log_file=log)
    assert c.ReplayLog(log) == 29
    c.InitRun([2, 3, 4])
    c.close()
    d = host_api.MLMCManager(3, callbacks=SyntheticPlugin(3).callbacks(), wall_time=False, batch=4)
    assert d.ReplayLog(log) == 29 + 9
    d.close()


def test_farm_logs_are_sharded_by_rank_and_replayed_together(tmp_path):
    """Every farm rank logs its own shard (rank r > 0 -> "<log>.rank<r>"); ReplayLog of a farm manager reads all shards, so
    the rebuilt sums / counters are the global ones and resumed sample ids do not overlap earlier ones.  (The reference
    logs on pid 0 only, src/MLMC_Manager.cpp:106-108.)"""
    log = str(tmp_path / "MLMC.dat")
    parts = []
    for rank in range(2):
        m = host_api.MLMCManager(3, callbacks=SyntheticPlugin(3).callbacks(), wall_time=False, batch=4, log_file=log)
        m.set_farm(2, rank, lambda buf: None)        # no exchange: each rank keeps its partial sums
        parts.append(m.InitRun([6, 9, 14])["sums"].copy())
        m.close()
    assert os.path.exists(log) and os.path.exists(log + ".rank1")
    n0 = len([ln for ln in open(log) if not ln.startswith("%")])
    n1 = len([ln for ln in open(log + ".rank1") if not ln.startswith("%")])
    assert n0 + n1 == 29 and 0 < n1 < 29
    serial = host_api.MLMCManager(3, callbacks=SyntheticPlugin(3).callbacks(), wall_time=False, batch=4)
    ref = serial.InitRun([6, 9, 14])
    assert np.allclose(parts[0] + parts[1], ref["sums"], rtol=1e-13, atol=1e-14)
    b = host_api.MLMCManager(3, callbacks=SyntheticPlugin(3).callbacks(), wall_time=False, batch=4)
    b.set_farm(2, 0, lambda buf: None)
    assert b.ReplayLog(log) == 29
    rb = b.result()
    assert np.allclose(rb["sums"], ref["sums"], rtol=1e-13, atol=1e-14) and list(rb["nsamples"]) == [6, 9, 14]
    b.close()
    # a missing shard is an error, not a silent partial resume
    os.remove(log + ".rank1")
    c = host_api.MLMCManager(3, callbacks=SyntheticPlugin(3).callbacks(), wall_time=False, batch=4)
    c.set_farm(2, 0, lambda buf: None)
    with pytest.raises(Exception):
        c.ReplayLog(log)
    c.close()
    serial.close()


# ---------------------------------------------------------------------------------- ratio estimator (Bayesian)
def _synthetic_likelihood(nl):
    def like(level, k):
        g = np.log(k).mean(axis=1)
        l = np.exp(-0.5 * (g - 0.1) ** 2 / 0.05) * (1.0 + 0.1 * 2.0 ** (-(nl - level)))
        q = g * 2.0 + 1.0 / (1 + level)
        return l, l * q, np.full(k.shape[0], float(NDOFS[level]))
    return like


@pytest.mark.parametrize("nl", [1, 2, 3])
def test_ratio_manager_sums_and_statistics(nl):
    """ML_BayesRatio_Manager::InitRun / computeNSamplesMSE (src/ML_BayesRatio_Manager.hpp:315-433,560-728) against the
    Python restatement; two independent prior draws per realization, coarse level evaluated on the same draws."""
    from oracle import ratio_oracle as ro
    pl = SyntheticPlugin(nl)
    like = _synthetic_likelihood(nl)
    mgr = host_api.RatioManager(nl, callbacks=pl.callbacks(), likelihood=like, wall_time=False, batch=4, eps2=1e-3)
    ns = [5, 8, 13][:nl]
    r = mgr.InitRun(ns)
    sums = np.zeros((nl, ro.NVAR))
    ref = SyntheticPlugin(nl)
    for lvl in range(nl - 1, -1, -1):
        for i in range(ns[lvl]):
            zxi, xi = ref.sample(lvl, (1 << 62) + i, 1), ref.sample(lvl, i, 1)
            zl, _, c1 = like(lvl, ref.eval(lvl, lvl, zxi, None, None)[0])
            _, rr, c2 = like(lvl, ref.eval(lvl, lvl, xi, None, None)[0])
            ctot, zc, rc = c1[0] + c2[0], 0.0, 0.0
            if lvl < nl - 1:
                zcl, _, c3 = like(lvl + 1, ref.eval(lvl + 1, lvl, zxi, None, None)[0])
                _, rcl, c4 = like(lvl + 1, ref.eval(lvl + 1, lvl, xi, None, None)[0])
                zc, rc, ctot = zcl[0], rcl[0], ctot + c3[0] + c4[0]
                ro.accumulate(sums, lvl, rr[0], rr[0] - rc, zl[0], zl[0] - zc, ctot)
            else:
                ro.accumulate(sums, lvl, rr[0], rr[0], zl[0], zl[0], ctot)
    assert np.allclose(r["sums"], sums, rtol=1e-12, atol=1e-14)
    st = ro.compute(sums, ns, NDOFS[:nl], 1e-3, 0.5)
    for key in ("eR", "varR", "eYR", "varYR", "eABS_YR", "eZ", "varZ", "eYZ", "varYZ", "eABS_YZ", "eC"):
        assert np.allclose(r[key], st[key], rtol=1e-10, atol=1e-14), key
    assert r["ratio_estimate"] == pytest.approx(st["ratio_estimate"], rel=1e-10)
    assert r["bias2"] == pytest.approx(st["bias2"], rel=1e-9, abs=1e-16)
    assert r["estimator_variance"] == pytest.approx(st["estimator_variance"], rel=1e-10)
    assert list(r["missing"]) == st["missing"]
    mgr.close()


@pytest.mark.parametrize("nl", [1, 3])
def test_ratio_splitting_manager_sums_and_statistics(nl):
    """ML_BayesRatio_Splitting_Manager / SL_BayesRatio_Splitting_Manager ("divide, then subtract":
    src/ML_BayesRatio_Splitting_Manager.hpp:297-432 InitRun, :595-737 computeNSamplesMSE) against the restatement."""
    from oracle import ratio_oracle as ro
    pl = SyntheticPlugin(nl)
    like = _synthetic_likelihood(nl)
    mgr = host_api.RatioManager(nl, callbacks=pl.callbacks(), likelihood=like, wall_time=False, batch=4, eps2=1e-3,
                                splitting=True)
    ns = [6, 9, 14][:nl]
    r = mgr.InitRun(ns)
    sums = np.zeros((nl, ro.NVAR))
    ref = SyntheticPlugin(nl)
    for lvl in range(nl - 1, -1, -1):
        for i in range(ns[lvl]):
            zxi, xi = ref.sample(lvl, (1 << 62) + i, 1), ref.sample(lvl, i, 1)
            zl, _, c1 = like(lvl, ref.eval(lvl, lvl, zxi, None, None)[0])
            _, rr, c2 = like(lvl, ref.eval(lvl, lvl, xi, None, None)[0])
            q = rr[0] / zl[0]
            if lvl < nl - 1:
                zcl, _, c3 = like(lvl + 1, ref.eval(lvl + 1, lvl, zxi, None, None)[0])
                _, rcl, c4 = like(lvl + 1, ref.eval(lvl + 1, lvl, xi, None, None)[0])
                ro.accumulate(sums, lvl, rr[0], rr[0] - rcl[0], zl[0], zl[0] - zcl[0], c1[0] + c2[0] + c3[0] + c4[0])
                ro.accumulate_ratio(sums, lvl, q, q - rcl[0] / zcl[0])
            else:
                ro.accumulate(sums, lvl, rr[0], rr[0], zl[0], zl[0], c1[0] + c2[0])
                ro.accumulate_ratio(sums, lvl, q, q)
    assert np.allclose(r["sums"], sums, rtol=1e-12, atol=1e-14)
    st = ro.compute_splitting(sums, ns, NDOFS[:nl], 1e-3, 0.5)
    for key in ("eRatio", "varRatio", "eYRatio", "varYRatio", "eABS_YRatio", "eC"):
        assert np.allclose(r[key], st[key], rtol=1e-10, atol=1e-14), key
    assert r["ratio_estimate"] == pytest.approx(st["ratio_estimate"], rel=1e-10)
    assert r["bias2"] == pytest.approx(st["bias2"], rel=1e-9, abs=1e-16)
    assert r["estimator_variance"] == pytest.approx(st["estimator_variance"], rel=1e-10)
    if nl > 2:
        assert r["alpha"] == pytest.approx(st["alpha"], rel=1e-9) and r["beta"] == pytest.approx(st["beta"], rel=1e-9)
    assert list(r["missing"]) == st["missing"]
    mgr.close()
    # adaptive loop terminates on the Ratio variance
    mgr = host_api.RatioManager(nl, callbacks=SyntheticPlugin(nl).callbacks(), likelihood=like, wall_time=False, batch=8,
                                eps2=2e-3, init_nsamples=10, splitting=True)
    r = mgr.Run()
    assert r["estimator_variance"] <= 0.5 * 2e-3 and np.isfinite(r["ratio_estimate"])
    mgr.close()


def test_ratio_manager_run_converges():
    pl = SyntheticPlugin(3)
    mgr = host_api.RatioManager(3, callbacks=pl.callbacks(), likelihood=_synthetic_likelihood(3), wall_time=False, batch=8,
                                eps2=5e-4, init_nsamples=10)
    r = mgr.Run()
    assert r["estimator_variance"] <= 0.5 * 5e-4 and np.isfinite(r["ratio_estimate"])
    mgr.close()


def test_show_me_prints_the_reference_table():
    """MLMC_Manager::ShowMe (src/MLMC_Manager.cpp:216-297): labels, widths and precision of the reference's table, so that
    its ctest regex on the estimate line ("Estimate" padded to 42 columns, examples/CMakeLists.txt:80) keeps matching."""
    import re
    pl = SyntheticPlugin(3)
    mgr = host_api.MLMCManager(3, callbacks=pl.callbacks(), wall_time=False, batch=4)
    r = mgr.InitRun([6, 9, 14])
    txt = mgr.ShowMe()
    lines = txt.splitlines()
    assert lines[0] == "=" * 79 and lines[1].startswith("MLMC Manager Errors:") and lines[2] == "-" * 79 and lines[-1] == "=" * 79
    m = re.search(r"^Estimate {34}(\S+)", txt, re.M)
    assert m and float(m.group(1)) == pytest.approx(r["estimate"], rel=1e-7)
    for label in ("Target MSE", "Actual MSE", "ML Estimator Variance", "Estimator Bias", "Alpha", "AlphaAbs", "Beta", "Gamma",
                  "DOFS in Forward Problem", "C_l ", "NumSamples ", "E[Y_l] ", "E[|Y_l|] ", "Var[Y_l] ", "E[Q_l] ", "E[|Q_l|] ",
                  "Var[Q_l] ", "V[Y_l]*C_l ", "Consistency ", "Kurtosis", "NNZ-Sampler", "NNZ-ForwardSolve"):
        assert any(ln.startswith(label.ljust(42)) for ln in lines), label
    ns_line = next(ln for ln in lines if ln.startswith("NumSamples"))
    assert ns_line.split()[1:] == ["6", "9", "14"]
    mgr.close()
