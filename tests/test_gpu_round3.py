"""GPU tests added in round 3: `bench.py --gpus N` starting its own ranks, the real sample farm (device plugins, two
processes), super-batches (several column groups of 32 realizations per launch) against narrow batches and the oracle, the
Darcy operator's in-loop timing and the per-phase device timers.  Run with -m gpu on an MI355X; everything goes through the
C ABI."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(b)


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus N` without a launcher (the driver's command) must run N ranks (the reference starts its
    ranks under mpirun, examples/MLMC.cpp:43-50).  Rehearsed with FOUR ranks x one lane on one GPU (the box admits at most six
    processes on its card, so eight cannot be rehearsed here; tests/test_farm_gloo.py runs eight ranks of the manager on the
    CPU): the ranks share device 0, torch.distributed falls back to gloo, RCCL refuses several ranks on one device and the
    farm's accumulators are summed over gloo instead - same manager, same sharding, same single reduction per round."""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    W = 4
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(W), "--streams", "1", "--refine", "3",
                        "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-r6"], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    # stdout carries the compact line (< 4 kB whatever the rank count); per-rank dicts and the farm block are in the full record
    assert len(lines[0]) < 4096
    line = json.loads(lines[0])
    assert line["n_gpus"] == W and line["scaling"] == "weak" and line["rank_seconds"]["max"] >= line["rank_seconds"]["min"] > 0
    assert line["extra"]["farm_allreduces_in_round"] == 1 and line["extra"]["farm_collective"].startswith("gloo")
    assert "ranks" not in line
    out = json.load(open(os.path.join(ROOT, line["full_record"])))
    assert abs(out["value"] - line["value"]) < 1e-3 * out["value"]
    assert out["n_gpus"] == W and out["scaling"] == "weak"
    ranks = out["ranks"]
    assert [x["rank"] for x in ranks] == list(range(W))
    # leap-frog split: rank r owns the global realization ids r, r + W, ... - same count, disjoint ranges
    per = 2 * out["config"]["batch"] * out["config"]["streams"]
    assert all(x["samples"] == per and x["id_stride"] == W for x in ranks)
    assert [x["global_ids"][0] % W for x in ranks] == list(range(W))
    ids = [set(range(x["global_ids"][0], x["global_ids"][1] + 1, W)) for x in ranks]
    assert all(len(s_) == per for s_ in ids) and len(set().union(*ids)) == W * per
    assert all("pinned" in x["cpu_affinity"] for x in ranks)          # every rank tried to bind to its GPU's NUMA node
    assert abs(out["value"] - W * per / (out["ms_per_step"] * 1e-3 * out["steps"])) < 1e-6 * out["value"]
    farm = out["extra"]["mlmc_farm"]
    assert "error" not in farm, farm
    assert farm["nsamples_after_allreduce"] == [64 * W, 256 * W, 1024 * W]
    assert farm["allreduces_in_round"] == 1 and farm["allreduce_ms"] >= 0.0 and len(farm["allreduce_ms_per_rank"]) == W
    assert farm["collective"].startswith("gloo")
    assert "cpu_baseline" not in out and "r6" not in out["extra"]


def test_super_batches_match_narrow_batches_and_the_oracle(gpu_ctx, hex_hierarchy, seeded_rng):
    """Levels too small to fill the chip with 32 realizations are solved as column groups of 32 in ONE launch (gridDim.y;
    batch_width in csrc/solver.hip: 64 / 128 / 256 realizations).  200 realizations in one call (chunks of 128 + 64 + 8)
    == the same realizations 32 per call == the oracle's direct solves, for the sampler on every level (the two coarse
    ones run the persistent one-workgroup-per-realization solver) and for Darcy; per-column freeze makes the longer joint
    iteration exact.  The loop being restated is src/MLMC_Manager.cpp:113-173."""
    from oracle.darcy_oracle import DarcyOracle
    from oracle.sampler_oracle import SamplerOracle
    from parelagmc_amd import capi
    from parelagmc_amd.fe import build_darcy_problem, build_sampler_problem
    sp_ = build_sampler_problem(hex_hierarchy, corlen=0.1, lognormal=True)
    dp = build_darcy_problem(hex_hierarchy, [0, 1, 1, 1, 1, 0], [1, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 1])
    o = capi.solver_opts(rel_tol=1e-12, abs_tol=1e-14)
    smp, ds = capi.PDESampler(gpu_ctx, sp_, o), capi.DarcySolver(gpu_ctx, dp, o)
    so, do = SamplerOracle(sp_), DarcyOracle(dp)
    nreal = 200
    xi = seeded_rng.standard_normal((nreal, sp_.levels[0].n_s))
    probe = (0, 31, 32, 63, 64, 127, 128, 191, 192, 199)        # both sides of every group / chunk boundary
    for lvl in range(3):
        sw, st = smp.Eval(lvl, xi, xi_level=0, return_stats=True)
        assert len(st) == nreal and all(t[1] == 1 for t in st)
        sn = np.vstack([smp.Eval(lvl, xi[i:i + 32], xi_level=0) for i in range(0, nreal, 32)])
        assert np.allclose(sw, sn, rtol=1e-9, atol=0)
        for b in probe:
            ref = so.eval(lvl, 0, xi[b])[0]
            assert np.linalg.norm(sw[b] - ref) <= 1e-8 * np.linalg.norm(ref), (lvl, b)
        Qw, Cw, stq = ds.SolveFwd(lvl, sw, return_stats=True)
        assert all(t[1] == 1 for t in stq) and np.all(Cw == dp.levels[lvl].ndofs)
        Qn = np.concatenate([ds.SolveFwd(lvl, sw[i:i + 32])[0] for i in range(0, nreal, 32)])
        assert np.allclose(Qw, Qn, rtol=1e-9)
        for b in probe[::2]:
            assert abs(Qw[b] - do.solve_fwd(lvl, sw[b])[0]) <= 1e-8 * abs(Qw[b]), (lvl, b)
    # warm start + coarse xi through the wide path: the level pair as the manager calls it, ONE device buffer for the
    # initial guess (in) and the embedded Gaussian field (out) over several chunks (src/MLMC_Manager.cpp:150-156)
    n0, n1 = sp_.levels[0].n_s, sp_.levels[1].n_s
    d_xi = gpu_ctx.array(xi)
    d_s, buf = gpu_ctx.empty(nreal * n0), gpu_ctx.empty(nreal * n0)
    smp.Eval(1, d_xi, xi_level=0, s_out=d_s, embed_out=buf)
    coarse = d_s.download()[: nreal * n1].reshape(nreal, n1)
    gauss1 = buf.download()[: nreal * n1].reshape(nreal, n1)
    assert np.allclose(np.exp(gauss1), coarse, rtol=1e-12)
    _, _, st = smp.Eval(0, d_xi, xi_level=0, init_s=buf, init_level=1, use_init=True, s_out=d_s, embed_out=buf,
                        return_stats=True)
    assert all(t[1] == 1 for t in st)
    cold = smp.Eval(0, xi, xi_level=0)
    assert np.allclose(d_s.download().reshape(nreal, n0), cold, rtol=1e-8)
    assert np.allclose(np.exp(buf.download().reshape(nreal, n0)), cold, rtol=1e-8)
    # K1: the white noise of a wide batch is the same stream as narrow batches
    a = smp.Sample(2, first_id=5, nbatch=256)
    b = np.vstack([smp.Sample(2, first_id=5 + i, nbatch=32) for i in range(0, 256, 32)])
    assert np.array_equal(a, b)
    ds.close()
    smp.close()


def test_manager_sums_do_not_depend_on_the_super_batch_width(gpu_ctx, hex_hierarchy):
    """MLMC_Manager::InitRun with 256 / 96 / 32 realizations per plugin call: same realization ids, same sums to the
    (tight) solver tolerance, same estimate; two lanes as well."""
    from parelagmc_amd import capi, host_api
    from parelagmc_amd.fe import build_darcy_problem, build_sampler_problem
    sp_ = build_sampler_problem(hex_hierarchy, corlen=0.1, lognormal=True)
    dp = build_darcy_problem(hex_hierarchy, [0, 1, 1, 1, 1, 0], [1, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 1])
    o = capi.solver_opts(rel_tol=1e-12, abs_tol=1e-14)
    smp, ds = capi.PDESampler(gpu_ctx, sp_, o), capi.DarcySolver(gpu_ctx, dp, o)
    out = []
    for batch in (256, 96, 32):
        mgr = host_api.MLMCManager(3, sampler=smp, solver=ds, wall_time=False, batch=batch)
        out.append(mgr.InitRun([70, 300, 520]))
        mgr.close()
    c2 = capi.Context(0, seed=20261003)
    smp2, ds2 = capi.PDESampler(c2, sp_, o), capi.DarcySolver(c2, dp, o)
    mgr = host_api.MLMCManager(3, sampler=smp, solver=ds, wall_time=False, batch=128)
    mgr.add_lane(smp2, ds2)
    out.append(mgr.InitRun([70, 300, 520]))
    mgr.close()
    for r in out[1:]:
        assert np.allclose(r["sums"], out[0]["sums"], rtol=1e-8, atol=1e-10)
        assert r["estimate"] == pytest.approx(out[0]["estimate"], rel=1e-9)
        assert list(r["nsamples"]) == list(out[0]["nsamples"])
    for h in (ds2, smp2, ds, smp):
        h.close()
    c2.close()


def test_phase_timers_operator_timing_and_abi_version(gpu_ctx, hex_hierarchy_small, seeded_rng):
    """(i) pmc_stats.solve_ms / setup_ms: device time of every solve, accumulated by the managers under the reference's
    TimeManager names "Sampler: Mult", "Darcy: Build Solver", "Darcy: Mult" -- Level i (src/PDESampler.cpp:328-333,
    src/DarcySolver.cpp:231-243, printed at examples/MLMC.cpp:275).  (ii) pmc_darcy_set_operator_timing brackets every in-loop
    launch of the u-rows [M(k) | B^T] x without changing a result; pmc_darcy_operator_bytes restates DESIGN section 4.
    (iii) a pmc_solver_opts of another ABI version is refused."""
    from parelagmc_amd import capi, host_api
    from parelagmc_amd.fe import build_darcy_problem, build_sampler_problem
    sp_ = build_sampler_problem(hex_hierarchy_small, corlen=0.1, lognormal=True)
    dp = build_darcy_problem(hex_hierarchy_small, [0, 1, 1, 1, 1, 0], [1, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 1])
    assert gpu_ctx.lib.pmc_abi_version() == 3
    bad = capi.solver_opts()
    bad.abi_version = 1
    with pytest.raises(capi.PmcError):
        capi.PDESampler(gpu_ctx, sp_, bad)
    with pytest.raises(capi.PmcError):
        capi.DarcySolver(gpu_ctx, dp, bad)
    smp, ds = capi.PDESampler(gpu_ctx, sp_, capi.solver_opts(mini_max_rows=0)), capi.DarcySolver(gpu_ctx, dp)
    xi = seeded_rng.standard_normal((16, sp_.levels[0].n_s))
    s = smp.Eval(0, xi)
    setup_ms, solve_ms = smp.last_phase_ms
    assert 0.0 < setup_ms < solve_ms < 1e4
    n0 = gpu_ctx.lib.pmc_kernel_launches()
    Q, C, st = ds.SolveFwd(0, s, return_stats=True)
    launches = gpu_ctx.lib.pmc_kernel_launches() - n0
    its = max(t[0] for t in st)
    assert launches > 5 * its                                   # several kernels per MINRES iteration
    build_ms, mult_ms = ds.last_phase_ms
    assert build_ms > 0.0 and mult_ms > build_ms
    # in-loop operator timing: same results, one bracket per operator application of the loop
    ds.set_operator_timing(True)
    ds.operator_time()
    Q2, _, st2 = ds.SolveFwd(0, s, return_stats=True)
    ms, n, gap = ds.operator_time()
    ds.set_operator_timing(False)
    assert np.allclose(Q2, Q, rtol=1e-12) and [t[:2] for t in st2] == [t[:2] for t in st]
    assert its - 2 <= n <= its + 2 and 0.0 < gap < ms
    assert ds.operator_time() == (0.0, 0, 0.0)
    L = dp.levels[0]
    nb = 16
    bytes_ = ds.operator_bytes(0, nb)
    lower = 8.0 * nb * (2 * L.n_u + L.n_p)                      # x_u, x_p read, y_u written
    assert lower < bytes_ < 4 * lower and ds.operator_bytes(0, 32) > bytes_
    # managers: per-level timers of a round
    mgr = host_api.MLMCManager(2, sampler=smp, solver=ds, wall_time=False)
    mgr.InitRun([20, 40])
    t0, t1 = mgr.phase_times(0), mgr.phase_times(1)
    assert t0["sampler_realizations"] == 20 and t0["darcy_realizations"] == 20
    assert t1["sampler_realizations"] == 40 + 20 and t1["darcy_realizations"] == 40 + 20   # level 1 also serves the pairs of level 0
    for t in (t0, t1):
        assert t["sampler_mult_ms"] > 0 and t["darcy_build_ms"] > 0 and t["darcy_mult_ms"] > 0
    txt = mgr.PrintTimers()
    for name in ("Sampler: Mult -- Level 0", "Darcy: Build Solver -- Level 1", "Darcy: Mult -- Level 1"):
        assert name in txt
    mgr.close()
    ds.close()
    smp.close()


def test_device_farm_of_two_ranks_matches_the_serial_manager(gpu_ctx, hex_hierarchy_small, tmp_path):
    """The real sample farm (src/MLMC_Manager.cpp:103-179 run under mpirun, examples/MLMC.cpp:43-50): two fresh processes
    share device 0, each with device PDESampler + DarcySolver plugins on two lanes, MLMC_Manager::SetFarm(2, r, reduce) with
    a gloo SUM all-reduce.  Sharded InitRun == serial InitRun: identical counts and allocation on both ranks, sums to 1e-12
    (same realizations, other summation order), every realization computed exactly once, and the rank-sharded logs replay to
    the same table."""
    import socket
    from parelagmc_amd import capi, host_api
    from parelagmc_amd.fe import build_darcy_problem, build_sampler_problem
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "farm_worker.py"), str(tmp_path)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-2000:] for o in outs)
    res = [np.load(tmp_path / f"rank{r}.npz") for r in range(2)]
    sp_ = build_sampler_problem(hex_hierarchy_small, corlen=0.1, lognormal=True)
    dp = build_darcy_problem(hex_hierarchy_small, [0, 1, 1, 1, 1, 0], [1, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 1])
    o = capi.solver_opts(rel_tol=1e-12, abs_tol=1e-14)
    smp, ds = capi.PDESampler(gpu_ctx, sp_, o), capi.DarcySolver(gpu_ctx, dp, o)
    serial = host_api.MLMCManager(2, sampler=smp, solver=ds, wall_time=False, eps2=1e-3)
    s1 = serial.InitRun([19, 37])
    s2 = serial.InitRun([6, 0])
    for r in res:
        assert np.allclose(r["sums1"], s1["sums"], rtol=1e-12, atol=1e-13)
        assert np.allclose(r["sums"], s2["sums"], rtol=1e-12, atol=1e-13)
        assert list(r["nsamples"]) == [25, 37] == list(s2["nsamples"])
        assert list(r["missing"]) == list(s2["missing"])                # both ranks derive the same allocation
        assert np.allclose(r["varY"], s2["varY"], rtol=1e-10) and float(r["estimate"]) == pytest.approx(s2["estimate"], rel=1e-12)
        assert int(r["reductions"]) == 2                                 # ONE all-reduce per InitRun round
    # the work was split: level-0 realizations 19 + 6 = 25 over the two ranks, none twice, none missing
    loc = [r["local_realizations"] for r in res]
    assert loc[0][0] + loc[1][0] == 25 and 0 < loc[0][0] < 25
    assert loc[0][1] + loc[1][1] == 37 + 25                              # level 1 also serves the level-0 pairs
    # rank-sharded logs -> one table
    log = str(tmp_path / "MLMC.dat")
    assert os.path.exists(log) and os.path.exists(log + ".rank1")
    b = host_api.MLMCManager(2, sampler=smp, solver=ds, wall_time=False, eps2=1e-3)
    b.set_farm(2, 0, lambda buf: None)
    assert b.ReplayLog(log) == 25 + 37
    rb = b.result()
    assert np.allclose(rb["sums"], s2["sums"], rtol=1e-12, atol=1e-13) and list(rb["nsamples"]) == [25, 37]
    b.close()
    serial.close()
    ds.close()
    smp.close()


def test_super_batch_edges_projections_and_observation_operator(gpu_ctx, hex_hierarchy_small, seeded_rng):
    """Edge cases of the column-group path (MLMC_Manager hands over whatever a round needs, src/MLMC_Manager.cpp:204-208):
    257 realizations on a 256-wide level (chunks 256 + 1), embedded (gather) and L2-projected outputs of a wide batch
    (src/EmbeddedPDESampler.cpp:552-556, src/L2ProjectionPDESampler.cpp:738-750), the full solution vector and the Bayesian
    observation operator (src/BayesianInverseProblem.cpp:178-186) of a wide Darcy batch - all against narrow batches and the
    oracle."""
    from oracle.bayes_oracle import compute_G, observation_functionals
    from oracle.darcy_oracle import DarcyOracle
    from oracle.sampler_oracle import SamplerOracle
    from parelagmc_amd import capi
    from parelagmc_amd.fe import (box_mesh, build_darcy_problem, build_hierarchy, build_sampler_problem, l2_projection_ops)
    tight = dict(rel_tol=1e-12, abs_tol=1e-14)
    # (1) 257 = 256 + 1 on the coarsest level; stats of every realization; same stream as narrow batches
    sp_ = build_sampler_problem(hex_hierarchy_small, corlen=0.1, lognormal=True)
    smp = capi.PDESampler(gpu_ctx, sp_, capi.solver_opts(**tight))
    assert smp.BatchWidth(1) == 256 and smp.BatchWidth(0) == 256
    xi = smp.Sample(1, first_id=3, nbatch=257)
    s, st = smp.Eval(1, xi, return_stats=True)
    assert s.shape == (257, sp_.levels[1].n_s) and len(st) == 257 and all(t[1] == 1 for t in st)
    s32 = np.vstack([smp.Eval(1, xi[i:i + 32]) for i in range(0, 257, 32)])
    assert np.allclose(s, s32, rtol=1e-9, atol=0)
    so = SamplerOracle(sp_)
    for b in (0, 255, 256):
        assert rel(s[b], so.eval(1, 1, xi[b])[0]) < 1e-8
    smp.close()
    # (2) projections of a wide batch
    m = box_mesh([6, 6, 6], [3.0, 3.0, 3.0], "hex", origin=[-0.5, -0.5, -0.5])
    cen = m.verts[m.elems].mean(1)
    m.elem_attr[:] = np.where(np.all((cen > 0) & (cen < 2), axis=1), 1, 2)
    h = build_hierarchy(m, 1)
    spe = build_sampler_problem(h, corlen=0.1, embedded=True, lognormal=True)
    soe = SamplerOracle(spe)
    ga = capi.PDESampler(gpu_ctx, spe, capi.solver_opts(**tight), projection="gather")
    pr = capi.PDESampler(gpu_ctx, spe, capi.solver_opts(**tight), projection="l2", l2_ops=l2_projection_ops(h, spe.orig_index))
    xe = seeded_rng.standard_normal((96, spe.levels[0].n_s))             # 96 = 64 + 32
    for lvl in range(2):
        a, emb = ga.Eval(lvl, xe, xi_level=0, want_embed=True)
        b = pr.Eval(lvl, xe, xi_level=0)
        assert rel(a, b) < 1e-12 and a.shape == (96, len(spe.orig_index[lvl]))
        assert np.allclose(np.exp(emb[:, spe.orig_index[lvl]]), a, rtol=1e-12)
        for i in (0, 63, 64, 95):
            assert rel(a[i], soe.eval(lvl, 0, xe[i], projection=("gather", spe.orig_index[lvl]))[0]) < 1e-8
    ga.close()
    pr.close()
    # (3) Darcy: full solution and observation operator of a wide batch
    dp = build_darcy_problem(hex_hierarchy_small, [0, 1, 1, 1, 1, 0], [1, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 1])
    do = DarcyOracle(dp)
    Gobs = observation_functionals(hex_hierarchy_small, np.array([[0.5, 0.5, 0.5], [1.5, 1.0, 0.4], [1.0, 1.6, 1.7]]), eps=0.3)
    ds = capi.DarcySolver(gpu_ctx, dp, capi.solver_opts(**tight))
    k = np.exp(0.5 * seeded_rng.standard_normal((70, dp.levels[0].n_p)))                # 70 = 64 + 4 + 2
    Q, C, sol = ds.SolveFwd(0, k, want_solution=True)
    assert sol.shape == (70, dp.levels[0].ndofs)
    for i in (0, 63, 64, 69):
        qr, _, ref = do.solve_fwd(0, k[i], return_solution=True)
        assert abs(Q[i] - qr) <= 1e-8 * abs(qr) and rel(sol[i], ref) < 1e-7
    assert np.allclose(sol @ dp.levels[0].obs, Q, rtol=1e-10)                           # Q = <obs, solution>
    ds.SetObservations(0, Gobs[0])
    G, _, Qg = ds.ComputeG(0, k)
    assert np.allclose(Qg, Q, rtol=1e-9)
    for i in (0, 64, 69):
        assert np.allclose(G[i], compute_G(do, Gobs, 0, k[i])[0], rtol=1e-8, atol=1e-10)
    ds.close()


def test_fp32_krylov_vectors_on_every_preconditioner_path(gpu_ctx, hex_hierarchy, seeded_rng):
    """The preconditioned MINRES vectors z are stored in fp32 by default (zvec, csrc/kernels.hpp; pmc_krylov_z_bytes): the last kernel
    of each preconditioner block writes them, the operator products and the w / x updates read them.  Solver configurations
    other than the default one end in other kernels: M-block degree 3 / 4 (typed last Chebyshev step, cheb_step_z; on Darcy
    any degree but 2 also leaves the element-grouped form: pair_spmm_z), V-cycle smoothing degree 3 (the general cycle),
    degree 1 (no typed kernel: fp64 scratch + k::convert_z), the hipGraph replay and the two-stream schedule.  Every one of them must still reproduce the oracle's
    direct solves at a 1e-12 solver tolerance - the stored precision of z only perturbs the preconditioner
    (src/PDESampler.cpp:279-333 and src/DarcySolver.cpp:472-649 are the solves being restated)."""
    from oracle.darcy_oracle import DarcyOracle
    from oracle.sampler_oracle import SamplerOracle
    from parelagmc_amd import capi
    from parelagmc_amd.fe import build_darcy_problem, build_sampler_problem
    assert gpu_ctx.lib.pmc_krylov_z_bytes() in (4, 8)
    sp_ = build_sampler_problem(hex_hierarchy, corlen=0.1, lognormal=True)
    dp = build_darcy_problem(hex_hierarchy, [0, 1, 1, 1, 1, 0], [1, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 1])
    so, do = SamplerOracle(sp_), DarcyOracle(dp)
    nreal = 16
    xi = seeded_rng.standard_normal((nreal, sp_.levels[0].n_s))
    ref_s = {lvl: [so.eval(lvl, 0, xi[b])[0] for b in (0, 7, 15)] for lvl in (0, 1)}
    tight = dict(rel_tol=1e-12, abs_tol=1e-14, mini_max_rows=0)
    variants = [dict(), dict(cheb_degree_M=4), dict(cheb_degree_M=3), dict(mg_smooth_degree=3), dict(use_graph=1),
                dict(two_streams=1), dict(two_streams=2, check_every=1), dict(mg_smooth_degree=1, cheb_degree_M=1, max_iter=900)]
    base_iters = None
    for kw in variants:
        o = capi.solver_opts(**tight, **kw)
        smp, ds = capi.PDESampler(gpu_ctx, sp_, o), capi.DarcySolver(gpu_ctx, dp, o)
        for lvl in (0, 1):
            s, st = smp.Eval(lvl, xi, xi_level=0, return_stats=True)
            assert all(t[1] == 1 for t in st), (kw, lvl, st[:3])
            for j, b in enumerate((0, 7, 15)):
                assert rel(s[b], ref_s[lvl][j]) <= 1e-8, (kw, lvl, b)
            Q, C, stq = ds.SolveFwd(lvl, s, return_stats=True)
            assert all(t[1] == 1 for t in stq), (kw, lvl, stq[:3])
            for b in (0, 15):
                assert abs(Q[b] - do.solve_fwd(lvl, s[b])[0]) <= 1e-8 * abs(Q[b]), (kw, lvl, b)
            if not kw and lvl == 0:
                base_iters = max(t[0] for t in st)
        ds.close()
        smp.close()
    assert base_iters is not None and base_iters > 0
