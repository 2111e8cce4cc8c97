"""hybridized against saddle-point sampler on the other BASELINE configurations (round 4): hex64 point, config 4, config 5"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from parelagmc_amd import capi  # noqa: E402
if os.environ.get("HYB_LIB"):
    capi.LIB_PATH = os.path.join(ROOT, "parelagmc_amd", "lib", os.environ["HYB_LIB"])
from parelagmc_amd.fe import (box_mesh, build_darcy_problem, build_hierarchy, build_hybrid_sampler_problem,  # noqa: E402
                              build_sampler_problem, l2_projection_hierarchy, mesh_from_json)

which = sys.argv[1].split(",") if len(sys.argv) > 1 else ["hex64", "c4", "c5"]
seed = 20261003
if "hex64" in which:
    h = build_hierarchy(box_mesh([4, 4, 4], [2, 2, 2], "hex"), 4)
    for name, fn in (("saddle", build_sampler_problem), ("hybrid", build_hybrid_sampler_problem)):
        p = fn(h, corlen=0.1, lognormal=True, n_mc_levels=3)
        r = bench.sampler_point(p, 0, seed, 32, 4, 6, "hex64", name, roofline=False)
        print(f"hex64 {name}: {r['value']:.1f} samples/s, iterations {r['mean_minres_iterations']:.1f}", flush=True)
    del h
if "c4" in which:
    h = build_hierarchy(mesh_from_json(os.path.join(ROOT, "tests", "golden", "meshes", "cube_tet_embed.json")), 4)
    for name, fn in (("saddle", build_sampler_problem), ("hybrid", build_hybrid_sampler_problem)):
        t0 = time.time()
        p = fn(h, corlen=0.1, embedded=True, lognormal=True, n_mc_levels=3)
        ts = time.time() - t0
        r = bench.config4(seed, p, cpu=False, nrep=3)
        print(f"c4 {name} (setup {ts:.1f} s):", [(round(x['realizations_per_s'], 1), round(x['sampler_iterations_mean'], 1)) for x in r["levels"]], flush=True)
    del h
if "c5" in which:
    nx, ny, nz = 7, 27, 10
    hx, hy, hz = 1200.0 / nx, 2200.0 / ny, 170.0 / nz
    ho = build_hierarchy(box_mesh([nx, ny, nz], [1200.0, 2200.0, 170.0], "hex"), 3)
    he = build_hierarchy(box_mesh([nx + 2, ny + 2, nz + 2], [1200.0 + 2 * hx, 2200.0 + 2 * hy, 170.0 + 2 * hz], "hex",
                                  origin=[-hx, -hy, -hz]), 3)
    ops = l2_projection_hierarchy(ho, he)
    dp = build_darcy_problem(ho, [1, 0, 1, 0, 1, 1], [0, 1, 0, 0, 0, 0], [0, 0, 0, 1, 0, 0])
    for name, fn in (("saddle", build_sampler_problem), ("hybrid", build_hybrid_sampler_problem)):
        t0 = time.time()
        p = fn(he, corlen=100.0, lognormal=True)
        ts = time.time() - t0
        r = bench.config5(seed, (p, ops, dp), cpu=False, nrep=2)
        print(f"c5 {name} (setup {ts:.1f} s):", [(round(x['realizations_per_s'], 1), round(x['sampler_iterations_mean'], 1), round(x.get('darcy_iterations_mean', 0), 1)) for x in r["levels"]],
              "round", round(r["mlmc_round"]["realizations_per_s"], 1), flush=True)
