"""Statistical comparison with the reference's RNG-dependent goldens (development aid -> tests/test_gpu_parity.py):
E[Q_l] of the effective permeability on the ctest problem (4^3 hex on [0,2]^3, 2 refinements, corlen 0.1, log-normal),
examples/CMakeLists.txt:76-80 (MLMC estimate 2.5599, MSE 1e-3) and :91-95 (10-sample means 2.391 / 2.103 / 1.998)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from parelagmc_amd.fe import box_mesh, build_darcy_problem, build_hierarchy, build_sampler_problem
from parelagmc_amd import capi
h = build_hierarchy(box_mesh([4, 4, 4], [2.0, 2.0, 2.0], "hex"), 2)
sp = build_sampler_problem(h, corlen=0.1, lognormal=True)
dp = build_darcy_problem(h, [0, 1, 1, 1, 1, 0], [1, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 1])
ctx = capi.Context(0, seed=2026)
if len(sys.argv) > 2 and sys.argv[2] == "l2":
    # DarcyTest_RandomInput.cpp uses the L2ProjectionPDESampler on the enlarged box (6^3 cells on [-0.5,2.5]^3, aligned)
    from parelagmc_amd.fe import l2_projection_hierarchy
    he = build_hierarchy(box_mesh([6, 6, 6], [3.0, 3.0, 3.0], "hex", origin=[-0.5, -0.5, -0.5]), 2)
    sp = build_sampler_problem(he, corlen=0.1, lognormal=True)
    smp = capi.PDESampler(ctx, sp, projection="l2", l2_ops=l2_projection_hierarchy(h, he))
else:
    smp = capi.PDESampler(ctx, sp)
ds = capi.DarcySolver(ctx, dp)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
for lvl in range(3):
    q = []
    for first in range(0, N, 256):
        s = smp.Eval(lvl, smp.Sample(lvl, first_id=first, nbatch=256))
        Q, _ = ds.SolveFwd(lvl, s)
        q.append(Q)
    q = np.concatenate(q)
    print(f"level {lvl}: E[Q] = {q.mean():.4f} +- {q.std() / np.sqrt(N):.4f} (std {q.std():.3f}, 10-sample std err {q.std() / np.sqrt(10):.3f})", flush=True)
