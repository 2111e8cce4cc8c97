"""which tail kernels survive a library variant: small-level sampler (mini kernel), mid-size sampler (LDS tail), hybrid (round 4)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from parelagmc_amd import capi  # noqa: E402
if os.environ.get("HYB_LIB"):
    capi.LIB_PATH = os.path.join(ROOT, "parelagmc_amd", "lib", os.environ["HYB_LIB"])
from oracle.sampler_oracle import SamplerOracle  # noqa: E402
from parelagmc_amd.fe import box_mesh, build_hierarchy, build_hybrid_sampler_problem, build_sampler_problem  # noqa: E402

h = build_hierarchy(box_mesh([4, 4, 4], [2, 2, 2], "hex"), 2)
sp, hp = build_sampler_problem(h, corlen=0.1), build_hybrid_sampler_problem(h, corlen=0.1)
so = SamplerOracle(sp)
ctx = capi.Context(0, seed=3)
xi = np.random.default_rng(1).standard_normal((3, sp.levels[0].n_s))
for name, prob, kw in (("saddle mini off", sp, dict(mini_max_rows=0)), ("saddle default", sp, dict()), ("hybrid", hp, dict())):
    smp = capi.PDESampler(ctx, prob, capi.solver_opts(rel_tol=1e-10, **kw))
    for lvl in range(3):
        s, st = smp.Eval(lvl, xi, xi_level=0, return_stats=True)
        ref = np.stack([so.eval(lvl, 0, x)[0] for x in xi])
        print(f"{os.environ.get('HYB_LIB', 'libpmc.so')} {name} level {lvl}: error {np.linalg.norm(s - ref) / np.linalg.norm(ref):.2e} iterations {[t[0] for t in st]} converged {[t[1] for t in st]}", flush=True)
    smp.close()
