"""Persistent small-level solver vs the batched kernels on a 16^3 level (development aid); run twice with
PMC_MINI_MAX_ROWS=0 / unset and compare."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from parelagmc_amd.fe import box_mesh, build_hierarchy, build_sampler_problem
from parelagmc_amd import capi
h = build_hierarchy(box_mesh([4, 4, 4], [2.0, 2.0, 2.0], "hex"), 2)
sp = build_sampler_problem(h, corlen=0.1, lognormal=False)
ctx = capi.Context(0, seed=1)
smp = capi.PDESampler(ctx, sp)
for lvl in (0, 1):
    n = sp.levels[lvl].n_s
    xi_d = ctx.array(np.random.default_rng(lvl).standard_normal(16 * n))
    s_d = ctx.empty(16 * n)
    s, st = smp.Eval(lvl, xi_d, xi_level=lvl, s_out=s_d, return_stats=True)
    ctx.timer_start()
    reps = 20
    for _ in range(reps):
        smp.Eval(lvl, xi_d, xi_level=lvl, s_out=s_d)
    ms = ctx.timer_stop() / reps
    out = s_d.download().reshape(16, n)
    print(f"mini_max={os.environ.get('PMC_MINI_MAX_ROWS', 'default')} level {lvl} ({sp.levels[lvl].n_u + n} DoF): {ms:.3f} ms per 16 realizations, "
          f"iterations {sorted(set(t[0] for t in st))}, checksum {np.abs(out).sum():.12e}", flush=True)
