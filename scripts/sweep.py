"""Solver-option sweep on the bench workload (development aid)."""
import itertools
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from bench import build_problem  # noqa: E402
from parelagmc_amd import capi  # noqa: E402

nref = int(sys.argv[1]) if len(sys.argv) > 1 else 5
problem = build_problem(nref)
ctx = capi.Context(0, seed=1)
n = problem.levels[0].n_s
nb = 16
xi_d, s_d = ctx.empty(nb * n), ctx.empty(nb * n)
for gam in (0.6, 0.8, 1.0, 1.2, 1.5, 2.0):
    degM, ratM, sdeg, srat = 2, 8, 2, 8
    opts = capi.solver_opts(schur_scale=gam)
    smp = capi.PDESampler(ctx, problem, opts)
    smp.Sample(0, first_id=0, nbatch=nb, out=xi_d)
    smp.Eval(0, xi_d, xi_level=0, s_out=s_d)
    ctx.synchronize()
    ctx.timer_start()
    reps = 4
    for _ in range(reps):
        st = smp.Eval(0, xi_d, xi_level=0, s_out=s_d, return_stats=True)[1]
    ms = ctx.timer_stop() / reps
    it = np.mean([t[0] for t in st])
    print(f"gamma={gam} degM={degM} ratM={ratM} sdeg={sdeg} srat={srat}: {ms:7.2f} ms/batch  iters {it:5.1f}  {ms / it * 1e3:6.1f} us/it  "
          f"{nb / ms * 1e3:7.1f} samples/s  conv {all(t[1] == 1 for t in st)}", flush=True)
    smp.close()
