"""Solver-option grid on the bench workload (development aid): ms per batch of 16 and iterations, one lane."""
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
rows = []
for degM, sdeg, srat, gam in itertools.product((1, 2, 3), (1, 2, 3), (4.0, 8.0, 16.0), (0.8, 1.0, 1.3)):
    opts = capi.solver_opts(cheb_degree_M=degM, mg_smooth_degree=sdeg, mg_smooth_ratio=srat, schur_scale=gam)
    try:
        smp = capi.PDESampler(ctx, problem, opts)
        smp.Sample(0, first_id=0, nbatch=nb, out=xi_d)
        smp.Eval(0, xi_d, xi_level=0, s_out=s_d)
        ctx.synchronize()
        ctx.timer_start()
        reps = 3
        for _ in range(reps):
            st = smp.Eval(0, xi_d, xi_level=0, s_out=s_d, return_stats=True)[1]
        ms = ctx.timer_stop() / reps
        it = np.mean([t[0] for t in st])
        ok = all(t[1] == 1 for t in st)
        smp.close()
    except Exception as e:  # noqa: BLE001
        print("failed", degM, sdeg, srat, gam, e, flush=True)
        continue
    rows.append((ms, degM, sdeg, srat, gam, it, ok))
    print(f"degM={degM} sdeg={sdeg} srat={srat} gamma={gam}: {ms:7.2f} ms/batch iters {it:5.1f} {ms / it * 1e3:6.1f} us/it conv {ok}", flush=True)
rows.sort()
print("best:")
for r in rows[:10]:
    print(r)
