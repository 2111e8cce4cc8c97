"""Anisotropy probe: SPE10-shaped box (cells ~21 x 10 x 2), sampler + Darcy iteration counts (development aid)."""
import sys
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from parelagmc_amd import capi  # noqa: E402
from parelagmc_amd.fe import box_mesh, build_darcy_problem, build_hierarchy, build_sampler_problem  # noqa: E402
nref = int(sys.argv[1]) if len(sys.argv) > 1 else 2
h = build_hierarchy(box_mesh([7, 27, 10], [1200.0, 2200.0, 170.0], "hex"), nref)
sp = build_sampler_problem(h, corlen=100.0, lognormal=True, n_mc_levels=min(3, nref + 1))
dp = build_darcy_problem(h, [1, 0, 1, 0, 1, 1], [0, 1, 0, 0, 0, 0], [0, 0, 0, 1, 0, 0], n_mc_levels=min(3, nref + 1))
print("dofs", [L.ndofs for L in dp.levels], flush=True)
ctx = capi.Context(0, seed=3)
smp0 = capi.PDESampler(ctx, sp, opts=capi.solver_opts(mg_coarsening=0))
smp = capi.PDESampler(ctx, sp, opts=capi.solver_opts(mg_coarsening=1))
ds0 = capi.DarcySolver(ctx, dp, opts=capi.solver_opts(mg_coarsening=0))
ds = capi.DarcySolver(ctx, dp, opts=capi.solver_opts(mg_coarsening=1))
for lvl in range(min(3, nref + 1)):
    xi = smp.Sample(lvl, first_id=0, nbatch=4)
    s0, st0 = smp0.Eval(lvl, xi, return_stats=True)
    s, st = smp.Eval(lvl, xi, return_stats=True)
    print(f"L{lvl}: geometric iters {[t[0] for t in st0]}  SA iters {[t[0] for t in st]}  rel diff "
          f"{np.abs(np.log(s) - np.log(s0)).max() / np.abs(np.log(s0)).max():.2e}", flush=True)
    Qg, _, stg = ds0.SolveFwd(lvl, s, return_stats=True)
    Q, C, st2 = ds.SolveFwd(lvl, s, return_stats=True)
    print(f"L{lvl}: darcy geometric iters {[t[0] for t in stg]} conv {[t[1] for t in stg]} Q {Qg}", flush=True)
    print(f"L{lvl}: darcy SA        iters {[t[0] for t in st2]} conv {[t[1] for t in st2]} Q {Q}", flush=True)
    Q1, _, st3 = ds.SolveFwd(lvl, np.ones((1, dp.levels[lvl].n_p)), return_stats=True)
    print(f"L{lvl}: sampler iters {[t[0] for t in st]} conv {[t[1] for t in st]} | darcy iters {[t[0] for t in st2]} conv {[t[1] for t in st2]} "
          f"| k=1: Q={Q1[0]:.6f} (exact {1200 * 170 / 2200:.6f}) iters {st3[0][0]}", flush=True)
