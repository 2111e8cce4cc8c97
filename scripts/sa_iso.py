"""Isotropic single-level problem, smoothed-aggregation variants (development aid)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from parelagmc_amd.fe import box_mesh, build_darcy_problem, build_hierarchy, build_sampler_problem
from parelagmc_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 1
nlev = int(sys.argv[3]) if len(sys.argv) > 3 else 0
h = build_hierarchy(box_mesh([n >> nlev] * 3, [2.0, 2.0, 2.0], "hex"), nlev)
sp = build_sampler_problem(h, corlen=0.5, lognormal=True, n_mc_levels=1)
dp = build_darcy_problem(h, [0, 1, 1, 1, 1, 0], [1, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 1], n_mc_levels=1)
ctx = capi.Context(0, seed=1)
o = capi.solver_opts(mg_coarsening=mode)
smp = capi.PDESampler(ctx, sp, o); ds = capi.DarcySolver(ctx, dp, o)
xi = smp.Sample(0, first_id=0, nbatch=4)
s, st = smp.Eval(0, xi, return_stats=True)
t0 = time.time(); s, st = smp.Eval(0, xi, return_stats=True); t1 = time.time()
Q, C, st2 = ds.SolveFwd(0, s, return_stats=True)
t2 = time.time(); Q, C, st2 = ds.SolveFwd(0, s, return_stats=True); t3 = time.time()
print(f"n={n} mode={mode} levels={nlev+1} env={ {k: v for k, v in os.environ.items() if k.startswith('PMC_SA')} }: sampler {max(t[0] for t in st)} its {1e3*(t1-t0)/4:.2f} ms/sample, "
      f"darcy {max(t[0] for t in st2)} its {1e3*(t3-t2)/4:.2f} ms/sample", flush=True)
