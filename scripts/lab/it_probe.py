import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import bench
from parelagmc_amd import capi
p = bench.build_problem(int(sys.argv[1]) if len(sys.argv) > 1 else 4)
n = p.levels[0].n_s
ctx = capi.Context(0, seed=3)
xi = np.random.default_rng(1).standard_normal((16, n))
for kw in [dict(), dict(cheb_degree_M=4), dict(cheb_degree_M=8), dict(mg_smooth_degree=3), dict(mg_smooth_degree=4, cheb_degree_M=8),
           dict(mg_smooth_degree=4, cheb_degree_M=8, mg_smooth_ratio=20.0), dict(mg_smooth_degree=6, cheb_degree_M=12, mg_smooth_ratio=30.0, mg_coarse_degree=30)]:
    smp = capi.PDESampler(ctx, p, capi.solver_opts(mini_max_rows=0, **kw))
    s, st = smp.Eval(0, xi, return_stats=True)
    print(kw, "iterations", sorted(set(t[0] for t in st)), flush=True)
    smp.close()
