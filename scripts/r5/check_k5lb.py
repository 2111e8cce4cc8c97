"""laboratory check: with PMC_K5_LB=1 the hybridized sampler (cube_tet r = 4 and r = 5, tight tolerance) returns the field of the
product path; run with HYB_LIB=libpmc_lab.so"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from parelagmc_amd import capi  # noqa: E402
capi.LIB_PATH = os.path.join(ROOT, "parelagmc_amd", "lib", os.environ.get("HYB_LIB", "libpmc_lab.so"))

for nref in (4, 5):
    hp = bench.build_hybrid_problem(nref)
    ctx = capi.Context(0, seed=5)
    o = capi.solver_opts(rel_tol=1e-12, abs_tol=1e-300)
    os.environ["PMC_K5_LB"] = "1"
    a = capi.PDESampler(ctx, hp, o)
    os.environ["PMC_K5_LB"] = "0"
    b = capi.PDESampler(ctx, hp, o)
    w = a.BatchWidth(0)
    xi = a.Sample(0, first_id=3, nbatch=w)
    sa, sta = a.Eval(0, xi, return_stats=True)
    sb, stb = b.Eval(0, xi, return_stats=True)
    err = np.linalg.norm(sa - sb) / np.linalg.norm(sb)
    print(f"r={nref}: width {w}, LDS-blocked K5 vs SELL K5: field difference {err:.2e}, iterations {sta[0][0]} / {stb[0][0]}", flush=True)
    assert err < 1e-9 and all(t[1] == 1 for t in sta)
    a.close(); b.close(); ctx.close()
print("k5lb ok")
