"""hybridized sampler, cube_tet r=5, 6 launches of 32 realizations: the process rocprofv3 wraps (round 4)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from parelagmc_amd import capi  # noqa: E402
if os.environ.get("HYB_LIB"):
    capi.LIB_PATH = os.path.join(ROOT, "parelagmc_amd", "lib", os.environ["HYB_LIB"])
from parelagmc_amd.fe import build_hierarchy, build_hybrid_sampler_problem, mesh_from_json  # noqa: E402

nref = int(sys.argv[1]) if len(sys.argv) > 1 else 5
h = build_hierarchy(mesh_from_json(os.path.join(ROOT, "tests", "golden", "meshes", "cube_tet.json")), nref)
hp = build_hybrid_sampler_problem(h, corlen=0.1, n_mc_levels=1)
ctx = capi.Context(0, seed=7)
b = capi.PDESampler(ctx, hp, capi.solver_opts())
nb = 32
dx = ctx.array(b.Sample(0, 0, nb))
ds = ctx.empty(nb * b.SampleSize(0))
for _ in range(6):
    b.Eval(0, dx, xi_level=0, s_out=ds)
ctx.synchronize()
