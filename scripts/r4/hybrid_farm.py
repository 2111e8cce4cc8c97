"""hybridized sampler in the harness shape of the headline (4 lanes x 32 realizations per launch), cube_tet r = 5 (round 4)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from parelagmc_amd import capi  # noqa: E402
if os.environ.get("HYB_LIB"):
    capi.LIB_PATH = os.path.join(ROOT, "parelagmc_amd", "lib", os.environ["HYB_LIB"])
from parelagmc_amd.fe import build_hierarchy, build_hybrid_sampler_problem, build_sampler_problem, mesh_from_json  # noqa: E402

nref = int(sys.argv[1]) if len(sys.argv) > 1 else 5
which = sys.argv[2] if len(sys.argv) > 2 else "both"
lanes = [int(a) for a in (sys.argv[3] if len(sys.argv) > 3 else "4,1").split(",")]
nbw = int(sys.argv[4]) if len(sys.argv) > 4 else 32      # realizations per plugin call and lane
h = build_hierarchy(mesh_from_json(os.path.join(ROOT, "tests", "golden", "meshes", "cube_tet.json")), nref)
probs = {}
if which in ("both", "saddle"):
    probs["saddle"] = build_sampler_problem(h, corlen=0.1, n_mc_levels=1)
if which in ("both", "hybrid"):
    probs["hybrid"] = build_hybrid_sampler_problem(h, corlen=0.1, n_mc_levels=1)
for name, p in probs.items():
    for ns in lanes:
        r = bench.sampler_point(p, 0, 1234, nbw, ns, max(5, 20 * 32 // nbw), name, name, roofline=False)
        print(f"{name} lanes {ns} x {nbw}: {r['value']:.1f} samples/s, iterations {r['mean_minres_iterations']:.1f}", flush=True)
