"""The Darcy operator of BASELINE config 3 as the MINRES loop launches it, one lane alone on the GPU: prints the roofline
block bench.py emits under extra.mlmc_config3.roofline.  Run under rocprofv3 by scripts/make_profiles.sh for the kernel-stats
row and the FETCH_SIZE / WRITE_SIZE passes of eg_pair_spmm_kernel<16, true, true>."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import darcy_operator_roofline  # noqa: E402
from parelagmc_amd import capi  # noqa: E402
from parelagmc_amd.fe import box_mesh, build_darcy_problem, build_hierarchy, build_sampler_problem  # noqa: E402

h = build_hierarchy(box_mesh([4, 4, 4], [2, 2, 2], "hex"), 4)
sp = build_sampler_problem(h, corlen=0.1, lognormal=True, n_mc_levels=3)
dp = build_darcy_problem(h, [0, 1, 1, 1, 1, 0], [1, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 1], n_mc_levels=3)
ctx = capi.Context(0, seed=20261003)
ts = int(os.environ.get("C3_TWO_STREAMS", "0"))        # 2 = one stream only: per-kernel times without overlap (profiling aid)
opts = capi.solver_opts(two_streams=ts) if ts else None
smp, ds = capi.PDESampler(ctx, sp, opts), capi.DarcySolver(ctx, dp, opts)
print(json.dumps(darcy_operator_roofline(ctx, smp, ds, 0, 16)))
ds.close()
smp.close()
ctx.close()
