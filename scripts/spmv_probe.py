"""K5 SpMV / SpMM timing probe (development aid)."""
import sys
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from bench import build_problem  # noqa: E402
from parelagmc_amd import capi  # noqa: E402
nref = int(sys.argv[1]) if len(sys.argv) > 1 else 5
p = build_problem(nref)
ctx = capi.Context(0)
smp = capi.PDESampler(ctx, p)
L = p.levels[0]
nbs = [int(a) for a in sys.argv[2:]] or [1, 2, 4, 8, 16]
for nb in nbs:
    x = ctx.array(np.random.default_rng(0).standard_normal(nb * (L.n_u + L.n_s)))
    _, ms, b = smp.Mult(0, x, repeat=100)
    print(f"r={nref} nb={nb:2d}: {ms * 1e3:7.1f} us  {b / ms / 1e6:7.0f} GB/s  ({b / 1e6:.1f} MB)", flush=True)
