"""Development check run on the GPU box: HIP path vs oracle on small cases + first timings.
(The judged tests live in tests/; this script is a quick progress probe.)"""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from parelagmc_amd import capi  # noqa: E402
from parelagmc_amd.fe import (box_mesh, build_darcy_problem, build_hierarchy, build_sampler_problem,  # noqa: E402
                              kuhn_cube_tet)
from oracle.darcy_oracle import DarcyOracle  # noqa: E402
from oracle.rng_oracle import normal_fill  # noqa: E402
from oracle.sampler_oracle import SamplerOracle  # noqa: E402


def rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def main():
    ctx = capi.Context(0, seed=1234)
    # RNG
    x = ctx.normal_fill(1001, nbatch=3, first_id=5, stream=2)
    ref = np.stack([normal_fill(1001, 1234, 5 + b, 2) for b in range(3)])
    print("rng max abs diff", np.abs(x - ref).max(), flush=True)

    h = build_hierarchy(box_mesh([4, 4, 4], [2, 2, 2], "hex"), 2)
    sp = build_sampler_problem(h, corlen=0.1)
    so = SamplerOracle(sp)
    rng = np.random.Generator(np.random.PCG64(20261003))
    xi = rng.standard_normal((4, sp.levels[0].n_s))
    for tol, name in ((1e-12, "tight"), (1e-6, "default")):
        smp = capi.PDESampler(ctx, sp, capi.solver_opts(rel_tol=tol, abs_tol=1e-30, max_iter=300))
        for lvl in range(3):
            s, emb, st = smp.Eval(lvl, xi, xi_level=0, want_embed=True, return_stats=True)
            refs = np.stack([so.eval(lvl, 0, xi[b])[0] for b in range(4)])
            print(f"sampler {name} L{lvl} rel err {rel(s, refs):.3e} iters {[t[0] for t in st]} conv {[t[1] for t in st]}", flush=True)
        # warm start from coarse
        s1, e1 = smp.Eval(1, xi, xi_level=0, want_embed=True)
        s0, e0, st = smp.Eval(0, xi, xi_level=0, init_s=e1, init_level=1, use_init=True, want_embed=True, return_stats=True)
        refs = np.stack([so.eval(0, 0, xi[b])[0] for b in range(4)])
        print(f"sampler {name} warm-start rel err {rel(s0, refs):.3e} iters {[t[0] for t in st]}", flush=True)
        smp.close()

    dp = build_darcy_problem(h, [0, 1, 1, 1, 1, 0], [1, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 1])
    do = DarcyOracle(dp)
    ds = capi.DarcySolver(ctx, dp)
    for lvl in range(3):
        Q, Cc, st = ds.SolveFwd(lvl, np.ones((1, dp.levels[lvl].n_p)), return_stats=True)
        print(f"darcy k=1 L{lvl} Q={Q[0]:.12f} C={Cc[0]:.0f} iters {st[0][0]} conv {st[0][1]}", flush=True)
    spl = build_sampler_problem(h, corlen=0.1, lognormal=True)
    sol = SamplerOracle(spl)
    for lvl in range(3):
        k = np.stack([sol.eval(lvl, 0, xi[b])[0] for b in range(4)])
        Q, Cc, st = ds.SolveFwd(lvl, k, return_stats=True)
        Qr = np.array([do.solve_fwd(lvl, k[b])[0] for b in range(4)])
        print(f"darcy lognormal L{lvl} Q={Q} ref={Qr} relerr={np.abs(Q - Qr).max() / np.abs(Qr).max():.2e} iters {[t[0] for t in st]}", flush=True)
    ds.close()

    # timing: cube_tet r=5 (config 2)
    nref = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    t0 = time.time()
    ht = build_hierarchy(kuhn_cube_tet(), nref)
    spt = build_sampler_problem(ht, corlen=0.1, n_mc_levels=1)
    print("setup fe", time.time() - t0, [(L.n_s, L.n_u) for L in spt.levels], flush=True)
    t0 = time.time()
    smp = capi.PDESampler(ctx, spt)
    print("create", time.time() - t0, flush=True)
    n = spt.levels[0].n_s
    for nb in (1, 4, 16):
        xi_d = ctx.empty(nb * n)
        s_d = ctx.empty(nb * n)
        smp.Sample(0, first_id=0, nbatch=nb, out=xi_d)
        smp.Eval(0, xi_d, xi_level=0, s_out=s_d)
        ctx.synchronize()
        ctx.timer_start()
        reps = 3
        for _ in range(reps):
            out = smp.Eval(0, xi_d, xi_level=0, s_out=s_d, return_stats=True)
        ms = ctx.timer_stop()
        print(f"tet r={nref} nb={nb}: {ms / reps:.2f} ms/batch, {nb * reps / ms * 1e3:.1f} samples/s, iters {out[1][0][0]}", flush=True)
    if nref <= 4:
        so_t = SamplerOracle(spt)
        xi_h = smp.Sample(0, first_id=0, nbatch=1)
        s = smp.Eval(0, xi_h, xi_level=0)
        print("tet rel err vs direct", rel(s[0], so_t.eval(0, 0, xi_h[0])[0]))


if __name__ == "__main__":
    main()
