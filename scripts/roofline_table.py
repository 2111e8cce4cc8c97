"""Per-kernel roofline table of the headline's solver (hybridized, cube_tet r = 5, one lane, pmc_sampler_batch_width realizations per launch):
every kernel row holding at least 2 % of the device time of the one-lane profile, priced with its ALGORITHMIC bytes (the sizes
come from the library: pmc_sampler_vcycle_info, pmc_sampler_smoother_bytes) against its rocprofv3 average duration, with the
HBM traffic of the separate FETCH_SIZE / WRITE_SIZE passes beside it.  Runs on the GPU box at the end of
scripts/make_profiles.sh (it needs a sampler handle for the sizes); writes gpurun_out/roofline_table.json, which
scripts/collect_profiles.py copies to profiles/rNN_roofline_table.json.

  python3 scripts/roofline_table.py gpurun_out/prof_s1_by_grid.csv gpurun_out/pmc_fetch_r5_by_grid.csv gpurun_out/pmc_write_r5_by_grid.csv
"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from parelagmc_amd import capi  # noqa: E402

PEAK = 8000.0


def rows_of(path):
    return list(csv.DictReader(open(path))) if path and os.path.exists(path) else []


def main():
    args = [a for i, a in enumerate(sys.argv[1:]) if a != "--sizes-from" and sys.argv[i] != "--sizes-from"]
    stats, fetch, write = (rows_of(a) for a in (args[:3] + [None, None, None])[:3])
    prev = None
    if "--sizes-from" in sys.argv:     # re-price an existing table off the GPU box: the sizes the library reported are in it
        prev = json.load(open(sys.argv[sys.argv.index("--sizes-from") + 1]))
        lv, zb, nb = prev["vcycle_levels"], prev.get("z_bytes", 4), prev["realizations_per_launch"]
        smp = ctx = None
    else:
        hp = bench.build_hybrid_problem(5)
        ctx = capi.Context(0, seed=1)
        smp = capi.PDESampler(ctx, hp)
        lv = smp.vcycle_levels(0)
        zb = smp.z_bytes()
        nb = smp.BatchWidth(0)    # what `bench.py --streams 1` hands over per call (64 = two column groups of 32 per launch)
    ngroups = max(1, nb // 32)
    n = [l["rows"] for l in lv]
    V, F = 8.0 * nb, 4.0 * nb

    def level_of(grid):           # slice kernels: grid = rows rounded up to a multiple of 256 threads (x column groups when
        for i, r in enumerate(n):  # the profiler reports the total grid)
            if (r + 255) // 256 * 256 in (grid, grid // ngroups):
                return i
        return None

    def price(name, grid):
        """(label, algorithmic bytes) of one launch, or None for kernels outside the iteration"""
        l = level_of(grid)
        if "mg_tail_kernel" in name:
            t = next(i for i, x in enumerate(lv) if x["in_tail"])
            passes = 4.0          # pre-smoothing, residual, two products of the post-smoothing on every tail level but the last
            l2 = sum(12.0 * x["slots"] * passes for x in lv[t:-1]) + 12.0 * lv[-1]["slots"]
            return (f"LDS tail (levels of {[x['rows'] for x in lv[t:]]} rows, one workgroup per realization)",
                    2 * V * n[t], {"l2_to_cu_bytes_per_workgroup_estimate": l2, "l2_to_cu_bytes_per_launch_estimate": l2 * nb,
                                   "note": "every workgroup re-reads the tail levels' (index, value) pairs from L2 in each of its "
                                           "passes; the HBM bytes are the level's right-hand side in and correction out"})
        if "lincomb3_kernel" in name:
            return ("Lanczos update v = c0 q + c1 v1 + c2 v0 (+ the fp32 copy the V-cycle reads)", (4 * V + (F if zb == 4 else 0)) * n[0], None)
        if "minres_wx_deferred_kernel" in name:
            return ("w / x update, eight iterations per launch", (8 * zb * nb + 6 * V) * n[0], None)
        if l is None:
            return None
        nxt = n[l + 1] if l + 1 < len(n) else 0
        if "sell_spmm_kernel<32, 0, 0, true, 1," in name:
            return ("K5 on H with the fused <u, Hu>", 12.0 * lv[0]["nnz"] + 4.0 * n[0] + (zb * nb + V) * n[0], None)
        if "vc_poly2_kernel<32, double, float, float, false" in name:
            return (f"pre-smoothing, V-cycle level {l}", 12.0 * lv[l]["nnz"] + 12.0 * n[l] + (V + F) * n[l], None)
        if "vc_poly2_kernel<32, float, float, float, false" in name and l == 0:
            return ("pre-smoothing from the fp32 copy of r, V-cycle level 0", 12.0 * lv[0]["nnz"] + 12.0 * n[0] + 2 * F * n[0], None)
        if "vc_residual_kernel<32, float, double, float, false" in name or "vc_residual_kernel<32, float, float, float, false" in name:
            fused = bool(lv[l]["fused_restriction"])
            rin = F if "float, float, float" in name else V
            b = 12.0 * lv[l]["nnz"] + 4.0 * n[l] + (rin + 2 * F) * n[l] + ((8.0 + V) * nxt if fused else 0.0)
            return (f"residual{' + fused restriction' if fused else ''}{' (fp32 r)' if rin == F else ''}, V-cycle level {l}", b, None)
        if "vc_residual_kernel<32, double, float, float, false" in name:
            return (f"res - (S P) xc, V-cycle level {l}", 12.0 * lv[l]["sp_nnz"] + 4.0 * n[l] + 2 * F * n[l] + V * nxt, None)
        if "vc_poly2_kernel<32, float, " in name:     # (DOT variants on level 0, the level-1 post-smoothing: gathers fp32 residuals)
            out = zb * nb if l == 0 else V
            return (f"post-smoothing + coarse correction{' + fused <r, z>' if l == 0 else ''}, V-cycle level {l}",
                    12.0 * lv[l]["nnz"] + 12.0 * n[l] + (2 * F + V + out) * n[l] + V * nxt, None)
        return None

    def price_restriction(grid):
        # the separate restriction's grid follows its COARSE rows: P^T of level l has n[l + 1] rows
        for i in range(len(n) - 1):
            if (n[i + 1] + 255) // 256 * 256 in (grid, grid // ngroups):
                return (f"restriction P^T res (separate product), V-cycle level {i}", 12.0 * n[i] + 4.0 * n[i + 1] + F * n[i] + V * n[i + 1])
        return None

    def counter(rows, name, grid):
        # the counter passes report the TOTAL grid (x times the column groups), the kernel trace its x extent
        v = [float(r["mean_counter_value_KB"]) for r in rows if r["kernel"] == name and int(r["grid"]) in (grid, grid * ngroups)]
        return v[0] if v else None

    total = sum(float(r["total_ns"]) for r in stats)
    table = []
    for r in sorted(stats, key=lambda r: -float(r["total_ns"])):
        share = float(r["total_ns"]) / total
        if share < 0.02:
            continue
        name, grid = r["kernel"], int(r["grid"])
        pr = price(name, grid)
        extra = None
        if pr is None and "sell_spmm_kernel<32, 0, 0, false, 0, false, false, false, float" in name:
            q = price_restriction(grid)
            pr = (q[0], q[1], None) if q else None
        row = {"kernel": name[:110], "grid_threads": grid, "calls": int(r["calls"]), "avg_us": float(r["avg_ns"]) / 1e3,
               "share_of_device_time": share}
        if pr:
            label, b, extra = pr
            tb = b / (float(r["avg_ns"]) * 1e-9) / 1e12
            row.update({"what": label, "algorithmic_bytes": b, "TB_per_s": tb, "frac_of_8TBs": tb * 1e3 / PEAK})
            f, w = counter(fetch, name, grid), counter(write, name, grid)
            if f is not None and w is not None:
                hbm = (2.0 * f + w) * 1024.0          # FETCH_SIZE x 2 on gfx950 (MI355X_MICROARCH.md, HBM section)
                row.update({"hbm_bytes_per_launch": hbm, "traffic_over_algorithmic": hbm / b})
            if extra:
                row.update(extra)
        table.append(row)
    out = {"workload": f"python bench.py --streams 1 --steps 40 --no-cpu-baseline --no-extras (hybridized, cube_tet r = 5, one lane x {nb})",
           "realizations_per_launch": nb,
           "peak_GBs": PEAK, "vcycle_levels": lv, "z_bytes": zb,
           "libpmc_sha256": prev["libpmc_sha256"] if prev else bench.lib_sha256(),
           "csrc_sha256": prev["csrc_sha256"] if prev else bench.csrc_sha256(),
           "rows_with_at_least_2_percent_of_device_time": table}
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "roofline_table.json"), "w"), indent=1)
    for row in table:
        print(f"{row['share_of_device_time']:6.1%} {row['avg_us']:8.1f} us  {row.get('frac_of_8TBs', float('nan')):5.3f}  "
              f"{row.get('traffic_over_algorithmic', float('nan')):5.2f}x  {row.get('what', row['kernel'][:60])}")
    if smp is not None:
        smp.close()
        ctx.close()


if __name__ == "__main__":
    main()
