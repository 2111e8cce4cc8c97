"""Copies the rocprofv3 summaries produced by scripts/make_profiles.sh (gpurun_out/) into profiles/ and recomputes
profiles/pmc_traffic.json: HBM bytes per launch of the block operator K5 from FETCH_SIZE / WRITE_SIZE, corrected as
MI355X_MICROARCH.md (HBM section) prescribes for gfx950 (FETCH_SIZE counts half the bytes of wide coalesced reads; the
factor is cross-checked in the same pass on the flat lincomb3 kernel, whose byte count is known exactly)."""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"


def find(d, suffix):
    """newest match (gpurun merges new files next to those of earlier runs)"""
    hits = [os.path.join(root, f) for root, _, files in os.walk(os.path.join(G, d)) for f in files if f.endswith(suffix)]
    if not hits:
        raise FileNotFoundError(f"{d}/*{suffix}")
    return max(hits, key=os.path.getmtime)


for src, dst, what in (("prof_bench", "bench", "python bench.py --steps 20 --warmup 5 --inline-setup"),
                       ("prof_l4", "bench_l4", "python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras"),
                       ("prof_s1", "bench_s1", "python bench.py --streams 1 --steps 40 --no-cpu-baseline --no-extras"),
                       ("prof_s1_saddle", "bench_s1_saddle", "python bench.py --solver saddle --streams 1 --steps 40 --no-cpu-baseline --no-extras"),
                       ("prof_s1_r6", "bench_s1_r6", "python bench.py --refine 6 --streams 1 --steps 8 --warmup 2 --no-cpu-baseline --no-extras"),
                       ("prof_s1_r6_saddle", "bench_s1_r6_saddle", "python bench.py --solver saddle --refine 6 --streams 1 --steps 8 --warmup 2 --no-cpu-baseline --no-extras"),
                       ("prof_c3", "c3_darcy_op", "python scripts/c3_darcy_op.py"),
                       ("prof_s1_onestream", "lab_s1_onestream",
                        "LABORATORY library (libpmc_lab.so, PMC_SPLIT_MIN=0: one lane on ONE stream, every kernel alone on the chip) "
                        "python bench.py --solver saddle --streams 1 --steps 20 --warmup 3 --no-cpu-baseline --no-extras")):
    try:
        stats = find(src, "kernel_stats.csv")
    except FileNotFoundError:
        print("no", src)
        continue
    shutil.copy(stats, os.path.join(P, f"{tag}_{dst}_kernel_stats.csv"))
    with open(os.path.join(G, f"{src}.log")) as f:
        lines = [ln for ln in f if ln.startswith("{")]
    with open(os.path.join(P, f"{tag}_{dst}_output.log"), "w") as f:
        f.write(f"# {what} under rocprofv3 --kernel-trace --stats (scripts/make_profiles.sh)\n" + lines[-1])


# per (kernel, grid) rows and the per-kernel roofline table (scripts/by_grid.py, scripts/roofline_table.py on the GPU box)
for src, dst in (("prof_s1_by_grid.csv", "bench_s1_by_grid.csv"), ("prof_l4_by_grid.csv", "bench_l4_by_grid.csv"),
                 ("roofline_table.json", "roofline_table.json"), ("roofline_table.txt", "roofline_table.txt")):
    if os.path.exists(os.path.join(G, src)):
        shutil.copy(os.path.join(G, src), os.path.join(P, f"{tag}_{dst}"))


def per_kernel(path):
    acc = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


def mean(d, key):
    v = [x for k, vals in d.items() if key in k for x in vals]
    return (sum(v) / len(v), len(v)) if v else (None, 0)


out = {}


def dump(name, acc, suffix):
    with open(os.path.join(P, f"{tag}_pmc_{name}_{suffix}.csv"), "w") as f:
        f.write("kernel,launches,mean_counter_value_KB\n")
        for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
            f.write(f"\"{k}\",{len(v)},{sum(v) / len(v):.3f}\n")


def entry(fetch, write, kern, what):
    fr, n = mean(fetch, kern)
    wr, _ = mean(write, kern)
    if fr is None or wr is None:
        return None
    return {"kernel": f"pmc::{kern} ...> {what}", "FETCH_SIZE_KB_raw": fr, "WRITE_SIZE_KB_raw": wr,
            "hbm_bytes_per_launch": (2.0 * fr + wr) * 1024.0, "launches_averaged": n}


def batch_of(logname, default=32):
    """realizations per launch of a bench pass (config.batch of its JSON line): the traffic keys carry it"""
    try:
        with open(os.path.join(G, logname)) as f:
            lines = [ln for ln in f if ln.startswith("{")]
        return int(json.loads(lines[-1])["config"]["batch"])
    except Exception:   # noqa: BLE001
        return default


for refine, nvec, nlam in ((5, 595968, 399360), (6, 4743168, 3170304)):
    # saddle-point passes (--solver saddle): K5 on A in the loop / isolated / one column; the flat lincomb3 kernel as the
    # cross-check of the FETCH_SIZE correction
    try:
        fetch = per_kernel(find(f"pmc_fetch_r{refine}s", "counter_collection.csv"))
        write = per_kernel(find(f"pmc_write_r{refine}s", "counter_collection.csv"))
    except FileNotFoundError as e:
        print("no saddle-point counter passes:", e)
        fetch = write = None
    if fetch is not None:
        dump("fetch_size", fetch, f"r{refine}_saddle")
        dump("write_size", write, f"r{refine}_saddle")
        # one lane alone on the GPU splits the Lanczos update into a u-row and an s-row launch (two streams): compare totals
        nbw = batch_of(f"pmc_fetch_r{refine}s.log")      # launches wider than 32 are column groups of the <32, ...> kernels
        lv = [x for k, vals in fetch.items() if "lincomb3_kernel<32" in k for x in vals]
        split = len(set(round(x / 1024.0) for x in lv)) > 1 and max(lv) > 1.5 * min(lv)
        lf = sum(lv) / len(lv) * (2 if split else 1)
        read_kb = 3 * nvec * nbw * 8 / 1024.0
        for key, kern in ((f"r{refine}_nb{nbw}_inloop", "sell_spmm_kernel<32, 0, 0, true, 1,"),
                          (f"r{refine}_nb{nbw}", "sell_spmm_kernel<32, 0, 0, false, 2,"),
                          (f"r{refine}_nb1", "sell_spmm_kernel<1, 0, 0, false, 2,")):
            e = entry(fetch, write, kern, "on A")
            if e:
                out[key] = e
        out[f"r{refine}_correction"] = ("FETCH_SIZE x2 on gfx950; cross-check in the same pass on the flat lincomb3_kernel: raw "
                                        f"{lf:.0f} KB for {read_kb:.0f} KB actually read (ratio {read_kb / lf:.3f})")
    # hybridized passes (the default solver of bench.py): post-smoothing of the finest V-cycle level, K5 on H
    try:
        fetch = per_kernel(find(f"pmc_fetch_r{refine}", "counter_collection.csv"))
        write = per_kernel(find(f"pmc_write_r{refine}", "counter_collection.csv"))
    except FileNotFoundError as e:
        print("no hybrid counter passes:", e)
        continue
    dump("fetch_size", fetch, f"r{refine}")
    dump("write_size", write, f"r{refine}")
    nbh = batch_of(f"pmc_fetch_r{refine}.log")
    for key, kern, what in ((f"r{refine}_hyb_post_nb{nbh}_inloop", "vc_poly2_kernel<32, float, float, float, true, true, 0>",
                             "post-smoothing of the finest level of the multiplier V-cycle, in the MINRES loop"),
                            (f"r{refine}_hyb_k5_nb{nbh}_inloop", "sell_spmm_kernel<32, 0, 0, true, 1,", "K5 on H, in the MINRES loop")):
        e = entry(fetch, write, kern.rstrip(">") if kern.endswith(">") else kern, what)
        if e:
            out[key] = e
    lv = [x for k, vals in fetch.items() if "lincomb3_kernel<32" in k for x in vals]
    if lv:
        read_kb = 3 * nlam * nbh * 8 / 1024.0
        out[f"r{refine}_hyb_correction"] = (f"lincomb3_kernel in the hybrid pass: raw {sum(lv) / len(lv):.0f} KB for {read_kb:.0f} KB "
                                            f"actually read (ratio {read_kb / (sum(lv) / len(lv)):.3f})")
# Darcy operator of config 3 (u-rows [M(k) | B^T] x with the fused dot, in the MINRES loop)
try:
    fetch = per_kernel(find("pmc_fetch_c3", "counter_collection.csv"))
    write = per_kernel(find("pmc_write_c3", "counter_collection.csv"))
    for name, acc in (("fetch_size", fetch), ("write_size", write)):
        with open(os.path.join(P, f"{tag}_pmc_{name}_c3.csv"), "w") as f:
            f.write("kernel,launches,mean_counter_value_KB\n")
            for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
                f.write(f"\"{k}\",{len(v)},{sum(v) / len(v):.3f}\n")
    for key, kern, what in (("c3_eg_nb16_inloop", "eg_pair_spmm_kernel<16, true, true,", "u-rows of the Darcy operator, level 0 of config 3"),
                            ("c3_egpoly_nb16_inloop", "eg_poly2_kernel<16, true, true,", "M-block polynomial of the Darcy preconditioner, level 0 of config 3")):
        fr, n = mean(fetch, kern)
        wr, _ = mean(write, kern)
        if fr is None or wr is None:
            raise FileNotFoundError(f"no rows of {kern} in the config-3 counter passes")
        out[key] = {"kernel": f"pmc::{kern} ...> ({what})", "FETCH_SIZE_KB_raw": fr, "WRITE_SIZE_KB_raw": wr,
                    "hbm_bytes_per_launch": (2.0 * fr + wr) * 1024.0, "launches_averaged": n}
except FileNotFoundError as e:
    print("no config-3 Darcy operator passes:", e)
# provenance: the stamp scripts/make_profiles.sh wrote ON THE GPU BOX before its passes (the library and sources those
# passes ran); the commit is named only when the tree's sources at HEAD hash to the same value
import hashlib
import subprocess
stamp = json.load(open(os.path.join(G, "profile_stamp.json")))
out["libpmc_sha256"] = stamp["libpmc_sha256"]
out["csrc_sha256"] = stamp["csrc_sha256"]
out["stamp"] = stamp["where"]
try:
    sys.path.insert(0, ROOT)
    import bench
    head = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip()
    dirty = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "parelagmc_amd/csrc", "include"],
                           capture_output=True, text=True).stdout.strip()
    same = bench.csrc_sha256() == stamp["csrc_sha256"]
    out["head"] = (head if same and not dirty else None)
    out["head_note"] = ("sources at this commit hash to csrc_sha256" if same and not dirty else
                        "the tree's sources differ from the ones the passes ran, or are uncommitted: no commit named")
except Exception:   # noqa: BLE001
    out["head"] = None
out["command"] = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE -- python3 bench.py [--solver saddle] [--refine 6] "
                  "--steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-extras (separate passes, scripts/make_profiles.sh)")
json.dump(out, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
