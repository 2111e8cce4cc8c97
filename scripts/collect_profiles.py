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
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"


def find(d, suffix):
    """newest match (gpurun merges new files next to those of earlier runs)"""
    hits = [os.path.join(root, f) for root, _, files in os.walk(os.path.join(G, d)) for f in files if f.endswith(suffix)]
    if not hits:
        raise FileNotFoundError(f"{d}/*{suffix}")
    return max(hits, key=os.path.getmtime)


for src, dst, what in (("prof_bench", "bench", "python bench.py --steps 20 --warmup 5 --inline-setup"),
                       ("prof_s1", "bench_s1", "python bench.py --streams 1 --steps 40 --no-cpu-baseline --no-extras"),
                       ("prof_s1_r6", "bench_s1_r6", "python bench.py --refine 6 --streams 1 --steps 8 --warmup 2 --no-cpu-baseline --no-extras"),
                       ("prof_c3", "c3_darcy_op", "python scripts/c3_darcy_op.py"),
                       ("prof_s1_onestream", "lab_s1_onestream",
                        "LABORATORY library (libpmc_lab.so, PMC_SPLIT_MIN=0: one lane on ONE stream, every kernel alone on the chip) "
                        "python bench.py --streams 1 --steps 20 --warmup 3 --no-cpu-baseline --no-extras")):
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
for refine, nvec in ((5, 595968), (6, 4743168)):
    fetch = per_kernel(find(f"pmc_fetch_r{refine}", "counter_collection.csv"))
    write = per_kernel(find(f"pmc_write_r{refine}", "counter_collection.csv"))
    for name, acc in (("fetch_size", fetch), ("write_size", write)):
        with open(os.path.join(P, f"{tag}_pmc_{name}_r{refine}.csv"), "w") as f:
            f.write("kernel,launches,mean_counter_value_KB\n")
            for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
                f.write(f"\"{k}\",{len(v)},{sum(v) / len(v):.3f}\n")
    # one lane alone on the GPU splits the Lanczos update into a u-row and an s-row launch (two streams): compare totals
    # (the bench runs the large levels 32 realizations per launch; older passes 16)
    nbw = 32 if any("lincomb3_kernel<32" in k for k in fetch) else 16
    lv = [x for k, vals in fetch.items() if f"lincomb3_kernel<{nbw}" in k for x in vals]
    split = len(set(round(x / 1024.0) for x in lv)) > 1 and max(lv) > 1.5 * min(lv)
    lf = sum(lv) / len(lv) * (2 if split else 1)
    read_kb = 3 * nvec * nbw * 8 / 1024.0
    for key, kern in ((f"r{refine}_nb{nbw}_inloop", f"sell_spmm_kernel<{nbw}, 0, 0, true, 1,"),
                      (f"r{refine}_nb{nbw}", f"sell_spmm_kernel<{nbw}, 0, 0, false, 2,"),
                      (f"r{refine}_nb1", "sell_spmm_kernel<1, 0, 0, false, 2,")):
        fr, n = mean(fetch, kern)
        wr, _ = mean(write, kern)
        if fr is None or wr is None:
            continue
        out[key] = {"kernel": f"pmc::{kern} ...> on A", "FETCH_SIZE_KB_raw": fr, "WRITE_SIZE_KB_raw": wr,
                    "hbm_bytes_per_launch": (2.0 * fr + wr) * 1024.0, "launches_averaged": n}
    out[f"r{refine}_correction"] = ("FETCH_SIZE x2 on gfx950; cross-check in the same pass on the flat lincomb3_kernel: raw "
                                    f"{lf:.0f} KB for {read_kb:.0f} KB actually read (ratio {read_kb / lf:.3f})")
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
out["command"] = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE -- python3 bench.py [--refine 6] --steps 2 --warmup 1 "
                  "--streams 1 --no-cpu-baseline --no-mlmc (separate passes, scripts/make_profiles.sh)")
json.dump(out, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
