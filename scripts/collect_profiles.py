"""Copies the rocprofv3 summaries produced by scripts/make_profiles.sh (gpurun_out/) into profiles/ and recomputes
profiles/pmc_traffic.json (HBM bytes per K5 launch from FETCH_SIZE / WRITE_SIZE with the gfx950 correction)."""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"


def find(d, suffix):
    for root, _, files in os.walk(os.path.join(G, d)):
        for f in files:
            if f.endswith(suffix):
                return os.path.join(root, f)
    raise FileNotFoundError(f"{d}/*{suffix}")


shutil.copy(find("prof_bench", "kernel_stats.csv"), os.path.join(P, f"{tag}_bench_kernel_stats.csv"))
with open(os.path.join(G, "prof_bench.log")) as f:
    lines = [ln for ln in f if ln.startswith("{")]
with open(os.path.join(P, f"{tag}_bench_output.log"), "w") as f:
    f.write("# python bench.py under rocprofv3 --kernel-trace --stats (scripts/make_profiles.sh)\n" + lines[-1])


def per_kernel(path):
    acc = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


fetch, write = per_kernel(find("pmc_fetch", "counter_collection.csv")), per_kernel(find("pmc_write", "counter_collection.csv"))
for name, src in (("fetch_size", "pmc_fetch"), ("write_size", "pmc_write")):
    acc = per_kernel(find(src, "counter_collection.csv"))
    with open(os.path.join(P, f"{tag}_pmc_{name}.csv"), "w") as f:
        f.write("kernel,launches,mean_counter_value_KB\n")
        for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
            f.write(f"\"{k}\",{len(v)},{sum(v) / len(v):.3f}\n")


def mean(d, key):
    v = [x for k, vals in d.items() if key in k for x in vals]
    return sum(v) / len(v), len(v)


out = {}
for nb in (16, 1):
    key = f"sell_spmm_kernel<{nb}, false, 0, false, 2>"
    fr, n = mean(fetch, key)
    wr, _ = mean(write, key)
    out[f"r5_nb{nb}"] = {"kernel": f"pmc::{key} on A", "FETCH_SIZE_KB_raw": fr, "WRITE_SIZE_KB_raw": wr,
                         "hbm_bytes_per_launch": (2.0 * fr + wr) * 1024.0, "launches_averaged": n}
lf, _ = mean(fetch, "lincomb3_kernel<16>")
out["r5_nb16"]["correction"] = ("FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md HBM section); cross-check in the same pass on the flat "
                                f"lincomb3_kernel<16>: raw {lf:.0f} KB for 223488 KB actually read")
out["r5_nb16"]["command"] = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE -- python3 bench.py --steps 2 --warmup 1 "
                             "--streams 1 --no-cpu-baseline --no-mlmc (two separate passes, scripts/make_profiles.sh)")
json.dump(out, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
