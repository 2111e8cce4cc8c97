"""per (kernel, grid) summary of a rocprofv3 output directory: durations from *kernel_trace.csv (calls, avg_ns, total_ns) or,
for a --pmc pass, the mean counter value from *counter_collection.csv.  scripts/make_profiles.sh calls it before it deletes the
traces: profiles/ keeps kernel rows of different V-cycle levels (same kernel, different grid) apart."""
import collections
import csv
import glob
import sys

d, out = sys.argv[1], sys.argv[2]
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
with open(out, "w") as f:
    if cc:
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(cc[0])):
            acc[(r["Kernel_Name"], int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
        w = csv.writer(f)
        w.writerow(["kernel", "grid", "launches", "mean_counter_value_KB"])
        for (k, g), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([k, g, len(v), f"{sum(v) / len(v):.3f}"])
    elif kt:
        acc = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(kt[0])):
            a = acc[(r["Kernel_Name"], int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"]))]
            a[0] += 1
            a[1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        w = csv.writer(f)
        w.writerow(["kernel", "grid", "calls", "avg_ns", "total_ns"])
        for (k, g), (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
            w.writerow([k, g, n, f"{t / n:.1f}", f"{t:.0f}"])
