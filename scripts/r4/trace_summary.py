"""per (kernel, grid) totals of a rocprofv3 kernel trace (csv)"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    d = float(r['End_Timestamp']) - float(r['Start_Timestamp'])
    k = (r['Kernel_Name'][:int(sys.argv[2]) if len(sys.argv) > 2 else 84], r['Grid_Size_X'])
    agg[k][0] += 1
    agg[k][1] += d
tot = sum(v[1] for v in agg.values())
print("total ms %.2f" % (tot / 1e6))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 28]:
    print(f"{k[0]:84s} grid {k[1]:>9s} calls {v[0]:5d} avg us {v[1] / v[0] / 1e3:7.1f} total ms {v[1] / 1e6:6.1f}")
