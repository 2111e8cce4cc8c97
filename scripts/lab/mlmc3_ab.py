"""config-3 MLMC round in the tree given by argv[1] (development aid for A/B runs)."""
import os, sys
root = os.path.abspath(sys.argv[1])
os.chdir(root)
sys.path.insert(0, root)
import bench
for i in range(4):
    b = int(os.environ.get('BATCHES', '32,16').split(',')[i % 2])
    r = bench.mlmc_config3(20261003, batch=b)
    r = r[0] if isinstance(r, tuple) else r
    print(root, 'batch', b, round(r["realizations_per_s"], 1), [round(x * 1e3, 3) for x in r["seconds_per_sample_per_level"]], flush=True)
