import os, sys
sys.path.insert(0, os.getcwd())
import bench
for lanes in (4, 6, 8, 4, 8):
    r = bench.mlmc_config3(20261003, lanes=lanes)[0]
    print("lanes", lanes, round(r["realizations_per_s"], 1), [round(x * 1e3, 3) for x in r["seconds_per_sample_per_level"]], flush=True)
