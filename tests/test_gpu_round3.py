"""GPU tests added in round 3: `bench.py --gpus N` starting its own ranks, the real sample farm (device plugins, two
processes), super-batches (several column groups of 32 realizations per launch) against narrow batches and the oracle, the
Darcy operator's in-loop timing and the per-phase device timers.  Run with -m gpu on an MI355X; everything goes through the
C ABI."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(b)


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher (the driver's command) must run TWO ranks (the reference starts its
    ranks under mpirun, examples/MLMC.cpp:43-50).  Rehearsed on one GPU: both ranks share device 0 and torch.distributed
    falls back to gloo (RCCL refuses two ranks on one device, so extra.mlmc_farm may carry that error text)."""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--refine", "3", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline", "--no-r6"], env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak"
    ranks = out["ranks"]
    assert [x["rank"] for x in ranks] == [0, 1]
    # leap-frog split: rank r owns the global realization ids r, r + 2, ... - same count, disjoint ranges
    assert ranks[0]["samples"] == ranks[1]["samples"] == 2 * out["config"]["batch"] * out["config"]["streams"]
    assert ranks[0]["id_stride"] == 2 and ranks[0]["global_ids"][0] % 2 == 0 and ranks[1]["global_ids"][0] % 2 == 1
    assert abs(out["value"] - 2 * ranks[0]["samples"] / (out["ms_per_step"] * 1e-3 * out["steps"])) < 1e-6 * out["value"]
    assert "mlmc_farm" in out["extra"]
    assert "cpu_baseline" not in out and "r6" not in out["extra"]
