"""bench.py command line without a GPU: the rank launcher of `--gpus N` propagates a failing rank's exit code, and a launcher
environment that contradicts `--gpus` is refused before anything heavy happens."""
import os
import subprocess
import sys

from conftest import ROOT


def _run(args, env_extra=None, drop=()):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT") + tuple(drop):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=600)


def test_spawned_ranks_propagate_failure():
    """No GPU here: both ranks die in torch.cuda.set_device; the parent must return non-zero and print no JSON line
    (the reference's mpirun does the same for a failing rank)."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("needs a box without a GPU")
    r = _run(["--gpus", "2", "--refine", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-r6", "--no-mlmc"])
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "No HIP GPUs" in r.stderr or "no HIP device" in r.stderr or "RuntimeError" in r.stderr


def test_launcher_world_size_must_match_gpus():
    r = _run(["--gpus", "2"], env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr


def _canned_full_record():
    """a complete record as bench.py assembled it in round 4 (21.7 kB as ONE line: the driver kept only its tail)"""
    import json
    with open(os.path.join(ROOT, "profiles", "r04_bench_output.log")) as f:
        return json.loads([ln for ln in f.read().splitlines() if ln.startswith("{")][-1])


REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


def test_compact_line_fits_the_drivers_window_and_keeps_the_contract():
    import json
    sys.path.insert(0, ROOT)
    import bench
    full = _canned_full_record()
    assert len(json.dumps(full)) > 20000
    line = bench.compact_line(full)
    assert "\n" not in line and len(line) < bench.LINE_CAP <= 4096
    d = json.loads(line)
    for k in REQUIRED:
        assert k in d, k
    assert d["value"] == float(f"{full['value']:.6g}") and d["metric"] == full["metric"]
    assert len(d["config"]["workload"]) <= 200 and len(d["roofline"]["kernel"]) <= 80
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "bytes_per_launch", "avg_kernel_ms", "operator_frac",
              "saddle_k5_frac", "solver_frac"):
        assert k in d["roofline"], k
    assert abs(d["roofline"]["frac"] - full["roofline"]["frac"]) < 1e-3
    assert abs(d["roofline"]["saddle_k5_frac"] - full["extra"]["saddle_point_minres"]["roofline"]["frac"]) < 1e-3
    for k in ("value", "cores", "kind", "unit", "sample"):
        assert k in d["cpu_baseline"], k
    # extra: scalars and flat lists of scalars only
    for k, v in d["extra"].items():
        assert not isinstance(v, dict), k
        if isinstance(v, list):
            assert all(not isinstance(x, (dict, list)) for x in v), k
    assert d["extra"]["c3_value"] > 0 and len(d["extra"]["c4_values"]) == 3 and len(d["extra"]["c5_values"]) == 4


def test_compact_line_with_eight_ranks_and_a_farm_block_stays_below_the_cap():
    import json
    sys.path.insert(0, ROOT)
    import bench
    full = _canned_full_record()
    full["n_gpus"] = 8
    full["ranks"] = [dict(full["ranks"][0], rank=r, device=r, cpu_affinity={"pinned": True, "numa_node": r // 4, "ncpus": 48})
                     for r in range(8)]
    full.pop("cpu_baseline")
    full["extra"] = {"mlmc_farm": dict(full["extra"]["mlmc_config3"], allreduce_ms=0.3, allreduces_in_round=1,
                                       allreduce_ms_per_rank=[0.3] * 8, collective="rccl (pmc_allreduce_sum_f64 of the "
                                       "library's communicator)", ranks_in_rccl_communicator=8)}
    line = bench.compact_line(full)
    assert len(line) < bench.LINE_CAP
    d = json.loads(line)
    assert d["n_gpus"] == 8 and d["rank_seconds"]["max"] >= d["rank_seconds"]["min"]
    assert d["extra"]["farm_allreduces_in_round"] == 1 and "ranks" not in d


def test_compact_line_survives_failed_secondary_figures():
    import json
    sys.path.insert(0, ROOT)
    import bench
    full = _canned_full_record()
    full["extra"] = {k: {"error": "RuntimeError('x' * 5000)" + "y" * 5000, "wall_s": 1.0} for k in full["extra"]}
    full["cpu_baseline"] = {"error": "z" * 9000}
    d = json.loads(bench.compact_line(full))
    assert "mlmc_config3" in d["extra"]["errors"] and d["value"] > 0


def test_compact_line_of_the_round5_record_carries_the_event_overhead_fields():
    """the full record of this round's gate run (profiles/r05_gate_bench_full.json, written before bench.py computed the net
    figure itself): raw frac is kept as it is, the net-of-event-overhead figure is passed through when the record has it"""
    import json
    sys.path.insert(0, ROOT)
    import bench
    with open(os.path.join(ROOT, "profiles", "r05_gate_bench_full.json")) as f:
        full = json.load(f)
    rf = full["roofline"]
    net = rf["bytes_per_launch"] / ((rf["avg_kernel_ms"] - rf["event_overhead_ms"]) * 1e-3) / 1e9 / bench.PEAK_GBS
    rf["frac_net_of_event_overhead"] = net
    line = bench.compact_line(full)
    assert len(line) < bench.LINE_CAP
    d = json.loads(line)
    for k in REQUIRED:
        assert k in d, k
    r = d["roofline"]
    assert abs(r["frac"] - rf["frac"]) < 1e-3 and abs(r["frac_net_of_event_overhead"] - net) < 1e-3
    assert r["frac"] < r["frac_net_of_event_overhead"] < r["frac"] * 1.06      # a 4-5 us bracket on a ~147 us launch
    assert 0.003 < r["event_overhead_ms"] < 0.008
    assert d["cpu_baseline"]["solver"].startswith("PCG") and d["config"]["solver"] == "hybridization"
