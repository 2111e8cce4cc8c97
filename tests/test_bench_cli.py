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
