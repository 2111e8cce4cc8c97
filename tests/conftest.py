import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_path(*p):
    return os.path.join(GOLDEN, *p)


@pytest.fixture(scope="session")
def hex_hierarchy():
    """ctest default problem: 4x4x4 hexes on [0,2]^3, 2 refinements -> 16^3/8^3/4^3."""
    from parelagmc_amd.fe import box_mesh, build_hierarchy
    return build_hierarchy(box_mesh([4, 4, 4], [2, 2, 2], "hex"), 2)


@pytest.fixture(scope="session")
def hex_hierarchy_small():
    from parelagmc_amd.fe import box_mesh, build_hierarchy
    return build_hierarchy(box_mesh([4, 4, 4], [2, 2, 2], "hex"), 1)


@pytest.fixture(scope="session")
def seeded_rng():
    return np.random.Generator(np.random.PCG64(20261003))


@pytest.fixture(scope="session")
def gpu_ctx():
    from parelagmc_amd import capi
    ctx = capi.Context(0, seed=20261003)
    yield ctx
    ctx.close()
