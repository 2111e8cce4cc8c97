"""The C-ABI libraries load and export every symbol include/*.h declares (no GPU needed)."""
import os
import re
import subprocess

import pytest

from conftest import ROOT
from parelagmc_amd import capi, host_api


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pmc_[A-Za-z0-9_]+)\s*\(", text)))


def _exported(lib):
    out = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
    return {ln.split()[-1] for ln in out.splitlines() if " T " in ln}


def test_libpmc_exports_every_declared_symbol():
    decl = _declared("pmc.h")
    assert len(decl) >= 30
    exp = _exported(capi.LIB_PATH)
    missing = [s for s in decl if s not in exp]
    assert not missing, missing
    assert sorted(capi.SYMBOLS) == decl            # the binding covers the header, nothing else
    capi.load_library()


def test_libpmc_host_exports_every_declared_symbol():
    decl = [s for s in _declared("pmc_host.h") if s not in ("pmc_reduce_fn", "pmc_cb_sample", "pmc_cb_eval", "pmc_cb_solve")]
    exp = _exported(host_api.HOST_LIB_PATH)
    missing = [s for s in decl if s not in exp]
    assert not missing, missing
    assert sorted(host_api.HOST_SYMBOLS) == sorted(decl)
    host_api.load_host_library()


def test_defaults_restate_reference_solver_settings():
    o = capi.solver_opts()
    # MINRES 300 / 1e-6 / 1e-12 (examples/example_helpers/CreateSamplerParameterList.hpp:54-66)
    assert (o.max_iter, o.rel_tol, o.abs_tol) == (300, 1e-6, 1e-12)


def test_no_cpu_fallback():
    """Without a GPU the product path must fail loudly, never fall back to the oracle."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.PmcError) as e:
        capi.Context(0)
    assert e.value.code == -2
    src = open(os.path.join(ROOT, "parelagmc_amd", "capi.py")).read() + open(os.path.join(ROOT, "parelagmc_amd", "host_api.py")).read()
    assert "oracle" not in src.replace("never fall back to the oracle", "")
