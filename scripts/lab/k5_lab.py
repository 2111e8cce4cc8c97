#!/usr/bin/env python3
"""Exports the config-2 block operator to /tmp and runs build/k5_lab on it (kernel laboratory, not part of the product)."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def export(refine, path):
    p = bench.build_problem(refine)
    L = p.levels[0]
    M, B = L.M.tocsr(), L.B.tocsr()
    with open(path, "wb") as f:
        np.array([L.n_u, L.n_s, M.nnz, B.nnz], dtype=np.int32).tofile(f)
        for A in (M, B):
            A.indptr.astype(np.int32).tofile(f)
            A.indices.astype(np.int32).tofile(f)
            A.data.astype(np.float64).tofile(f)
        L.w_diag.astype(np.float64).tofile(f)
        np.array([p.alpha], dtype=np.float64).tofile(f)


if __name__ == "__main__":
    refine = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    path = f"/tmp/k5_r{refine}.bin"
    export(refine, path)
    sys.exit(subprocess.call([os.path.join(ROOT, "parelagmc_amd", "lib", "k5_lab"), path] + sys.argv[2:]))
