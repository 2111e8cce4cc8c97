"""Does loading libpmc.so (dlopen of the HIP runtime, no HIP call) open /dev/kfd?  Decides whether bench.py's setup worker
processes may call the host-only pmc_hybrid_build without counting as processes that use the GPU."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def fds():
    out = []
    for f in os.listdir("/proc/self/fd"):
        try:
            out.append(os.readlink(f"/proc/self/fd/{f}"))
        except OSError:
            pass
    return [x for x in out if "kfd" in x or "dri" in x]


print("before:", fds())
from parelagmc_amd import capi
lib = capi.load_library()
print("after dlopen:", fds())
from parelagmc_amd.fe import box_mesh, build_hierarchy
h = build_hierarchy(box_mesh([4, 4, 4], [2, 2, 2], "hex"), 1)
H, G, z = capi.library_hybrid_builder(h.spaces[0], 100.0)
print("after pmc_hybrid_build:", fds(), H.nnz)
