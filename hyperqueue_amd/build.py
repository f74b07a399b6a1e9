"""Builds libhqtick.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc.  No JIT cache: the .so travels with the repo snapshot."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libhqtick.so")
SOURCES = ["hqtick.cpp", "host_model.cpp", "milp.cpp", "debug_capi.cpp", "kernels.hip", "graph.hip", "wire.hip"]
HEADERS = ["kernels.h", "graph.h", "devbuf.h", "host_model.h", "milp.h", "hb_order.h", "wire_core.h", os.path.join("..", "..", "include", "hqwire.h"), os.path.join("..", "..", "include", "hqtick.h"), os.path.join("..", "..", "include", "hqtick_debug.h")]


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


ALLOC_LIB = os.path.join(HERE, "libhqalloc.so")
ALLOC_SOURCES = ["allocator.cpp", "milp.cpp"]
ALLOC_HEADERS = ["hb_table.h", "milp.h", os.path.join("..", "..", "include", "hqalloc.h")]


def build_alloc(force: bool = False, verbose: bool = False) -> str:
    """libhqalloc.so: the worker-side allocator (include/hqalloc.h).  Host-only on purpose -- worker nodes have no MI355X -- so
    it is compiled with g++ and has no HIP dependency."""
    srcs = [os.path.join(CSRC, s) for s in ALLOC_SOURCES]
    deps = srcs + [os.path.join(CSRC, h) for h in ALLOC_HEADERS]
    if not force and os.path.exists(ALLOC_LIB) and all(os.path.getmtime(ALLOC_LIB) >= os.path.getmtime(d) for d in deps):
        return ALLOC_LIB
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", ALLOC_LIB] + srcs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return ALLOC_LIB


def build(force: bool = False, verbose: bool = False) -> str:
    build_alloc(force, verbose)
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.join(CSRC, h) for h in HEADERS]
    if not force and os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in deps):
        return LIB
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-result", "-Wno-unused-value", "-o", LIB] + srcs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
