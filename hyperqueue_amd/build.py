"""Builds libhqtick.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc.  No JIT cache: the .so travels with the repo snapshot."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libhqtick.so")
SOURCES = ["hqtick.cpp", "host_model.cpp", "milp.cpp", "debug_capi.cpp", "kernels.hip", "graph.hip"]
HEADERS = ["kernels.h", "graph.h", "devbuf.h", "host_model.h", "milp.h", "hb_order.h", os.path.join("..", "..", "include", "hqtick.h"), os.path.join("..", "..", "include", "hqtick_debug.h")]


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def build(force: bool = False, verbose: bool = False) -> str:
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
