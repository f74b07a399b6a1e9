"""Builds the native pieces in-tree for gfx950 with hipcc.  No JIT cache: the .so files travel with the repo snapshot.

  libhqtick.so       the product: include/hqtick.h + include/hqwire.h, nothing else (HIP kernels + C ABI + host stages)
  libhqtick_test.so  the same objects plus the CPU test hooks of include/hqtick_debug.h (-DHQTICK_TEST_HOOKS): what the `-m "not gpu"`
                     tests load to exercise host logic and the kernels' phase functions without a GPU.  Nothing in the product loads it.
  libhqalloc.so      worker-side allocator (include/hqalloc.h), host-only g++
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_obj")
LIB = os.path.join(HERE, "libhqtick.so")
TEST_LIB = os.path.join(HERE, "libhqtick_test.so")
SOURCES = ["hqtick.cpp", "host_model.cpp", "milp.cpp", "price.cpp", "price_shard.cpp", "kernels.hip", "graph.hip", "wire.hip", "block_solve.hip", "price.hip"]
HOOKED = ["hqtick.cpp", "wire.hip"]          # sources that carry #ifdef HQTICK_TEST_HOOKS sections
TEST_ONLY = ["debug_capi.cpp", "price_emul.cpp"]               # sources of the test library only
HEADERS = ["kernels.h", "block_core.h", "price_core.h", "price.h", "price_emul.h", "price_dev.h", "lp_tab.h", "dev_wave.h", "block_solve.h", "graph.h", "devbuf.h", "host_model.h", "milp.h", "hb_order.h", "wire_core.h",
           os.path.join("..", "..", "include", "hqwire.h"), os.path.join("..", "..", "include", "hqtick.h"), os.path.join("..", "..", "include", "hqtick_debug.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-fvisibility-inlines-hidden", "-Wall", "-Wno-unused-result", "-Wno-unused-value",
         "-msse4.1"]  # host side: floor / round / nearbyint as one instruction instead of a call into libm (every x86-64 host of an MI355X has it; no FMA, so no result changes)
FLAGS += os.environ.get("HQTICK_EXTRA_CXXFLAGS", "").split()  # experiments (tools/exp/sprof.py --lines wants -g); a change of this variable needs --force
# The price sweeps choose among tied block optima by floating-point comparisons: no fused multiply-add contraction in the two places that run that
# arithmetic (the kernel and its CPU emulation), so that a GPU tick and the emulated tick of the CPU suite walk the same sequence of prices.
# (block_core.h's explicit fma() calls are the same operation on both sides.)
EXTRA_FLAGS = {"price.hip": ["-ffp-contract=off"], "price_emul.cpp": ["-ffp-contract=off"]}


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


ALLOC_LIB = os.path.join(HERE, "libhqalloc.so")
EXPORTS = os.path.join(CSRC, "exports.map")  # linker version script: only hqtick_* / hqwire_* / hqalloc_* leave the libraries (libstdc++'s instantiations carry default visibility of their own)
ALLOC_SOURCES = ["allocator.cpp", "milp.cpp", "price.cpp"]
ALLOC_HEADERS = ["hb_table.h", "milp.h", "lp_tab.h", "price.h", os.path.join("..", "..", "include", "hqalloc.h")]


def build_alloc(force: bool = False, verbose: bool = False) -> str:
    """libhqalloc.so: the worker-side allocator (include/hqalloc.h).  Host-only on purpose -- worker nodes have no MI355X -- so
    it is compiled with g++ and has no HIP dependency."""
    srcs = [os.path.join(CSRC, s) for s in ALLOC_SOURCES]
    deps = srcs + [os.path.join(CSRC, h) for h in ALLOC_HEADERS] + [EXPORTS]
    if not force and os.path.exists(ALLOC_LIB) and all(os.path.getmtime(ALLOC_LIB) >= os.path.getmtime(d) for d in deps):
        return ALLOC_LIB
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-fvisibility-inlines-hidden", "-shared", "-Wall", "-Wl,--version-script=" + EXPORTS, "-o", ALLOC_LIB] + srcs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return ALLOC_LIB


def _compile(src: str, hooks: bool, force: bool, verbose: bool) -> str:
    """one source -> one object under _obj/ (rebuilt when the source or any header is newer)"""
    os.makedirs(OBJ, exist_ok=True)
    obj = os.path.join(OBJ, src.replace(".", "_") + ("_hooks" if hooks else "") + ".o")
    deps = [os.path.join(CSRC, src)] + [os.path.join(CSRC, h) for h in HEADERS if os.path.exists(os.path.join(CSRC, h))]
    if not force and os.path.exists(obj) and all(os.path.getmtime(obj) >= os.path.getmtime(d) for d in deps):
        return obj
    cmd = [hipcc()] + FLAGS + EXTRA_FLAGS.get(src, []) + (["-DHQTICK_TEST_HOOKS=1"] if hooks else []) + ["-c", os.path.join(CSRC, src), "-o", obj]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return obj


def _link(lib: str, objs, verbose: bool) -> str:
    if os.path.exists(lib) and all(os.path.getmtime(lib) >= os.path.getmtime(o) for o in list(objs) + [EXPORTS]):
        return lib
    tmp = f"{lib}.{os.getpid()}.tmp"  # linked beside its place and moved there: a process that has the old library mapped keeps it, one that loads meanwhile never sees half a file
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + EXPORTS, "-o", tmp] + list(objs) + ["-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(tmp, lib)
    return lib


def build(force: bool = False, verbose: bool = False) -> str:
    """everything: libhqalloc.so, libhqtick.so (product), libhqtick_test.so (product objects + CPU test hooks)"""
    build_alloc(force, verbose)
    jobs = [(s, False) for s in SOURCES] + [(s, True) for s in HOOKED + TEST_ONLY]
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 2)) as ex:
        objs = dict(zip(jobs, ex.map(lambda j: _compile(j[0], j[1], force, verbose), jobs)))
    _link(LIB, [objs[(s, False)] for s in SOURCES], verbose)
    _link(TEST_LIB, [objs[(s, s in HOOKED)] for s in SOURCES] + [objs[(s, True)] for s in TEST_ONLY], verbose)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
