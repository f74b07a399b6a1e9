"""Times libhqalloc.so (through ctypes) and the Python oracle on three allocation mixes of one worker; CPU only.

    python tools/alloc_bench.py

The allocator is a worker-side, latency-type component (one call per task start / end); the figure of interest is the
microseconds per `try_allocate`, dominated by the group model when a request is coupled.
"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from hyperqueue_amd import allocator as api  # noqa: E402
from oracle import alloc_oracle as ora  # noqa: E402


def mixes(mod):
    numa = [mod.regular_sockets(8, 16), mod.regular_sockets(8, 1)]  # 8 NUMA groups x 16 cores, one GPU each (BASELINE C4's worker)
    coupling = [(0, g, 1, g, 256) for g in range(8)]
    E = mod.Entry
    return [
        ("1 cpu, flat 128-core worker", mod.Descriptor([mod.simple_indices(128)]), [E(0, mod.COMPACT, 10_000)], 128),
        ("4 cpus + 0.5 gpu, 8 NUMA groups, coupled", mod.Descriptor(numa, coupling), [E(0, mod.COMPACT, 40_000), E(1, mod.COMPACT, 5_000)], 16),
        ("4 cpus compact! + 1 gpu compact!, coupled", mod.Descriptor(numa, coupling), [E(0, mod.FORCE_COMPACT, 40_000), E(1, mod.FORCE_COMPACT, 10_000)], 8),
    ]


def run(mod, desc, rq, fill, seconds):
    ac = mod.ResourceAllocator(desc)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        live = []
        for _ in range(fill):
            al = ac.try_allocate(rq)
            if al is None:
                break
            live.append(al)
        for al in live:
            ac.release_allocation(al)
        n += len(live)
    return (time.perf_counter() - t0) / max(1, n) * 1e6


def native():
    """tools/alloc_bench.cpp: the same mixes from C, without ctypes between the timer and the library."""
    import subprocess
    import tempfile

    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    lib = os.path.abspath(os.path.join(root, "hyperqueue_amd"))
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "alloc_bench")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(root, "tools", "alloc_bench.cpp"), "-L" + lib, "-lhqalloc", "-Wl,-rpath," + lib])
        print(subprocess.run([exe], capture_output=True, text=True, check=True).stdout, end="")


if __name__ == "__main__":
    if "--native" in sys.argv:
        native()
        sys.exit(0)
    for (name, d1, r1, fill), (_, d2, r2, _) in zip(mixes(api), mixes(ora)):
        a = run(api, d1, r1, fill, 1.0)
        o = run(ora, d2, r2, fill, 1.0)
        print(f"{name:48s} libhqalloc {a:8.1f} us per allocate+release (ctypes included)   oracle {o:10.1f} us")
