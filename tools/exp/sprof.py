#!/usr/bin/env python
"""Flat profile of the tick's HOST stages on the CPU (tools/exp/sprof.c): python tools/exp/sprof.py [workload] [iterations] [--all]
Runs hqtick_debug_host_stages on the workload with the emulated price sweeps, samples the program counter every 200 us of CPU time and
prints the functions of libhqtick_test.so by share — without the emulated kernels' own functions (price_emul.cpp), which the GPU runs."""
import bisect
import collections
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from hyperqueue_amd import _testhooks, abi, workloads  # noqa: E402
from host_stages import HostStages  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "c3p"
iters = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 20
lib = _testhooks.load()
lib.hqtick_debug_set_price_emulation.argtypes = [C.c_int, C.c_uint32]
lib.hqtick_debug_set_price_emulation(1, 0)
if "--blocks" in sys.argv:  # class blocks through the emulated k_block_solve (its own functions are filtered out like the sweeps')
    lib.hqtick_debug_set_block_emulation.argtypes = [C.c_int, C.c_uint32]
    lib.hqtick_debug_set_block_emulation(1, 4096)
if name in ("wave", "0.2", "0.45", "c3ps"):  # tools/price_probe.py's coupled snapshots (the config-5 first wave, the unsaturated 1024-worker probes)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import price_probe
    snap = price_probe.snapshot(name)
elif name == "unsat":  # bench.py's config4_unsaturated: c4's cluster, fewer ready tasks than it could run — one coupled model of all 4096 workers
    snap = workloads.make("c4", seed=8, n_workers=4096, n_tasks=56_761)
else:
    snap = workloads.make_steady(name[:-7]) if name.endswith("_steady") else workloads.make(name)
import numpy as np  # noqa: E402
from host_stages import scan_outputs  # noqa: E402

hs = HostStages(abi.make_config(time_limit_s=20.0))
hs.stages(snap)
sp = C.CDLL(os.path.join(ROOT, "tools", "exp", "bin", "libsprof.so"))
sp.sprof_samples.restype = C.POINTER(C.c_uint64)
sp.sprof_returns.restype = C.POINTER(C.c_uint64)
sc = snap.to_c()
flags, tmc, levels, hist = scan_outputs(sc)  # (the python restatement of the scan kernels' outputs: once, outside the samples)
PERIOD = 100
pcs, rets = [], []
for _ in range(iters):
    out = abi.ResultC()
    sp.sprof_start(PERIOD)  # only the library call is sampled
    rc = hs.lib.hqtick_debug_host_stages(C.byref(hs.cfg), C.byref(sc), flags.ctypes.data_as(abi.u8p), tmc.ctypes.data_as(abi.u32p), len(levels), levels.ctypes.data_as(abi.u64p), hist.ctypes.data_as(abi.u32p), C.byref(out))
    k = sp.sprof_stop()
    assert rc >= 0
    pcs += [sp.sprof_samples()[i] for i in range(k)]
    rets += [sp.sprof_returns()[i] for i in range(k)]
n = len(pcs)
libpath = os.path.join(ROOT, "hyperqueue_amd", "libhqtick_test.so")
base, text_end, others = None, 0, []
for line in open("/proc/self/maps"):
    f = line.split()
    lo, hi = (int(v, 16) for v in f[0].split("-"))
    if line.rstrip().endswith("libhqtick_test.so"):
        off = int(f[2], 16)
        if base is None or lo - off < base:
            base = lo - off
        if "x" in f[1]:
            text_end = max(text_end, hi)
    elif "x" in f[1] and len(f) >= 6:
        others.append((lo, hi, os.path.basename(f[5])))
syms = []
for line in subprocess.run(["nm", "-C", "--defined-only", "-n", libpath], capture_output=True, text=True).stdout.splitlines():
    parts = line.split(" ", 2)
    if len(parts) == 3 and parts[1] in "tTwW":
        syms.append((int(parts[0], 16), parts[2]))
addrs = [a for a, _ in syms]
count = collections.Counter()
for pc in pcs:
    rel = pc - base
    if base <= pc < text_end:
        i = bisect.bisect_right(addrs, rel) - 1
        count[syms[i][1] if i >= 0 else "?"] += 1
    else:
        where = next((nm for lo, hi, nm in others if lo <= pc < hi), "?")
        count[f"(called code in {where})"] += 1
skip = lambda s: ("emul" in s.lower() or "hqblock::" in s or "solve_priced_block" in s or "HostWave" in s) and "--all" not in sys.argv
host = {k: v for k, v in count.items() if not skip(k)}
tot = sum(host.values())
print(f"{n} samples, {tot} outside the emulated kernels ({iters} iterations: {tot * PERIOD / 1e3 / iters:.2f} ms of host work per tick incl. the python driver)")
for k, v in sorted(host.items(), key=lambda kv: -kv[1])[:45]:
    print(f"{100.0 * v / tot:6.2f} %  {v * PERIOD / 1e3 / iters:7.3f} ms  {k[:170]}")

if "--lines" in sys.argv:  # needs a library built with HQTICK_EXTRA_CXXFLAGS=-g: samples by source line (the innermost frame under csrc/)
    rels = collections.Counter(pc - base for pc in pcs if base <= pc < text_end)
    callers = collections.Counter(rt - base for pc, rt in zip(pcs, rets) if not (base <= pc < text_end) and base <= rt < text_end)  # a leaf of libc / libm called from the library
    keys = list(rels)
    res = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-symbolizer", "--obj=" + libpath, "--inlines", "--no-demangle"], input="\n".join(hex(k) for k in keys) + "\n", capture_output=True, text=True).stdout
    lines = collections.Counter()
    for k, blockt in zip(keys, res.strip().split("\n\n")):
        frames = [ln for ln in blockt.splitlines() if ln.startswith("/")]
        own = [f for f in frames if "/csrc/" in f]
        if not own:
            continue
        f = own[0].split("/csrc/")[1]
        if any(x in f for x in ("price_emul", "block_core", "price_core", "dev_wave")):
            continue
        lines[":".join(f.split(":")[:2])] += rels[k]
    print("\nby source line (innermost frame under csrc/):")
    for k, v in sorted(lines.items(), key=lambda kv: -kv[1])[:90]:
        print(f"{v * PERIOD / 1e3 / iters:7.3f} ms  {k}")

    keys = list(callers)
    res = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-symbolizer", "--obj=" + libpath, "--inlines", "--no-demangle"], input="\n".join(hex(k) for k in keys) + "\n", capture_output=True, text=True).stdout
    lines = collections.Counter()
    for k, blockt in zip(keys, res.strip().split("\n\n")):
        own = [ln for ln in blockt.splitlines() if ln.startswith("/") and "/csrc/" in ln]
        if not own:
            continue
        f = own[0].split("/csrc/")[1]
        if any(x in f for x in ("price_emul", "block_core", "price_core", "dev_wave")):
            continue
        lines[":".join(f.split(":")[:2])] += callers[k]
    print("\nlibc / libm leaves by the line that called them (return address on top of the stack):")
    for k, v in sorted(lines.items(), key=lambda kv: -kv[1])[:40]:
        print(f"{v * PERIOD / 1e3 / iters:7.3f} ms  {k}")
