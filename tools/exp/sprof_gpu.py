#!/usr/bin/env python
"""Flat profile of the REAL tick's host side on the GPU box (tools/exp/sprof.c): python tools/exp/sprof_gpu.py [c3p|c4u|c4p|wave|c3_steady] [iterations]
The resident cold tick of bench.py's headline loop (ready set + cluster tables in HBM, HQTICK_FLAG_NO_TICK_CACHES) through libhqtick.so; the program counter is sampled
every 50 us of wall time while the library call runs.  Waiting for a kernel shows up as the function that spins (DeviceSweeper::wait_done, the stream synchronisations)."""
import bisect, collections, ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: F401
import price_probe
from hyperqueue_amd import abi
from hyperqueue_amd.tick import Tick

name = sys.argv[1] if len(sys.argv) > 1 else "c3p"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-o", "/tmp/libsprof.so", os.path.join(ROOT, "tools", "exp", "sprof.c")], check=True)
sp = C.CDLL("/tmp/libsprof.so")
sp.sprof_samples.restype = C.POINTER(C.c_uint64); sp.sprof_returns.restype = C.POINTER(C.c_uint64)
if name.endswith("_steady"):   # a cluster mid-run: every worker its own free vector, the class blocks through k_block_solve (workloads.make_steady)
    from hyperqueue_amd import workloads
    snap = workloads.make_steady(name[:-7])
else:
    snap = price_probe.snapshot(name)
t = Tick(abi.make_config(time_limit_s=5.0, flags=getattr(abi, "HQTICK_FLAG_NO_TICK_CACHES", 0)))
t.upload_ready(snap.task_id, snap.task_priority, snap.task_rq)
sc = snap.to_c()
t.cluster_upload(sc)
t.set_kernel_timing(False)
for _ in range(5):
    t.tick_raw(sc, resident=True)
PERIOD = 50
pcs, rets = [], []
import time
t0 = time.perf_counter()
for _ in range(iters):
    sp.sprof_start(PERIOD)
    t.tick_raw(sc, resident=True)
    k = sp.sprof_stop()
    pcs += [sp.sprof_samples()[i] for i in range(k)]
    rets += [sp.sprof_returns()[i] for i in range(k)]
wall = (time.perf_counter() - t0) / iters
libpath = os.path.join(ROOT, "hyperqueue_amd", "libhqtick.so")
base, text_end, others = None, 0, []
for line in open("/proc/self/maps"):
    f = line.split()
    lo, hi = (int(v, 16) for v in f[0].split("-"))
    if line.rstrip().endswith("libhqtick.so"):
        off = int(f[2], 16)
        if base is None or lo - off < base: base = lo - off
        if "x" in f[1]: text_end = max(text_end, hi)
    elif "x" in f[1] and len(f) >= 6:
        others.append((lo, hi, os.path.basename(f[5])))
syms = []
for line in subprocess.run(["nm", "-C", "--defined-only", "-n", libpath], capture_output=True, text=True).stdout.splitlines():
    parts = line.split(" ", 2)
    if len(parts) == 3 and parts[1] in "tTwW": syms.append((int(parts[0], 16), parts[2]))
addrs = [a for a, _ in syms]
count = collections.Counter()
for pc in pcs:
    if base is not None and base <= pc < text_end:
        i = bisect.bisect_right(addrs, pc - base) - 1
        count[syms[i][1] if i >= 0 else "?"] += 1
    else:
        count["(" + next((nm for lo, hi, nm in others if lo <= pc < hi), "?") + ")"] += 1
n = len(pcs)
print(f"{name}: {iters} ticks, {wall * 1e3:.3f} ms per tick incl. the python call, {n} samples ({n * PERIOD / 1e3 / iters:.3f} ms per tick sampled)")
for k, v in sorted(count.items(), key=lambda kv: -kv[1])[:40]:
    print(f"{100.0 * v / n:6.2f} %  {v * PERIOD / iters:7.1f} us  {k[:150]}")

if "--lines" in sys.argv:  # needs a library built with HQTICK_EXTRA_CXXFLAGS=-g (python hyperqueue_amd/build.py --force): samples by source line
    def by_line(counter, title):
        keys = list(counter)
        res = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-symbolizer", "--obj=" + libpath, "--inlines", "--no-demangle"], input="\n".join(hex(k) for k in keys) + "\n", capture_output=True, text=True).stdout
        lines = collections.Counter()
        for k, blockt in zip(keys, res.strip().split("\n\n")):
            own = [ln for ln in blockt.splitlines() if ln.startswith("/") and "/csrc/" in ln]
            if own:
                lines[":".join(own[0].split("/csrc/")[1].split(":")[:2])] += counter[k]
        print("\n" + title)
        for k, v in sorted(lines.items(), key=lambda kv: -kv[1])[:70]:
            print(f"{v * PERIOD / iters:7.1f} us  {k}")
    by_line(collections.Counter(pc - base for pc in pcs if base <= pc < text_end), "by source line (innermost frame under csrc/):")
    by_line(collections.Counter(rt - base for pc, rt in zip(pcs, rets) if not (base <= pc < text_end) and base <= rt < text_end), "library leaves (libc, HIP) by the line that called them:")
