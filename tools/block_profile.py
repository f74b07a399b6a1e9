#!/usr/bin/env python
"""Where k_block_solve spends its time: per-class stage timestamps of one steady-state tick (HQTICK_BLOCK_PROFILE=1)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
os.environ["HQTICK_BLOCK_PROFILE"] = "1"
from hyperqueue_amd import abi, workloads
from hyperqueue_amd.tick import Tick
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
snap = workloads.make_steady(name, seed=0)
t = Tick(abi.make_config(time_limit_s=5.0), measure=True)
t.upload_ready(snap.task_id, snap.task_priority, snap.task_rq)
sc = snap.to_c()
for _ in range(3):
    t.tick_raw(sc, resident=True)
n = C.c_uint32()
t._lib.hqtick_block_profile_last.restype = C.POINTER(C.c_uint64)
t._lib.hqtick_block_profile_last.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
p = t._lib.hqtick_block_profile_last(t._ctx, C.byref(n))
a = np.ctypeslib.as_array(p, shape=(n.value, 8)).astype(np.int64)
ts = a[:, :6] * 0.01  # us
t0 = ts[:, 0].min()
print("classes", n.value, "kernel span us", ts[:, 5].max() - t0, "start spread us", ts[:, 0].max() - t0)
names = ["build", "duals", "greedy", "phase1", "phase2"]
d = np.diff(ts, axis=1)
for i, nm in enumerate(names):
    print(f"{nm:8s} median {np.median(d[:, i]):7.2f} us  p99 {np.percentile(d[:, i], 99):7.2f}  max {d[:, i].max():7.2f}")
tot = ts[:, 5] - ts[:, 0]
print(f"per class total median {np.median(tot):.2f} p99 {np.percentile(tot, 99):.2f} max {tot.max():.2f}; phase-1 steps max {a[:, 6].max()}, pool max {a[:, 7].max()}")
print(t.kernel_stats())
order = np.argsort(-tot)
print("launch positions of the 20 slowest classes (0 = launched first):", order[:20].tolist())
print("their times us:", [round(float(tot[i]), 1) for i in order[:20]])
