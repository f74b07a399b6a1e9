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
if n.value <= 1024:  # launch order = class order = order of first appearance of the free vector among the workers
    free = np.asarray(snap.worker_free, np.uint64).reshape(len(snap.worker_id), -1)
    seen, cls_free = {}, []
    for row in free:
        k = tuple(int(x) for x in row)
        if k not in seen:
            seen[k] = len(cls_free); cls_free.append(k)
    if len(cls_free) == n.value:
        cf = np.asarray(cls_free, np.float64) / 10000.0
        print("free (cpu, gpu, mem) of the 12 slowest:", [tuple(round(float(v), 2) for v in cf[i]) for i in order[:12]])
        print("free of 8 typical (median time):", [tuple(round(float(v), 2) for v in cf[i]) for i in order[len(order) // 2: len(order) // 2 + 8]])
        for r in range(cf.shape[1]):
            print(f"corr(time, free[{r}]) = {np.corrcoef(tot, cf[:, r])[0, 1]:.3f}")
        steps = a[:, 6].astype(np.float64)
        print("corr(time, phase-1 steps) =", round(float(np.corrcoef(tot, steps)[0, 1]), 3))
        tot_row = np.asarray(snap.worker_total, np.float64).reshape(len(snap.worker_id), -1)[0] / 10000.0
        share = cf / tot_row
        for name_, key in (("min share", share.min(axis=1)), ("product of shares", share.prod(axis=1)), ("cpu share", share[:, 0]), ("sum of shares", share.sum(axis=1))):
            rank = np.argsort(-key, kind="stable")
            pos = {int(c): i for i, c in enumerate(rank)}
            print(f"predictor {name_}: ranks of the 20 slowest:", sorted(pos[int(i)] for i in order[:20]), " corr", round(float(np.corrcoef(tot, key)[0, 1]), 3))
