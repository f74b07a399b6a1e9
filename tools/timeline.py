#!/usr/bin/env python
"""Prints the median internal stage timeline of hqtick_run_resident (bench tooling; needs a GPU)."""
import ctypes as C
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hyperqueue_amd import abi, workloads
from hyperqueue_amd.tick import Tick

args = [a for a in sys.argv[1:] if not a.startswith("--")]
name = args[0] if args else "c3"
n = int(args[1]) if len(args) > 1 else 30
snap = workloads.make(name)
t = Tick(abi.make_config(time_limit_s=5.0, flags=abi.HQTICK_FLAG_COMPACT_RECORDS | (0 if "--u32" in sys.argv else abi.HQTICK_FLAG_COMPACT_DELTA16)), measure=True)  # as bench.py runs it (measure: hqtick_timeline lives in the test library)
t.upload_ready(snap.task_id, snap.task_priority, snap.task_rq)
sc = snap.to_c()
t.cluster_upload(sc)
if "--timed" not in sys.argv:
    t.set_kernel_timing(False)
t._lib.hqtick_timeline.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int]
rows, ks = [], []
for i in range(n + 5):
    r = t.tick_raw(sc, resident=True)
    buf = (C.c_double * 32)()
    k = t._lib.hqtick_timeline(t._ctx, buf, 32)
    if i >= 5:
        rows.append([buf[j] for j in range(k)] + [r.t_total_us])
        ks.append(t.kernel_stats())
m = np.median(np.asarray(rows), axis=0)
labels = ["phaseA", "batches", "solve", "keytables", "prefillplan", "k5tables", "pack", "C_enqueued", "C_synced", "assembled", "total"]
prev = 0.0
for l, v in zip(labels, m):
    print(f"{l:12s} at {v:8.1f} us  (+{v - prev:7.1f})")
    prev = v
print({k: round(float(np.median([s[k] for s in ks])), 2) for k in ks[0]})
print("status", int(r.status), "is_optimal", int(r.is_optimal), "assigned+prefilled records", int(r.n_records) if hasattr(r, "n_records") else "?")
