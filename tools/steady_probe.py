#!/usr/bin/env python
"""Steady-state tick (heterogeneous workers) on one MI355X: stage timeline with the class blocks on the device vs on the host.
  python tools/steady_probe.py [c3|c4] [ticks]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (one HIP runtime per process: torch's first)

from hyperqueue_amd import abi, workloads
from hyperqueue_amd.tick import Tick

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
snap = workloads.make_steady(name, seed=0)
sc = snap.to_c()
print("distinct free vectors:", len(np.unique(np.asarray(snap.worker_free), axis=0)), "of", len(snap.worker_id), "workers")
for label, env in (("device blocks", {}), ("host blocks", {"HQTICK_BLOCK_MIN_CLASSES": str(1 << 30)})):
    os.environ.pop("HQTICK_BLOCK_MIN_CLASSES", None)
    os.environ.update(env)
    t = Tick(abi.make_config(time_limit_s=5.0), measure=True)
    t.upload_ready(snap.task_id, snap.task_priority, snap.task_rq)
    t._lib.hqtick_timeline.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int]
    rows, ks = [], []
    for i in range(n + 3):
        r = t.tick_raw(sc, resident=True)
        buf = (C.c_double * 32)()
        k = t._lib.hqtick_timeline(t._ctx, buf, 32)
        if i >= 3:
            rows.append([buf[j] for j in range(k)] + [r.t_total_us])
            ks.append(t.kernel_stats())
    m = np.median(np.asarray(rows), axis=0)
    labels = ["phaseA", "batches", "solve", "keytables", "prefillplan", "k5tables", "pack", "C_enqueued", "C_synced", "assembled", "total"]
    prev = 0.0
    print(f"--- {label}: status {int(r.status)} optimal {int(r.is_optimal)} canonical {int(r.is_canonical)}")
    for l, v in zip(labels, m):
        print(f"{l:12s} at {v:8.1f} us  (+{v - prev:7.1f})")
        prev = v
    print({k: round(float(np.median([s[k] for s in ks])), 2) for k in ks[0]})
    t.close()
