#!/usr/bin/env python
"""Stage timeline of the tick INSIDE the add -> tick -> consume loop of bench.py's steady_state block (needs a GPU): where the tick of a changing ready set
spends more than the repeated identical tick of the headline.   python tools/loop_timeline.py [steps]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401

from hyperqueue_amd import abi, workloads
from hyperqueue_amd.tick import Tick

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
snap = workloads.make("c3")
W = len(snap.worker_id)
cfg = abi.make_config(time_limit_s=5.0, flags=abi.HQTICK_FLAG_COMPACT_RECORDS | abi.HQTICK_FLAG_COMPACT_DELTA16)
ts = Tick(cfg, measure=True)
ts.upload_ready(snap.task_id, snap.task_priority, snap.task_rq, sorted_=True)
sc = snap.to_c()
ts.cluster_upload(sc)
if "--timed" not in sys.argv:
    ts.set_kernel_timing(False)
ts._lib.hqtick_timeline.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int]
rq_of = snap.task_rq.copy()
res = ts.tick_raw(sc, resident=True)
gone = abi.record_task_ids(res, W)
ts.ready_consume_last()
next_id = int(snap.task_id[-1]) + 1
rows, t_add, t_tick, t_cons = [], [], [], []
for i in range(n + 3):
    k = len(gone)
    new_rq = rq_of[(gone & np.uint64(0xFFFFFFFF)).astype(np.int64) - 1]
    rq_of = np.concatenate([rq_of, new_rq])
    if "--packed" in sys.argv:
        rq16 = new_rq.astype(np.uint16)
        a = time.perf_counter(); ts.ready_add_packed([(next_id, k)], [(int(snap.task_priority[0]), k)], rq16); next_id += k
    else:
        v_id, v_prio, v_rq = ts.ready_add_stage(k)
        v_id[:] = np.arange(next_id, next_id + k, dtype=np.uint64); next_id += k
        v_prio[:] = snap.task_priority[0]
        v_rq[:] = new_rq
        a = time.perf_counter(); ts.ready_add_staged(k)
    b = time.perf_counter(); res = ts.tick_raw(sc, resident=True)
    c = time.perf_counter(); ts.ready_consume_last(); torch.cuda.synchronize()
    d = time.perf_counter()
    buf = (C.c_double * 32)()
    kk = ts._lib.hqtick_timeline(ts._ctx, buf, 32)
    gone = abi.record_task_ids(res, W)
    if i >= 3:
        rows.append([buf[j] for j in range(kk)] + [res.t_total_us]); t_add.append(b - a); t_tick.append(c - b); t_cons.append(d - c)
m = np.median(np.asarray(rows), axis=0)
labels = ["phaseA", "batches", "solve", "keytables", "prefillplan", "k5tables", "pack", "C_enqueued", "C_synced", "assembled", "total"]
prev = 0.0
for l, v in zip(labels, m):
    print(f"{l:12s} at {v:8.1f} us  (+{v - prev:7.1f})")
    prev = v
print(f"add {1e6 * np.median(t_add):.1f} us, tick {1e6 * np.median(t_tick):.1f} us (python call), consume {1e6 * np.median(t_cons):.1f} us")
print({k: round(float(v), 2) for k, v in ts.kernel_stats().items() if v})
