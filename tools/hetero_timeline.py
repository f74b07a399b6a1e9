#!/usr/bin/env python
"""Stage timeline of the tick inside bench.py's heterogeneous steady state (a random 10 % of the running tasks finishes per tick, every worker
its own free vector, ~140 worker classes per tick through k_block_solve).  Needs a GPU.   python tools/hetero_timeline.py [steps]"""
import ctypes as C
import dataclasses
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401

from hyperqueue_amd import abi, workloads
from hyperqueue_amd.tick import Tick

steps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 30
snap = workloads.make("c3")
W, R, Q = len(snap.worker_id), snap.n_resources, len(snap.requests)
need = np.zeros((Q, R), np.int64)
for q, variants in enumerate(snap.requests):
    for (r, _k, a) in variants[0]["entries"]:
        need[q, r] = int(a)
total = np.asarray(snap.worker_total, np.int64).reshape(W, R)
running = np.zeros((W, Q), np.int64)
rng = np.random.default_rng(0)
ts = Tick(abi.make_config(time_limit_s=5.0, flags=abi.HQTICK_FLAG_COMPACT_RECORDS | abi.HQTICK_FLAG_COMPACT_DELTA16), measure=True)
ts.set_kernel_timing("--timed" in sys.argv)
ts._lib.hqtick_timeline.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int]
ts.upload_ready(snap.task_id, snap.task_priority, snap.task_rq, sorted_=True)
rq_of = snap.task_rq.copy()
next_id = int(snap.task_id[-1]) + 1
n_staged, prev_free, rows, stats = 0, None, [], []
for step in range(steps + 4):
    free = total - running @ need
    assigned = [[(int(q), 0) for q in np.repeat(np.arange(Q), running[w])] for w in range(W)]
    cur = dataclasses.replace(snap, _keep=[], worker_free=free.astype(np.uint64), assigned=assigned, task_id=np.zeros(0, np.uint64), task_priority=np.zeros(0, np.uint64), task_rq=np.zeros(0, np.uint32))
    sc = cur.to_c()
    if n_staged:
        ts.ready_add_staged(n_staged)
    if prev_free is None:
        ts.cluster_upload(sc)
    else:
        changed = np.nonzero((free != prev_free).any(axis=1))[0].astype(np.uint32)
        ts.cluster_update_workers(changed, free[changed].astype(np.uint64))
    prev_free = free.copy()
    res = ts.tick_raw(sc, resident=True)
    buf = (C.c_double * 32)()
    kk = ts._lib.hqtick_timeline(ts._ctx, buf, 32)
    ks = ts.kernel_stats()
    if step >= 4:
        rows.append([buf[j] for j in range(kk)] + [res.t_total_us]); stats.append(ks)
    ts.ready_consume_last()
    n_cnt = int(res.n_counts)
    cw = np.ctypeslib.as_array(res.count_worker, shape=(n_cnt,)).astype(np.int64); cq = np.ctypeslib.as_array(res.count_rq, shape=(n_cnt,)).astype(np.int64); cv = np.ctypeslib.as_array(res.count_value, shape=(n_cnt,)).astype(np.int64)
    np.add.at(running, (cw, cq), cv)
    gone = abi.record_task_ids(res, W)
    idx = (gone & np.uint64(0xFFFFFFFF)).astype(np.int64) - 1
    n_staged = len(idx)
    v_id, v_prio, v_rq = ts.ready_add_stage(n_staged)
    new_rq = rq_of[idx]; v_rq[:] = new_rq; rq_of = np.concatenate([rq_of, new_rq])
    v_id[:] = np.arange(next_id, next_id + n_staged, dtype=np.uint64); next_id += n_staged
    v_prio[:] = snap.task_priority[0]
    running -= rng.binomial(running, 0.10)
m = np.median(np.asarray(rows), axis=0)
labels = ["phaseA", "batches", "solve", "keytables", "prefillplan", "k5tables", "pack", "C_enqueued", "C_synced", "assembled", "total"]
prev = 0.0
for l, v in zip(labels, m):
    print(f"{l:12s} at {v:8.1f} us  (+{v - prev:7.1f})")
    prev = v
med = lambda k: float(np.median([s[k] for s in stats]))
print({k: round(med(k), 2) for k in ("n_classes", "n_classes_device", "n_classes_host", "n_classes_memo", "solve_classify_us", "solve_blocks_us", "solve_decode_us", "block_solve_us", "level_hist_us", "select_us", "sweep_us", "scan_us", "other_us") if k in stats[0]})
