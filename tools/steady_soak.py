#!/usr/bin/env python
"""Soak of the steady loop at BASELINE size (needs a GPU): consume -> add (packed, fresh ids) -> tick, thousands of steps on the resident ready set — through the
appends behind the columns, the merges when their room is used up and the compactions when tombstones outnumber the live tasks.  Checked WITHOUT the oracle, by the
queue's own law: what a tick hands out of a request class are that class's LOWEST ids (take_tasks pops in id order within a priority level), as many as the first
tick handed out of it (the workers are empty again before every tick: every tick solves the same placement).
  python tools/steady_soak.py [steps=2000]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401

from hyperqueue_amd import abi, workloads
from hyperqueue_amd.tick import Tick

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
snap = workloads.make("c3")
W, Q = len(snap.worker_id), len(snap.requests)
sc = snap.to_c()
t = Tick(abi.make_config(time_limit_s=5.0, flags=abi.HQTICK_FLAG_COMPACT_RECORDS | abi.HQTICK_FLAG_COMPACT_DELTA16))
t.set_kernel_timing(False)
t.upload_ready(snap.task_id, snap.task_priority, snap.task_rq, sorted_=True)
t.cluster_upload(sc)
# per class: its ids in ascending order, as a growing array with a head index
ids_of, head, tail = [], [0] * Q, [0] * Q
for q in range(Q):
    mine = snap.task_id[snap.task_rq == q]
    buf = np.empty(2 * len(mine) + 1024, np.uint64)
    buf[: len(mine)] = mine; ids_of.append(buf); tail[q] = len(mine)
next_id = int(snap.task_id[-1]) + 1
per_class = None
t0 = time.time(); bad = 0; tick_us = []
for step in range(steps):
    a = time.perf_counter(); res = t.tick_raw(sc, resident=True); tick_us.append(1e6 * (time.perf_counter() - a))
    gone = np.sort(abi.record_task_ids(res, W))
    # which class each handed-out id belongs to: by membership in the expected prefixes
    taken = 0
    counts = []
    for q in range(Q):
        n_q = per_class[q] if per_class is not None else None
        if n_q is None:  # first tick: learn how many of each class a tick hands out
            cand = ids_of[q][head[q]: tail[q]]
            n_q = int(np.isin(cand[: 300_000], gone, assume_unique=True).sum())
        want = ids_of[q][head[q]: head[q] + n_q]
        ok = np.isin(want, gone, assume_unique=True).all()
        if not ok:
            bad += 1
            if bad <= 3:
                print(f"step {step} class {q}: the tick did not hand out the class's lowest {n_q} ids", flush=True)
        counts.append(n_q); taken += n_q
        head[q] += n_q
    if per_class is None:
        per_class = counts
    if taken != len(gone):
        bad += 1
        if bad <= 3:
            print(f"step {step}: {len(gone)} ids handed out, {taken} expected", flush=True)
    t.ready_consume_last()
    # arrivals: as many of each class as left, fresh consecutive ids, classes interleaved by a fixed pattern
    k = len(gone)
    new_rq = np.repeat(np.arange(Q, dtype=np.uint16), per_class)
    new_rq = new_rq[(np.arange(k) * 7919) % k] if k > 1 else new_rq
    new_ids = np.arange(next_id, next_id + k, dtype=np.uint64)
    t.ready_add_packed([(next_id, k)], [(int(snap.task_priority[0]), k)], new_rq)
    next_id += k
    for q in range(Q):
        mine = new_ids[new_rq == q]
        if tail[q] + len(mine) > len(ids_of[q]):  # drop the consumed prefix, double the room
            live = ids_of[q][head[q]: tail[q]]
            ids_of[q] = np.empty(2 * (len(live) + len(mine)) + 1024, np.uint64)
            ids_of[q][: len(live)] = live; tail[q] = len(live); head[q] = 0
        ids_of[q][tail[q]: tail[q] + len(mine)] = mine; tail[q] += len(mine)
    if t.ready_count() != sum(tail[q] - head[q] for q in range(Q)):
        bad += 1
        if bad <= 3:
            print(f"step {step}: ready_count {t.ready_count()} != mirror {sum(tail[q] - head[q] for q in range(Q))}", flush=True)
ks = t.kernel_stats()
tu = np.asarray(tick_us[5:])
print(f"{steps} steps in {time.time() - t0:.1f} s: problems {bad}; per step {sum(per_class)} tasks handed out ({per_class} per class); {int(ks['ready_appends'])} of {steps} add batches appended; "
      f"tick p50 {np.median(tu):.1f} us, p99 {np.percentile(tu, 99):.1f} us, max {tu.max():.1f} us")
