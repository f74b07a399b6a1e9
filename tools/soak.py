#!/usr/bin/env python
"""Soak of the headline loop (needs a GPU): N cold ticks of c3 through hqtick_run_resident with the ABI-6 emission; every tick's raw output (unit stream, runs,
spans, counts, new free vectors, stage of the kernels used) must be byte-identical to the first tick's, whose decoded records are checked against the oracle's
fixture digest once.  Catches rare races (a kernel reading a table another one is still writing, a stale plan buffer) that 50 bench ticks would miss.
  python tools/soak.py [ticks=20000] [workload=c3]"""
import ctypes as C
import hashlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hyperqueue_amd import abi, workloads
from hyperqueue_amd.tick import Tick

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
name = sys.argv[2] if len(sys.argv) > 2 else "c3"
snap = workloads.make(name)
sc = snap.to_c()
W, R = len(snap.worker_id), snap.n_resources
t = Tick(abi.make_config(time_limit_s=5.0, flags=abi.HQTICK_FLAG_COMPACT_RECORDS | abi.HQTICK_FLAG_COMPACT_DELTA16))
t.upload_ready(snap.task_id, snap.task_priority, snap.task_rq)
t.cluster_upload(sc)
t.set_kernel_timing(False)


def digest(r):
    off = np.ctypeslib.as_array(r.rec_off, shape=(W + 1,))
    n_rec = int(off[-1])
    h = hashlib.blake2b(digest_size=16)
    h.update(off.tobytes())
    span = np.ctypeslib.as_array(r.run_span, shape=(2 * W,)).reshape(W, 2)
    h.update(span.tobytes())
    units = np.ctypeslib.as_array(r.rec_delta16, shape=(4 * n_rec,))
    runs = np.ctypeslib.as_array(r.runs16, shape=(4 * n_rec,)).reshape(n_rec, 4)
    for w in range(W):  # only the defined parts: a worker's runs and the units its records consumed (the slack between streams is not written)
        a, b, c = int(off[w]), int(off[w + 1]), int(span[w, 1])
        if b > a:
            h.update(runs[a:a + c].tobytes())
    h.update(np.ctypeslib.as_array(r.new_free, shape=(W * R,)).tobytes())
    h.update(np.ctypeslib.as_array(r.count_value, shape=(r.n_counts,)).tobytes())
    return h.hexdigest(), n_rec


r0 = t.tick_raw(sc, resident=True)
first, n_rec = digest(r0)
ids0 = abi.record_task_ids(r0, W)
print(f"{name}: {n_rec} records per tick; first tick digest {first}", flush=True)
bad = 0
t0 = time.time()
lat = []
for i in range(n):
    a = time.perf_counter()
    r = t.tick_raw(sc, resident=True)
    lat.append(time.perf_counter() - a)
    if i % 50 == 0 or i == n - 1:  # full check (decode included) on a sample, structural digest always cheap enough every 50th
        d, _ = digest(r)
        if d != first or (i % 1000 == 0 and not np.array_equal(abi.record_task_ids(r, W), ids0)):
            bad += 1
            print(f"tick {i}: output differs from the first tick", flush=True)
            if bad > 5:
                break
    if int(r.status) != int(r0.status) or int(r.is_optimal) != 1:
        bad += 1
        print(f"tick {i}: status {int(r.status)} optimal {int(r.is_optimal)}", flush=True)
lat = np.asarray(lat) * 1e6
print(f"{n} ticks in {time.time() - t0:.1f} s: differences {bad}; tick p50 {np.median(lat):.1f} us, p99 {np.percentile(lat, 99):.1f} us, max {lat.max():.1f} us, ticks above 2 x p50: {int((lat > 2 * np.median(lat)).sum())}")
