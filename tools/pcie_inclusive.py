#!/usr/bin/env python
"""hqtick_run (the whole ready set handed over as host buffers every tick) vs hqtick_run_resident on c3 (bench tooling; needs a GPU)."""
import os, sys, time
import numpy as np
import torch  # noqa
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hyperqueue_amd import abi, workloads
from hyperqueue_amd.tick import Tick
snap = workloads.make("c3")
t = Tick(abi.make_config(time_limit_s=5.0))
sc = snap.to_c()
for mode in ("host buffers every tick (hqtick_run)", "resident (hqtick_run_resident)"):
    res = mode.startswith("resident")
    if res:
        t.upload_ready(snap.task_id, snap.task_priority, snap.task_rq)
    lat = []
    for i in range(25):
        t0 = time.perf_counter(); t.tick_raw(sc, resident=res); lat.append(time.perf_counter() - t0)
    p50 = float(np.median(lat[5:]))
    print(f"{mode}: p50 {1e3 * p50:.3f} ms  -> {65536 / p50 / 1e6:.1f} M tasks assigned/s")
