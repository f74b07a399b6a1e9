#!/usr/bin/env python
"""Tick latency with FEW ready tasks on MANY workers (the everyday regime between bursts): stage times of the resident tick.  Needs a GPU."""
import sys, os, time
import numpy as np
import torch  # noqa
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hyperqueue_amd import abi, workloads
from hyperqueue_amd.tick import Tick

ids, prio, rq, off, dep = workloads.make_dag(200_000, seed=0)
for W in (64, 1024):
    for n_ready, ncls in ((96, 2), (96, 8), (2000, 2), (2000, 8), (20000, 2)):
        drv = workloads.DagChurn(n_workers=W, churn=0.1, seed=0)
        snap = drv.snapshot(ids[:n_ready], prio[:n_ready], (rq[:n_ready] % ncls).astype(np.uint32))
        t = Tick(abi.make_config(time_limit_s=5.0))
        rows = []
        for _ in range(6):
            t0 = time.perf_counter(); r = t.tick(snap); dt = time.perf_counter() - t0
            rows.append((dt * 1e3, r.times_us["scan"], r.times_us["batches"], r.times_us["solve"], r.times_us["mapping"], r.times_us["total"]))
        m = np.median(np.asarray(rows[1:]), axis=0)
        print(f"W={W:5d} ready={n_ready:6d} classes={ncls}: tick {m[0]:8.3f} ms | library us: scan {m[1]:8.1f} batches {m[2]:8.1f} solve {m[3]:10.1f} mapping {m[4]:8.1f} total {m[5]:10.1f} | optimal={int(r.is_optimal)} "
              f"assigned={sum(len(x) for x in r.records)}", flush=True)
        t.close()
