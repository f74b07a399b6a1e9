#!/usr/bin/env python
"""How many worker classes make the k_block_solve launch worth it?  Placement stage (host wall clock) with the class blocks on the device vs
on the host, for clusters of 1..256 heterogeneous workers (c3 classes, steady state)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from hyperqueue_amd import abi, workloads
from hyperqueue_amd.tick import Tick

for nw in (1, 2, 4, 8, 16, 32, 64, 256):
    snap = workloads.make_steady("c3", seed=1, n_tasks=200_000, n_workers=nw)
    sc = snap.to_c()
    row = [f"workers {nw:4d}"]
    for label, env in (("device", "0"), ("host", str(1 << 30))):
        os.environ["HQTICK_BLOCK_MIN_CLASSES"] = env
        t = Tick(abi.make_config(time_limit_s=5.0))
        t.upload_ready(snap.task_id, snap.task_priority, snap.task_rq)
        solve, tot, ks = [], [], None
        for i in range(23):
            r = t.tick_raw(sc, resident=True)
            if i >= 3:
                solve.append(r.t_solve_us); tot.append(r.t_total_us); ks = t.kernel_stats()
        row.append(f"{label}: classes {int(ks['n_classes']):4d} placement {np.median(solve):7.1f} us tick {np.median(tot):7.1f} us kernel {ks['block_solve_us']:6.1f}")
        t.close()
    print(" | ".join(row))
