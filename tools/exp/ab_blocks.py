#!/usr/bin/env python
"""A/B of k_block_solve on steady-state ticks: python tools/exp/ab_blocks.py [old]   (old: the library under tools/exp/bin/old/)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
from hyperqueue_amd import tick as tk
if "old" in sys.argv:
    tk.LIB_PATH = os.path.join(ROOT, "tools", "exp", "bin", "old", "libhqtick.so")
from hyperqueue_amd import abi, workloads
from hyperqueue_amd.tick import Tick
for name, kw in (("c3", dict(seed=0)), ("c4", dict(seed=0)), ("c3", dict(seed=0, n_workers=256, n_tasks=250_000))):
    snap = workloads.make_steady(name, **kw)
    t = Tick(abi.make_config(time_limit_s=5.0))
    t.upload_ready(snap.task_id, snap.task_priority, snap.task_rq)
    sc = snap.to_c()
    us, tot, ncls = [], [], 0
    for i in range(12):
        r = t.tick_raw(sc, resident=True)
        ks = t.kernel_stats()
        if i >= 3:
            us.append(ks["block_solve_us"]); tot.append(r.t_total_us); ncls = ks["n_classes_device"]
    print(f"{'old' if 'old' in sys.argv else 'new'} {name} {kw}: classes on the device {int(ncls)}, k_block_solve {np.median(us):.1f} us, tick {np.median(tot):.1f} us", flush=True)
    t.close()
