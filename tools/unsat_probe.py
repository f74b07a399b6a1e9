#!/usr/bin/env python
"""Unsaturated multi-class ticks (fewer ready tasks than the cluster holds: no batch is saturated, the batch-size rows couple all workers):
product vs the HiGHS oracle at the reference's 5 s limit — objective, proven optimality, time.  Needs a GPU."""
import sys, os, time
import numpy as np
import torch  # noqa
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hyperqueue_amd import abi, workloads
from hyperqueue_amd.tick import Tick
from oracle.oracle import Oracle

ids, prio, rq, off, dep = workloads.make_dag(1_000_000, seed=0)
src = np.nonzero((off[1:] - off[:-1]) == 0)[0]
ncls = int(os.environ.get("NCLS", "8"))
for W in [int(a) for a in sys.argv[1:]] or [4, 8, 16, 32, 64, 128, 1024]:
    for fill in (0.2, 0.45, 0.8):
        k = min(len(src), int(len(src) * W / 1024 * fill / 0.45))
        sel = src[:k]
        drv = workloads.DagChurn(n_workers=W, churn=0.1, seed=0)
        snap = drv.snapshot(ids[sel], prio[sel], (rq[sel] % ncls).astype(np.uint32))
        t = Tick(abi.make_config(time_limit_s=5.0))
        t0 = time.time(); g = t.tick(snap); tg = time.time() - t0
        o = Oracle(abi.make_config(time_limit_s=5.0))
        t0 = time.time(); w = o.tick(snap); to = time.time() - t0
        m = o.last_model()
        cd = g.counts_dict()
        x = np.zeros(len(m["obj"]))
        for j in range(len(x)):
            if m["ctype"][j] == 0:
                x[j] = cd.get((int(m["crq"][j]), int(m["cvariant"][j]), int(m["cworker"][j])), 0)
        mine = float(np.dot(m["obj"], x))
        print(f"W={W:5d} fill={fill:.2f} ready={k:6d} cols={len(m['obj']):5d} | product {tg:6.2f}s opt={int(g.is_optimal)} obj={mine:.7f} | HiGHS {to:6.2f}s opt={int(w.is_optimal)} obj={m['objective']:.7f}"
              f" | rel {(mine - m['objective']) / m['objective']:+.2e}", flush=True)
        t.close()
