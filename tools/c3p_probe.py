#!/usr/bin/env python
"""How far does the coupled (multi-priority) path go?  c3p at growing worker counts: product vs plain-HiGHS oracle objective."""
import sys, os, time
import numpy as np
import torch  # noqa
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hyperqueue_amd import abi, workloads
from hyperqueue_amd.tick import Tick
from oracle.oracle import Oracle
for W in [int(a) for a in sys.argv[1:]] or [8, 16, 32, 64]:
    snap = workloads.make("c3p", n_tasks=20_000 * max(1, W // 8), n_workers=W)
    cfg = abi.make_config(time_limit_s=5.0)
    t = Tick(cfg)
    t0 = time.time(); g = t.tick(snap); tg = time.time() - t0
    o = Oracle(abi.make_config(time_limit_s=60.0))
    t0 = time.time(); w = o.tick(snap); to = time.time() - t0
    m = o.last_model()
    cd = g.counts_dict()
    x = np.zeros(len(m["obj"]))
    for j in range(len(x)):
        if m["ctype"][j] == 0:
            x[j] = cd.get((int(m["crq"][j]), int(m["cvariant"][j]), int(m["cworker"][j])), 0)
    # blocker flags are not in counts: objective only over placement columns (flags have zero cost)
    mine = float(np.dot(m["obj"], x))
    print(f"W={W:5d} cols={len(m['obj'])} rows={len(m['rhs'])}  product: {tg:6.2f}s status={g.status} opt={g.is_optimal} obj={mine:.6f} assigned={sum(len(r) for r in g.records)}"
          f"   oracle(HiGHS): {to:6.2f}s opt={w.is_optimal} obj={m['objective']:.6f} assigned={sum(len(r) for r in w.records)}", flush=True)
