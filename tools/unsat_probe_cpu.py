#!/usr/bin/env python
"""tools/unsat_probe.py without a GPU: the product's HOST stages (hqtick_debug_host_stages: batches + placement) on the unsaturated
multi-class ticks, next to the HiGHS oracle.  `python tools/unsat_probe_cpu.py 8 16 32 [--limit 5] [--no-oracle]`"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from host_stages import HostStages  # noqa: E402
from hyperqueue_amd import abi, workloads  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
limit = float(sys.argv[sys.argv.index("--limit") + 1]) if "--limit" in sys.argv else 5.0
with_oracle = "--no-oracle" not in sys.argv
ids, prio, rq, off, dep = workloads.make_dag(1_000_000, seed=0)
src = np.nonzero((off[1:] - off[:-1]) == 0)[0]
ncls = int(os.environ.get("NCLS", "8"))
for W in [int(a) for a in args] or [4, 8, 16, 32]:
    for fill in (0.2, 0.45, 0.8):
        k = min(len(src), int(len(src) * W / 1024 * fill / 0.45))
        sel = src[:k]
        drv = workloads.DagChurn(n_workers=W, churn=0.1, seed=0)
        snap = drv.snapshot(ids[sel], prio[sel], (rq[sel] % ncls).astype(np.uint32))
        hs = HostStages(abi.make_config(time_limit_s=limit))
        t0 = time.time(); g = hs.stages(snap); tg = time.time() - t0
        line = f"W={W:5d} fill={fill:.2f} ready={k:6d} | product {tg:6.2f}s opt={int(g.is_optimal)} canonical={int(g.is_canonical)} assigned={sum(c for *_, c in g.counts)}"
        if with_oracle:
            o = Oracle(abi.make_config(time_limit_s=limit), reference_solver_options="--exact-oracle" not in sys.argv)  # HiGHS as the reference configures it (mip_rel_gap 1e-4)
            t0 = time.time(); w = o.tick(snap); to = time.time() - t0
            m = o.last_model()
            cd = g.counts_dict()
            x = np.zeros(len(m["obj"]))
            for j in range(len(x)):
                if m["ctype"][j] == 0:
                    x[j] = cd.get((int(m["crq"][j]), int(m["cvariant"][j]), int(m["cworker"][j])), 0)
            mine = float(np.dot(m["obj"], x))
            line += f" obj={mine:.7f} | HiGHS {to:6.2f}s opt={int(w.is_optimal)} obj={m['objective']:.7f} | rel {(mine - m['objective']) / m['objective']:+.2e}"
        print(line, flush=True)
