import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from host_stages import HostStages
from hyperqueue_amd import abi, workloads
W = int(sys.argv[1]); fill = float(sys.argv[2]); limit = float(sys.argv[3]) if len(sys.argv) > 3 else 5.0
ids, prio, rq, off, dep = workloads.make_dag(1_000_000, seed=0)
src = np.nonzero((off[1:] - off[:-1]) == 0)[0]
k = min(len(src), int(len(src) * W / 1024 * fill / 0.45))
sel = src[:k]
drv = workloads.DagChurn(n_workers=W, churn=0.1, seed=0)
snap = drv.snapshot(ids[sel], prio[sel], (rq[sel] % 8).astype(np.uint32))
hs = HostStages(abi.make_config(time_limit_s=limit))
t0 = time.time(); g = hs.stages(snap); tg = time.time() - t0
print(f"W={W} fill={fill} ready={k} | {tg:.2f}s opt={int(g.is_optimal)} canonical={int(g.is_canonical)} assigned={sum(c for *_, c in g.counts)}")
