import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from host_stages import HostStages
from hyperqueue_amd import abi, workloads
from oracle.oracle import Oracle
args = [a for a in sys.argv[1:] if not a.startswith("--")]
W = int(args[0]); ntasks = int(args[1]) if len(args) > 1 else 20_000 * max(1, W // 8)
snap = workloads.make("c3p", n_tasks=ntasks, n_workers=W)
hs = HostStages(abi.make_config(time_limit_s=5.0))
t0 = time.time(); g = hs.stages(snap); tg = time.time() - t0
line = f"W={W} tasks={ntasks} | {tg:.2f}s opt={int(g.is_optimal)} canonical={int(g.is_canonical)} assigned={sum(c for *_, c in g.counts)}"
if "--oracle" in sys.argv:
    o = Oracle(abi.make_config(time_limit_s=float(os.environ.get("OLIMIT", "5"))), reference_solver_options=True)
    t0 = time.time(); w = o.tick(snap); to = time.time() - t0
    m = o.last_model(); cd = g.counts_dict(); x = np.zeros(len(m["obj"]))
    for j in range(len(x)):
        if m["ctype"][j] == 0: x[j] = cd.get((int(m["crq"][j]), int(m["cvariant"][j]), int(m["cworker"][j])), 0)
    line += f" obj={float(np.dot(m['obj'], x)):.6f} | HiGHS {to:.2f}s opt={int(w.is_optimal)} obj={m['objective']:.6f}"
print(line)
