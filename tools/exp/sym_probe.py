"""Experiment: do value-ordering rows between identical workers (symmetry breaking) help the B&B on the half-full plateaus?"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from hyperqueue_amd import abi, workloads
from oracle.oracle import Oracle
from scipy.optimize import milp, LinearConstraint, Bounds
import scipy.sparse as sp
W = int(sys.argv[1]); fill = float(sys.argv[2])
ids, prio, rq, off, dep = workloads.make_dag(1_000_000, seed=0)
src = np.nonzero((off[1:] - off[:-1]) == 0)[0]
k = min(len(src), int(len(src) * W / 1024 * fill / 0.45)); sel = src[:k]
drv = workloads.DagChurn(n_workers=W, churn=0.1, seed=0)
snap = drv.snapshot(ids[sel], prio[sel], (rq[sel] % 8).astype(np.uint32))
o = Oracle(abi.make_config(time_limit_s=0.5)); o.tick(snap); m = o.last_model()
n = len(m["obj"]); nr = len(m["rhs"])
A = sp.csr_matrix((m["rcoef"], m["rcol"], m["roff"]), shape=(nr, n)).tolil()
cw = m["cworker"]; workers = sorted(set(cw.tolist()))
cols_w = {w: np.nonzero(cw == w)[0] for w in workers}
f = {w: (len(workers) - i) / len(workers) for i, w in enumerate(workers)}
def solve(extra):
    cons = [LinearConstraint(A.tocsr(), -np.inf, m["rhs"])]
    if extra:
        rows = sp.lil_matrix((len(workers) - 1, n))
        for i in range(len(workers) - 1):
            a, b = workers[i], workers[i + 1]
            for j in cols_w[a]: rows[i, j] = m["obj"][j] / f[a]
            for j in cols_w[b]: rows[i, j] -= m["obj"][j] / f[b]
        cons.append(LinearConstraint(rows.tocsr(), 0, np.inf))
    t0 = time.time()
    r = milp(-m["obj"], constraints=cons, integrality=np.ones(n), bounds=Bounds(0, np.inf), options=dict(mip_rel_gap=1e-4, time_limit=20))
    return time.time() - t0, -r.fun if r.x is not None else None, r.status, getattr(r, "mip_node_count", None), getattr(r, "mip_dual_bound", None)
print("plain   ", solve(False))
print("ordered ", solve(True))
