#!/usr/bin/env python
"""Experiment: how tight is the Dantzig-Wolfe (per-worker integer hull) bound on the unsaturated ticks?  CPU only, scipy."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from hyperqueue_amd import abi, workloads
from oracle.oracle import Oracle
from scipy.optimize import milp, linprog, LinearConstraint, Bounds
import scipy.sparse as sp

def instance(W, fill, ncls=8):
    ids, prio, rq, off, dep = workloads.make_dag(1_000_000, seed=0)
    src = np.nonzero((off[1:] - off[:-1]) == 0)[0]
    k = min(len(src), int(len(src) * W / 1024 * fill / 0.45))
    sel = src[:k]
    drv = workloads.DagChurn(n_workers=W, churn=0.1, seed=0)
    return drv.snapshot(ids[sel], prio[sel], (rq[sel] % ncls).astype(np.uint32))

W = int(sys.argv[1]); fill = float(sys.argv[2])
snap = instance(W, fill)
o = Oracle(abi.make_config(time_limit_s=20.0))
t0 = time.time(); o.tick(snap); print("oracle", time.time() - t0)
m = o.last_model()
n = len(m["obj"]); nr = len(m["rhs"])
A = sp.csr_matrix((m["rcoef"], m["rcol"], m["roff"]), shape=(nr, n))
print("cols", n, "rows", nr, "rtypes", np.bincount(m["rtype"]), "ctype", np.bincount(m["ctype"]), "objective", m["objective"])
# LP bound
ub = np.full(n, np.inf)
res = linprog(-m["obj"], A_ub=A, b_ub=m["rhs"], bounds=[(0, None)] * n, method="highs")
print("LP bound", -res.fun, "gap", (-res.fun - m["objective"]) / m["objective"])
# rows by type: which rows involve a single worker?
cw = m["cworker"]
rows_w = {}
coupling = []
for r in range(nr):
    cols = m["rcol"][m["roff"][r]:m["roff"][r + 1]]
    ws = set(cw[cols].tolist())
    if len(ws) == 1: rows_w.setdefault(ws.pop(), []).append(r)
    else: coupling.append(r)
print("coupling rows", len(coupling), "per-worker rows", {w: len(v) for w, v in list(rows_w.items())[:3]})
workers = sorted(set(cw.tolist()))
cols_w = {w: np.nonzero(cw == w)[0] for w in workers}
Ad = A.toarray()
# column generation
pats = {w: [np.zeros(len(cols_w[w]))] for w in workers}
def price(w, pi):
    c = cols_w[w]
    red = m["obj"][c] - pi @ Ad[np.ix_(coupling, c)]
    rr = rows_w.get(w, [])
    cons = LinearConstraint(Ad[np.ix_(rr, c)], -np.inf, m["rhs"][rr])
    r = milp(-red, constraints=cons, integrality=np.ones(len(c)), bounds=Bounds(0, np.inf))
    return np.round(r.x), -r.fun
best = np.inf
for it in range(0 if os.environ.get("NOCG") else 200):
    # master: max sum_w sum_p lam v(p)  s.t. sum_p lam_wp <= 1, sum coupling(p) lam <= rhs
    cols = [(w, p) for w in workers for p in pats[w]]
    cvec = np.array([m["obj"][cols_w[w]] @ p for w, p in cols])
    Aconv = np.zeros((len(workers), len(cols)))
    for j, (w, p) in enumerate(cols): Aconv[workers.index(w), j] = 1
    Acoup = np.array([[Ad[r, cols_w[w]] @ p for (w, p) in cols] for r in coupling])
    res = linprog(-cvec, A_ub=np.vstack([Aconv, Acoup]), b_ub=np.concatenate([np.ones(len(workers)), m["rhs"][coupling]]), bounds=[(0, None)] * len(cols), method="highs")
    duals = -res.ineqlin.marginals
    mu, pi = duals[:len(workers)], duals[len(workers):]
    # Lagrangian bound at pi
    lb = pi @ m["rhs"][coupling]; added = 0
    for i, w in enumerate(workers):
        p, val = price(w, pi)
        lb += max(val, 0)
        if val > mu[i] + 1e-9:
            pats[w].append(p); added += 1
    best = min(best, lb)
    print(f"it {it} master {-res.fun:.8f} lagr {lb:.8f} best {best:.8f} added {added} gap {(best - m['objective']) / m['objective']:.2e}", flush=True)
    if added == 0: break
if os.environ.get("DUMP"):
    np.set_printoptions(linewidth=250, precision=6, suppress=True)
    for w in workers[:2]:
        c = cols_w[w]; print("worker", w, "obj", m["obj"][c]); print(Ad[np.ix_(rows_w[w], c)], m["rhs"][rows_w[w]])
    print("size rhs", m["rhs"][coupling])
    print("x", m["x"].reshape(len(workers), -1))
if os.environ.get("HIGHSLOG"):
    r = milp(-m["obj"], constraints=LinearConstraint(A, -np.inf, m["rhs"]), integrality=np.ones(n), bounds=Bounds(0, np.inf), options=dict(disp=True, mip_rel_gap=1e-4))
