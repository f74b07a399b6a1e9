"""debug: test_fresh_batches_are_appended's sequence, printing where the resident tick, a fresh tick on the full snapshot and the oracle part ways"""
import dataclasses, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from hyperqueue_amd import abi, workloads
from hyperqueue_amd.tick import Tick
from oracle.oracle import Oracle

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rng = np.random.default_rng(900 + seed)
cfg = abi.make_config(time_limit_s=20.0)
snap = workloads.make("c3", n_tasks=80_000, n_workers=24, seed=seed)
t = Tick(cfg)
t.upload_ready(snap.task_id, snap.task_priority, snap.task_rq)
ids, prio, rq = snap.task_id.copy(), snap.task_priority.copy(), snap.task_rq.copy()
empty = dataclasses.replace(snap, _keep=[], task_id=np.zeros(0, np.uint64), task_priority=np.zeros(0, np.uint64), task_rq=np.zeros(0, np.uint32))
p0 = int(snap.task_priority[0]); next_id = int(ids[-1]) + 1
print("worker ids", snap.worker_id[:6], "...", "prio distinct", np.unique(snap.task_priority)[:5])
for step in range(8):
    kind = ["between", "plain", "packed", "packed_off"][step % 4]
    n = int(rng.integers(1, 3000))
    if kind == "between":
        new_ids = np.uint64(1 + 10_000 * step) + np.arange(n, dtype=np.uint64)
    elif kind == "packed_off":
        off = np.cumsum(rng.integers(1, 4, n)).astype(np.uint32); new_ids = np.uint64(next_id) + off.astype(np.uint64)
    else:
        new_ids = np.uint64(next_id) + np.arange(n, dtype=np.uint64)
    new_rq = rng.integers(0, 8, n).astype(np.uint32); new_prio = np.full(n, p0, np.uint64)
    if kind in ("plain", "between"): t.ready_add(new_ids, new_prio, new_rq)
    elif kind == "packed":
        half = max(1, n // 2); runs = [(next_id, half), (next_id + half, n - half)] if n - half else [(next_id, n)]
        t.ready_add_packed(runs, [(p0, n)], new_rq.astype(np.uint16))
    else: t.ready_add_packed([(next_id, n)], [(p0, n)], new_rq.astype(np.uint16), off)
    ids, prio, rq = np.concatenate([ids, new_ids]), np.concatenate([prio, new_prio]), np.concatenate([rq, new_rq])
    order = np.argsort(ids, kind="stable"); ids, prio, rq = ids[order], prio[order], rq[order]
    next_id = max(next_id, int(ids[-1]) + 1)
    full = dataclasses.replace(snap, _keep=[], task_id=ids, task_priority=prio, task_rq=rq)
    got = t.tick(empty, resident=True)
    fresh = Tick(cfg).tick(full)
    want = Oracle(cfg, canonical=True).tick(full)
    ks = t.kernel_stats()
    print(f"step {step} {kind} n {n}: resident==fresh counts {got.counts == fresh.counts} records {got.records == fresh.records}; fresh==oracle counts {fresh.counts == want.counts} records {fresh.records == want.records}; batches eq {got.batches == want.batches}; "
          f"optimal {got.is_optimal}/{want.is_optimal} canonical {got.is_canonical} classes {ks['n_classes']} host {ks['n_classes_host']} memo {ks['n_classes_memo']} appends {ks['ready_appends']}")
    if got.counts != want.counts:
        print("  got ", got.counts[:12]); print("  want", want.counts[:12]); print("  batches", got.batches[:10])
    t.ready_consume_last()
    gone = np.asarray(sorted(tt for recs in got.records for (tt, _, _) in recs), np.uint64)
    keep = ~np.isin(ids, gone); ids, prio, rq = ids[keep], prio[keep], rq[keep]
