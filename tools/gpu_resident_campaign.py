#!/usr/bin/env python
"""One-off GPU campaign of the resident ready set's deltas (needs a GPU): random sequences of arrivals in every form — plain, staged, packed with consecutive ids,
packed with gaps, batches BETWEEN resident ids (merged), fresh batches (appended behind the columns, round 4), now and then a new priority level or a cancel — with a tick
and a consume after each, every tick against the oracle on the equivalent full snapshot: the batches always (they are the ready set's histogram), everything else
(counts, records, free vectors) whenever both answers are canonical.
    python tools/gpu_resident_campaign.py [first_seed] [count]"""
import dataclasses
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401

from hyperqueue_amd import abi, workloads
from hyperqueue_amd.core import priority_from_user
from hyperqueue_amd.tick import Tick
from oracle.oracle import Oracle


def one(seed):
    rng = np.random.default_rng(31_000 + seed)
    W = int(rng.choice([8, 16, 24, 48]))
    in_tick = seed % 2 == 1  # every other scenario: the tick takes what it hands out itself (HQTICK_FLAG_CONSUME_IN_TICK), consume_last is a no-op
    cfg = abi.make_config(time_limit_s=20.0, flags=abi.HQTICK_FLAG_CONSUME_IN_TICK if in_tick else 0)
    snap = workloads.make("c3", n_tasks=int(rng.integers(60_000, 200_000)), n_workers=W, seed=seed)
    t = Tick(cfg)
    t.upload_ready(snap.task_id, snap.task_priority, snap.task_rq)
    ids, prio, rq = snap.task_id.copy(), snap.task_priority.copy(), snap.task_rq.copy()
    empty = dataclasses.replace(snap, _keep=[], task_id=np.zeros(0, np.uint64), task_priority=np.zeros(0, np.uint64), task_rq=np.zeros(0, np.uint32))
    p0 = int(snap.task_priority[0])
    next_id = int(ids[-1]) + 1
    low = 1
    stats = dict(seed=seed, W=W, in_tick=in_tick, steps=0, appended=0, merged=0, full_compares=0, levels=1, removed=0)
    o = Oracle(cfg, canonical=True)
    for step in range(int(rng.integers(6, 12))):
        kind = str(rng.choice(["plain", "staged", "packed", "packed_off", "between", "plain", "packed"]))
        n = int(rng.integers(1, 5000))
        pr = p0
        if rng.random() < 0.12:  # a new priority level: K1's validation flag, level table rebuilt, tick retried
            pr = int(priority_from_user(int(rng.integers(1, 4)))); n = int(rng.integers(1, 40)); stats["levels"] += 1
        before = t.kernel_stats()["ready_appends"]
        if kind == "between":
            new_ids = np.uint64(low) + np.arange(n, dtype=np.uint64); low += n + int(rng.integers(1, 100))
        elif kind == "packed_off":
            off = np.cumsum(rng.integers(1, 4, n)).astype(np.uint32); new_ids = np.uint64(next_id) + off.astype(np.uint64)
        else:
            new_ids = np.uint64(next_id) + np.arange(n, dtype=np.uint64)
        new_rq = rng.integers(0, 8, n).astype(np.uint32); new_prio = np.full(n, pr, np.uint64)
        if kind in ("plain", "between"):
            t.ready_add(new_ids, new_prio, new_rq)
        elif kind == "staged":
            a, b, c = t.ready_add_stage(n); a[:] = new_ids; b[:] = new_prio; c[:] = new_rq; t.ready_add_staged(n)
        elif kind == "packed":
            cut = int(rng.integers(1, n)) if n > 1 else n
            runs = [(next_id, cut), (next_id + cut, n - cut)] if n - cut else [(next_id, n)]
            t.ready_add_packed(runs, [(pr, n)], new_rq.astype(np.uint16))
        else:
            t.ready_add_packed([(next_id, n)], [(pr, n)], new_rq.astype(np.uint16), off)
        took = t.kernel_stats()["ready_appends"] - before
        stats["appended"] += took; stats["merged"] += 1 - took
        ids, prio, rq = np.concatenate([ids, new_ids]), np.concatenate([prio, new_prio]), np.concatenate([rq, new_rq])
        order = np.argsort(ids, kind="stable"); ids, prio, rq = ids[order], prio[order], rq[order]
        next_id = max(next_id, int(ids[-1]) + 1)
        if rng.random() < 0.25:  # a cancel: tombstones in the middle of the columns (an appended region included)
            victims = rng.choice(ids, size=min(len(ids), int(rng.integers(1, 300))), replace=False)
            assert t.ready_remove(np.sort(victims)) == len(victims)
            keep = ~np.isin(ids, victims); ids, prio, rq = ids[keep], prio[keep], rq[keep]; stats["removed"] += len(victims)
        assert t.ready_count() == len(ids), ("count", step, kind)
        got = t.tick(empty, resident=True)
        want = o.tick(dataclasses.replace(snap, _keep=[], task_id=ids, task_priority=prio, task_rq=rq))
        assert got.status == want.status and got.batches == want.batches, ("batches", step, kind)
        if got.is_canonical and want.is_canonical and got.is_optimal and want.is_optimal:
            assert got.counts == want.counts and got.records == want.records and (got.new_free == want.new_free).all(), ("records", step, kind)
            stats["full_compares"] += 1
        t.ready_consume_last()
        gone = np.asarray(sorted(tt for recs in got.records for (tt, _, _) in recs), np.uint64)
        keep = ~np.isin(ids, gone); ids, prio, rq = ids[keep], prio[keep], rq[keep]
        assert t.ready_count() == len(ids), ("count after consume", step, kind)
        stats["steps"] += 1
    t.close()
    return stats


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    bad, tot = [], dict(steps=0, appended=0, merged=0, full_compares=0, removed=0)
    t0 = time.time()
    for seed in range(first, first + count):
        try:
            s = one(seed)
            for k in tot: tot[k] += s[k]
            print("ok  ", s, flush=True)
        except Exception as e:  # noqa: BLE001
            bad.append(seed); print("FAIL", seed, type(e).__name__, str(e)[:300], flush=True)
    print(f"{count} scenarios, {len(bad)} failed {bad}; {tot}; {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
