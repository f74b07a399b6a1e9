#!/usr/bin/env python
"""Generates the committed golden fixtures in this directory (run from the repo root: python tests/golden/make_fixtures.py).

Each fixture is one snapshot (the ABI's SoA columns) plus the result the CPU oracle — pinned by the reference's own unit
tests, tests/test_oracle_golden.py — computes for it in canonical mode: batches, per-(rq, variant, worker) counts, the
per-worker record lists, retracts, redirects, multi-node placements and the free vectors.  The GPU tests replay the
snapshots through libhqtick.so and compare with these files, so the HIP path is checked against data that does not change
when the oracle's code does (tests/test_gpu_fixtures.py); the CPU suite checks that the oracle still reproduces them.
Large snapshots store the ready set by its generator arguments instead of the columns.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

from hyperqueue_amd import abi, workloads  # noqa: E402
from hyperqueue_amd.core import SchedEnv, TaskBuilder as TB, WorkerBuilder as WB  # noqa: E402


def snapshot_to_json(snap: abi.Snapshot, cfg: abi.Config, gen=None) -> dict:
    d = dict(
        config=dict(reserve=cfg.proactive_filling_reserve, fill_max=cfg.proactive_filling_max),
        n_resources=snap.n_resources, worker_id=snap.worker_id.tolist(), worker_total=np.asarray(snap.worker_total).tolist(),
        worker_free=np.asarray(snap.worker_free).tolist(), worker_remaining_ns=snap.worker_remaining_ns.tolist(),
        worker_min_utilization=[float(x) for x in snap.worker_min_utilization], worker_flags=snap.worker_flags.tolist(),
        worker_group=snap.worker_group.tolist(), n_groups=snap.n_groups, blocked=[list(b) for b in snap.blocked],
        assigned=[[list(x) for x in a] for a in snap.assigned], prefilled=[list(a) for a in snap.prefilled], requests=snap.requests,
        prefill={str(k): [v[0], [list(x) for x in v[1]]] for k, v in snap.prefill.items()},
        worker_map_rank=None if snap.worker_map_rank is None else snap.worker_map_rank.tolist(),
    )
    if gen is None:
        d.update(task_id=snap.task_id.tolist(), task_priority=snap.task_priority.tolist(), task_rq=snap.task_rq.tolist())
    else:
        d["ready_set_generator"] = gen
    return d


def snapshot_from_json(d: dict):
    cfg = abi.make_config(reserve=d["config"]["reserve"], fill_max=d["config"]["fill_max"], time_limit_s=60.0)
    if "ready_set_generator" in d:
        g = d["ready_set_generator"]
        base = workloads.make(g["workload"], seed=g["seed"], n_tasks=g["n_tasks"], n_workers=g["n_workers"])  # (c3s / c4s: only the ready set is taken from here)
        ids, prio, rq = base.task_id, base.task_priority, base.task_rq
    else:
        ids, prio, rq = np.asarray(d["task_id"], np.uint64), np.asarray(d["task_priority"], np.uint64), np.asarray(d["task_rq"], np.uint32)
    W, R = len(d["worker_id"]), d["n_resources"]
    snap = abi.Snapshot(
        n_resources=R, worker_id=np.asarray(d["worker_id"], np.uint32), worker_total=np.asarray(d["worker_total"], np.uint64).reshape(W, R),
        worker_free=np.asarray(d["worker_free"], np.uint64).reshape(W, R), worker_remaining_ns=np.asarray(d["worker_remaining_ns"], np.int64),
        worker_min_utilization=np.asarray(d["worker_min_utilization"], np.float32), worker_flags=np.asarray(d["worker_flags"], np.uint8),
        worker_group=np.asarray(d["worker_group"], np.uint32), n_groups=d["n_groups"], blocked=[tuple(b) for b in d["blocked"]],
        assigned=[[tuple(x) for x in a] for a in d["assigned"]], prefilled=[list(a) for a in d["prefilled"]],
        requests=[[dict(entries=[tuple(e) for e in v["entries"]], n_nodes=v["n_nodes"], min_time_ns=v["min_time_ns"], weight=v["weight"]) for v in vs] for vs in d["requests"]],
        task_id=ids, task_priority=prio, task_rq=rq,
        prefill={int(k): (v[0], [tuple(x) for x in v[1]]) for k, v in d["prefill"].items()},
        worker_map_rank=None if d["worker_map_rank"] is None else np.asarray(d["worker_map_rank"], np.uint32),
    )
    return snap, cfg


def result_to_json(r: abi.Result) -> dict:
    return dict(
        status=r.status, is_optimal=r.is_optimal,
        batches=[dict(rq=b.rq, size=b.size, limit=b.limit, limit_reached=b.limit_reached, is_blocker=b.is_blocker, cuts=[[c[0], [list(x) for x in c[1]]] for c in b.cuts]) for b in r.batches],
        counts=[list(c) for c in r.counts], records=[[list(x) for x in recs] for recs in r.records], retracts=r.retracts,
        redirects=sorted(list(x) for x in r.redirects), mn=[[t, ws] for (t, ws) in r.mn], new_free=np.asarray(r.new_free).tolist(),
    )


def result_digest(r: abi.Result) -> dict:
    """For big cases: counts in full, records as per-worker (n, first, last, xor-of-ids) + the total."""
    recs = []
    for w in r.records:
        x = 0
        for (t, v, k) in w:
            x ^= (t * 1000003 + v * 101 + k) & 0xFFFFFFFFFFFFFFFF
        recs.append([len(w), w[0][0] if w else 0, w[-1][0] if w else 0, x])
    d = result_to_json(r)
    d["records"] = None
    d["records_digest"] = recs
    return d


def big_digest(r: abi.Result) -> dict:
    """BASELINE-size results: SHA-256 of the count list, of every record in emission order and of the free vectors, plus the small parts in full"""
    import hashlib

    def sha(a):
        return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()

    counts = np.asarray(r.counts, np.uint32).reshape(-1, 4)
    recs = np.asarray([(t, v, k, w) for w, lst in enumerate(r.records) for (t, v, k) in lst], np.uint64).reshape(-1, 4)
    per_key = {}
    for (q, v, w, c) in r.counts:
        a = per_key.setdefault(f"{q}/{v}", [0, 0]); a[0] += c; a[1] += 1
    return dict(status=r.status, is_optimal=r.is_optimal,
                batches=[dict(rq=b.rq, size=b.size, limit=b.limit, limit_reached=b.limit_reached, is_blocker=b.is_blocker, n_cuts=len(b.cuts)) for b in r.batches],
                n_counts=len(counts), counts_sha256=sha(counts), tasks_and_workers_per_key=per_key, n_records=len(recs), records_sha256=sha(recs),
                n_assigned=int((recs[:, 2] == abi.HQ_REC_ASSIGN).sum()) if len(recs) else 0, new_free_sha256=sha(np.asarray(r.new_free, np.uint64)),
                n_retracts=sum(len(x) for x in r.retracts), n_redirects=len(r.redirects), n_mn=len(r.mn))


def big_snapshot(gen: dict):
    """BASELINE-size snapshots are stored by their generator call only"""
    if gen.get("steady"):
        snap = workloads.make_steady(gen["workload"], seed=gen["seed"], n_tasks=gen.get("n_tasks"), n_workers=gen.get("n_workers"))
    else:
        snap = workloads.make(gen["workload"], seed=gen["seed"], n_tasks=gen.get("n_tasks"), n_workers=gen.get("n_workers"))
    return snap, abi.make_config(time_limit_s=60.0)


BIG = {  # name -> generator call; the oracle needs seconds (c4: one distinct worker class) to minutes (steady state: ~1000 distinct classes, HiGHS each)
    "c4_full": dict(workload="c4", seed=0),
    "c2_full": dict(workload="c2", seed=0),
    "c3_full": dict(workload="c3", seed=0),
    "c3_steady_256": dict(workload="c3", seed=4, steady=True, n_tasks=300_000, n_workers=256),
    "c3_steady_full": dict(workload="c3", seed=0, steady=True),   # 1 M ready tasks, 1024 workers, 929 distinct worker classes
    "c4_steady_512": dict(workload="c4", seed=2, steady=True, n_tasks=500_000, n_workers=512),
}


def make_big(names=None):
    from oracle.oracle import Oracle

    os.makedirs(os.path.join(HERE, "big"), exist_ok=True)
    for name, gen in BIG.items():
        if names and name not in names:
            continue
        snap, cfg = big_snapshot(gen)
        r = Oracle(cfg, canonical=True).tick(snap)
        assert r.is_optimal, name
        with open(os.path.join(HERE, "big", name + ".json"), "w") as f:
            json.dump(dict(generator=gen, expect=big_digest(r)), f, indent=1)
        print(name, "written")


def multi_tick_env(seed: int) -> SchedEnv:
    """A SchedEnv after two ticks with finished tasks in between: carries prefill sets, assigned tasks, changed free vectors."""
    rng = np.random.default_rng(seed)
    env = SchedEnv(abi.make_config(reserve=2, fill_max=5, time_limit_s=60.0))
    env.new_named_resource("gpus/amd")
    for _ in range(int(rng.integers(3, 7))):
        env.new_worker(WB(int(rng.integers(4, 13))).res_sum("gpus/amd", int(rng.integers(0, 3))))
    for _ in range(int(rng.integers(40, 90))):
        c = int(rng.integers(0, 3))
        env.new_task([TB().cpus(1), TB().cpus(2).user_priority(1), TB().cpus(1).add_resource(1, 0.5)][c])
    return env


def main(only=None):
    from oracle.oracle import Oracle
    from test_gpu_parity import random_env

    out = {}
    for seed in (1, 5, 8, 13, 21, 34):
        env = random_env(seed)
        snap = env.snapshot()
        r = Oracle(env.config, canonical=True).tick(snap)
        out[f"random_{seed}"] = dict(snapshot=snapshot_to_json(snap, env.config), expect=result_to_json(r))
    for seed in (2, 3):  # second tick of a scripted scenario: prefill sets + retract/redirect paths
        env = multi_tick_env(seed)
        o = Oracle(env.config, canonical=True)
        env.schedule(o)
        done = 0
        for t in sorted(env.tasks.values(), key=lambda t: t.id):
            if t.state == 1 and done < 3:
                env.finish_task(t.id, t.worker); done += 1
        env.new_worker(WB(6).res_sum("gpus/amd", 1))
        snap = env.snapshot()
        r = o.tick(snap)
        out[f"second_tick_{seed}"] = dict(snapshot=snapshot_to_json(snap, env.config), expect=result_to_json(r))
    for name, kw in (("c2", dict(n_tasks=100_000, n_workers=256)), ("c3", dict(n_tasks=60_000, n_workers=48)), ("c4", dict(n_tasks=40_000, n_workers=24))):
        cfg = abi.make_config(time_limit_s=60.0)
        snap = workloads.make(name, seed=0, **kw)
        r = Oracle(cfg, canonical=True).tick(snap)
        out[f"{name}_{kw['n_tasks']}x{kw['n_workers']}"] = dict(
            snapshot=snapshot_to_json(snap, cfg, gen=dict(workload=name, seed=0, **kw)), expect=result_digest(r))
    # BASELINE configs[0] at full size: 1 000 single-core tasks on 4 workers x 4 cores (benchmarks/experiment-per-task-overhead.py:33-55) — small enough to be
    # stored with its columns and every record
    cfg = abi.make_config(time_limit_s=60.0)
    snap = workloads.make("c1", seed=0)
    out["c1_1000x4"] = dict(snapshot=snapshot_to_json(snap, cfg), expect=result_to_json(Oracle(cfg, canonical=True).tick(snap)))
    if only:
        out = {k: v for k, v in out.items() if k in only}
    for k, v in out.items():
        with open(os.path.join(HERE, k + ".json"), "w") as f:
            json.dump(v, f, separators=(",", ":"))
        print(k, os.path.getsize(os.path.join(HERE, k + ".json")), "bytes")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "big":
        make_big(sys.argv[2:])
    else:
        main(sys.argv[1:])  # names given: only those files are rewritten
