"""Seeded synthetic snapshots of the shapes BASELINE.json names.  All integer; the RNG is splitmix64 so a
C++/Rust harness can regenerate the same inputs.

  c1  1 000 tasks cpus=1, 4 workers x 4 cores                       (plumbing case of benchmarks/experiment-per-task-overhead.py)
  c2  100 000 tasks cpus=1, 256 workers x 128 cores, one priority   (uniform bin-packing)
  c3  1 000 000 tasks over 8 request classes on {cpus, gpus/amd, mem} incl. 0.5 / 0.25 GPU fractions, 1024 workers
      (128 c / 8 g / 512 m), one priority level, every class saturated  -> the placement model is separable per worker
  c3p c3 with three user-priority levels (80/15/5 %): couples all workers through priority cuts (reported, not benched)
  c4  c3 classes as 2-variant OR-lists, 4096 workers                (sharded case; one priority level)
  c4p c4 with c3p's three priority levels: configs[3] as BASELINE.md §3 writes it ("as C3") — 65 536 placement columns plus cut / blocker rows over 4096 blocks
  c5  make_dag(): 1 000 000-node random DAG over the c3 classes, fan-in ~ Poisson(3) from lower ids (dependency-graph case)
  c3s / c4s  make_steady(): c3 / c4 in the STEADY STATE of SURVEY.md §8(d) — every worker is running a packed mix of tasks of which a random
      10 % have just finished, so the free vectors differ from worker to worker (about one worker class per worker) while the ready set is
      still large enough to saturate every request class
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np

from . import abi
from .core import FR, priority_from_user, task_id

MASK = (1 << 64) - 1


def splitmix64_stream(seed: int, n: int) -> np.ndarray:
    """n outputs of splitmix64 seeded with `seed` (vectorised: state_i = seed + (i+1)*golden)."""
    with np.errstate(over="ignore"):
        z = (np.uint64(seed) + (np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _variant(entries: List[Tuple[int, float]], weight: int = 10_000) -> dict:
    return dict(entries=[(r, abi.HQ_ENTRY_AMOUNT, int(round(a * FR))) for r, a in entries], n_nodes=0, min_time_ns=0, weight=weight)


# resource ids: 0 cpus, 1 gpus/amd, 2 mem
C3_CLASSES = [
    ([(0, 1)], 0.33),                          # 1 cpu
    ([(0, 4)], 0.12),                          # 4 cpus
    ([(0, 2), (1, 1)], 0.08),                  # 2 cpus + 1 gpu
    ([(0, 1), (1, 0.5)], 0.08),                # 1 cpu + half a gpu
    ([(0, 1), (1, 0.25)], 0.05),               # 1 cpu + quarter gpu
    ([(0, 8), (2, 64)], 0.08),                 # 8 cpus + 64 mem
    ([(0, 16), (1, 2), (2, 128)], 0.06),       # 16 cpus + 2 gpus + 128 mem
    ([(0, 1), (2, 1)], 0.20),                  # 1 cpu + 1 mem
]
C4_ALTERNATIVES = [[(0, 2)], [(0, 16)], [(0, 8)], [(0, 4)], [(0, 2)], [(0, 32)], [(0, 64)], [(0, 2)]]


def _uniform_workers(n: int, first_id: int, total_units: List[float]):
    R = len(total_units)
    total = np.tile(np.asarray([int(round(u * FR)) for u in total_units], np.uint64), (n, 1))
    return dict(
        n_resources=R, worker_id=np.arange(first_id, first_id + n, dtype=np.uint32), worker_total=total, worker_free=total.copy(),
        worker_remaining_ns=np.full(n, abi.HQ_NO_TIME_LIMIT, np.int64), worker_min_utilization=np.zeros(n, np.float32),
        worker_flags=np.full(n, abi.HQ_WORKER_SN, np.uint8), worker_group=np.zeros(n, np.uint32), n_groups=1, blocked=[],
        assigned=[[] for _ in range(n)], prefilled=[[] for _ in range(n)],
    )


def _tasks(n: int, class_weights: List[float], seed: int, priorities: Optional[List[Tuple[int, float]]] = None):
    ids = (np.uint64(1) << np.uint64(32)) | np.arange(1, n + 1, dtype=np.uint64)  # job 1, task 1..n: ascending
    r = splitmix64_stream(seed, 2 * n)
    u = (r[:n] >> np.uint64(11)).astype(np.float64) / float(1 << 53)
    edges = np.cumsum(class_weights) / np.sum(class_weights)
    rq = np.searchsorted(edges, u, side="right").clip(0, len(class_weights) - 1).astype(np.uint32)
    if priorities:
        u2 = (r[n:] >> np.uint64(11)).astype(np.float64) / float(1 << 53)
        pe = np.cumsum([p[1] for p in priorities]) / sum(p[1] for p in priorities)
        pi = np.searchsorted(pe, u2, side="right").clip(0, len(priorities) - 1)
        prio = np.asarray([priority_from_user(p[0]) for p in priorities], np.uint64)[pi]
    else:
        prio = np.full(n, priority_from_user(0), np.uint64)
    return ids, prio, rq


def make(name: str, seed: int = 0, n_tasks: Optional[int] = None, n_workers: Optional[int] = None) -> abi.Snapshot:
    name = name.lower()
    if name == "c1":
        w = _uniform_workers(n_workers or 4, 1, [4])
        ids, prio, rq = _tasks(n_tasks or 1_000, [1.0], seed)
        return abi.Snapshot(requests=[[_variant([(0, 1)])]], task_id=ids, task_priority=prio, task_rq=rq, **w)
    if name == "c2":
        w = _uniform_workers(n_workers or 256, 1, [128])
        ids, prio, rq = _tasks(n_tasks or 100_000, [1.0], seed)
        return abi.Snapshot(requests=[[_variant([(0, 1)])]], task_id=ids, task_priority=prio, task_rq=rq, **w)
    if name in ("c3", "c3p"):
        w = _uniform_workers(n_workers or 1024, 1, [128, 8, 512])
        pr = [(0, 0.80), (1, 0.15), (2, 0.05)] if name == "c3p" else None
        ids, prio, rq = _tasks(n_tasks or 1_000_000, [c[1] for c in C3_CLASSES], seed, pr)
        return abi.Snapshot(requests=[[_variant(c[0])] for c in C3_CLASSES], task_id=ids, task_priority=prio, task_rq=rq, **w)
    if name in ("c4", "c4p"):   # c4p: BASELINE configs[3] as BASELINE.md §3 / SURVEY §8(d) write it — "as C3" (three priority levels at 80/15/5 %) with 2-variant OR-lists
        w = _uniform_workers(n_workers or 4096, 1, [128, 8, 512])
        pr = [(0, 0.80), (1, 0.15), (2, 0.05)] if name == "c4p" else None
        ids, prio, rq = _tasks(n_tasks or 1_000_000, [c[1] for c in C3_CLASSES], seed, pr)
        reqs = [[_variant(c[0]), _variant(alt)] for c, alt in zip(C3_CLASSES, C4_ALTERNATIVES)]
        return abi.Snapshot(requests=reqs, task_id=ids, task_priority=prio, task_rq=rq, **w)
    raise ValueError(f"unknown workload {name}")


def make_steady(name: str = "c3", seed: int = 0, n_tasks: Optional[int] = None, n_workers: Optional[int] = None, release: float = 0.10) -> abi.Snapshot:
    """SURVEY.md §8(d) "steady-state variant": the cluster of `name` (c3 / c4) mid-run.  Every worker is first filled with a random mix of
    tasks drawn by the class weights until five draws in a row no longer fit, then each running task finishes with probability `release`;
    `worker_free` = total - what still runs, `assigned` = the running (rq, variant 0) list.  The ready set is the one of `name`."""
    snap = make(name, seed=seed, n_tasks=n_tasks, n_workers=n_workers)
    W, R = len(snap.worker_id), snap.n_resources
    need = np.zeros((len(C3_CLASSES), R), np.int64)
    for q, (entries, _) in enumerate(C3_CLASSES):
        for r, a in entries:
            need[q, r] = int(round(a * FR))
    edges = np.cumsum([c[1] for c in C3_CLASSES]) / np.sum([c[1] for c in C3_CLASSES])
    free = np.asarray(snap.worker_total, np.int64).copy()
    assigned = []
    draws = splitmix64_stream(seed ^ 0x57EAD, W * 512).reshape(W, 512)
    for w in range(W):
        running, misses, i = [], 0, 0
        while misses < 5 and i < 384:
            u = float(draws[w, i] >> np.uint64(11)) / float(1 << 53); i += 1
            q = int(np.searchsorted(edges, u, side="right").clip(0, len(C3_CLASSES) - 1))
            if (free[w] >= need[q]).all():
                free[w] -= need[q]; running.append(q); misses = 0
            else:
                misses += 1
        keep = []
        for k, q in enumerate(running):
            u = float(draws[w, 384 + (k % 128)] >> np.uint64(11)) / float(1 << 53)
            if u < release:
                free[w] += need[q]
            else:
                keep.append((q, 0))
        assigned.append(keep)
    snap.worker_free = free.astype(np.uint64)
    snap.assigned = assigned
    return snap


def make_dag(n: int = 1_000_000, seed: int = 0, mean_fan_in: float = 3.0):
    """BASELINE config 5: `n` tasks of the c3 classes in submission order; task i depends on ~Poisson(mean_fan_in)
    distinct tasks drawn uniformly from the ones before it.  Returns (task_id, priority, rq, dep_off u32[n + 1], dep_task_id u64[E])."""
    ids, prio, rq = _tasks(n, [c[1] for c in C3_CLASSES], seed)
    r = splitmix64_stream(seed ^ 0x5EED, n)
    u = (r >> np.uint64(11)).astype(np.float64) / float(1 << 53)
    kmax = 24
    pmf = np.exp(-mean_fan_in) * np.cumprod(np.concatenate([[1.0], mean_fan_in / np.arange(1, kmax)]))
    k = np.searchsorted(np.cumsum(pmf), u, side="right").clip(0, kmax - 1)
    k = np.minimum(k, np.arange(n))  # task i has only i predecessors
    cons = np.repeat(np.arange(n, dtype=np.int64), k)
    r2 = splitmix64_stream(seed ^ 0xDA6, len(cons))
    u2 = (r2 >> np.uint64(11)).astype(np.float64) / float(1 << 53)
    dep = np.minimum((u2 * cons).astype(np.int64), cons - 1)
    pair = np.unique(cons * n + dep)  # distinct (consumer, dependency) pairs, sorted by consumer then dependency
    cons, dep = pair // n, pair % n
    off = np.zeros(n + 1, np.uint32)
    off[1:] = np.cumsum(np.bincount(cons, minlength=n))
    return ids, prio, rq, off, ids[dep]


def make_dag_layered(n: int = 1_000_000, width: int = 20_000, seed: int = 0, fan_in: int = 3):
    """A second DAG shape next to SURVEY.md's: layers of `width` tasks in submission order, every task of layer L depends on
    `fan_in` distinct tasks drawn uniformly from layer L - 1.  Its frontier is a whole layer (>= 10^4 ready tasks per tick) where make_dag's narrows to
    ~100 after the sources.  Returns the same tuple as make_dag."""
    ids, prio, rq = _tasks(n, [c[1] for c in C3_CLASSES], seed)
    idx = np.arange(n, dtype=np.int64)
    layer = idx // width
    has = layer > 0
    cons = np.repeat(idx[has], fan_in)
    r = splitmix64_stream(seed ^ 0x1A7E6, len(cons))
    u = (r >> np.uint64(11)).astype(np.float64) / float(1 << 53)
    lo = (np.repeat(layer[has], fan_in) - 1) * width
    dep = lo + np.minimum((u * width).astype(np.int64), width - 1)
    pair = np.unique(cons * n + dep)
    cons, dep = pair // n, pair % n
    off = np.zeros(n + 1, np.uint32)
    off[1:] = np.cumsum(np.bincount(cons, minlength=n))
    return ids, prio, rq, off, ids[dep]


class DagChurn:
    """Driver of BASELINE config 5 (DAG + 10 % worker churn per tick), independent of the backend that ticks.

    Between two ticks every task the last tick handed out (assigned or prefilled) finishes — the `sleep 0` model of
    benchmarks/experiment-per-task-overhead.py — except the tasks on the workers that are lost in between: those go back to the ready
    queues (on_remove_worker, reactor.rs:64-186) and the lost workers are replaced by fresh ones (new, larger ids, same resources).
    """

    def __init__(self, n_workers: int = 1024, churn: float = 0.10, seed: int = 0):
        w = _uniform_workers(n_workers, 1, [128, 8, 512])
        self.kw = dict(w)
        self.requests = [[_variant(c[0])] for c in C3_CLASSES]
        self.worker_id = np.asarray(w["worker_id"], np.uint32).copy()
        self.next_worker = int(self.worker_id.max()) + 1
        self.n_lost = int(round(churn * n_workers))
        self.seed, self.step = seed, 0

    def snapshot(self, task_id=None, task_priority=None, task_rq=None) -> abi.Snapshot:
        """the current workers; without task columns the ready set is the resident one"""
        kw = dict(self.kw); kw["worker_id"] = self.worker_id.copy()
        z = np.zeros(0, np.uint64)
        return abi.Snapshot(requests=self.requests, task_id=z if task_id is None else np.asarray(task_id, np.uint64),
                            task_priority=z if task_priority is None else np.asarray(task_priority, np.uint64),
                            task_rq=np.zeros(0, np.uint32) if task_rq is None else np.asarray(task_rq, np.uint32), **kw)

    def after_tick(self, rec_off: np.ndarray, rec_task: np.ndarray):
        """-> (finished ids, ids returned to the ready queues ascending); replaces the lost workers"""
        W = len(self.worker_id)
        self.step += 1
        r = splitmix64_stream(self.seed * 1_000_003 + self.step, W)
        lost = np.sort(np.argsort(r, kind="stable")[: self.n_lost])
        on_lost = np.zeros(len(rec_task), bool)
        for w in lost:
            on_lost[int(rec_off[w]) : int(rec_off[w + 1])] = True
        returned = np.sort(rec_task[on_lost])
        finished = rec_task[~on_lost]
        keep = np.ones(W, bool); keep[lost] = False
        fresh = np.arange(self.next_worker, self.next_worker + len(lost), dtype=np.uint32)
        self.next_worker += len(lost)
        self.last_lost_ids, self.last_fresh_ids = self.worker_id[lost].copy(), fresh  # what a delta-driven host forwards (hqtick_cluster_remove_workers / _add_workers)
        self.worker_id = np.concatenate([self.worker_id[keep], fresh])
        return finished, returned


def shard_workers(snap: abi.Snapshot, rank: int, world: int) -> abi.Snapshot:
    """Worker shard of one rank: workers whose FxHash(worker_id) mod world == rank (north_star: hash-partitioned workers).
    Every rank keeps the whole ready set and request table."""
    from .hbmap import hash_u32

    keep = np.asarray([hash_u32(int(w)) % world == rank for w in snap.worker_id], bool)
    idx = np.nonzero(keep)[0]
    return abi.Snapshot(
        n_resources=snap.n_resources, worker_id=snap.worker_id[idx], worker_total=snap.worker_total[idx], worker_free=snap.worker_free[idx],
        worker_remaining_ns=snap.worker_remaining_ns[idx], worker_min_utilization=snap.worker_min_utilization[idx], worker_flags=snap.worker_flags[idx],
        worker_group=snap.worker_group[idx], n_groups=snap.n_groups, blocked=[], assigned=[snap.assigned[i] for i in idx], prefilled=[snap.prefilled[i] for i in idx],
        requests=snap.requests, task_id=snap.task_id, task_priority=snap.task_priority, task_rq=snap.task_rq,
    )
