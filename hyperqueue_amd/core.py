"""TEST SUPPORT (not part of the product, which is libhqtick.so): host-side mirror of the slice of tako's `Core` that the scheduling tick reads and writes.

The reference keeps this state in Rust (`Core`, `Worker`, `Task`, `TaskQueues`; paths relative to
/root/reference/crates/tako/src/internal/): server/core.rs, server/worker.rs:41-78, server/task.rs:22-43,
scheduler/taskqueue.rs:27-119.  No Rust toolchain exists in this image, so the host side above the C ABI is
mirrored here with the same names and the same builder vocabulary as the reference's own unit-test harness
(tests/utils/env.rs `TestEnv`, tests/utils/task.rs `TaskBuilder`, tests/utils/worker.rs `WorkerBuilder`), so
parity tests read like the reference's tests.

This module only HOLDS state, FLATTENS it into the ABI snapshot (include/hqtick.h) and APPLIES the returned
mapping the way create_task_mapping()/process_proactive_filling() mutate `Core` (scheduler/mapping.rs:23-234).
All scheduling decisions come from the backend passed to `schedule()`: the HIP library (hyperqueue_amd.tick)
or — in tests only — the CPU oracle.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import abi
from .hbmap import task_id_set, worker_id_set

FR = abi.HQ_FRACTIONS_PER_UNIT


def amount(x) -> int:
    """ResourceAmount from units (int), fractional units (float) or a pre-scaled ('raw', n) tuple."""
    if isinstance(x, tuple) and x[0] == "raw":
        return int(x[1])
    return int(round(x * FR))


def priority_from_user(user_priority: int) -> int:
    """Priority::from_user_priority  common/priority.rs:43-47."""
    return (((user_priority & 0xFFFF_FFFF_FFFF_FFFF) ^ 0x8000_0000) << 32) & 0xFFFF_FFFF_FFFF_FFFF


def task_id(job_id: int, job_task_id: int) -> int:
    """TaskId{job_id, job_task_id} packed so that integer order == `Ord`  common/ids.rs:17-21."""
    return (job_id << 32) | job_task_id


# ------------------------------------------------------------------------------------------------------
# builders (tests/utils/resources.rs, tests/utils/task.rs, tests/utils/worker.rs)
# ------------------------------------------------------------------------------------------------------
class ResBuilder:
    def __init__(self):
        self.n_nodes_ = 0
        self.entries: List[Tuple[int, int, int]] = []  # (resource id, kind, amount)
        self.min_time_ns = 0
        self.weight_ = 10_000

    def copy(self) -> "ResBuilder":
        r = ResBuilder()
        r.n_nodes_, r.entries, r.min_time_ns, r.weight_ = self.n_nodes_, list(self.entries), self.min_time_ns, self.weight_
        return r

    def finish(self) -> dict:
        entries = list(self.entries)
        if not any(e[0] == 0 for e in entries):  # "Add 1 cpu if no cpu exists"  resources.rs:105-115
            entries.insert(0, (0, abi.HQ_ENTRY_AMOUNT, FR))
        entries.sort(key=lambda e: e[0])  # ResourceRequest::new sorts by resource id  request.rs:144
        return dict(entries=entries, n_nodes=self.n_nodes_, min_time_ns=self.min_time_ns, weight=self.weight_)


class TaskBuilder:
    """tests/utils/task.rs:23-101."""

    def __init__(self):
        self.finished: List[dict] = []
        self.rb = ResBuilder()
        self.user_priority_ = 0
        self.deps: List[int] = []

    def _c(self) -> "TaskBuilder":
        t = TaskBuilder()
        t.finished, t.rb, t.user_priority_, t.deps = list(self.finished), self.rb.copy(), self.user_priority_, list(self.deps)
        return t

    def user_priority(self, p: int) -> "TaskBuilder":
        t = self._c(); t.user_priority_ = p; return t

    def task_deps(self, deps) -> "TaskBuilder":
        t = self._c(); t.deps = list(deps); return t

    def next_variant(self) -> "TaskBuilder":
        t = self._c(); t.finished.append(t.rb.finish()); t.rb = ResBuilder(); return t

    def n_nodes(self, n: int) -> "TaskBuilder":
        t = self._c(); t.rb.n_nodes_ = n; return t

    def cpus(self, n) -> "TaskBuilder":
        return self.add_resource(0, n)

    def cpus_all(self) -> "TaskBuilder":
        t = self._c(); t.rb.entries.append((0, abi.HQ_ENTRY_ALL, 0)); return t

    def add_resource(self, res_id: int, n) -> "TaskBuilder":
        t = self._c(); t.rb.entries.append((res_id, abi.HQ_ENTRY_AMOUNT, amount(n))); return t

    def add_all(self, res_id: int) -> "TaskBuilder":
        t = self._c(); t.rb.entries.append((res_id, abi.HQ_ENTRY_ALL, 0)); return t

    def weight(self, w: float) -> "TaskBuilder":
        t = self._c(); t.rb.weight_ = int(round(np.float32(w) * np.float32(10_000))); return t  # ResourceWeight::try_from request.rs:112-119

    def time_request(self, secs: int) -> "TaskBuilder":
        t = self._c(); t.rb.min_time_ns = secs * 1_000_000_000; return t

    def build_rqv(self) -> Tuple:
        vs = self.finished + [self.rb.finish()]
        return tuple((tuple(v["entries"]), v["n_nodes"], v["min_time_ns"], v["weight"]) for v in vs)


class WorkerBuilder:
    """tests/utils/worker.rs:8-92."""

    def __init__(self, cpus: Optional[int] = None):
        self.res: List[Tuple[str, int]] = [] if cpus is None else [("cpus", amount(cpus))]
        self.time_limit_ns: Optional[int] = None
        self.group_ = "default"
        self.min_utilization_ = 0.0

    @staticmethod
    def empty() -> "WorkerBuilder":
        return WorkerBuilder(None)

    def _c(self) -> "WorkerBuilder":
        w = WorkerBuilder(None)
        w.res, w.time_limit_ns, w.group_, w.min_utilization_ = list(self.res), self.time_limit_ns, self.group_, self.min_utilization_
        return w

    def res_sum(self, name: str, n) -> "WorkerBuilder":
        w = self._c(); w.res.append((name, amount(n))); return w

    def res_sum_raw(self, name: str, raw: int) -> "WorkerBuilder":
        w = self._c(); w.res.append((name, raw)); return w

    def res_range(self, name: str, start: int, end: int) -> "WorkerBuilder":
        w = self._c(); w.res.append((name, amount(end + 1 - start) if end >= start else 0)); return w  # descriptor.rs:140-143

    def time_limit_s(self, s: float) -> "WorkerBuilder":
        w = self._c(); w.time_limit_ns = int(s * 1_000_000_000); return w

    def group(self, g: str) -> "WorkerBuilder":
        w = self._c(); w.group_ = g; return w

    def min_utilization(self, v: float) -> "WorkerBuilder":
        w = self._c(); w.min_utilization_ = v; return w


# ------------------------------------------------------------------------------------------------------
# state
# ------------------------------------------------------------------------------------------------------
WAITING, ASSIGNED, RUNNING, PREFILLED, RETRACTING, RUNNING_MN, FINISHED = range(7)


@dataclass
@dataclass
class WorkerTypeQuery:
    """control.rs `WorkerTypeQuery` (crates/tako/src/control.rs:98-123) with the descriptor reduced to (name, units) sums."""

    resources: List[Tuple[str, float]] = field(default_factory=list)
    partial: bool = False
    time_limit_s: Optional[float] = None
    max_sn_workers: int = 1
    max_workers_per_allocation: int = 1
    min_utilization: float = 0.0

    @staticmethod
    def cpus(n, **kw) -> "WorkerTypeQuery":
        return WorkerTypeQuery(resources=[("cpus", n)], **kw)


@dataclass
class Task:
    id: int
    rq: int
    priority: int
    state: int = WAITING
    worker: Optional[int] = None  # worker id (Assigned/Running/Prefilled/Retracting)
    rv: Optional[int] = None
    mn_workers: Optional[List[int]] = None
    unfinished_deps: int = 0
    consumers: List[int] = field(default_factory=list)

    def is_waiting(self): return self.state == WAITING
    def is_assigned(self): return self.state == ASSIGNED
    def is_sn_running(self): return self.state == RUNNING
    def is_prefilled(self): return self.state == PREFILLED
    def is_retracting(self): return self.state == RETRACTING
    def is_mn_running(self): return self.state == RUNNING_MN


@dataclass
class Worker:
    id: int
    total: List[int]  # padded lazily to n_resources when flattened (workerload.rs:51-75)
    free: List[int]
    termination_ns: Optional[int]  # absolute, on the env clock
    group: str
    min_utilization: float
    assigned_tasks: set = field(default_factory=set)
    prefilled_tasks: set = field(default_factory=set)
    blocked_requests: set = field(default_factory=set)
    mn_task: Optional[Tuple[int, bool]] = None
    stopping: bool = False

    def sn(self) -> bool:
        return self.mn_task is None


class SchedEnv:
    """Mirror of tests/utils/env.rs `TestEnv` (ids: workers from 50, tasks from 1 in job 1)."""

    def __init__(self, config: Optional[abi.Config] = None):
        self.config = config or abi.make_config()
        self.resource_names: Dict[str, int] = {"cpus": 0}  # map.rs:22-32
        self.rq_ids: Dict[Tuple, int] = {}
        self.requests: List[List[dict]] = []
        self.tasks: Dict[int, Task] = {}
        self.workers: Dict[int, Worker] = {}
        self.worker_map = worker_id_set()  # iteration order of core.worker_map
        self.ready: Dict[int, set] = {}  # rq -> ids in TaskQueue.queue
        self.prefill: Dict[int, Tuple[int, object]] = {}  # rq -> (priority, HbSet of task ids)
        self.redirects: Dict[int, Tuple[int, int]] = {}
        self.retaken_variant: Dict[int, int] = {}  # Retracting task put back on its own worker by a tick -> the variant chosen
        self.retract_messages: List[Tuple[int, int]] = []  # (worker, task) of RetractTasks sent outside the tick (prefill disposal)
        self.groups: Dict[str, int] = {}
        self.job_id = 1
        self.task_id_counter = 1
        self.worker_id_counter = 50
        self.now_ns = 0
        self.last_result: Optional[abi.Result] = None

    # -- resources / requests ------------------------------------------------------------------------
    def new_named_resource(self, name: str) -> int:
        if name not in self.resource_names:
            self.resource_names[name] = len(self.resource_names)
        return self.resource_names[name]

    def new_generic_resource(self, count: int):
        for i in range(count):
            self.new_named_resource(f"Res{i}")

    def rq_id(self, builder: TaskBuilder) -> int:
        key = builder.build_rqv()
        if key not in self.rq_ids:
            self.rq_ids[key] = len(self.requests)
            self.requests.append([dict(entries=list(v[0]), n_nodes=v[1], min_time_ns=v[2], weight=v[3]) for v in key])
            self.ready[self.rq_ids[key]] = set()
        return self.rq_ids[key]

    @property
    def n_resources(self) -> int:
        # GlobalResourceMapping::n_resources() counts NAMED resources (map.rs:76); a request may still carry an id nobody named
        # (tests/test_query.rs:735-755) — workers simply hold 0 of it.  The ABI wants every entry id < R, so R covers both.
        used = max([e[0] + 1 for vs in self.requests for v in vs for e in v["entries"]], default=0)
        return max(len(self.resource_names), used)

    # -- tasks -----------------------------------------------------------------------------------------
    def new_task(self, builder: Optional[TaskBuilder] = None) -> int:
        builder = builder or TaskBuilder()
        tid = task_id(self.job_id, self.task_id_counter)
        self.task_id_counter += 1
        rq = self.rq_id(builder)
        t = Task(tid, rq, priority_from_user(builder.user_priority_))
        for d in builder.deps:
            if self.tasks[d].state != FINISHED:
                t.unfinished_deps += 1
                self.tasks[d].consumers.append(tid)
        self.tasks[tid] = t
        if t.unfinished_deps == 0:
            self._add_ready(t)
        return tid

    def _add_ready(self, t: Task):
        """TaskQueues::add_ready_task  taskqueue.rs:37-43: a higher-priority arrival first dissolves every prefill set of lower
        priority (check_dispose_prefill :148-154): its tasks go back into their queue in state Retracting{worker}, leave the worker's
        prefilled set, and a RetractTasks message goes to the worker (process_retracted, server/reactor.rs:34-62)."""
        for rq, (p, s) in list(self.prefill.items()):
            if len(s) and p < t.priority:
                for tid in list(s):
                    pt = self.tasks[tid]
                    assert pt.state == PREFILLED
                    self.workers[pt.worker].prefilled_tasks.discard(tid)
                    pt.state = RETRACTING
                    self.ready[rq].add(tid)
                    self.retract_messages.append((pt.worker, tid))
                del self.prefill[rq]
        self.ready[t.rq].add(t.id)

    def retract_response(self, wid: int, task_ids: List[int]):
        """on_retract_response  server/reactor.rs:462-508: a redirected task becomes Assigned on its target, any other goes back to
        Waiting (it already sits in its queue)."""
        for tid in task_ids:
            t = self.tasks[tid]
            if not (t.state == RETRACTING and t.worker == wid):
                continue
            if tid in self.redirects:
                target, v = self.redirects.pop(tid)
                t.state, t.worker, t.rv = ASSIGNED, target, v
            else:
                t.state, t.worker = WAITING, None

    def new_tasks(self, n: int, builder: Optional[TaskBuilder] = None) -> List[int]:
        return [self.new_task(builder) for _ in range(n)]

    def new_task_cpus(self, cpus) -> int:
        return self.new_task(TaskBuilder().cpus(cpus))

    def new_tasks_cpus(self, cpus: List[int]) -> List[int]:
        return [self.new_task_cpus(c) for c in cpus]

    def task(self, tid: int) -> Task:
        return self.tasks[tid]

    # -- workers ---------------------------------------------------------------------------------------
    def new_worker(self, builder: WorkerBuilder) -> int:
        wid = self.worker_id_counter
        self.worker_id_counter += 1
        for name, _ in builder.res:
            self.new_named_resource(name)
        n = max([self.resource_names[name] + 1 for name, _ in builder.res], default=0)
        total = [0] * n
        for name, a in builder.res:
            total[self.resource_names[name]] = a
        term = None if builder.time_limit_ns is None else self.now_ns + builder.time_limit_ns
        self.workers[wid] = Worker(wid, total, list(total), term, builder.group_, builder.min_utilization_)
        self.worker_map.insert(wid)
        if builder.group_ not in self.groups:
            self.groups[builder.group_] = len(self.groups)
        return wid

    def new_workers(self, n: int, builder: WorkerBuilder) -> List[int]:
        return [self.new_worker(builder) for _ in range(n)]

    def remove_worker(self, wid: int) -> List[Tuple[int, int, int]]:
        """on_remove_worker  server/reactor.rs:64-147, what the scheduler state sees of it.  The lost worker's assigned / running tasks go back to their queues
        as Waiting (add_ready_task: may dissolve lower-priority prefill sets); a task that is Retracting{old} and was REDIRECTED to the lost worker loses the
        redirect and goes back into its queue, still Retracting{old} (:89-94); its prefilled tasks move from the prefill set into the queue (:97-104).  Then every
        task Retracting{lost worker}: with a redirect it becomes Assigned{target} and the target gets its ComputeTasks message, without one it is Waiting
        (:124-145).  Returns those messages as (task, target worker id, variant)."""
        w = self.workers.pop(wid)
        self.worker_map.remove(wid)
        if w.sn():
            for tid in sorted(w.assigned_tasks):
                t = self.tasks[tid]
                if t.state == RETRACTING:
                    assert self.redirects.pop(tid, None) is not None
                else:
                    assert t.state in (ASSIGNED, RUNNING) and t.worker == wid
                    t.state, t.worker, t.rv = WAITING, None, None
                self._add_ready(t)
            for tid in sorted(w.prefilled_tasks):
                t = self.tasks[tid]
                assert t.state == PREFILLED and t.worker == wid
                t.state, t.worker = WAITING, None
                self.prefill[t.rq][1].remove(tid)  # move_prefilled_task_to_ready  taskqueue.rs:263-271: plain add(), no prefill disposal
                self.ready[t.rq].add(tid)
        else:
            tid, _is_root = w.mn_task
            t = self.tasks[tid]
            assert t.state == RUNNING_MN
            if t.mn_workers[0] == wid:  # root: the task returns to its queue, the other nodes are free again
                for other in t.mn_workers[1:]:
                    if other in self.workers:
                        self.workers[other].mn_task = None
                t.state, t.mn_workers = WAITING, None
                self._add_ready(t)
            else:
                t.mn_workers = [x for x in t.mn_workers if x != wid]
        sent = []
        for t in sorted(self.tasks.values(), key=lambda t: t.id):
            if t.state == RETRACTING and t.worker == wid:
                if t.id in self.redirects:
                    target, v = self.redirects.pop(t.id)
                    t.state, t.worker, t.rv = ASSIGNED, target, v
                    sent.append((t.id, target, v))
                else:
                    t.state, t.worker = WAITING, None
                self.retaken_variant.pop(t.id, None)
        return sent

    def new_worker_cpus(self, cpus: int) -> int:
        return self.new_worker(WorkerBuilder(cpus))

    def new_workers_cpus(self, cpus: List[int]) -> List[int]:
        return [self.new_worker_cpus(c) for c in cpus]

    def worker(self, wid: int) -> Worker:
        return self.workers[wid]

    def worker_tasks(self, wid: int) -> set:
        return self.workers[wid].assigned_tasks

    # -- WorkerResources arithmetic (server/workerload.rs:156-200) ------------------------------------
    def _variant(self, rq: int, v: int) -> dict:
        return self.requests[rq][v]

    @staticmethod
    def _pad(vec: List[int], n: int):
        while len(vec) < n:
            vec.append(0)

    def _remove(self, w: Worker, rq: int, v: int):
        for (r, kind, a) in self._variant(rq, v)["entries"]:
            self._pad(w.free, r + 1)
            w.free[r] = max(0, w.free[r] - a) if kind == abi.HQ_ENTRY_AMOUNT else 0

    def _add(self, w: Worker, rq: int, v: int):
        for (r, kind, a) in self._variant(rq, v)["entries"]:
            self._pad(w.free, r + 1)
            self._pad(w.total, r + 1)
            w.free[r] = w.free[r] + a if kind == abi.HQ_ENTRY_AMOUNT else w.total[r]

    # -- TestEnv task life-cycle helpers (tests/utils/env.rs:156-248) -----------------------------------
    def assign_task(self, tid: int, wid: int, variant: int = 0):
        t = self.tasks[tid]
        assert t.state == WAITING and t.unfinished_deps == 0
        t.state, t.worker, t.rv = ASSIGNED, wid, variant
        w = self.workers[wid]
        self._remove(w, t.rq, variant)  # insert_sn_task  server/worker.rs:188-196
        w.assigned_tasks.add(tid)
        self.ready[t.rq].discard(tid)

    def start_task(self, tid: int, variant: int = 0):
        t = self.tasks[tid]
        assert t.state == ASSIGNED
        t.state, t.rv = RUNNING, variant

    def assign_and_start_task(self, tid: int, wid: int, variant: int = 0):
        self.assign_task(tid, wid, variant)
        self.start_task(tid, variant)

    def new_task_assigned(self, builder: TaskBuilder, wid: int) -> int:
        tid = self.new_task(builder)
        self.assign_task(tid, wid)
        return tid

    def new_task_running(self, builder: TaskBuilder, wid: int) -> int:
        tid = self.new_task_assigned(builder, wid)
        self.start_task(tid, 0)
        return tid

    def finish_task(self, tid: int, wid: int):
        """on_task_update(Finished) restricted to what the tick observes  server/reactor.rs:510-590."""
        t = self.tasks[tid]
        if t.state == RUNNING_MN:
            for w_id in t.mn_workers:
                self.workers[w_id].mn_task = None  # reset_mn_task  server/worker.rs:168-171
        else:
            assert t.state in (ASSIGNED, RUNNING) and t.worker == wid
            w = self.workers[wid]
            w.assigned_tasks.discard(tid)
            self._add(w, t.rq, t.rv)  # remove_sn_task  server/worker.rs:223-234
        t.state = FINISHED
        for c in t.consumers:
            ct = self.tasks[c]
            ct.unfinished_deps -= 1
            if ct.unfinished_deps == 0 and ct.state == WAITING:
                self._add_ready(ct)

    def start_task_mn(self, tid: int, workers: List[int]):
        t = self.tasks[tid]
        for i, w in enumerate(workers):
            self.workers[w].mn_task = (tid, i == 0)
        t.state, t.mn_workers = RUNNING_MN, list(workers)
        self.ready[t.rq].discard(tid)

    def block_request(self, wid: int, rq: int, variant: int):
        self.workers[wid].blocked_requests.add((rq, variant))

    def reject_task(self, tid: int, wid: int, variant: Optional[int] = 0):
        """on_task_update(RejectRequest) = task_reject  server/reactor.rs:365-442, the Assigned and Prefilled arms: the worker blocks the
        (request, variant), gives the task's resources back, and the task returns to its queue as Waiting (the Retracting arm with a redirect
        is `retract_response`'s twin and is not needed by the scheduler tests)."""
        t = self.tasks[tid]
        w = self.workers[wid]
        if variant is not None:
            w.blocked_requests.add((t.rq, variant))
        if t.state == ASSIGNED:
            if t.worker == wid and variant == t.rv:
                w.assigned_tasks.discard(tid)
                self._add(w, t.rq, t.rv)  # remove_sn_task  server/worker.rs:223-234
        elif t.state == PREFILLED:
            w.prefilled_tasks.discard(tid)
            p, sset = self.prefill[t.rq]
            sset.remove(tid)
        else:
            raise AssertionError("unreachable in the reference (reactor.rs:434-439)")
        t.state, t.worker = WAITING, None
        self._add_ready(t)

    def enable_request(self, wid: int, rq: int, variant: int):
        """on_task_update(EnableRequest) = request_enabled  server/reactor.rs:444-458."""
        self.workers[wid].blocked_requests.discard((rq, variant))

    # -- flatten ---------------------------------------------------------------------------------------
    def snapshot(self) -> abi.Snapshot:
        R = self.n_resources
        wids = sorted(self.workers)
        index = {w: i for i, w in enumerate(wids)}
        W = len(wids)
        total = np.zeros((W, R), np.uint64)
        free = np.zeros((W, R), np.uint64)
        rem = np.full(W, abi.HQ_NO_TIME_LIMIT, np.int64)
        mu = np.zeros(W, np.float32)
        flags = np.zeros(W, np.uint8)
        group = np.zeros(W, np.uint32)
        assigned, prefilled, blocked = [], [], []
        for i, wid in enumerate(wids):
            w = self.workers[wid]
            total[i, : len(w.total)] = w.total
            free[i, : len(w.free)] = w.free
            if w.termination_ns is not None:
                rem[i] = w.termination_ns - self.now_ns
            mu[i] = w.min_utilization
            flags[i] = (abi.HQ_WORKER_SN if w.sn() else 0) | (abi.HQ_WORKER_STOPPING if w.stopping else 0)
            group[i] = self.groups[w.group]
            assigned.append([(self.tasks[t].rq, self._assigned_variant(t, wid)) for t in sorted(w.assigned_tasks)] if w.sn() else [])
            prefilled.append([self.tasks[t].rq for t in sorted(w.prefilled_tasks)] if w.sn() else [])
            for (rq, v) in sorted(w.blocked_requests):
                blocked.append((i, rq, v))
        rank = np.zeros(W, np.uint32)
        for pos, wid in enumerate(self.worker_map):
            rank[index[wid]] = pos
        ids, prio, rqs = [], [], []
        for rq, s in self.ready.items():
            for t in s:
                ids.append(t); prio.append(self.tasks[t].priority); rqs.append(rq)
        ids = np.asarray(ids, np.uint64)
        order = np.argsort(ids, kind="stable")
        retracting = []
        for rq, ids_ in self.ready.items():
            for tid in ids_:
                tt = self.tasks[tid]
                if tt.state == RETRACTING:
                    tw, tv = self.redirects.get(tid, (None, 0))
                    retracting.append((tid, index[tt.worker], abi.HQ_NO_WORKER if tw is None else index[tw], tv))
        retracting.sort()
        prefill = {}
        for rq, (p, s) in self.prefill.items():
            if len(s):
                prefill[rq] = (p, [(t, index[self.tasks[t].worker]) for t in s])
        snap = abi.Snapshot(
            n_resources=R, worker_id=np.asarray(wids, np.uint32), worker_total=total, worker_free=free,
            worker_remaining_ns=rem, worker_min_utilization=mu, worker_flags=flags, worker_group=group,
            n_groups=max(1, len(self.groups)), blocked=blocked, assigned=assigned, prefilled=prefilled,
            requests=self.requests, task_id=ids[order], task_priority=np.asarray(prio, np.uint64)[order],
            task_rq=np.asarray(rqs, np.uint32)[order], prefill=prefill, worker_map_rank=rank, retracting=retracting,
        )
        snap.config = self.config
        return snap


    def _assigned_variant(self, tid: int, wid: int) -> int:
        t = self.tasks[tid]
        if t.state == RETRACTING:  # sanity_check: Retracting tasks count on their redirect target  server/worker.rs:249-252
            if tid in self.redirects:
                return self.redirects[tid][1]
            return self.retaken_variant[tid]  # retaken by the worker it is retracting from: no redirect entry (mapping.rs:69)
        return t.rv

    # -- apply -----------------------------------------------------------------------------------------
    def apply(self, res: abi.Result):
        """What create_task_mapping + process_proactive_filling leave behind in `Core` (mapping.rs:36-234)."""
        wids = sorted(self.workers)
        if res.status < 0:
            raise RuntimeError(f"tick failed: {res.status}")
        # create_task_mapping drains the prefill sets (take_tasks) for every worker before process_proactive_filling refills any
        # (mapping.rs:36-157 then :159-234): all retracts first, then the records
        for i, wid in enumerate(wids):
            w = self.workers[wid]
            for tid in res.retracts[i]:  # Prefilled{old} -> Retracting{old}  mapping.rs:81-101
                t = self.tasks[tid]
                assert t.state == PREFILLED and t.worker == wid
                w.prefilled_tasks.discard(tid)
                t.state = RETRACTING
                p, s = self.prefill[t.rq]
                s.remove(tid)
        for i, wid in enumerate(wids):
            w = self.workers[wid]
            for (tid, v, kind) in res.records[i]:
                t = self.tasks[tid]
                if kind == abi.HQ_REC_ASSIGN:  # Waiting -> Assigned  mapping.rs:53-65
                    assert t.state == WAITING, (tid, t.state)
                    t.state, t.worker, t.rv = ASSIGNED, wid, v
                    self._remove(w, t.rq, v)
                    w.assigned_tasks.add(tid)
                    self.ready[t.rq].discard(tid)
                else:  # prefill  mapping.rs:217-232, taskqueue.rs:304-318
                    assert t.state == WAITING
                    t.state, t.worker = PREFILLED, wid
                    w.prefilled_tasks.add(tid)
                    self.ready[t.rq].discard(tid)
                    if t.rq not in self.prefill or len(self.prefill[t.rq][1]) == 0:
                        self.prefill[t.rq] = (t.priority, task_id_set())
                    assert self.prefill[t.rq][0] == t.priority
                    self.prefill[t.rq][1].insert(tid)
        kinds = res.redirect_kinds or [abi.HQ_REDIRECT_FROM_PREFILL] * len(res.redirects)
        for (tid, widx, v), kind in zip(res.redirects, kinds):  # insert_sn_task on the new target + the redirect table  mapping.rs:51,66-100
            wid = wids[widx]
            t = self.tasks[tid]
            w = self.workers[wid]
            if kind != abi.HQ_REDIRECT_FROM_PREFILL:  # the task was Retracting in its queue: take_tasks removed it
                assert t.state == RETRACTING and tid in self.ready[t.rq], (tid, t.state)
                self.ready[t.rq].discard(tid)
            if kind == abi.HQ_REDIRECT_RETARGET and tid in self.redirects:  # remove_sn_task on the previous target  mapping.rs:70-77
                ow, ov = self.redirects[tid]
                self.workers[ow].assigned_tasks.discard(tid)
                self._add(self.workers[ow], t.rq, ov)
            if kind != abi.HQ_REDIRECT_SAME_WORKER:
                self.redirects[tid] = (wid, v)
            else:
                self.retaken_variant[tid] = v
            self._remove(w, t.rq, v)
            w.assigned_tasks.add(tid)
        for (tid, widxs) in res.mn:  # mapping.rs:133-154
            t = self.tasks[tid]
            assert t.state == WAITING
            ws = [wids[k] for k in widxs]
            for k, wid in enumerate(ws):
                assert self.workers[wid].sn() and not self.workers[wid].assigned_tasks
                self.workers[wid].mn_task = (tid, k == 0)
            t.state, t.mn_workers = RUNNING_MN, ws
            self.ready[t.rq].discard(tid)
        # the ABI also reports the free vectors it computed: they must agree with ours
        R = self.n_resources
        for i, wid in enumerate(wids):
            w = self.workers[wid]
            if w.sn():
                mine = list(w.free) + [0] * (R - len(w.free))
                assert mine == [int(x) for x in res.new_free[i]], (wid, mine, res.new_free[i])
        self.last_result = res

    def schedule(self, backend) -> abi.Result:
        """run_scheduling_inner  scheduler/main.rs:50-72 through a backend exposing tick(Snapshot) -> Result."""
        res = backend.tick(self.snapshot())
        self.apply(res)
        return res

    def new_worker_query(self, backend, queries: List[WorkerTypeQuery]):
        """compute_new_worker_query  scheduler/query.rs:12-131: fake workers per query (ids above every real id, `MAX` amounts
        for the resources a partial query does not name), batches + solver through `backend.query`, then the counts per query
        and the multi-node allocations (pure queue bookkeeping, query.rs:97-124).
        Returns (single_node_workers_per_query, [(worker_type, worker_per_allocation, max_allocations)])."""
        for q in queries:  # query.rs:22-26: every named resource gets an id
            for name, _ in q.resources:
                self.new_named_resource(name)
        R = self.n_resources
        names = sorted(self.resource_names, key=lambda n: self.resource_names[n])
        ids, totals, rem, mu = [], [], [], []
        next_id = max([self.worker_id_counter] + list(self.workers)) + 1
        for q in queries:
            for _ in range(q.max_sn_workers):
                named = set(self.resource_names.values())
                row = [abi.HQ_AMOUNT_MAX if (q.partial and r in named) else 0 for r in range(R)]  # query.rs:33-47: only NAMED resources
                for name, units in q.resources:
                    row[self.resource_names[name]] = amount(units)
                ids.append(next_id); next_id += 1
                totals.append(row)
                rem.append(abi.HQ_NO_TIME_LIMIT if q.time_limit_s is None else int(round(q.time_limit_s * 1e9)))
                mu.append(q.min_utilization)
        snap = self.snapshot()
        if ids:
            loaded, _opt = backend.query(snap, np.asarray(ids, np.uint32), np.asarray(totals, np.uint64).reshape(len(ids), R), np.asarray(rem, np.int64), np.asarray(mu, np.float32))
        else:
            loaded = np.zeros(0, bool)
        counts, k = [], 0
        for q in queries:
            counts.append(int(np.sum(loaded[k:k + q.max_sn_workers]))); k += q.max_sn_workers
        allocs = []
        for rq, variants in enumerate(self.requests):  # task_queues.iter(): one queue per request id
            v0 = variants[0]
            if v0["n_nodes"] == 0:
                continue
            for i, q in enumerate(queries):
                if q.time_limit_s is not None and v0["min_time_ns"] > int(round(q.time_limit_s * 1e9)):
                    continue
                if q.max_workers_per_allocation >= v0["n_nodes"]:
                    allocs.append((i, v0["n_nodes"], len(self.ready[rq])))
                    break
        allocs.sort(key=lambda a: (a[0], a[1]))
        return counts, allocs

    def cancel_task(self, tid: int):
        """on_cancel_tasks restricted to what the tick observes (server/reactor.rs:706-790): a waiting task leaves its queue."""
        t = self.tasks[tid]
        assert t.state == WAITING
        self.ready[t.rq].discard(tid)
        t.state = FINISHED

    # -- bookkeeping used by tests ----------------------------------------------------------------------
    def assigned_counts(self) -> List[int]:
        counts = [0] * len(self.requests)
        for t in self.tasks.values():
            if t.state == ASSIGNED:
                counts[t.rq] += 1
        return counts

    def prefill_count(self, wid: int) -> int:
        n = len(self.workers[wid].prefilled_tasks)
        assert n == sum(1 for t in self.tasks.values() if t.state == PREFILLED and t.worker == wid)
        return n

    def queue_priority_sizes(self, rq: int) -> List[Tuple[int, int]]:
        """TaskQueue::iter_priority_sizes  taskqueue.rs:273-302."""
        levels: Dict[int, int] = {}
        for t in self.ready[rq]:
            p = self.tasks[t].priority
            levels[p] = levels.get(p, 0) + 1
        out = sorted(levels.items(), key=lambda kv: -kv[0])
        if rq in self.prefill and len(self.prefill[rq][1]):
            p, s = self.prefill[rq]
            if out and out[0][0] == p:
                out[0] = (p, out[0][1] + len(s))
            else:
                out.insert(0, (p, len(s)))
        return out
