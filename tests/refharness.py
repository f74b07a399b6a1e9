"""Python rendition of the reference's declarative scheduler-test DSL
(/root/reference/crates/tako/src/internal/tests/utils/scheduler.rs:10-219) on top of hyperqueue_amd.core.SchedEnv,
so the transcribed golden tests read like the reference's own.  Works with any backend exposing
`tick(Snapshot) -> Result` (the CPU oracle or the HIP library).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

from hyperqueue_amd.core import SchedEnv, TaskBuilder, WorkerBuilder


class TestWorker:
    __test__ = False

    def __init__(self, case: "TestCase", wid: int):
        self.case, self.wid = case, wid
        self.expect: Tuple[str, object] = ("empty", None)
        self.eq: Optional[int] = None

    def eq_class(self, k: int) -> "TestWorker":
        self.eq = k
        return self

    def expect_tasks(self, tasks: List[int]) -> "TestWorker":
        self.expect = ("tasks", [(t, 0) for t in tasks])
        return self

    def expect_request(self, count: int, builder: TaskBuilder) -> "TestWorker":
        return self.expect_request_v(count, builder, 0)

    def expect_request_v(self, count: int, builder: TaskBuilder, variant: int) -> "TestWorker":
        rq = self.case.rt.rq_id(builder)
        if self.expect[0] != "requests":
            self.expect = ("requests", {})
        m: Dict[Tuple[int, int], int] = self.expect[1]
        m[(rq, variant)] = m.get((rq, variant), 0) + count
        return self

    def running(self, builder: TaskBuilder) -> "TestWorker":
        self.case.rt.new_task_running(builder, self.wid)
        return self

    def running_c(self, cpus: int) -> "TestWorker":
        return self.running(TaskBuilder().cpus(cpus))

    def check(self, assigned: List[Tuple[int, int]]):
        kind, val = self.expect
        if kind == "empty":
            assert assigned == [], (self.wid, assigned)
        elif kind == "tasks":
            assert assigned == val, (self.wid, assigned, val)
        else:
            got: Dict[Tuple[int, int], int] = {}
            for (t, v) in assigned:
                k = (self.case.rt.task(t).rq, v)
                got[k] = got.get(k, 0) + 1
            assert got == val, (self.wid, got, val)


class TestCase:
    __test__ = False

    def __init__(self, backend):
        self.rt = SchedEnv()
        self.backend = backend
        self.workers: List[TestWorker] = []

    def resources(self, names: List[str]) -> "TestCase":
        for n in names:
            self.rt.new_named_resource(n)
        return self

    def w(self, builder: WorkerBuilder) -> TestWorker:
        tw = TestWorker(self, self.rt.new_worker(builder))
        self.workers.append(tw)
        return tw

    def t(self, builder: TaskBuilder) -> int:
        return self.rt.new_task(builder)

    def ts(self, n: int, builder: TaskBuilder) -> List[int]:
        return [self.t(builder) for _ in range(n)]

    def c_tasks(self, cpus: List[int]) -> List[int]:
        return self.rt.new_tasks_cpus(cpus)

    def pc_tasks(self, pc: List[Tuple[int, int]]) -> List[int]:
        return [self.rt.new_task(TaskBuilder().cpus(c).user_priority(p)) for (p, c) in pc]

    def check(self):
        """TestCase::check  scheduler.rs:34-66 (schedule_mapping + eq-class normalisation)."""
        res = self.rt.schedule(self.backend)
        wids = sorted(self.rt.workers)
        assigned = {wid: res.assigned(i) for i, wid in enumerate(wids)}
        for cls in {w.eq for w in self.workers if w.eq is not None}:
            ids = [w.wid for w in self.workers if w.eq == cls]
            lists = sorted(assigned[i] for i in ids)  # normalize_workers  scheduler.rs:97-106
            for i, l in zip(ids, lists):
                assigned[i] = l
        for w in self.workers:
            w.check(assigned[w.wid])
        return res
