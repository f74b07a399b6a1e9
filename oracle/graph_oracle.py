"""TEST INFRASTRUCTURE — CPU restatement of the dependency bookkeeping of tako's reactor.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(hqtick_graph_* in libhqtick.so) never does.

Restates, with plain dicts and sets (crates/tako/src/internal):
  on_new_tasks                 server/reactor.rs:188-220
  Core::add_task               server/core.rs:213-218
  task_finished (consumer loop) server/reactor.rs:570-590
  Task::decrease_unfinished_deps server/task.rs:207-216
  Core::remove_task            server/core.rs:222-240
  Task::collect_recursive_consumers server/task.rs:235-250
Pinned by the reference's own tests transcribed in tests/test_graph_oracle.py (test_submit_jobs, test_task_deps,
test_running_task_on_error, task_recursive_consumers).
"""
from __future__ import annotations

from typing import Iterable, Sequence


class _Task:
    __slots__ = ("id", "priority", "rq", "deps", "consumers", "unfinished")

    def __init__(self, id_, priority, rq, deps):
        self.id, self.priority, self.rq = id_, priority, rq
        self.deps = list(deps)
        self.consumers: set[int] = set()
        self.unfinished = 0


class GraphOracle:
    def __init__(self):
        self.tasks: dict[int, _Task] = {}
        self.ready: dict[int, tuple[int, int]] = {}  # ids sitting in the ready queues -> (priority, rq)

    # reactor.rs:188-220
    def on_new_tasks(self, tasks: Sequence[tuple[int, int, int, Iterable[int]]]) -> list[int]:
        ready_now = []
        for id_, priority, rq, deps in tasks:
            assert id_ not in self.tasks, "core.rs:217 assert!(self.tasks.insert(task).is_none())"
            t = _Task(id_, priority, rq, deps)
            kept = []
            for d in t.deps:  # task_deps.retain(...)  :193-203
                dep = self.tasks.get(d)
                if dep is not None:
                    dep.consumers.add(id_)
                    t.unfinished += 1  # a task in the map is never in state Finished between two events
                    kept.append(d)
            t.deps = kept
            self.tasks[id_] = t
            if t.unfinished == 0:  # Core::add_task -> add_ready_task
                self.ready[id_] = (priority, rq)
                ready_now.append(id_)
        return sorted(ready_now)

    def take_from_ready(self, ids: Iterable[int]):
        """the tick handed these out (take_tasks): they leave the queues but stay in the task map"""
        for i in ids:
            del self.ready[i]

    # reactor.rs:510-590 for a batch of updates, in order
    def task_finished(self, ids: Iterable[int]) -> tuple[list[int], int]:
        released, unknown = [], 0
        for id_ in ids:
            t = self.tasks.get(id_)
            if t is None:  # "Unknown task finished"  :565-567
                unknown += 1
                continue
            assert t.unfinished == 0 and id_ not in self.ready, "reactor.rs:551-555 unreachable!()"
            for c in t.consumers:  # :575-580
                ct = self.tasks[c]
                assert ct.unfinished > 0
                ct.unfinished -= 1
                if ct.unfinished == 0:
                    self.ready[c] = (ct.priority, ct.rq)
                    released.append(c)
            del self.tasks[id_]  # core.remove_task  :587
        return sorted(released), unknown

    # task.rs:235-250
    def collect_recursive_consumers(self, id_: int, out: set[int]):
        stack = [id_]
        while stack:
            t = self.tasks[stack.pop()]
            for c in t.consumers:
                if c not in out:
                    out.add(c)
                    stack.append(c)

    # core.rs:222-240 (+ the recursive collection the cancel / fail paths do first: reactor.rs:669-672, 722-724)
    def remove(self, ids: Iterable[int], recursive: bool) -> tuple[list[int], int]:
        todo: set[int] = set()
        unknown = 0
        for id_ in ids:
            if id_ not in self.tasks:
                unknown += 1
                continue
            todo.add(id_)
            if recursive:
                self.collect_recursive_consumers(id_, todo)
        for id_ in todo:
            t = self.tasks[id_]
            self.ready.pop(id_, None)  # TaskQueue::remove
            if t.unfinished > 0:
                for d in t.deps:
                    dep = self.tasks.get(d)
                    if dep is not None:
                        dep.consumers.discard(id_)
        for id_ in todo:
            del self.tasks[id_]
        return sorted(todo), unknown

    def unfinished(self, id_: int) -> int:
        t = self.tasks.get(id_)
        return 0xFFFFFFFF if t is None else t.unfinished

    # ---- EXTENSION, no reference counterpart (the reference writes no scheduler priority; common/priority.rs:43-66) — parity unpinned -------------
    def blevels(self) -> dict[int, int]:
        """b-level of every task in the map: 0 without a consumer in the map, else 1 + the largest b-level among its consumers (longest path to a sink).
        The checker of hqtick_graph_blevel (include/hqtick.h); a definition, not a restatement: the reference has nothing to restate here."""
        bl: dict[int, int] = {}
        # the consumers of a task are minted after it (a dependency names an existing task): descending id order visits consumers first — but ids need not be
        # minted in order across calls, so this is a plain memoised depth-first walk with an explicit stack
        for root in self.tasks:
            if root in bl:
                continue
            stack = [(root, iter(self.tasks[root].consumers))]
            while stack:
                node, it = stack[-1]
                advanced = False
                for c in it:
                    if c in self.tasks and c not in bl:
                        stack.append((c, iter(self.tasks[c].consumers)))
                        advanced = True
                        break
                if advanced:
                    continue
                live = [bl[c] for c in self.tasks[node].consumers if c in self.tasks]
                bl[node] = 1 + max(live) if live else 0
                stack.pop()
        return bl

    def apply_blevels(self) -> int:
        """Priority's low 32 bits := b-level (what hqtick_graph_blevel does to the device graph, and — with HQTICK_BLEVEL_UPDATE_READY — to the ready set).  Returns the depth."""
        bl = self.blevels()
        for id_, t in self.tasks.items():
            t.priority = (t.priority & 0xFFFFFFFF00000000) | min(bl[id_], 0xFFFFFFFF)
            if id_ in self.ready:
                self.ready[id_] = (t.priority, self.ready[id_][1])
        return max(bl.values(), default=0)
