"""Dependency-graph scenarios transcribed from the reference's own tests (crates/tako/src/internal), shared by the oracle
test (CPU) and the device test (GPU).  A scenario is a list of steps; every step names the reference call it stands for and the
values the reference test asserts after it.

  ("add", [(id, deps...)...], {"unfinished": {id: n}, "ready": [ids]})        on_new_tasks
  ("take", [ids])                                                              the scheduler handed these out (assign_task)
  ("finish", [ids], {"released": [ids], "unfinished": {...}})                  on_task_update(Finished)
  ("fail", id, {"removed": [ids]})                                             task_failed -> recursive consumers removed
  ("collect", id, {"consumers": [ids]})                                        Task::collect_recursive_consumers (graph unchanged)
"""


def tid(job, task):
    return (job << 32) | task


def _example_1():
    # tests/utils/workflows.rs:5-23
    t = [tid(1, i + 1) for i in range(7)]
    deps = [[], [], [t[0], t[1]], [t[1]], [t[2], t[3]], [t[2]], [t[5]]]
    return t, deps


def _example_3():
    # tests/utils/workflows.rs:25-42
    t = [tid(1, i + 1) for i in range(6)]
    deps = [[], [], [t[0]], [t[0], t[1]], [t[1]], [t[0], t[2], t[4]]]
    return t, deps


def submit_jobs():
    # tests/test_reactor.rs:141-193
    t1, t2 = tid(100, 501), tid(100, 502)
    t3, t4, t5, t6 = tid(100, 604), tid(100, 503), tid(100, 603), tid(100, 601)
    return [
        ("add", [(t1, []), (t2, [t1])], {"unfinished": {t1: 0, t2: 1}, "ready": [t1]}),
        ("add", [(t3, []), (t4, [t1, t3]), (t5, [t3]), (t6, [t3, t4, t5, t2])], {"unfinished": {t1: 0, t2: 1, t4: 2, t6: 4, t3: 0, t5: 1}, "ready": [t3]}),
    ]


def task_deps():
    # tests/test_reactor.rs:635-648: new_task() submits one task per on_new_tasks call
    t, deps = _example_3()
    steps = [("add", [(t[i], deps[i])], {"unfinished": {t[i]: len(deps[i])}, "ready": [t[i]] if not deps[i] else []}) for i in range(6)]
    steps += [
        ("take", [t[1]]),
        ("finish", [t[1]], {"released": [t[4]], "unfinished": {t[2]: 1, t[3]: 1, t[5]: 3, t[4]: 0}}),   # assert_waiting(2,3,5) assert_ready(4)
        ("take", [t[0]]),
        ("finish", [t[0]], {"released": [t[2], t[3]], "unfinished": {t[5]: 2, t[2]: 0, t[3]: 0, t[4]: 0}}),  # assert_waiting(5) assert_ready(2,3,4)
    ]
    return steps


def running_task_on_error():
    # tests/test_reactor.rs:308-341
    t, deps = _example_1()
    steps = [("add", [(t[i], deps[i])], {"unfinished": {t[i]: len(deps[i])}}) for i in range(7)]
    steps += [
        ("take", [t[0]]), ("finish", [t[0]], {"released": []}),
        ("take", [t[1]]), ("finish", [t[1]], {"released": [t[2], t[3]]}),
        ("take", [t[2]]),
        ("fail", t[2], {"removed": [t[2], t[4], t[5], t[6]]}),   # on_task_error(id = wf[2], consumers = [wf4, wf5, wf6])
        ("exists", {t[3]: True, t[4]: False, t[5]: False, t[6]: False, t[2]: False}),
    ]
    return steps


def recursive_consumers():
    # server/task.rs:475-489
    a, b, c, d, e = (tid(1, i) for i in range(1, 6))
    return [
        ("add", [(a, [])], {}), ("add", [(b, [a])], {}), ("add", [(c, [b])], {}), ("add", [(d, [b])], {}), ("add", [(e, [c, d])], {"unfinished": {e: 2}}),
        ("collect", a, {"consumers": [b, c, d, e]}),
    ]


def assignments_and_finish():
    # the dependency part of tests/test_reactor.rs:199-306:  t1 t2 -> t3 ;  t4 -> t7 ;  t5
    t1, t2, t3, t4, t5, t7 = (tid(1, i) for i in (1, 2, 3, 4, 5, 6))
    return [
        ("add", [(t1, [])], {}), ("add", [(t2, [])], {}), ("add", [(t3, [t1, t2])], {}), ("add", [(t4, [])], {}), ("add", [(t5, [])], {}),
        ("add", [(t7, [t4])], {"unfinished": {t3: 2, t7: 1}}),
        ("take", [t1, t5, t2]),
        ("finish", [t5], {"released": []}), ("exists", {t5: False}),
        ("finish", [t2], {"released": [], "unfinished": {t3: 1}}),
        ("finish", [t1], {"released": [t3]}),
        ("take", [t3, t4]),
        ("finish", [t3], {"released": []}),
    ]


CASES = {
    "submit_jobs": submit_jobs, "task_deps": task_deps, "running_task_on_error": running_task_on_error,
    "recursive_consumers": recursive_consumers, "assignments_and_finish": assignments_and_finish,
}
