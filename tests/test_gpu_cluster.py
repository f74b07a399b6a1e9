"""Cluster tables resident in HBM (hqtick_cluster_*, ABI 5; row f1): the tick reads worker rows and request tables from the device copy that the
host keeps current with row deltas, instead of re-packing every worker per tick — same results as the per-tick path on the same snapshots, the
deltas applied, a forgotten delta caught (HQTICK_CHECK_CLUSTER=1), a changed worker set refused, new request classes picked up."""
import dataclasses
import os

import numpy as np
import pytest

from hyperqueue_amd import abi, workloads

pytestmark = pytest.mark.gpu


def _tick(**env):
    from hyperqueue_amd.tick import Tick

    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return Tick(abi.make_config(time_limit_s=20.0))
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _same(a, b):
    assert a.status == b.status and a.is_optimal == b.is_optimal and a.batches == b.batches
    assert a.counts == b.counts and a.records == b.records and a.retracts == b.retracts
    assert (a.new_free == b.new_free).all()


def _with_free(snap, free):
    return dataclasses.replace(snap, _keep=[], worker_free=np.ascontiguousarray(free, np.uint64))


@pytest.mark.parametrize("name,n_workers,seed", [("c3", 48, 0), ("c4", 40, 1)])
def test_resident_cluster_equals_per_tick_packing(name, n_workers, seed):
    plain, res = _tick(), _tick(HQTICK_CHECK_CLUSTER=1)
    snap = workloads.make_steady(name, seed=seed, n_tasks=60_000, n_workers=n_workers)
    res.cluster_upload(snap)
    _same(res.tick(snap), plain.tick(snap))
    # a few workers finish tasks (free goes up to the total), a few start some (free goes down): rows as deltas
    rng = np.random.default_rng(seed)
    W, R = len(snap.worker_id), snap.n_resources
    free = np.array(snap.worker_free, np.uint64).reshape(W, R).copy()
    total = np.array(snap.worker_total, np.uint64).reshape(W, R)
    for step in range(4):
        idx = np.sort(rng.choice(W, size=max(1, W // 5), replace=False)).astype(np.uint32)
        for w in idx:
            free[w] = total[w] if rng.random() < 0.5 else free[w] // np.uint64(2)
        snap2 = _with_free(snap, free.reshape(-1))
        res.cluster_update_workers(idx, free[idx])
        _same(res.tick(snap2), plain.tick(snap2))
    ks = res.kernel_stats()
    assert ks["n_assigned"] >= 0
    plain.close(); res.close()


def test_missed_delta_is_caught_and_changed_worker_set_refused():
    from hyperqueue_amd.tick import HqTickError

    t = _tick(HQTICK_CHECK_CLUSTER=1)
    snap = workloads.make_steady("c3", seed=3, n_tasks=20_000, n_workers=24)
    t.cluster_upload(snap)
    t.tick(snap)
    W, R = len(snap.worker_id), snap.n_resources
    free = np.array(snap.worker_free, np.uint64).reshape(W, R).copy()
    free[5] = np.array(snap.worker_total, np.uint64).reshape(W, R)[5]
    if (free[5] == np.array(snap.worker_free, np.uint64).reshape(W, R)[5]).all():
        free[5] = free[5] // np.uint64(2)
    stale = _with_free(snap, free.reshape(-1))
    with pytest.raises(HqTickError) as e:
        t.tick(stale)
    assert e.value.code == abi.HQTICK_E_INVALID and "cluster tables" in str(e.value)
    t.cluster_update_workers([5], free[5:6])
    t.tick(stale)  # now current
    other = workloads.make_steady("c3", seed=3, n_tasks=20_000, n_workers=25)
    with pytest.raises(HqTickError) as e:
        t.tick(other)
    assert e.value.code == abi.HQTICK_E_INVALID
    with pytest.raises(HqTickError):
        t.cluster_update_workers([24], free[5:6])  # row out of range
    t.cluster_drop()
    t.tick(other)  # per-tick packing again
    t.close()


def test_new_request_classes_reach_the_resident_tables():
    """the request tables are part of the resident block: a snapshot that brings more request classes than the uploaded one is served from HBM too"""
    plain, res = _tick(), _tick(HQTICK_CHECK_CLUSTER=1)
    small = workloads.make("c2", n_tasks=5_000, n_workers=16)  # one request class, one resource
    res.cluster_upload(small)
    _same(res.tick(small), plain.tick(small))
    # same workers, different request table: two classes (1 cpu, 4 cpus)
    ids, prio, rq = small.task_id, small.task_priority, (np.arange(len(small.task_id)) % 2).astype(np.uint32)
    two = dataclasses.replace(small, _keep=[], requests=[small.requests[0], [workloads._variant([(0, 4)])]], task_id=ids, task_priority=prio, task_rq=rq)
    _same(res.tick(two), plain.tick(two))
    _same(res.tick(small), plain.tick(small))  # and back
    plain.close(); res.close()


def _subset(snap, keep_idx, extra=None):
    """the snapshot with only the workers `keep_idx` (row indices), optionally followed by new workers extra = (ids, total rows, free rows)"""
    W, R = len(snap.worker_id), snap.n_resources
    tot = np.array(snap.worker_total, np.uint64).reshape(W, R); fre = np.array(snap.worker_free, np.uint64).reshape(W, R)
    k = np.asarray(keep_idx, np.int64)
    remap = {int(w): i for i, w in enumerate(k)}
    ids, t, f = snap.worker_id[k], tot[k], fre[k]
    rem, mu, fl, gr = snap.worker_remaining_ns[k], snap.worker_min_utilization[k], snap.worker_flags[k], snap.worker_group[k]
    asg, pre = [snap.assigned[i] for i in k], [snap.prefilled[i] for i in k]
    if extra is not None:
        eids, et, ef = extra
        n = len(eids)
        ids = np.concatenate([ids, np.asarray(eids, np.uint32)]); t = np.concatenate([t, et]); f = np.concatenate([f, ef])
        rem = np.concatenate([rem, np.full(n, abi.HQ_NO_TIME_LIMIT, np.int64)]); mu = np.concatenate([mu, np.zeros(n, np.float32)])
        fl = np.concatenate([fl, np.full(n, abi.HQ_WORKER_SN, np.uint8)]); gr = np.concatenate([gr, np.zeros(n, np.uint32)])
        asg = asg + [[] for _ in range(n)]; pre = pre + [[] for _ in range(n)]
    blocked = [(remap[b[0]], b[1], b[2]) for b in snap.blocked if b[0] in remap]
    return dataclasses.replace(snap, _keep=[], worker_id=ids, worker_total=t.reshape(-1), worker_free=f.reshape(-1), worker_remaining_ns=rem, worker_min_utilization=mu,
                               worker_flags=fl, worker_group=gr, assigned=asg, prefilled=pre, blocked=blocked)


def test_workers_join_leave_and_reject_without_a_re_upload(oracle_free=None):
    """ABI 7 (row f1: on_new_worker / on_remove_worker / task_reject as deltas): after ONE hqtick_cluster_upload the worker set changes only through
    hqtick_cluster_remove_workers / _add_workers / _set_blocked / _update_workers, and the ticks' snapshots carry no worker arrays at all — every
    tick equals the plain tick (and the oracle) on the full snapshot of the same state."""
    from oracle.oracle import Oracle

    plain, res = _tick(), _tick(HQTICK_CHECK_CLUSTER=1)
    o = Oracle(abi.make_config(time_limit_s=20.0), canonical=True)
    snap = workloads.make_steady("c3", seed=5, n_tasks=40_000, n_workers=32)
    W, R = len(snap.worker_id), snap.n_resources
    res.cluster_upload(snap)
    _same(res.tick(snap, resident_workers=True), plain.tick(snap))
    # three workers are lost (on_remove_worker): rows leave, later rows move up
    lost = [3, 10, 31]
    keep = [w for w in range(W) if w not in lost]
    res.cluster_remove_workers(snap.worker_id[lost])
    s1 = _subset(snap, keep)
    assert res.cluster_workers().tolist() == s1.worker_id.tolist()
    got = res.tick(s1, resident_workers=True)
    _same(got, plain.tick(s1))
    want = o.tick(s1)
    assert got.counts == want.counts and got.records == want.records
    # two fresh workers join (on_new_worker): larger ids, idle
    top = int(snap.worker_id.max())
    tot_row = np.array(snap.worker_total, np.uint64).reshape(W, R)[0]
    extra = ([top + 1, top + 5], np.stack([tot_row, tot_row]), np.stack([tot_row, tot_row // np.uint64(2)]))
    res.cluster_add_workers(extra[0], extra[1], extra[2])
    s2 = _subset(snap, keep, extra)
    _same(res.tick(s2, resident_workers=True), plain.tick(s2))
    # a worker rejects (rq 0, variant 0) and (rq 3, variant 0) (task_reject -> blocked_requests); another one rejected and was re-enabled
    wid = int(s2.worker_id[4])
    res.cluster_set_blocked(wid, [(0, 0), (3, 0)])
    res.cluster_set_blocked(int(s2.worker_id[7]), [(1, 0)])
    res.cluster_set_blocked(int(s2.worker_id[7]), [])
    s3 = dataclasses.replace(s2, _keep=[], blocked=[(4, 0, 0), (4, 3, 0)])
    got = res.tick(s3, resident_workers=True)
    _same(got, plain.tick(s3))
    want = o.tick(s3)
    assert got.counts == want.counts and got.records == want.records
    # rows change (tasks start / finish) by index of the CURRENT set, then the blocked worker leaves: its pairs go with it
    W3 = len(s3.worker_id)
    free = np.array(s3.worker_free, np.uint64).reshape(W3, R).copy()
    free[[1, 4, W3 - 1]] = np.array(s3.worker_total, np.uint64).reshape(W3, R)[[1, 4, W3 - 1]]
    res.cluster_update_workers([1, 4, W3 - 1], free[[1, 4, W3 - 1]])
    s4 = _with_free(s3, free.reshape(-1))
    _same(res.tick(s4, resident_workers=True), plain.tick(s4))
    res.cluster_remove_workers([wid])
    s5 = _subset(s4, [w for w in range(W3) if w != 4])
    assert s5.blocked == []
    _same(res.tick(s5, resident_workers=True), plain.tick(s5))
    # errors: unknown id, ids that do not ascend above the set
    from hyperqueue_amd.tick import HqTickError

    with pytest.raises(HqTickError):
        res.cluster_remove_workers([wid])
    with pytest.raises(HqTickError):
        res.cluster_add_workers([1], tot_row.reshape(1, -1))
    plain.close(); res.close()


@pytest.mark.parametrize("seed", range(40))
def test_retracting_table_resident_in_the_library(seed):
    """ABI 7 (row f1: process_retracted / on_retract_response as deltas, server/reactor.rs:34-62,462-508): the Retracting tasks and their redirects live
    in the library — hqtick_retracting_add when a higher-priority arrival dissolves a prefill set, the ticks' own effects on task states and redirects
    (mapping.rs:66-101) applied by the library itself, hqtick_retract_response when a worker answers — and every tick run with
    n_retracting = HQ_RETRACTING_RESIDENT equals the tick on the snapshot that carries the retracting arrays (the scenario family of
    tests/test_gpu_fuzz.py::test_fuzz_prefill_disposal)."""
    from hyperqueue_amd.core import SchedEnv, TaskBuilder as TB, WorkerBuilder as WB
    from hyperqueue_amd.tick import HqTickError, Tick

    rng = np.random.default_rng(9000 + seed)
    cfg = abi.make_config(reserve=int(rng.integers(0, 2)), fill_max=int(rng.integers(1, 4)), time_limit_s=20.0)
    e = SchedEnv(cfg)
    g, r = Tick(cfg), Tick(cfg)
    shapes = [TB().cpus(1), TB().cpus(2)]
    for c in [int(x) for x in np.random.default_rng(seed).integers(1, 5, size=3)]:
        e.new_worker(WB(c))
    prio, seen, n_msgs = 0, 0, 0
    for round_ in range(5):
        n_new = int(rng.integers(1, 7)) if round_ else int(rng.integers(8, 16)); which = [int(rng.integers(0, 2)) for _ in range(n_new)]
        if round_ and rng.random() < 0.7:
            prio += 1
        for c in which:
            e.new_task(shapes[c].user_priority(prio))
        new_msgs = e.retract_messages[n_msgs:]; n_msgs = len(e.retract_messages)  # prefill sets dissolved by these arrivals: (worker id, task)
        if new_msgs:
            r.retracting_add([t for (_, t) in new_msgs], [w for (w, _) in new_msgs])
        snap = e.snapshot()
        seen += len(snap.retracting)
        try:
            want = g.tick(snap)
        except HqTickError as err:  # a Retracting task reached the prefill step: the reference asserts there
            assert err.code == abi.HQTICK_E_UNSUPPORTED
            with pytest.raises(HqTickError):
                r.tick(snap, resident_retracting=True)
            break
        got = r.tick(snap, resident_retracting=True)
        _same(got, want)
        assert got.redirects == want.redirects and got.redirect_kinds == want.redirect_kinds
        e.apply(want)
        k = int(rng.integers(1, 7)); answer = rng.random() < 0.6
        done = 0
        for t in sorted(e.tasks.values(), key=lambda t: t.id):
            if t.state == 1 and done < k:
                e.finish_task(t.id, t.worker); done += 1
        if answer:
            for t in [t for t in sorted(e.tasks.values(), key=lambda t: t.id) if t.state == 4 and t.id not in e.retaken_variant][:2]:
                expect = [(t.id,) + tuple(e.redirects[t.id])] if t.id in e.redirects else []
                wid = t.worker
                e.retract_response(wid, [t.id])
                assert r.retract_response(wid, [t.id]) == expect
        # the table holds exactly the tasks the mirror of the reactor has in state Retracting (those a tick put back on their own worker included)
        assert r.retracting_count() == sum(1 for t in e.tasks.values() if t.state == 4)
    g.close(); r.close()


@pytest.mark.parametrize("seed", range(30))
def test_worker_removal_with_the_resident_retracting_table(seed):
    """hqtick_cluster_remove_workers against on_remove_worker (server/reactor.rs:86-147) on the resident Retracting table.  The scenario family of
    the test above, with the worker set resident too; once redirected Retracting tasks exist a worker is lost — the one such a task is retracting FROM (the task
    becomes Assigned{target}: the library must report it, hqtick_cluster_last_reassigned, exactly as SchedEnv.remove_worker sends the ComputeTasks message) or the
    redirect's TARGET (the task loses the redirect and sits in its queue again, still Retracting{old}: the next resident tick must see it as such).  Every tick
    on the resident state equals the plain tick on the mirror's full snapshot."""
    from hyperqueue_amd.core import SchedEnv, TaskBuilder as TB, WorkerBuilder as WB
    from hyperqueue_amd.tick import HqTickError, Tick

    rng = np.random.default_rng(12_000 + seed)
    cfg = abi.make_config(reserve=int(rng.integers(0, 2)), fill_max=int(rng.integers(1, 4)), time_limit_s=20.0)
    e = SchedEnv(cfg)
    g, r = Tick(cfg), Tick(cfg)
    shapes = [TB().cpus(1), TB().cpus(2)]
    for c in [int(x) for x in np.random.default_rng(seed).integers(1, 5, size=4)]:
        e.new_worker(WB(c))
    prio, n_msgs, removed, uploaded = 0, 0, 0, False

    def sync_rows(snap):  # the reactor's row deltas: every row, each tick (the table is a handful of workers)
        W = len(snap.worker_id)
        r.cluster_update_workers(list(range(W)), np.asarray(snap.worker_free, np.uint64).reshape(W, snap.n_resources))

    for round_ in range(7):
        n_new = int(rng.integers(1, 7)) if round_ else int(rng.integers(8, 16))
        if round_ and rng.random() < 0.7:
            prio += 1
        for _ in range(n_new):
            e.new_task(shapes[int(rng.integers(0, 2))].user_priority(prio))
        new_msgs = e.retract_messages[n_msgs:]; n_msgs = len(e.retract_messages)
        if new_msgs:
            r.retracting_add([t for (_, t) in new_msgs], [w for (w, _) in new_msgs])
        snap = e.snapshot()
        if not uploaded:
            r.cluster_upload(snap); uploaded = True
        else:
            sync_rows(snap)
        try:
            want = g.tick(snap)
        except HqTickError as err:
            assert err.code == abi.HQTICK_E_UNSUPPORTED
            break
        got = r.tick(snap, resident_workers=True, resident_retracting=True)
        _same(got, want)
        assert got.redirects == want.redirects and got.redirect_kinds == want.redirect_kinds
        e.apply(want)
        done = 0
        for t in sorted(e.tasks.values(), key=lambda t: t.id):
            if t.state == 1 and done < int(rng.integers(1, 5)):
                e.finish_task(t.id, t.worker); done += 1
        # lose a worker that matters to a redirected Retracting task, if there is one (never the last two workers)
        red = [(t.id, t.worker, e.redirects[t.id][0]) for t in sorted(e.tasks.values(), key=lambda t: t.id) if t.state == 4 and t.id in e.redirects]
        if red and len(e.workers) > 2 and removed < 2:
            tid, old, target = red[int(rng.integers(0, len(red)))]
            wid = old if rng.random() < 0.5 else target
            if any(e.tasks[x].state == 4 and x not in e.redirects for x in e.workers[wid].assigned_tasks):
                continue  # a Retracting task retaken by its own worker sits there: the reference's assert fires in on_remove_worker (reactor.rs:90) — not a scenario
            sent = e.remove_worker(wid)
            assert r.cluster_remove_workers([wid]) == sent
            removed += 1
            new_msgs = e.retract_messages[n_msgs:]; n_msgs = len(e.retract_messages)  # add_ready_task of the returned tasks may have dissolved prefill sets
            if new_msgs:
                r.retracting_add([t for (_, t) in new_msgs], [w for (w, _) in new_msgs])
            assert r.cluster_workers().tolist() == sorted(e.workers)
        assert r.retracting_count() == sum(1 for t in e.tasks.values() if t.state == 4)
    g.close(); r.close()


def test_resident_workers_without_a_worker_set_fail_loudly():
    """n_workers = HQ_WORKERS_RESIDENT with nothing resident (never uploaded / dropped) is HQTICK_E_INVALID, not a legitimate tick of zero workers"""
    from hyperqueue_amd.tick import HqTickError

    t = _tick()
    snap = workloads.make_steady("c3", seed=3, n_tasks=5_000, n_workers=8)
    with pytest.raises(HqTickError) as err:
        t.tick(snap, resident_workers=True)
    assert err.value.code == abi.HQTICK_E_INVALID
    t.cluster_upload(snap)
    a = t.tick(snap, resident_workers=True)
    assert sum(len(x) for x in a.records) > 0
    t.cluster_drop()
    with pytest.raises(HqTickError) as err:
        t.tick(snap, resident_workers=True)
    assert err.value.code == abi.HQTICK_E_INVALID and "HQ_WORKERS_RESIDENT" in str(err.value)
    t.close()


def test_membership_deltas_at_cluster_scale_on_coupled_ticks():
    """The same deltas at BASELINE scale, on ticks the price sweeps solve: 1024 workers mid-run, three priority levels (every tick one coupled model of the whole
    cluster), 100 workers lost and 100 fresh ones joining per step, rejects, row changes — the resident tick (no worker arrays in the snapshot) equals the plain tick
    on the full snapshot of the same state at every step, and the last state's mapping equals the oracle's on the same counts (T3)."""
    from oracle.oracle import Oracle

    plain, res = _tick(), _tick(HQTICK_CHECK_CLUSTER=1)
    snap = workloads.make_steady("c3p", seed=11, n_tasks=300_000, n_workers=1024)
    R = snap.n_resources
    res.cluster_upload(snap)
    a, b = res.tick(snap, resident_workers=True), plain.tick(snap)
    _same(a, b)
    assert res.kernel_stats()["price_sweeps"] > 0 and a.is_optimal
    rng = np.random.default_rng(11)
    cur = snap
    for step in range(3):
        W = len(cur.worker_id)
        lost = np.sort(rng.choice(W, 100, replace=False))
        keep = [w for w in range(W) if w not in set(lost.tolist())]
        res.cluster_remove_workers(cur.worker_id[lost])
        top = int(cur.worker_id.max())
        tot_row = np.array(cur.worker_total, np.uint64).reshape(W, R)[0]
        ids = [top + 1 + i for i in range(100)]
        extra = (ids, np.stack([tot_row] * 100), np.stack([tot_row] * 100))
        res.cluster_add_workers(extra[0], extra[1], extra[2])
        cur = _subset(cur, keep, extra)
        # a few rejects and row changes on the new set
        wb = int(rng.integers(0, len(cur.worker_id)))
        res.cluster_set_blocked(int(cur.worker_id[wb]), [(0, 0), (5, 0)])
        cur = dataclasses.replace(cur, _keep=[], blocked=[bl for bl in cur.blocked if bl[0] != wb] + [(wb, 0, 0), (wb, 5, 0)])
        Wn = len(cur.worker_id)
        rows = np.sort(rng.choice(Wn, 40, replace=False))
        free = np.array(cur.worker_free, np.uint64).reshape(Wn, R).copy()
        free[rows] = np.array(cur.worker_total, np.uint64).reshape(Wn, R)[rows]
        res.cluster_update_workers(rows.tolist(), free[rows])
        cur = _with_free(cur, free.reshape(-1))
        assert res.cluster_workers().tolist() == cur.worker_id.tolist()
        a, b = res.tick(cur, resident_workers=True), plain.tick(cur)
        _same(a, b)
        assert a.is_optimal
    o = Oracle(abi.make_config(time_limit_s=5.0))
    want = o.tick_given(cur, a.counts, is_optimal=True)
    assert a.counts == want.counts and a.records == want.records and a.retracts == want.retracts and (a.new_free == want.new_free).all()
    plain.close(); res.close()
