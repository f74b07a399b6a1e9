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
