"""Malformed input never crashes or hangs the library: every case comes back as the documented negative code (include/hqtick.h)
and the context stays usable afterwards."""
import numpy as np
import pytest

from hyperqueue_amd import abi, workloads
from hyperqueue_amd.tick import HqTickError, Tick

pytestmark = pytest.mark.gpu


@pytest.fixture()
def good():
    return workloads.make("c3", n_tasks=5_000, n_workers=8)


def expect(code, fn):
    with pytest.raises(HqTickError) as e:
        fn()
    assert e.value.code == code, (e.value.code, str(e.value))


def test_bad_snapshots_are_rejected_and_ctx_survives(good):
    t = Tick(abi.make_config())
    ok = t.tick(good)
    cases = []
    s = workloads.make("c3", n_tasks=5_000, n_workers=8); s.worker_id = s.worker_id[::-1].copy(); cases.append(("worker ids descending", s, abi.HQTICK_E_INVALID))
    s = workloads.make("c3", n_tasks=5_000, n_workers=8); s.task_id = s.task_id[::-1].copy(); cases.append(("ready set unsorted", s, abi.HQTICK_E_INVALID))
    s = workloads.make("c3", n_tasks=5_000, n_workers=8); s.task_id[10] = s.task_id[9]; cases.append(("duplicate task id", s, abi.HQTICK_E_INVALID))
    s = workloads.make("c3", n_tasks=5_000, n_workers=8); s.task_rq[123] = 77; cases.append(("request id out of range (found by K1)", s, abi.HQTICK_E_INVALID))
    s = workloads.make("c3", n_tasks=5_000, n_workers=8); s.requests[2][0]["entries"][0] = (9, abi.HQ_ENTRY_AMOUNT, 10_000); cases.append(("resource id out of range", s, abi.HQTICK_E_INVALID))
    s = workloads.make("c3", n_tasks=5_000, n_workers=8); s.requests[1][0]["entries"][0] = (0, abi.HQ_ENTRY_AMOUNT, 0); cases.append(("zero amount", s, abi.HQTICK_E_INVALID))
    s = workloads.make("c3", n_tasks=5_000, n_workers=8); s.requests[0] = [s.requests[0][0]] * 33; cases.append(("33 variants", s, abi.HQTICK_E_INVALID))
    s = workloads.make("c3", n_tasks=5_000, n_workers=8); s.blocked = [(99, 0, 0)]; cases.append(("blocked worker index", s, abi.HQTICK_E_INVALID))
    for name, snap, code in cases:
        expect(code, lambda: t.tick(snap))
        again = t.tick(good)  # the context is still good
        assert again.records == ok.records, name


def test_resident_misuse(good):
    t = Tick(abi.make_config())
    expect(abi.HQTICK_E_INVALID, lambda: t.tick(good, resident=True))  # no hqtick_upload_ready yet
    expect(abi.HQTICK_E_INVALID, lambda: t.ready_add(good.task_id[:3], good.task_priority[:3], good.task_rq[:3]))
    expect(abi.HQTICK_E_INVALID, lambda: t.ready_consume_last())
    expect(abi.HQTICK_E_INVALID, lambda: t.upload_ready(good.task_id[::-1].copy(), good.task_priority, good.task_rq, sorted_=True))
    t.upload_ready(good.task_id, good.task_priority, good.task_rq)
    t.ready_consume_last()  # nothing handed out yet: a no-op
    r1 = t.tick(good, resident=True)
    t.ready_consume_last(); t.ready_consume_last()  # second call: no-op
    n = sum(len(x) for x in r1.records)
    assert t.ready_count() == 5_000 - n
    expect(abi.HQTICK_E_INVALID, lambda: t.ready_add(np.asarray([5, 5], np.uint64), np.zeros(2, np.uint64), np.zeros(2, np.uint32)))
    expect(abi.HQTICK_E_INVALID, lambda: t.ready_add(np.asarray([5], np.uint64), np.zeros(1, np.uint64), np.asarray([0xFFFFFFFF], np.uint32)))


def test_empty_inputs():
    t = Tick(abi.make_config())
    s = workloads.make("c3", n_tasks=10, n_workers=4)
    s.task_id, s.task_priority, s.task_rq = s.task_id[:0], s.task_priority[:0], s.task_rq[:0]
    r = t.tick(s)
    assert r.status == abi.HQTICK_DONE and r.batches == [] and all(not x for x in r.records)
    s = workloads.make("c3", n_tasks=100, n_workers=4)
    for f in ("worker_id", "worker_total", "worker_free", "worker_remaining_ns", "worker_min_utilization", "worker_flags", "worker_group"):
        setattr(s, f, getattr(s, f)[:0])
    s.assigned, s.prefilled = [], []
    r = t.tick(s)  # tasks but no workers
    assert r.status == abi.HQTICK_DONE and r.counts == []
    t.upload_ready(np.zeros(0, np.uint64), np.zeros(0, np.uint64), np.zeros(0, np.uint32))
    assert t.ready_count() == 0
    t.ready_add(np.asarray([7, 9], np.uint64), np.full(2, 1 << 63, np.uint64), np.zeros(2, np.uint32))  # first tasks of an empty resident set
    assert t.ready_count() == 2


def test_unsorted_upload_is_sorted_on_the_device(good):
    """hqtick_upload_ready(sorted = 0): any order in, same ticks out; duplicates are rejected."""
    rng = np.random.default_rng(3)
    perm = rng.permutation(len(good.task_id))
    a, b = Tick(abi.make_config()), Tick(abi.make_config())
    a.upload_ready(good.task_id, good.task_priority, good.task_rq, sorted_=True)
    b.upload_ready(good.task_id[perm], good.task_priority[perm], good.task_rq[perm], sorted_=False)
    ra, rb = a.tick(good, resident=True), b.tick(good, resident=True)
    assert ra.records == rb.records and ra.counts == rb.counts and sum(len(x) for x in ra.records) > 0
    ids = good.task_id[perm].copy(); ids[7] = ids[1234]
    expect(abi.HQTICK_E_INVALID, lambda: b.upload_ready(ids, good.task_priority[perm], good.task_rq[perm], sorted_=False))
    expect(abi.HQTICK_E_INVALID, lambda: b.tick(good, resident=True))  # a failed upload leaves no resident set behind
    for n in (1, 2, 3, 255, 257, 4097):  # sizes around the power-of-two padding
        s = workloads.make("c3", n_tasks=n, n_workers=2)
        p = rng.permutation(n)
        b.upload_ready(s.task_id[p], s.task_priority[p], s.task_rq[p], sorted_=False)
        a.upload_ready(s.task_id, s.task_priority, s.task_rq, sorted_=True)
        assert a.tick(s, resident=True).records == b.tick(s, resident=True).records
