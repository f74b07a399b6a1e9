"""Compact record emission (HQTICK_FLAG_COMPACT_RECORDS, include/hqtick.h): the records cross PCIe as the u32 low halves of the task ids plus
one run per stretch of equal (job, variant, kind).  Expanded again (abi.expand_compact = what the host shim does), they must be the very
records of the default emission: checked on the committed fixtures (small and BASELINE size), on scenarios with several jobs per worker, prefill
sets and priority levels, and on the randomised family."""
import glob
import json
import os

import numpy as np
import pytest

from hyperqueue_amd import abi, workloads
from hyperqueue_amd.core import SchedEnv, TaskBuilder as TB, WorkerBuilder as WB

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


COMPACT = abi.HQTICK_FLAG_COMPACT_RECORDS
DELTA16 = abi.HQTICK_FLAG_COMPACT_RECORDS | abi.HQTICK_FLAG_COMPACT_DELTA16  # ABI 6: 16-bit differences instead of the u32 low halves
MODES = pytest.mark.parametrize("mode", [COMPACT, DELTA16], ids=["u32", "delta16"])


def _pair(cfg, mode=COMPACT):
    from hyperqueue_amd.tick import Tick

    c2 = abi.make_config(reserve=cfg.proactive_filling_reserve, fill_max=cfg.proactive_filling_max, time_limit_s=cfg.mip_time_limit_s, flags=mode)
    return Tick(cfg), Tick(c2)


@MODES
@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "*.json"))), ids=lambda p: os.path.basename(p)[:-5])
def test_compact_reproduces_fixture(path, mode):
    from test_fixtures import _check, _load

    snap, cfg, exp = _load(path)
    plain, comp = _pair(cfg, mode)
    try:
        rc = comp.tick_raw(snap.to_c())
        assert (((rc.rec_task_lo and not rc.rec_delta16) if mode == COMPACT else (rc.rec_delta16 and rc.runs16 and not rc.rec_task_lo and not rc.runs)) and not rc.rec_task) or int(np.ctypeslib.as_array(rc.rec_off, shape=(len(snap.worker_id) + 1,))[-1]) == 0
        _check(comp.tick(snap), exp)
    finally:
        plain.close(); comp.close()


@MODES
@pytest.mark.parametrize("name", ["c4_full", "c3_steady_full", "c2_full"])
def test_compact_reproduces_big_fixture(name, mode):
    from test_fixtures import _check_big, _load_big

    snap, cfg, exp = _load_big(os.path.join(GOLDEN, "big", name + ".json"))
    plain, comp = _pair(cfg, mode)
    try:
        _check_big(comp.tick(snap), exp)
    finally:
        plain.close(); comp.close()


@MODES
def test_compact_runs_split_on_job_variant_and_kind(mode):
    """several jobs interleaved in one queue, two variants, prefills and two priority levels on the same workers"""
    env = SchedEnv(abi.make_config(reserve=2, fill_max=6, time_limit_s=20.0))
    env.new_named_resource("gpus/amd")
    env.new_workers(5, WB(12).res_sum("gpus/amd", 2))
    env.new_tasks(120, TB().cpus(1))
    env.new_tasks(25, TB().cpus(2).user_priority(2))
    env.new_tasks(10, TB().cpus(1).add_resource(1, 1).next_variant().cpus(3))
    snap = env.snapshot()
    # spread the tasks over several jobs: rewrite the ids' high halves, keeping the column sorted
    ids = snap.task_id.copy()
    lo = ids & np.uint64(0xFFFFFFFF)
    job = (np.arange(len(ids)) // 17 + 1).astype(np.uint64)
    snap.task_id = (job << np.uint64(32)) | lo
    assert (np.diff(snap.task_id.astype(np.int64)) > 0).all()
    plain, comp = _pair(env.config, mode)
    try:
        a, b = plain.tick(snap), comp.tick(snap)
        assert a.records == b.records and a.counts == b.counts and a.retracts == b.retracts
        rc = comp.tick_raw(snap.to_c())
        W = len(snap.worker_id)
        import records_c  # the header-only C walker of include/hqtick_records.h on the library's own output, in this emission form and in the full one

        flat = [(t, v, k) for recs in a.records for (t, v, k) in recs]
        n, ct, cv, ck, cw = records_c.walk(rc, W)
        assert n == len(flat) and list(zip(ct, cv, ck)) == flat and cw == [w for w, recs in enumerate(a.records) for _ in recs]
        n, ct, cv, ck, cw = records_c.walk(plain.tick_raw(snap.to_c()), W)
        assert n == len(flat) and list(zip(ct, cv, ck)) == flat
        cnt = np.ctypeslib.as_array(rc.run_span, shape=(2 * W,)).reshape(W, 2)[:, 1]
        off = np.ctypeslib.as_array(rc.rec_off, shape=(W + 1,))
        assert max(int(cnt[w]) for w in range(W) if off[w + 1] > off[w]) >= 3  # jobs / kinds really split the runs
    finally:
        plain.close(); comp.close()


@MODES
@pytest.mark.parametrize("seed", range(20))
def test_compact_equals_default_on_fuzz_family(seed, mode):
    import test_gpu_fuzz as f

    cfg, envs, _rng = f.build(seed)
    snap = envs[1].snapshot()
    plain, comp = _pair(cfg, mode)
    try:
        try:
            a = plain.tick(snap)
        except Exception as e:  # scenarios the library refuses (E_UNSUPPORTED): refused in both modes
            with pytest.raises(type(e)):
                comp.tick(snap)
            return
        b = comp.tick(snap)
        assert a.status == b.status and a.records == b.records and a.retracts == b.retracts and sorted(a.redirects) == sorted(b.redirects) and a.mn == b.mn
    finally:
        plain.close(); comp.close()


@pytest.mark.parametrize("spacing", [1, 65534, 65535, 70000, 3_000_000])
def test_delta16_escapes_and_limits(spacing):
    """ids so far apart that the differences do not fit 16 bits (every record then takes the three-unit form: the stream's full capacity), at the
    limit values of the one-unit form, and low halves up to 2^32 - 1; negative differences come from the key boundaries inside a worker's records"""
    env = SchedEnv(abi.make_config(reserve=2, fill_max=8, time_limit_s=20.0))
    env.new_named_resource("gpus/amd")
    env.new_workers(12, WB(32).res_sum("gpus/amd", 2))
    env.new_tasks(900, TB().cpus(1))
    env.new_tasks(120, TB().cpus(2).user_priority(1))
    env.new_tasks(60, TB().cpus(1).add_resource(1, 1))
    snap = env.snapshot()
    n = len(snap.task_id)
    lo = (np.arange(1, n + 1, dtype=np.uint64) * np.uint64(spacing))
    if spacing == 3_000_000:
        lo = lo + (np.uint64(0xFFFFFFFF) - lo[-1])  # the last id's low half is 2^32 - 1
    assert int(lo[-1]) <= 0xFFFFFFFF
    snap.task_id = (np.uint64(7) << np.uint64(32)) | lo
    plain, comp = _pair(env.config, DELTA16)
    try:
        a, b = plain.tick(snap), comp.tick(snap)
        assert sum(len(r) for r in a.records) > 300
        assert a.records == b.records and a.counts == b.counts and a.retracts == b.retracts
        import records_c

        n, ct, cv, ck, _cw = records_c.walk(comp.tick_raw(snap.to_c()), len(snap.worker_id))
        assert list(zip(ct, cv, ck)) == [(t, v, k) for recs in a.records for (t, v, k) in recs]
    finally:
        plain.close(); comp.close()
