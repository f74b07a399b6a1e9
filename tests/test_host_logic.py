"""Host-side building blocks of libhqtick.so that do not need a GPU: the exact MILP solver (vs HiGHS through the
oracle's solver entry) and the hashbrown iteration-order helper (vs the oracle's independent emulation)."""
import ctypes as C

import numpy as np
import pytest

from hyperqueue_amd import _testhooks, abi
from oracle import oracle as orc


def product_milp(obj, kind, rtype, rhs, roff, rcol, rcoef, canonical=True, time_limit=30.0):
    lib = _testhooks.load()
    n, m = len(obj), len(rhs)
    obj = np.ascontiguousarray(obj, np.float64); kind = np.ascontiguousarray(kind, np.uint8)
    rtype = np.ascontiguousarray(rtype, np.uint8); rhs = np.ascontiguousarray(rhs, np.float64)
    roff = np.ascontiguousarray(roff, np.int32); rcol = np.ascontiguousarray(rcol, np.int32); rcoef = np.ascontiguousarray(rcoef, np.float64)
    x = np.zeros(max(n, 1)); z = C.c_double(); opt = C.c_int(); nodes = C.c_long()
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
    ok = lib.hqtick_debug_milp_solve(
        C.c_int(n), obj.ctypes.data_as(dp), kind.ctypes.data_as(abi.u8p), C.c_int(m), rtype.ctypes.data_as(abi.u8p), rhs.ctypes.data_as(dp),
        roff.ctypes.data_as(ip), rcol.ctypes.data_as(ip), rcoef.ctypes.data_as(dp), C.c_double(time_limit), C.c_int(1 if canonical else 0),
        x.ctypes.data_as(dp), C.byref(z), C.byref(opt), C.byref(nodes))
    return (x[:n].copy(), z.value, bool(opt.value)) if ok else None


def random_model(rng):
    """Shaped like the tick's model: non-negative objective, packing rows, a few big-M rows with bool columns."""
    n = int(rng.integers(2, 10)); m = int(rng.integers(1, 6))
    obj = rng.integers(0, 6, n) / 4.0
    kind = (rng.random(n) < 0.25).astype(np.uint8)
    rtype, rhs, roff, rcol, rcoef = [], [], [0], [], []
    for i in range(m):
        cols = np.nonzero(rng.random(n) < 0.6)[0]
        if len(cols) == 0:
            cols = np.array([int(rng.integers(0, n))])
        rtype.append(1); rhs.append(float(rng.integers(1, 12)))
        rcol += cols.tolist(); rcoef += (rng.integers(1, 5, len(cols)) * 0.5).tolist(); roff.append(len(rcol))
    # every nat column bounded by a row
    rtype.append(1); rhs.append(float(rng.integers(3, 20))); rcol += list(range(n)); rcoef += [1.0] * n; roff.append(len(rcol))
    if rng.random() < 0.5 and kind.any():  # a Min row through a bool ("blocker short" flag shape, solver.rs:243-250)
        b = int(np.nonzero(kind)[0][0]); s = float(rng.integers(1, 4))
        cols = [j for j in range(n) if j != b and not kind[j]][:3]
        if cols:
            rtype.append(0); rhs.append(s); rcol += cols + [b]; rcoef += [1.0] * len(cols) + [s]; roff.append(len(rcol))
    return obj, kind, np.array(rtype, np.uint8), np.array(rhs), np.array(roff, np.int32), np.array(rcol, np.int32), np.array(rcoef)


@pytest.mark.parametrize("seed", range(60))
def test_milp_matches_highs(seed):
    rng = np.random.default_rng(seed)
    mdl = random_model(rng)
    want = orc.solve_milp(*mdl, time_limit=30.0, canonical=False)
    got = product_milp(*mdl, canonical=False)
    assert (want is None) == (got is None)
    if want is None:
        return
    assert got[2] and want[2]
    assert abs(got[1] - want[1]) <= 1e-7 * max(1.0, abs(want[1]))  # same optimum value
    # and the canonical optimum is exactly the oracle's canonicalised HiGHS optimum
    wantc = orc.solve_milp(*mdl, time_limit=30.0, canonical=True)
    gotc = product_milp(*mdl, canonical=True)
    assert np.array_equal(np.round(gotc[0]), np.round(wantc[0])), (gotc[0], wantc[0])
    assert abs(gotc[1] - want[1]) <= 1e-7 * max(1.0, abs(want[1]))


def test_milp_infeasible_is_none():
    # x0 >= 3 and x0 <= 1
    r = product_milp([1.0], [0], [0, 1], [3.0, 1.0], [0, 1, 2], [0, 0], [1.0, 1.0])
    assert r is None


@pytest.mark.parametrize("n", [1, 2, 3, 4, 7, 8, 15, 28, 29, 57, 200, 1024, 4096])
def test_map_order_matches_oracle_emulation(n):
    lib = _testhooks.load()
    rng = np.random.default_rng(n)
    for keys in (np.arange(50, 50 + n, dtype=np.uint32), np.sort(rng.choice(1 << 20, n, replace=False)).astype(np.uint32)):
        out = np.zeros(n, np.uint32)
        lib.hqtick_debug_map_order_u32(keys.ctypes.data_as(abi.u32p), C.c_uint32(n), out.ctypes.data_as(abi.u32p))
        assert sorted(out.tolist()) == list(range(n))
        assert keys[out].tolist() == orc.hb_order_u32(keys)


def test_map_order_reference_pins():
    """workers {50, 51} iterate (50, 51) — pinned by test_schedule_no_priorities (test_scheduler_sn.rs:183-187)."""
    assert orc.hb_order_u32([50, 51]) == [50, 51]
    assert orc.hb_order_u32([50, 51, 52]) == [52, 50, 51]


@pytest.mark.parametrize("n_workers,n_ready", [(4, 86), (4, 193), (8, 172), (16, 344)])
def test_unsaturated_multi_class_models_are_proven(n_workers, n_ready):
    """Fewer ready tasks than the cluster holds, all eight c3 classes: no batch is saturated and the batch-size rows couple every worker
    ("Unsaturated ticks").  Most-fractional branching timed out on every one of these; the value-ordered branching + restart
    portfolio must prove them, with HiGHS's objective."""
    from hyperqueue_amd import workloads

    ids, prio, rq, off, dep = workloads.make_dag(100_000, seed=0)
    src = np.nonzero((off[1:] - off[:-1]) == 0)[0][:n_ready]
    assert len(src) == n_ready
    drv = workloads.DagChurn(n_workers=n_workers, churn=0.1, seed=0)
    snap = drv.snapshot(ids[src], prio[src], rq[src])
    o = orc.Oracle(abi.make_config(time_limit_s=60.0))
    w = o.tick(snap)
    m = o.last_model()
    assert w.is_optimal and len(m["obj"]) == 8 * n_workers
    got = product_milp(m["obj"], m["kind"], m["rtype"], m["rhs"], m["roff"], m["rcol"], m["rcoef"], canonical=False, time_limit=60.0)
    assert got is not None and got[2], "not proven optimal"
    assert abs(got[1] - m["objective"]) <= 1e-9 * abs(m["objective"])


def _encode_delta16(records_per_worker):
    """The emission format of HQTICK_FLAG_COMPACT_DELTA16 as include/hqtick.h states it, written here from the specification (NOT the kernel):
    per worker runs of equal (job, variant, kind) carrying their first low id, every other record one 16-bit difference or the three-unit form."""
    W = len(records_per_worker)
    off = np.zeros(W + 1, np.uint32)
    for w, recs in enumerate(records_per_worker):
        off[w + 1] = off[w] + len(recs)
    n = int(off[-1])
    units = np.zeros(4 * max(n, 1), np.uint16); runs = np.zeros((max(n, 1), 4), np.uint32); span = np.zeros((W, 2), np.uint32)
    for w, recs in enumerate(records_per_worker):
        a = int(off[w]); u = 4 * a; nr = 0
        for i, (task, var, kind) in enumerate(recs):
            job, lo = task >> 32, task & 0xFFFFFFFF
            meta = var | (kind << 8)
            if i == 0 or (recs[i - 1][0] >> 32) != job or (recs[i - 1][1] | (recs[i - 1][2] << 8)) != meta:
                runs[a + nr] = (i, job, meta, lo); nr += 1
                continue
            d = (lo - (recs[i - 1][0] & 0xFFFFFFFF)) & 0xFFFFFFFF
            if d < 0xFFFF:
                units[u] = d; u += 1
            else:
                units[u:u + 3] = (0xFFFF, lo & 0xFFFF, lo >> 16); u += 3
        span[w] = (a, nr)
    return off, units, runs, span


@pytest.mark.parametrize("seed", range(6))
def test_delta16_decoder_round_trip(seed):
    """abi.expand_delta16 (the host shim's decoder, which the GPU tests rely on) against an encoder written from the header's text: jobs, variants and
    kinds that split runs, differences at the limits of the one-unit form, negative differences, low halves up to 2^32 - 1, workers without records"""
    rng = np.random.default_rng(seed)
    recs = []
    for w in range(9):
        if w % 4 == 3:
            recs.append([]); continue
        out = []
        lo = int(rng.integers(1, 1000))
        for i in range(int(rng.integers(1, 80))):
            step = int(rng.choice([1, 7, 8191, 65534, 65535, 65536, 70000, -5, -100000])) if seed else int(rng.integers(1, 60000))  # seed 0: no escape (the vectorised path)
            lo = (lo + step) % (1 << 32)
            if rng.random() < 0.05 and seed:
                lo = 0xFFFFFFFF
            out.append(((int(rng.integers(1, 3)) << 32) | lo, int(rng.choice([0, 1, 0xFF])), int(rng.integers(0, 2))))
        recs.append(out)
    off, units, runs, span = _encode_delta16(recs)
    r = abi.ResultC()
    r.rec_off = off.ctypes.data_as(abi.u32p); r.run_span = span.ctypes.data_as(abi.u32p); r.runs16 = runs.ctypes.data_as(abi.u32p)
    r.rec_delta16 = units.ctypes.data_as(C.POINTER(C.c_uint16))
    t, v, k = abi.expand_delta16(r, len(recs), off)
    flat = [x for rr in recs for x in rr]
    assert t == [x[0] for x in flat] and v == [x[1] for x in flat] and k == [x[2] for x in flat]
    # the header-only C walker of include/hqtick_records.h (what a host shim would use) visits the same records, worker by worker
    import records_c

    n, ct, cv, ck, cw = records_c.walk(r, len(recs))
    assert n == len(flat) and ct == t and cv == v and ck == k and cw == [w for w, rr in enumerate(recs) for _ in rr]


def test_record_walker_full_and_u32_forms():
    """include/hqtick_records.h on the other two emission forms (built here from the header's text) and on a result without host records"""
    import records_c

    recs = [[((3 << 32) | 10, 0xFF, 0), ((3 << 32) | 99, 0xFF, 0), ((3 << 32) | 7, 1, 1), ((4 << 32) | 8, 1, 1)], [], [((9 << 32) | 0xFFFFFFFF, 0, 1)]]
    flat = [x for rr in recs for x in rr]
    off = np.asarray([0, 4, 4, 5], np.uint32)
    task = np.asarray([x[0] for x in flat], np.uint64); var = np.asarray([x[1] for x in flat], np.uint8); kind = np.asarray([x[2] for x in flat], np.uint8)
    r = abi.ResultC()
    r.rec_off = off.ctypes.data_as(abi.u32p); r.rec_task = task.ctypes.data_as(abi.u64p); r.rec_variant = var.ctypes.data_as(abi.u8p); r.rec_kind = kind.ctypes.data_as(abi.u8p)
    n, ct, cv, ck, cw = records_c.walk(r, 3)
    assert (n, ct, cv, ck, cw) == (5, task.tolist(), var.tolist(), kind.tolist(), [0, 0, 0, 0, 2])
    lo = (task & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    runs = np.asarray([[0, 3, 0xFF | (0 << 8)], [2, 3, 1 | (1 << 8)], [3, 4, 1 | (1 << 8)], [0, 0, 0], [0, 9, 0 | (1 << 8)]], np.uint32)  # a worker's runs sit in the slots of its own records
    span = np.asarray([[0, 3], [0, 0], [4, 1]], np.uint32)
    r2 = abi.ResultC()
    r2.rec_off = off.ctypes.data_as(abi.u32p); r2.rec_task_lo = lo.ctypes.data_as(abi.u32p); r2.run_span = span.ctypes.data_as(abi.u32p); r2.runs = runs.ctypes.data_as(abi.u32p)
    assert records_c.walk(r2, 3) == (5, task.tolist(), var.tolist(), kind.tolist(), [0, 0, 0, 0, 2])
    assert abi.expand_compact(r2, 3, off) == (task.tolist(), var.tolist(), kind.tolist())
    r3 = abi.ResultC(); r3.rec_off = off.ctypes.data_as(abi.u32p)   # records left in a device sink: nothing to walk
    assert records_c.walk(r3, 3)[0] == -1
