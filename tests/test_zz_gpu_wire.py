"""hqwire_encode_device on a real MI355X: tables and records in HBM, three kernels, the bytes compared with the bincode oracle.
(File name sorts last on purpose: this kernel family was written at the end of round 1, after the round's GPU budget was spent; its phase
functions are covered on the CPU by tests/test_wire.py, the launches themselves are first exercised here.)"""
import random

import pytest

import wire_cases as wc
from hyperqueue_amd import wire

pytestmark = pytest.mark.gpu


# ---- the five late end-to-end pins of tests/golden_cases.py::E2E_EXTRA_CASES through the HIP C ABI (kept out of test_gpu_golden.py so that
# a surprise here cannot stop the GPU suite early)
def _extra():
    import golden_cases

    return golden_cases.E2E_EXTRA_CASES


@pytest.fixture(scope="module")
def gpu_backend():
    from test_gpu_golden import GpuBackend

    return GpuBackend()  # one set of contexts for all cases, as in test_gpu_golden.py


@pytest.mark.parametrize("case", _extra(), ids=lambda f: f.__name__)
def test_e2e_extra_gpu(case, gpu_backend):
    gpu_backend.flags.clear()
    case(gpu_backend)
    assert all(gpu_backend.flags)


@pytest.fixture(scope="module")
def wire_canary():
    """First contact of the wire kernels with hardware happens in a SUBPROCESS: a device fault there kills that process, not the GPU suite.
    The in-process tests below run only if the canary came back clean."""
    import os
    import subprocess
    import sys

    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import wire_cases as wc\nfrom hyperqueue_amd import wire\n"
            "for seed in (0, 1, 2):\n    wc.check_scenario(wire.encode_device, wc.random_scenario(seed))\nprint('canary ok')\n") % (
        os.path.join(os.path.dirname(__file__), ".."), os.path.dirname(__file__))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    if p.returncode != 0 or "canary ok" not in p.stdout:
        pytest.fail("wire kernels failed their first hardware run (subprocess): exit %s\n%s" % (p.returncode, (p.stdout + p.stderr)[-1500:]))
    return True


@pytest.mark.parametrize("seed", range(12))
def test_device_matches_oracle(seed, wire_canary):
    wc.check_scenario(wire.encode_device, wc.random_scenario(seed))


def test_device_tick_mapping(wire_canary):
    wc.check_scenario(wire.encode_device, wc.tick_scenario())


@pytest.mark.parametrize("seed", range(40, 46))
def test_device_roundtrip_through_independent_decoder(seed, wire_canary):
    sc = wc.random_scenario(seed)
    t, r = wc.tables_and_records(*sc)
    wc.check_roundtrip(sc, wire.encode_device(t, r, 1 << 22).messages(r))


def test_device_wide_message(wire_canary):
    rnd = random.Random(3)
    configs = [(None if i % 2 else (60 * i, 0), bytes([i]) * (50 * i)) for i in range(12)]
    attrs = {((5 << 32) | i): (1, i, 7, rnd.randrange(12), None if i % 3 else b"e%d" % i) for i in range(1, 2001)}
    wc.check_scenario(wire.encode_device, (attrs, configs, [77], [[(t, 0, 1) for t in attrs]], [[]], []), capacity=1 << 24)


def test_device_c3_shape(wire_canary):
    """BASELINE C3's cold tick shape: 1024 workers x (120 prefills + 64 assigned) records, 8 request classes = 8 configurations"""
    rnd = random.Random(11)
    configs = [((3600, 0), b"body-of-class-%d" % i * 8) for i in range(8)]
    W, per = 1024, 184
    attrs, records = {}, []
    tid = 1
    for w in range(W):
        recs = []
        for j in range(per):
            t = (1 << 32) | tid
            tid += 1
            attrs[t] = (rnd.randrange(8), 0, 0x8000000000000000, rnd.randrange(8), None)
            recs.append((t, 0xFF, 0) if j < 120 else (t, 0, 1))
        records.append(recs)
    res = wc.check_scenario(wire.encode_device, (attrs, configs, list(range(1, W + 1)), records, [[] for _ in range(W)], []), capacity=1 << 25)
    assert res.total_bytes > W * per * 42


def test_device_slot_conditions_and_capacity(wire_canary):
    attrs = {1: (0, 0, 0, 0, None), 3: (0, 0, 0, 0, None)}
    configs = [(None, b"small")]
    records = [[(1, 0, 1)], [(99, 0, 1)], [(3, 0, 1)] * (wire.HQWIRE_MAX_RECORDS + 1)]
    t, r = wc.tables_and_records(attrs, configs, [10, 11, 12], records, [[], [5], []], [])
    res = wire.encode_device(t, r, 1 << 16)
    assert res.status == wire.HQWIRE_OK
    assert res.slot_status.tolist() == [wire.SLOT_OK, wire.SLOT_UNKNOWN, wire.SLOT_TOO_MANY]
    assert len(res.messages(r)) == 2
    small = wire.encode_device(t, r, res.total_bytes - 1)
    assert small.status == wire.HQWIRE_CAPACITY and small.total_bytes == res.total_bytes


def test_tick_to_bytes_through_the_record_sink(wire_canary):
    """DESIGN.md 8d end to end: a real tick (HIP library) leaves its records in a device record sink, hqwire_encode_device reads them there, and
    the bytes equal what the oracle's tick + the bincode oracle produce for the same snapshot."""
    import numpy as np

    from hyperqueue_amd import sharded
    from hyperqueue_amd.core import SchedEnv, TaskBuilder as TB, WorkerBuilder as WB
    from oracle.oracle import Oracle

    env = SchedEnv()
    env.new_named_resource("gpus/amd")
    env.new_workers(6, WB(16).res_sum("gpus/amd", 2))
    env.new_tasks(200, TB().cpus(1))
    env.new_tasks(40, TB().cpus(4).user_priority(1))
    env.new_tasks(30, TB().cpus(2).add_resource(1, 0.5))
    snap = env.snapshot()
    want = Oracle(env.config, canonical=True).tick(snap)
    W, cap = len(snap.worker_id), 4096
    st = sharded.ShardedTick(env.config, rank=0, world=1, records_per_shard=cap)
    res_c, sink = st.tick_local(snap.to_c(), W)
    n_records = int(np.ctypeslib.as_array(res_c.rec_off, shape=(W + 1,))[W])
    assert n_records == sum(len(r) for r in want.records) > 0
    rnd = random.Random(9)
    configs = [(None, b"prog-a" * 30), ((600, 0), b"prog-b" * 70), ((5, 250), b"")]
    attrs = {t: (rnd.randrange(4), rnd.randrange(50), (0x80000000 + rnd.randrange(3)) << 32, rnd.randrange(3), None if rnd.random() < 0.6 else b"e-%d" % (t & 0xFFFF))
             for recs in want.records for (t, v, k) in recs}
    worker_ids = [int(w) for w in snap.worker_id]
    sc = (attrs, configs, worker_ids, want.records, want.retracts, [])
    tables, side = wc.tables_and_records(*sc)
    got = wire.encode_from_sink(tables, sink, W, cap, n_records, side, 1 << 22)
    assert got.status == wire.HQWIRE_OK and (got.slot_status == 0).all()
    assert got.messages(side) == wc.oracle_messages(*sc)
    wc.check_roundtrip(sc, got.messages(side))
    st.t.close()


@pytest.mark.parametrize("seed,limit", [(s, l) for s in range(100, 106) for l in (700, 4000)])
def test_device_fragmentation_matches_the_builder(seed, limit, wire_canary):
    """create_message_on_overflow (server/task.rs:388-400) on the device: with the builder's limit lowered on both sides, the kernels cut every
    worker's ComputeTasks message where the oracle's builder cuts it, each fragment with its own shared-data list and shared_index numbering"""
    sc = wc.random_scenario(seed, max_rec=60)
    res = wc.check_scenario(lambda t, r, cap: wire.encode_device(t, r, cap, limit=limit), sc, limit=limit)
    wc.check_roundtrip(sc, res.messages(wc.tables_and_records(*sc)[1]))


def test_device_fragmentation_really_cuts(wire_canary):
    rnd = random.Random(3)
    configs = [(None if k % 2 else (k, 7), bytes([65 + k]) * (10 + 13 * k)) for k in range(5)]
    attrs = {(1 << 32) | i: (i % 3, i, (0x80000000 + i % 2) << 32, rnd.randrange(5), None if i % 4 else b"e" * (i % 9)) for i in range(1, 101)}
    recs = [((1 << 32) | i, 0xFF if i % 5 == 0 else i % 2, 0 if i % 5 == 0 else 1) for i in range(1, 101)]
    sc = (attrs, configs, [42, 43], [recs, recs[:3]], [[(1 << 32) | 500], []], [])
    res = wc.check_scenario(lambda t, r, cap: wire.encode_device(t, r, cap, limit=600), sc, limit=600)
    assert 5 <= int(res.slot_nfrag[0]) <= wire.HQWIRE_MAX_FRAGMENTS and int(res.slot_nfrag[1]) == 1
    wc.check_roundtrip(sc, res.messages(wc.tables_and_records(*sc)[1]))
