"""Scenario builders shared by the CPU (debug hook) and GPU (hqwire_encode_device) tests of the wire encoding."""
import random

import numpy as np

from hyperqueue_amd import wire
from oracle import wire_oracle as wo


def random_scenario(seed, n_workers=None, max_rec=30):
    rnd = random.Random(seed)
    n_cfg = rnd.randint(1, 12)
    configs = []
    for _ in range(n_cfg):
        tl = None if rnd.random() < 0.5 else (rnd.randrange(0, 10**6), rnd.randrange(0, 10**9))
        configs.append((tl, rnd.randbytes(rnd.choice([0, 1, 7, 64, 300, 1500]))))
    n_tasks = rnd.randint(1, 400)
    ids = sorted(rnd.sample(range(1, 5000), n_tasks))
    ids = [(rnd.randint(1, 3) << 32) | t for t in ids]
    ids = sorted(set(ids))
    attrs = {}
    for t in ids:
        e = rnd.choice([None, None, b"", rnd.randbytes(rnd.randint(1, 40))])
        attrs[t] = (rnd.randrange(0, 50), rnd.randrange(0, 1 << 32), rnd.randrange(0, 1 << 64), rnd.randrange(n_cfg), e)
    W = n_workers if n_workers is not None else rnd.randint(1, 9)
    worker_ids = sorted(rnd.sample(range(1, 500), W))
    pool = ids[:]
    rnd.shuffle(pool)
    records, retracts = [], []
    for _ in range(W):
        k = min(len(pool), rnd.choice([0, 1, 2, rnd.randint(0, max_rec)]))
        recs = []
        for _ in range(k):
            kind = rnd.choice([0, 1, 1])
            recs.append((pool.pop(), 0xFF if kind == 0 else rnd.randrange(0, 4), kind))
        records.append(recs)
        retracts.append([rnd.choice(ids) for _ in range(rnd.choice([0, 0, 1, 5]))])
    mn = []
    for _ in range(rnd.choice([0, 0, 1, 3])):
        if not pool or W < 2:
            break
        mn.append((pool.pop(), rnd.sample(range(W), rnd.randint(2, min(W, 4)))))
    return attrs, configs, worker_ids, records, retracts, mn


def oracle_messages(attrs, configs, worker_ids, records, retracts, mn):
    a = {t: wo.TaskAttr(*v) for t, v in attrs.items()}
    c = [wo.Config(tl, body) for (tl, body) in configs]
    return wo.send_messages(a, c, worker_ids, records, retracts, mn)


def tables_and_records(attrs, configs, worker_ids, records, retracts, mn):
    return wire.WireTables.build(attrs, configs), wire.WireRecords.build(worker_ids, records, retracts, mn)


def check_scenario(encode, sc, capacity=1 << 22):
    """`encode(tables, records, capacity) -> WireResult` against the oracle: same messages, byte for byte, in send order."""
    t, r = tables_and_records(*sc)
    res = encode(t, r, capacity)
    assert res.status == wire.HQWIRE_OK
    assert (res.slot_status == 0).all()
    got, want = res.messages(r), oracle_messages(*sc)
    assert len(got) == len(want)
    for (gw, gb), (ww, wb) in zip(got, want):
        assert gw == ww
        assert gb == wb, (gw, gb[:64].hex(), wb[:64].hex())
    assert res.total_bytes == sum(len(b) for _, b in want)
    return res


def tick_scenario():
    """A real tick's mapping (oracle tick on a small cluster with priorities, prefill and a multi-node task) dressed with task attributes."""
    from hyperqueue_amd.core import SchedEnv, TaskBuilder as TB, WorkerBuilder as WB
    from oracle.oracle import Oracle

    env = SchedEnv()
    env.new_workers(5, WB(8))
    env.new_tasks(120, TB().cpus(1))
    env.new_tasks(10, TB().cpus(2).user_priority(3))
    snap = env.snapshot()
    res = Oracle(env.config, canonical=True).tick(snap)
    rnd = random.Random(5)
    configs = [(None, b"program-a" * 20), ((3600, 0), b"program-b" * 50), ((1, 500), b"")]
    attrs = {}
    for recs in res.records:
        for (t, v, k) in recs:
            attrs[t] = (rnd.randrange(4), rnd.randrange(100), (0x80000000 + rnd.randrange(4)) << 32, rnd.randrange(3), None if rnd.random() < 0.7 else b"entry-%d" % (t & 0xFFFF))
    worker_ids = [int(w) for w in snap.worker_id]
    assert sum(len(r) for r in res.records) > 40
    return attrs, configs, worker_ids, res.records, res.retracts, []
