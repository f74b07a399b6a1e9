"""Parity of the HIP path (through the C ABI) with the CPU oracle on the same snapshots.  Integer work: bit-exact.

The oracle runs in `canonical` mode: HiGHS for every optimum, plus the documented tie-break so that
instances with several optimal placements have one well-defined answer.
"""
import numpy as np
import pytest

from hyperqueue_amd import abi, workloads
from hyperqueue_amd.core import SchedEnv, TaskBuilder as TB, WorkerBuilder as WB

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    from hyperqueue_amd.tick import Tick

    return Tick(abi.make_config(time_limit_s=20.0))


@pytest.fixture(scope="module")
def oracle():
    from oracle.oracle import Oracle

    return Oracle(abi.make_config(time_limit_s=20.0), canonical=True)


def assert_same(got: abi.Result, want: abi.Result):
    assert got.status == want.status
    assert got.batches == want.batches
    assert got.counts == want.counts
    assert got.records == want.records
    assert got.retracts == want.retracts
    assert sorted(got.redirects) == sorted(want.redirects)
    assert got.mn == want.mn
    assert (got.new_free == want.new_free).all()


def random_env(seed: int) -> SchedEnv:
    rng = np.random.default_rng(seed)
    env = SchedEnv(abi.make_config(reserve=int(rng.integers(0, 6)), fill_max=int(rng.integers(1, 12)), time_limit_s=20.0))
    names = ["gpus", "mem"][: int(rng.integers(0, 3))]
    for n in names:
        env.new_named_resource(n)
    n_classes = int(rng.integers(1, 5))
    builders = []
    for _ in range(n_classes):
        b = TB().cpus(int(rng.integers(1, 5)))
        for ri, _n in enumerate(names):
            if rng.random() < 0.5:
                b = b.add_resource(ri + 1, [0.25, 0.5, 1, 2][int(rng.integers(0, 4))])
        if rng.random() < 0.25:
            b = b.next_variant().cpus(int(rng.integers(1, 7)))
        builders.append(b)
    same_prio = rng.random() < 0.5
    for _ in range(int(rng.integers(1, 120))):
        b = builders[int(rng.integers(0, n_classes))]
        env.new_task(b if same_prio else b.user_priority(int(rng.integers(-2, 3))))
    for _ in range(int(rng.integers(1, 7))):
        wb = WB(int(rng.integers(2, 17)))
        for n in names:
            if rng.random() < 0.7:
                wb = wb.res_sum(n, int(rng.integers(1, 9)))
        env.new_worker(wb)
    return env


@pytest.mark.parametrize("seed", range(40))
def test_random_snapshot(seed, gpu, oracle):
    env = random_env(seed)
    snap = env.snapshot()
    cfg = env.config
    from hyperqueue_amd.tick import Tick
    from oracle.oracle import Oracle

    g, o = Tick(cfg), Oracle(cfg, canonical=True)
    assert_same(g.tick(snap), o.tick(snap))


@pytest.mark.parametrize("seed", range(12))
def test_random_multi_tick(seed):
    """schedule -> finish a few tasks -> new worker / new tasks -> schedule ...: exercises prefill sets, retracts, redirects."""
    from hyperqueue_amd.tick import Tick
    from oracle.oracle import Oracle

    rng = np.random.default_rng(1000 + seed)
    envs = []
    for _ in range(2):
        e = SchedEnv(abi.make_config(reserve=2, fill_max=5, time_limit_s=20.0))
        envs.append(e)
    g, o = Tick(envs[0].config), Oracle(envs[1].config, canonical=True)
    script = []
    n_classes = int(rng.integers(1, 3))
    for _ in range(5):
        script.append(("workers", [int(rng.integers(1, 7)) for _ in range(int(rng.integers(1, 3)))]))
        script.append(("tasks", [(int(rng.integers(0, n_classes)) + 1) for _ in range(int(rng.integers(5, 40)))]))
        script.append(("tick", None))
        script.append(("finish", int(rng.integers(0, 4))))
    for (op, arg) in script:
        results = []
        for e, be in ((envs[0], g), (envs[1], o)):
            if op == "workers":
                for c in arg:
                    e.new_worker(WB(c))
            elif op == "tasks":
                for c in arg:
                    e.new_task(TB().cpus(c))
            elif op == "tick":
                results.append(e.schedule(be))
            elif op == "finish":
                done = 0
                for t in sorted(e.tasks.values(), key=lambda t: t.id):
                    if done >= arg:
                        break
                    if t.state == 1:  # ASSIGNED
                        e.finish_task(t.id, t.worker)
                        done += 1
        if op == "tick":
            assert_same(results[0], results[1])


def test_c2_exact(gpu, oracle):
    snap = workloads.make("c2")
    got, want = gpu.tick(snap), oracle.tick(snap)
    assert_same(got, want)
    # hand-derived expectation: one batch 32768/32768 limit reached, 128 per worker, 40 prefills per worker
    assert [(b.size, b.limit, b.limit_reached) for b in got.batches] == [(32768, 32768, True)]
    assert all(len(got.assigned(w)) == 128 and len(got.prefills(w)) == 40 for w in range(256))


def test_c1_exact_and_drained_like_the_benchmark(gpu, oracle):
    """BASELINE configs[0] at full size (benchmarks/experiment-per-task-overhead.py:33-55: 1 000 single-core `sleep 0` tasks, 4 workers x 4 cores): the first
    tick against the oracle and the committed fixture's shape, then the benchmark's own loop — everything a tick hands out finishes before the next tick —
    until the queue is empty, HIP and oracle side by side, every tick record for record."""
    snap = workloads.make("c1")
    got, want = gpu.tick(snap), oracle.tick(snap)
    assert_same(got, want)
    assert [(b.size, b.limit, b.limit_reached) for b in got.batches] == [(16, 16, True)]          # 4 workers x 4 cores of 1-cpu tasks
    assert all(len(got.assigned(w)) == 4 and len(got.prefills(w)) == 40 for w in range(4))      # + proactive filling: max 40 per worker
    cfg = abi.make_config(time_limit_s=60.0)
    envs = [SchedEnv(cfg), SchedEnv(cfg)]
    for e in envs:
        e.new_workers(4, WB(4))
        e.new_tasks(1000, TB().cpus(1))
    from hyperqueue_amd.tick import Tick
    from oracle.oracle import Oracle

    backends = [Tick(cfg), Oracle(cfg, canonical=True)]
    handed, ticks = 0, 0
    while any(t.state == 0 for t in envs[0].tasks.values()) and ticks < 400:
        rs = [e.schedule(b) for e, b in zip(envs, backends)]
        assert_same(rs[0], rs[1])
        handed += sum(1 for recs in rs[0].records for (_, _, k) in recs if k == abi.HQ_REC_ASSIGN)
        for e in envs:
            # the workers answer the tick's retracts (a task the tick took out of a prefill set is Retracting{old} until `old` lets it go: reactor.rs:462-508) ...
            by_worker = {}
            for t in e.tasks.values():
                if t.state == 4:
                    by_worker.setdefault(t.worker, []).append(t.id)
            for wid, tids in sorted(by_worker.items()):
                e.retract_response(wid, sorted(tids))
            # ... and `sleep 0`: every assigned task has finished before the next tick
            for t in sorted(e.tasks.values(), key=lambda t: t.id):
                if t.state == 1:
                    e.finish_task(t.id, t.worker)
        ticks += 1
    assert not any(t.state == 0 for t in envs[0].tasks.values()) and handed >= 1000 - 4 * 40
    backends[0].close()


def test_c3_reduced_exact(gpu, oracle):
    snap = workloads.make("c3", n_tasks=60_000, n_workers=48)
    assert_same(gpu.tick(snap), oracle.tick(snap))


def test_c4_reduced_exact(gpu, oracle):
    snap = workloads.make("c4", n_tasks=40_000, n_workers=24)
    assert_same(gpu.tick(snap), oracle.tick(snap))


def test_c3_full_properties(gpu):
    """Full BASELINE size: size-independent properties + equal MILP objective with plain HiGHS on the oracle's model."""
    from oracle.oracle import Oracle

    snap = workloads.make("c3")
    got = gpu.tick(snap)
    W, R = len(snap.worker_id), snap.n_resources
    ids = snap.task_id
    rq_of = dict(zip(ids.tolist(), snap.task_rq.tolist()))
    seen = set()
    used = np.zeros((W, R), np.int64)
    per_rq = {}
    for w in range(W):
        for (t, v, k) in got.records[w]:
            assert t in rq_of and t not in seen  # every record is a distinct ready task
            seen.add(t)
            q = rq_of[t]
            per_rq.setdefault(q, []).append(t)
            if k == abi.HQ_REC_ASSIGN:
                for (r, kind, a) in snap.requests[q][v]["entries"]:
                    used[w, r] += a
    assert (used <= snap.worker_free.astype(np.int64)).all()  # no worker is oversubscribed
    assert (snap.worker_free.astype(np.int64) - used == got.new_free.astype(np.int64)).all()
    for q, ts in per_rq.items():  # take_tasks: the lowest ids of every queue (one priority level in c3), no holes
        mine = np.sort(np.asarray(ts, np.uint64))
        allq = ids[snap.task_rq == q]
        assert (mine == allq[: len(mine)]).all()
    cd = got.counts_dict()
    for w in range(W):  # records agree with the counts
        c = {}
        for (t, v, k) in got.records[w]:
            if k == abi.HQ_REC_ASSIGN:
                c[(rq_of[t], v)] = c.get((rq_of[t], v), 0) + 1
        assert c == {(q, v): n for (q, v, ww), n in cd.items() if ww == w}
    # objective equality with HiGHS (tier T2): evaluate our counts in the oracle's model
    o = Oracle(abi.make_config(time_limit_s=60.0))
    o.tick(snap)
    m = o.last_model()
    x = np.zeros(len(m["obj"]))
    for j in range(len(x)):
        if m["ctype"][j] == 0:
            x[j] = cd.get((int(m["crq"][j]), int(m["cvariant"][j]), int(m["cworker"][j])), 0)
    mine_obj = float(np.dot(m["obj"], x))
    # HiGHS stops inside its own absolute gap (1e-6): ours must be at least as good, and close
    assert mine_obj >= m["objective"] - 1e-9 and abs(mine_obj - m["objective"]) <= 1e-4 * abs(m["objective"]), (mine_obj, m["objective"])
    for i in range(len(m["rhs"])):  # and every row of the reference's model holds
        a, b = m["roff"][i], m["roff"][i + 1]
        act = float(np.dot(m["rcoef"][a:b], x[m["rcol"][a:b]]))
        if m["rtype"][i] == 1:
            assert act <= m["rhs"][i] + 1e-6


# ------------------------------------------------------------------------------------------------------ worker shards
def _merged_from_sequential_shards(snap, cfg, world, cap):
    """Runs the `world` shards one after the other on this GPU (no collective) and merges their device sinks."""
    from hyperqueue_amd import sharded

    W = len(snap.worker_id)
    sinks, last = [], None
    for r in range(world):
        st = sharded.ShardedTick(cfg, rank=r, world=world, records_per_shard=cap)
        res_c, sink = st.tick_local(snap.to_c(), W)
        sinks.append(sink.cpu().numpy().copy())
        last = abi.parse_result(res_c, W, snap.n_resources)
        owned = [len(x) for x in last.records]
        assert sum(owned) == 0  # records live in the sink, not in host memory
        st.t.close()
    return last, sharded.merge_shards(np.concatenate(sinks), world, W, cap)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_random_snapshots(world, oracle):
    from oracle.oracle import Oracle

    for seed in (3, 11, 17, 29):
        env = random_env(seed)
        snap = env.snapshot()
        want = Oracle(env.config, canonical=True).tick(snap)
        last, records = _merged_from_sequential_shards(snap, env.config, world, 512)
        assert records == want.records
        assert last.counts == want.counts and last.batches == want.batches and (last.new_free == want.new_free).all()


def test_sharded_c3_reduced(gpu, oracle):
    from hyperqueue_amd import sharded

    snap = workloads.make("c3", n_tasks=60_000, n_workers=48)
    want = oracle.tick(snap)
    cfg = abi.make_config(time_limit_s=20.0)
    last, records = _merged_from_sequential_shards(snap, cfg, 4, 8192)
    assert records == want.records
    # every shard emitted only its own workers
    st = sharded.ShardedTick(cfg, rank=1, world=4, records_per_shard=8192)
    res_c, sink = st.tick_local(snap.to_c(), len(snap.worker_id))
    off = np.ctypeslib.as_array(res_c.rec_off, shape=(len(snap.worker_id) + 1,))
    for w in range(len(snap.worker_id)):
        mine = sharded.owner_of(int(snap.worker_id[w]), 4) == 1
        assert (off[w + 1] - off[w] == len(want.records[w])) if mine else (off[w + 1] == off[w])


def test_sharded_sink_too_small():
    from hyperqueue_amd import sharded
    from hyperqueue_amd.tick import HqTickError

    snap = workloads.make("c2", n_tasks=5_000, n_workers=16)
    st = sharded.ShardedTick(abi.make_config(), rank=0, world=2, records_per_shard=8)
    with pytest.raises(HqTickError) as e:
        st.tick_local(snap.to_c(), len(snap.worker_id))
    assert e.value.code == abi.HQTICK_E_CAPACITY


# ------------------------------------------------------------------------------------------------------ BASELINE sizes
def test_c4_full_sharded_properties():
    """BASELINE configs[3]: 1 M tasks with 2-variant OR-lists, 4096 workers, 8 worker shards.  The oracle needs minutes at this size,
    so: every shard through the HIP library, merged, checked for the size-independent properties + agreement of the replicated parts."""
    from hyperqueue_amd import sharded

    snap = workloads.make("c4")
    cfg = abi.make_config(time_limit_s=20.0)
    W, R, world, cap = len(snap.worker_id), snap.n_resources, 8, 1 << 18
    sinks, results = [], []
    for r in range(world):
        st = sharded.ShardedTick(cfg, rank=r, world=world, records_per_shard=cap)
        res_c, sink = st.tick_local(snap.to_c(), W)
        sinks.append(sink.cpu().numpy().copy())
        results.append(abi.parse_result(res_c, W, R))
        st.t.close()
    records = sharded.merge_shards(np.concatenate(sinks), world, W, cap)
    for r in results[1:]:  # replicated stages agree on every rank
        assert r.counts == results[0].counts and r.batches == results[0].batches and (r.new_free == results[0].new_free).all()
    got = results[0]
    ids = snap.task_id
    rq_of = snap.task_rq
    idx_of = lambda t: int(t & 0xFFFFFFFF) - 1  # job 1, task 1..n
    seen = np.zeros(len(ids), bool)
    used = np.zeros((W, R), np.int64)
    cd = got.counts_dict()
    per_rq_max = {}
    for w in range(W):
        c = {}
        for (t, v, k) in records[w]:
            i = idx_of(t)
            assert ids[i] == t and not seen[i]
            seen[i] = True
            q = int(rq_of[i])
            per_rq_max[q] = max(per_rq_max.get(q, -1), i)
            if k == abi.HQ_REC_ASSIGN:
                c[(q, v)] = c.get((q, v), 0) + 1
                for (res, kind, a) in snap.requests[q][v]["entries"]:
                    used[w, res] += a
        assert c == {(q, v): n for (q, v, ww), n in cd.items() if ww == w}  # records agree with the counts
    assert (used <= snap.worker_free.astype(np.int64)).all()
    assert (snap.worker_free.astype(np.int64) - used == got.new_free.astype(np.int64)).all()
    for q, last in per_rq_max.items():  # take_tasks: a prefix of every queue, no holes (one priority level)
        mine = seen[: last + 1][rq_of[: last + 1] == q]
        assert mine.all()
    assert seen.sum() == sum(len(r) for r in records) > 0
    # ... and the COUNTS, record for record: the merged shards against the committed full-size fixture (the canonical oracle's answer:
    # HiGHS on the one distinct 16-column worker block, the batch-size rows verified afterwards — tests/golden/make_fixtures.py big)
    import json, os, sys

    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sys.path.insert(0, golden)
    from make_fixtures import big_digest

    got.records = records
    exp = json.load(open(os.path.join(golden, "big", "c4_full.json")))["expect"]
    g = json.loads(json.dumps(big_digest(got)))
    for key in exp:
        assert g[key] == exp[key], key


def test_c4_full_equals_oracle_cold_tick(gpu):
    """BASELINE configs[3] at full size against the canonical oracle run here (a few seconds: one distinct worker class)."""
    from oracle.oracle import Oracle

    snap = workloads.make("c4")
    got = gpu.tick(snap)
    want = Oracle(abi.make_config(time_limit_s=60.0), canonical=True).tick(snap)
    assert want.is_optimal
    assert_same(got, want)


def test_c3_full_equals_oracle_cold_tick(gpu):
    """BASELINE configs[2] at full size against the canonical oracle: 1 M tasks x 1024 workers (the oracle needs ~2 s here)."""
    from oracle.oracle import Oracle

    snap = workloads.make("c3")
    got = gpu.tick(snap)
    want = Oracle(abi.make_config(time_limit_s=60.0), canonical=True).tick(snap)
    assert_same(got, want)


def test_library_allgather_single_rank(gpu):
    """hqtick_comm_unique_id / hqtick_comm_init / hqtick_shard_allgather (RCCL, loaded by libhqtick.so itself) with a world of one: the gathered
    buffer is this rank's record sink, and the merged records equal the unsharded tick's.  (More ranks need more GPUs: tests/test_gpu_multi.py.)"""
    from hyperqueue_amd import sharded

    snap = workloads.make("c2", n_tasks=5000, n_workers=16)
    want = gpu.tick(snap)
    st = sharded.ShardedTick(abi.make_config(time_limit_s=20.0), rank=0, world=1, records_per_shard=1 << 14, collective="library")
    got = st.tick(snap)
    assert st.collective == "library"
    assert got.records == want.records and got.counts == want.counts
    st.t.close()


@pytest.mark.parametrize("tpw", [512, 1024, 2048])
def test_slice_size_is_not_observable(tpw, oracle, monkeypatch):
    """The ready set is scanned in per-wavefront slices (256 tasks; larger ones from 8 M tasks on, `HQTICK_TPW` as the tuning knob): every
    output must be the same whatever the slice size — selection order (`pop_first`, taskqueue.rs:273-302) included."""
    from hyperqueue_amd.tick import Tick
    from oracle.oracle import Oracle

    monkeypatch.setenv("HQTICK_TPW", str(tpw))
    snap = workloads.make("c3", n_tasks=60_000, n_workers=64)
    assert_same(Tick(abi.make_config(time_limit_s=20.0)).tick(snap), oracle.tick(snap))
    for seed in (3, 11, 27, 33):
        env = random_env(seed)
        s = env.snapshot()
        assert_same(Tick(env.config).tick(s), Oracle(env.config, canonical=True).tick(s))


def test_long_rows_scan_kernel_equals_short_rows_kernel(monkeypatch):
    """K1b has two forms (kernels.hip): a wavefront per group row for rows of up to 1024 slices, a workgroup per row above.  2.5 M ready tasks in
    slices of 256 are rows of 9 766 slices — three 4096-entry steps of the workgroup form — and rows of 611 with slices of 4096: same tick, same bytes."""
    from hyperqueue_amd.tick import Tick

    snap = workloads.make("c3", n_tasks=2_500_000, n_workers=512)
    results = []
    for tpw in (256, 4096):
        monkeypatch.setenv("HQTICK_TPW", str(tpw))
        results.append(Tick(abi.make_config(time_limit_s=20.0)).tick(snap))
    a, b = results
    assert a.status == b.status == 0 and a.is_optimal and b.is_optimal
    assert a.batches == b.batches and a.counts == b.counts
    assert a.records == b.records and sum(len(r) for r in a.records) > 50_000
    assert (a.new_free == b.new_free).all()


def test_c4p_full_sharded_replicas_agree():
    """BASELINE configs[3] AS WRITTEN (c4p: three priority levels x 4096 workers x 2-variant OR-lists, 65 552 columns) the way the contract runs it — workers
    hash-sharded, here over EIGHT shards: every rank solves the same model on its own context, the replicated stages must agree bit for bit, and the merged shard
    records must be the plain tick's records."""
    from hyperqueue_amd import sharded
    from hyperqueue_amd.tick import Tick

    snap = workloads.make("c4p")
    cfg = abi.make_config(time_limit_s=20.0)
    W, R, world, cap = len(snap.worker_id), snap.n_resources, 8, 1 << 16
    sinks, results = [], []
    for r in range(world):
        st = sharded.ShardedTick(cfg, rank=r, world=world, records_per_shard=cap)
        res_c, sink = st.tick_local(snap.to_c(), W)
        sinks.append(sink.cpu().numpy().copy())
        results.append(abi.parse_result(res_c, W, R))
        assert st.t.kernel_stats()["price_sweeps"] > 0
        st.t.close()
    records = sharded.merge_shards(np.concatenate(sinks), world, W, cap)
    for r in results[1:]:
        assert r.is_optimal and r.counts == results[0].counts and r.batches == results[0].batches and (r.new_free == results[0].new_free).all()
    t = Tick(cfg)
    try:
        plain = t.tick(snap)
    finally:
        t.close()
    assert plain.status == abi.HQTICK_DONE and plain.is_optimal
    assert plain.counts == results[0].counts and records == plain.records and (plain.new_free == results[0].new_free).all()


def test_c3p_full_sharded_replicas_agree():
    """BASELINE C3 with three priority levels (one coupled model of 8 205 columns, solved by the price sweeps) as FOUR worker shards: every rank solves the same
    model on its own context — the replicated stages must agree bit for bit (counts, batches, free vectors: the price path has no clock on a tick that certifies), and
    the merged shard records must be the plain tick's records."""
    from hyperqueue_amd import sharded

    snap = workloads.make("c3p")
    cfg = abi.make_config(time_limit_s=20.0)
    W, R, world, cap = len(snap.worker_id), snap.n_resources, 4, 1 << 17
    sinks, results = [], []
    for r in range(world):
        st = sharded.ShardedTick(cfg, rank=r, world=world, records_per_shard=cap)
        res_c, sink = st.tick_local(snap.to_c(), W)
        sinks.append(sink.cpu().numpy().copy())
        results.append(abi.parse_result(res_c, W, R))
        assert st.t.kernel_stats()["price_sweeps"] > 0
        st.t.close()
    records = sharded.merge_shards(np.concatenate(sinks), world, W, cap)
    for r in results[1:]:
        assert r.is_optimal and r.counts == results[0].counts and r.batches == results[0].batches and (r.new_free == results[0].new_free).all()
    from hyperqueue_amd.tick import Tick

    t = Tick(cfg)
    try:
        plain = t.tick(snap)
    finally:
        t.close()
    assert plain.counts == results[0].counts and records == plain.records and (plain.new_free == results[0].new_free).all()


@pytest.mark.parametrize("n_levels,n_classes", [(1, 8), (3, 8), (4, 8), (5, 8), (9, 3), (3, 20), (70, 8)])
def test_a_tick_that_discovers_its_levels_scans_right_behind_the_discovery(n_levels, n_classes, oracle, monkeypatch):
    """Round 6: a tick without a level table (the first tick on a ready set; every tick under HQTICK_FLAG_NO_TICK_CACHES) launches K1 behind the discovery kernels
    without waiting for them, sized for four levels, and K1 reads the table from HBM.  <= 4 levels x <= 16 requests: that scan stands.  More levels: K1 refuses, the
    tick reads the table and scans again with the general variant.  More than 16 requests: no speculation.  Either way the answer is the oracle's — and the same as
    with the speculation switched off (HQTICK_NO_SPEC_SCAN), on the first tick and on the second."""
    from hyperqueue_amd.core import priority_from_user
    from hyperqueue_amd.tick import Tick

    snap = workloads.make("c3", seed=5, n_tasks=6_000, n_workers=6)
    rng = np.random.default_rng(n_levels * 100 + n_classes)
    if n_classes != 8:   # requests cpus = 1 .. n_classes, so that Q differs from the c3 table
        snap.requests = [[workloads._variant([(0, 1 + (q % 4))])] for q in range(n_classes)]
    snap.task_rq = rng.integers(0, n_classes, len(snap.task_id)).astype(np.uint32)
    snap.task_priority = np.asarray([priority_from_user(int(p)) for p in rng.integers(0, n_levels, len(snap.task_id))], np.uint64)
    want = oracle.tick(snap)
    outs = []
    for no_spec in (False, True):
        if no_spec:
            monkeypatch.setenv("HQTICK_NO_SPEC_SCAN", "1")
        else:
            monkeypatch.delenv("HQTICK_NO_SPEC_SCAN", raising=False)
        for flags in (0, abi.HQTICK_FLAG_NO_TICK_CACHES):
            t = Tick(abi.make_config(time_limit_s=20.0, flags=flags))
            try:
                t.upload_ready(snap.task_id, snap.task_priority, snap.task_rq, sorted_=True)
                first = t.tick(snap, resident=True)
                second = t.tick(snap, resident=True)
            finally:
                t.close()
            # what the scan decides — the batches (create_task_batches on the histogram) — must be the oracle's; the placement too wherever the product's answer is the
            # canonical one (a coupled model of several levels may stop at its certificate: then the four runs are compared with one another instead)
            assert first.status == want.status and first.batches == want.batches and second.batches == want.batches
            if first.is_canonical:
                assert_same(first, want)
            assert_same(second, first)
            outs.append(first)
    for o in outs[1:]:
        assert_same(o, outs[0])
