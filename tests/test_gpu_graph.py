"""Device-resident dependency graph (hqtick_graph_*) against oracle/graph_oracle.py: the reference's own
dependency tests, randomised DAG churn, hub / chain shapes, error behaviour, and the interplay with the resident ready set."""
import numpy as np
import pytest

from graph_cases import CASES
from hyperqueue_amd import abi, workloads

pytestmark = pytest.mark.gpu


@pytest.fixture()
def T():
    from hyperqueue_amd.tick import Tick

    t = Tick(abi.make_config(time_limit_s=20.0))
    t.upload_ready(np.zeros(0, np.uint64), np.zeros(0, np.uint64), np.zeros(0, np.uint32))
    return t


def _add(t, tasks):
    return t.graph_add_tasks([i for i, *_ in tasks], [p for _, p, _, _ in tasks], [q for _, _, q, _ in tasks], [list(d) for *_, d in tasks]).tolist()


@pytest.mark.parametrize("name", sorted(CASES))
def test_reference_case(T, name):
    n_ready = 0
    for st in CASES[name]():
        op = st[0]
        if op == "add":
            ready = _add(T, [(i, 0, 0, deps) for i, deps in st[1]])
            n_ready += len(ready)
            if "ready" in st[2]:
                assert ready == sorted(st[2]["ready"])
            for i, n in st[2].get("unfinished", {}).items():
                assert T.graph_unfinished([i])[0] == n
        elif op == "take":
            assert T.ready_remove(np.asarray(st[1], np.uint64)) == len(st[1])  # stands for the tick handing them out
            n_ready -= len(st[1])
        elif op == "finish":
            rel, unknown = T.graph_finish(st[1])
            assert unknown == 0 and rel.tolist() == sorted(st[2]["released"])
            n_ready += len(rel)
            for i, n in st[2].get("unfinished", {}).items():
                assert T.graph_unfinished([i])[0] == n
        elif op == "fail":
            removed = T.graph_remove([st[1]], recursive=True).tolist()
            assert removed == sorted(st[2]["removed"])
        elif op == "collect":
            removed = T.graph_remove([st[1]], recursive=True).tolist()
            assert removed == sorted([st[1]] + st[2]["consumers"])
            n_ready -= 1
        elif op == "exists":
            for i, e in st[1].items():
                assert (T.graph_unfinished([i])[0] != 0xFFFFFFFF) == e
        if op not in ("fail",):
            assert T.ready_count() == n_ready


def test_dep_later_in_batch_is_dropped(T):
    assert _add(T, [(1, 0, 0, [2]), (2, 0, 0, [1]), (3, 0, 0, [3])]) == [1, 3]
    assert T.graph_unfinished([1, 2, 3, 4]).tolist() == [0, 1, 0, 0xFFFFFFFF]


def test_errors_leave_the_graph_unchanged(T):
    from hyperqueue_amd.tick import HqTickError

    assert _add(T, [(5, 1, 0, []), (6, 1, 0, [5])]) == [5]
    with pytest.raises(HqTickError) as e:
        _add(T, [(9, 1, 0, []), (6, 1, 0, [])])  # 6 exists: core.rs:217
    assert e.value.code == abi.HQTICK_E_INVALID
    with pytest.raises(HqTickError) as e:
        _add(T, [(10, 1, 0, []), (11, 1, 0, []), (10, 1, 0, [])])
    assert e.value.code == abi.HQTICK_E_INVALID
    assert T.graph_unfinished([5, 6, 9, 10, 11]).tolist() == [0, 1, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF]
    assert T.graph_stats()["n_tasks"] == 2 and T.ready_count() == 1
    assert _add(T, [(9, 1, 0, [6]), (10, 1, 0, [])]) == [10]
    T.ready_remove([5])
    rel, unknown = T.graph_finish([5, 77, 5])
    assert rel.tolist() == [6] and unknown == 2
    with pytest.raises(HqTickError):  # 9 still waits for 6: reactor.rs:551-555 unreachable!()
        T.graph_finish([9])


def _random_dag_run(T, seed, n_rounds, batch, p_unknown=0.05):
    from oracle.graph_oracle import GraphOracle

    rng = np.random.default_rng(seed)
    g = GraphOracle()
    next_id, running = 1, []
    for rnd in range(n_rounds):
        # --- submit a batch: deps on live tasks (any state), on finished ones, on later ones of the same batch
        n = int(rng.integers(1, batch))
        ids = [(int(rng.integers(1, 4)) << 32) | (next_id + k) for k in range(n)]
        next_id += n
        if rng.random() < 0.5:
            rng.shuffle(ids)
        live = list(g.tasks)
        tasks = []
        for k, i in enumerate(ids):
            pool = live + ids  # earlier AND later ids of the batch
            nd = int(min(len(pool), rng.poisson(2.0)))
            deps = set(int(x) for x in rng.choice(pool, nd, replace=False)) if nd else set()
            if rng.random() < p_unknown:
                deps.add(int(rng.integers(1 << 40, 1 << 41)))
            tasks.append((i, int(rng.integers(0, 5)) << 32, int(rng.integers(0, 3)), sorted(deps)))
        want = g.on_new_tasks(tasks)
        assert _add(T, tasks) == want
        # --- the scheduler hands out some ready tasks
        rd = sorted(g.ready)
        take = [i for i in rd if rng.random() < 0.6]
        if take:
            g.take_from_ready(take)
            assert T.ready_remove(np.asarray(take, np.uint64)) == len(take)
            running += take
        # --- some running tasks finish (plus unknown ids and a duplicate)
        rng.shuffle(running)
        k = int(rng.integers(0, len(running) + 1))
        fin, running = running[:k], running[k:]
        extra = [int(rng.integers(1 << 42, 1 << 43))] if rng.random() < 0.3 else []
        if fin and rng.random() < 0.3:
            extra.append(fin[0])
        rel_w, unk_w = g.task_finished(fin + extra)
        rel, unk = T.graph_finish(fin + extra) if fin + extra else (np.zeros(0, np.uint64), 0)
        assert rel.tolist() == rel_w and unk == unk_w
        # --- sometimes a task fails / is cancelled: it and its transitive consumers leave
        if rng.random() < 0.35 and g.tasks:
            victims = [int(x) for x in rng.choice(list(g.tasks), int(min(len(g.tasks), rng.integers(1, 4))), replace=False)]
            rec = bool(rng.random() < 0.8)
            if not rec:  # the reference only removes a lone task when nothing depends on it
                victims = [v for v in victims if not g.tasks[v].consumers]
            if victims:
                rm_w, _ = g.remove(victims, rec)
                assert T.graph_remove(victims, recursive=rec).tolist() == rm_w
                running = [r for r in running if r in g.tasks]
        assert T.ready_count() == len(g.ready)
        assert T.graph_stats()["n_tasks"] == len(g.tasks)
        probe = list(g.tasks)[:200] + [next_id + 10_000]
        assert T.graph_unfinished(probe).tolist() == [g.unfinished(i) for i in probe]
    return g


@pytest.mark.parametrize("seed", range(12))
def test_random_dag_churn(T, seed):
    _random_dag_run(T, 9000 + seed, n_rounds=40, batch=60)


def test_random_dag_churn_long_with_pool_growth(T):
    g = _random_dag_run(T, 4242, n_rounds=120, batch=400)
    st = T.graph_stats()
    assert st["n_slots"] < 120 * 400  # slots were recycled
    assert st["n_edges_live"] <= st["n_edges_pool"]


def test_ready_set_after_graph_ops_ticks_like_the_oracle(T):
    """the tasks the graph released are really in the resident ready set: a resident tick equals the oracle's tick on the oracle's ready set"""
    from oracle.oracle import Oracle

    g = _random_dag_run(T, 77, n_rounds=30, batch=80)
    base = workloads.make("c1")
    ids = np.asarray(sorted(g.ready), np.uint64)
    assert len(ids) > 0
    snap_kw = {f: getattr(base, f) for f in ("n_resources", "worker_id", "worker_total", "worker_free", "worker_remaining_ns", "worker_min_utilization", "worker_flags",
                                             "worker_group", "n_groups", "blocked", "assigned", "prefilled", "prefill", "worker_map_rank")}
    reqs = [base.requests[0]] * 3  # rq 0..2 all `cpus = 1`
    full = abi.Snapshot(**snap_kw, requests=reqs, task_id=ids, task_priority=np.asarray([g.ready[int(i)][0] for i in ids], np.uint64),
                        task_rq=np.asarray([g.ready[int(i)][1] for i in ids], np.uint32))
    stripped = abi.Snapshot(**snap_kw, requests=reqs, task_id=np.zeros(0, np.uint64), task_priority=np.zeros(0, np.uint64), task_rq=np.zeros(0, np.uint32))
    got = T.tick(stripped, resident=True)
    want = Oracle(abi.make_config(time_limit_s=20.0), canonical=True).tick(full)
    assert got.batches == want.batches and got.counts == want.counts and got.records == want.records


def test_hub_and_chain(T):
    """one producer with 100 000 consumers (the wide-run kernel) and a 300-deep chain (level-synchronous recursive removal)"""
    hub = 1
    cons = list(range(2, 100_002))
    assert _add(T, [(hub, 0, 0, [])] + [(c, 0, 0, [hub]) for c in cons]) == [hub]
    chain = list(range(200_000, 200_300))
    assert _add(T, [(chain[0], 0, 1, [hub])] + [(chain[k], 0, 1, [chain[k - 1]]) for k in range(1, 300)]) == []
    assert T.graph_unfinished([cons[0], cons[-1], chain[0], chain[5]]).tolist() == [1, 1, 1, 1]
    removed = T.graph_remove([chain[0]], recursive=True).tolist()
    assert removed == chain
    T.ready_remove([hub])
    rel, unk = T.graph_finish([hub])
    assert unk == 0 and rel.tolist() == cons
    assert T.ready_count() == len(cons) and T.graph_stats()["n_tasks"] == len(cons)


def test_c5_shape_drains_level_by_level(T):
    """BASELINE config 5 shape, reduced: random DAG with Poisson(3) fan-in from lower ids; finishing everything that is ready, wave after
    wave, releases every task exactly once and in the same waves as the oracle"""
    from oracle.graph_oracle import GraphOracle

    n = 60_000
    ids, prio, rq, off, dep = workloads.make_dag(n, seed=5)
    g = GraphOracle()
    want = g.on_new_tasks([(int(ids[i]), int(prio[i]), int(rq[i]), [int(x) for x in dep[off[i]:off[i + 1]]]) for i in range(n)])
    ready = T.graph_add_tasks(ids, prio, rq, (off, dep)).tolist()
    assert ready == want
    unf = T.graph_unfinished(ids)
    assert unf.tolist() == [g.unfinished(int(i)) for i in ids]
    done, waves = 0, 0
    while ready:
        assert T.ready_remove(np.asarray(ready, np.uint64)) == len(ready)
        g.take_from_ready(ready)
        rel_w, _ = g.task_finished(ready)
        rel, unk = T.graph_finish(ready)
        assert unk == 0 and rel.tolist() == rel_w
        done += len(ready); waves += 1
        ready = rel.tolist()
    assert done == n and waves > 3 and T.graph_stats()["n_tasks"] == 0


def test_c5_dag_with_worker_churn_equals_oracle(T):
    """BASELINE config 5, reduced: DAG release + 25 % worker churn per tick; every tick of the resident path equals the oracle's tick on the
    full snapshot of the same moment, and every release equals the graph oracle's"""
    from oracle.graph_oracle import GraphOracle
    from oracle.oracle import Oracle

    n, W = 3_000, 4
    ids, prio, rq, off, dep = workloads.make_dag(n, seed=3)
    rq = (rq % np.uint32(3)).astype(np.uint32)  # 1c / 4c / 2c+1g: with all eight classes the unsaturated model is beyond a 20 s exact solve
    meta = {int(i): (int(p), int(q)) for i, p, q in zip(ids, prio, rq)}
    g = GraphOracle()
    g.on_new_tasks([(int(ids[i]), int(prio[i]), int(rq[i]), [int(x) for x in dep[off[i]:off[i + 1]]]) for i in range(n)])
    T.graph_add_tasks(ids, prio, rq, (off, dep))
    drv = workloads.DagChurn(n_workers=W, churn=0.25, seed=1)
    o = Oracle(abi.make_config(time_limit_s=20.0), canonical=True)
    handed, steps = 0, 0
    while g.tasks and steps < 200:
        rid = sorted(g.ready)
        want = o.tick(drv.snapshot(rid, [g.ready[i][0] for i in rid], [g.ready[i][1] for i in rid]))
        got = T.tick(drv.snapshot(), resident=True)
        T.ready_consume_last()
        assert got.batches == want.batches and got.counts == want.counts and got.records == want.records
        rec_off = np.zeros(W + 1, np.int64)
        rec_off[1:] = np.cumsum([len(r) for r in got.records])
        rec_task = np.asarray([t for r in got.records for (t, _, _) in r], np.uint64)
        g.take_from_ready(rec_task.tolist())
        finished, returned = drv.after_tick(rec_off, rec_task)
        for i in returned.tolist():
            g.ready[i] = meta[i]
        if len(returned):
            T.ready_add(returned, [meta[int(i)][0] for i in returned], [meta[int(i)][1] for i in returned])
        rel_w, _ = g.task_finished(finished.tolist())
        rel, unk = T.graph_finish(finished) if len(finished) else (np.zeros(0, np.uint64), 0)
        assert unk == 0 and rel.tolist() == rel_w
        assert T.ready_count() == len(g.ready) and T.graph_stats()["n_tasks"] == len(g.tasks)
        handed += len(finished); steps += 1
    assert not g.tasks and handed == n and steps > 5


# ---- EXTENSION: hqtick_graph_blevel (include/hqtick.h) — no reference counterpart, parity unpinned; the checker is oracle/graph_oracle.py's definition --------------
def test_the_low_priority_bits_stay_zero_unless_the_host_asks(T):
    """the reference never writes Priority's low 32 bits (common/priority.rs:43-66): neither does a graph nobody called hqtick_graph_blevel on"""
    ids, prio, rq, off, dep = workloads.make_dag(5_000, seed=2)
    ready = T.graph_add_tasks(ids, prio, rq, (off, dep))
    assert (T.graph_priorities(ids) & np.uint64(0xFFFFFFFF) == 0).all()
    T.ready_remove(ready)
    rel, _ = T.graph_finish(ready)
    assert len(rel) and (T.graph_priorities(rel) & np.uint64(0xFFFFFFFF) == 0).all()


@pytest.mark.parametrize("shape,seed", [("random", 1), ("random", 4), ("layered", 2)])
def test_blevel_equals_the_oracles_definition(T, shape, seed):
    from oracle.graph_oracle import GraphOracle

    n = 30_000
    ids, prio, rq, off, dep = workloads.make_dag(n, seed=seed) if shape == "random" else workloads.make_dag_layered(n, width=600, seed=seed)
    g = GraphOracle()
    g.on_new_tasks([(int(ids[i]), int(prio[i]), int(rq[i]), [int(x) for x in dep[off[i]:off[i + 1]]]) for i in range(n)])
    ready = T.graph_add_tasks(ids, prio, rq, (off, dep)).tolist()
    for wave in range(3):  # on the full graph, and again after waves of finishes and a recursive removal have eaten into it
        info = T.graph_blevel(update_ready=True)
        depth = g.apply_blevels()
        live = np.asarray(sorted(g.tasks), np.uint64)
        assert info["max_level"] == depth and info["sweeps"] >= 1
        assert T.graph_priorities(live).tolist() == [g.tasks[int(i)].priority for i in live]
        assert info["ready_updated"] == len(g.ready)
        # the ready set carries the new priorities: the next tick ranks by them.  (checked through the graph's released tasks below and through a tick in the test after this one)
        take = ready[: max(1, len(ready) // 2)]
        assert T.ready_remove(np.asarray(take, np.uint64)) == len(take)
        g.take_from_ready(take)
        rel_w, _ = g.task_finished(take)
        rel, unk = T.graph_finish(take)
        assert unk == 0 and rel.tolist() == rel_w
        if wave == 1 and len(g.tasks) > 10:
            victim = sorted(g.tasks)[len(g.tasks) // 2]
            gone_w, _ = g.remove([victim], recursive=True)
            gone = T.graph_remove(np.asarray([victim], np.uint64), recursive=True)
            assert gone.tolist() == gone_w
        ready = sorted(g.ready)
    assert depth >= 1


def test_a_tick_ranks_by_the_b_levels_once_they_are_in_the_ready_set(T):
    """after hqtick_graph_blevel(UPDATE_READY) the resident ready set's priorities carry the b-levels: the tick on it equals the oracle's tick on a snapshot with those priorities"""
    from oracle.graph_oracle import GraphOracle
    from oracle.oracle import Oracle

    n, W = 4_000, 4
    ids, prio, rq, off, dep = workloads.make_dag(n, seed=6)
    rq = (rq % np.uint32(3)).astype(np.uint32)
    g = GraphOracle()
    g.on_new_tasks([(int(ids[i]), int(prio[i]), int(rq[i]), [int(x) for x in dep[off[i]:off[i + 1]]]) for i in range(n)])
    T.graph_add_tasks(ids, prio, rq, (off, dep))
    T.graph_blevel(update_ready=True)
    g.apply_blevels()
    drv = workloads.DagChurn(n_workers=W, churn=0.0, seed=1)
    rid = sorted(g.ready)
    assert len({g.ready[i][0] for i in rid}) > 1  # several b-levels among the sources: the tick has levels to rank
    want = Oracle(abi.make_config(time_limit_s=20.0), canonical=True).tick(drv.snapshot(rid, [g.ready[i][0] for i in rid], [g.ready[i][1] for i in rid]))
    got = T.tick(drv.snapshot(), resident=True)
    assert got.batches == want.batches and got.counts == want.counts and got.records == want.records


def test_blevel_between_a_tick_and_its_consume_leaves_the_consume_alone(T):
    """a host on the two-call protocol may call hqtick_graph_blevel(UPDATE_READY) between hqtick_run_resident and hqtick_ready_consume_last — the pending
    consume replays the tick's selection (group keys, per-slice counters, plan), which the priority rewrite does not touch: what the tick handed out leaves the set."""
    n, W = 4_000, 4
    ids, prio, rq, off, dep = workloads.make_dag(n, seed=9)
    rq = (rq % np.uint32(3)).astype(np.uint32)
    ready = T.graph_add_tasks(ids, prio, rq, (off, dep))
    drv = workloads.DagChurn(n_workers=W, churn=0.0, seed=1)
    before = T.ready_count()
    assert before == len(ready)
    snap = drv.snapshot()   # (kept alive: the C view points into its arrays)
    res = T.tick_raw(snap.to_c(), resident=True)
    handed = abi.record_task_ids(res, W).copy()
    assert len(handed)
    info = T.graph_blevel(update_ready=True)
    assert info["ready_updated"] == before
    T.ready_consume_last()   # used to fail with E_INVALID ('needs a preceding hqtick_run_resident'): the tasks stayed live and the next tick handed them out again
    assert T.ready_count() == before - len(handed)
    res2 = T.tick_raw(snap.to_c(), resident=True)
    again = set(abi.record_task_ids(res2, W).tolist()) & set(handed.tolist())
    assert not again
