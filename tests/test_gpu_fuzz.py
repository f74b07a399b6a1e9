"""Wider randomised parity campaign: multi-node requests, worker groups, time limits, min_utilization, blocked requests, `All`
entries, weights, prior assignments — features the basic random_env of test_gpu_parity.py does not draw.  HIP path vs canonical oracle,
bit-exact, several ticks per scenario."""
import numpy as np
import pytest

from hyperqueue_amd import abi
from hyperqueue_amd.core import SchedEnv, TaskBuilder as TB, WorkerBuilder as WB

pytestmark = pytest.mark.gpu


def assert_same(got, want):
    assert got.status == want.status and got.batches == want.batches and got.counts == want.counts
    assert got.records == want.records and got.retracts == want.retracts and sorted(got.redirects) == sorted(want.redirects)
    assert got.mn == want.mn and (got.new_free == want.new_free).all()


def build(seed: int):
    rng = np.random.default_rng(seed)
    cfg = abi.make_config(reserve=int(rng.integers(0, 4)), fill_max=int(rng.integers(1, 6)), time_limit_s=20.0)
    envs = [SchedEnv(cfg), SchedEnv(cfg)]
    names = ["gpus", "mem"][: int(rng.integers(0, 3))]
    shapes = []
    for _ in range(int(rng.integers(2, 5))):
        kind = rng.random()
        if kind < 0.12:
            b = TB().n_nodes(int(rng.integers(1, 4)))
        elif kind < 0.2:
            b = TB().cpus_all()
        else:
            b = TB().cpus(int(rng.integers(1, 5)))
            for ri in range(len(names)):
                if rng.random() < 0.4:
                    b = b.add_resource(ri + 1, [0.5, 1, 2][int(rng.integers(0, 3))])
            if rng.random() < 0.25:
                b = b.time_request(int(rng.integers(10, 200)))
            if rng.random() < 0.2:
                b = b.weight([0.5, 1.5, 2.0][int(rng.integers(0, 3))])
            if rng.random() < 0.2:
                b = b.next_variant().cpus(int(rng.integers(1, 7)))
        shapes.append(b)
    workers = []
    for _ in range(int(rng.integers(2, 7))):
        wb = WB(int(rng.integers(2, 13)))
        for n in names:
            if rng.random() < 0.6:
                wb = wb.res_sum(n, int(rng.integers(1, 5)))
        if rng.random() < 0.25:
            wb = wb.time_limit_s(int(rng.integers(20, 300)))
        if rng.random() < 0.2:
            wb = wb.group(["g1", "g2"][int(rng.integers(0, 2))])
        if rng.random() < 0.15:
            wb = wb.min_utilization([0.3, 0.5, 0.9][int(rng.integers(0, 3))])
        workers.append(wb)
    tasks = [(int(rng.integers(0, len(shapes))), int(rng.integers(-1, 2)) if rng.random() < 0.5 else 0) for _ in range(int(rng.integers(4, 60)))]
    blocks = [(int(rng.integers(0, len(workers))), int(rng.integers(0, len(shapes)))) for _ in range(int(rng.integers(0, 3)))]
    for e in envs:
        for n in names:
            e.new_named_resource(n)
        wids = [e.new_worker(wb) for wb in workers]
        for (si, pr) in tasks:
            e.new_task(shapes[si].user_priority(pr))
        for (wi, si) in blocks:
            rq = e.rq_id(shapes[si].user_priority(0))
            e.block_request(wids[wi], rq, 0)
    return cfg, envs, rng


@pytest.mark.parametrize("seed", range(120))
def test_fuzz_scenario(seed):
    from hyperqueue_amd.tick import Tick
    from oracle.oracle import Oracle

    cfg, envs, rng = build(seed)
    g, o = Tick(cfg), Oracle(cfg, canonical=True)
    for round_ in range(3):
        rg, ro = envs[0].schedule(g), envs[1].schedule(o)
        assert_same(rg, ro)
        k = int(rng.integers(0, 4))
        for e in envs:
            done = 0
            for t in sorted(e.tasks.values(), key=lambda t: t.id):
                if done >= k:
                    break
                if t.state == 1:
                    e.finish_task(t.id, t.worker); done += 1
                elif t.state == 5 and t.mn_workers:  # RUNNING_MN
                    e.finish_task(t.id, t.mn_workers[0]); done += 1


@pytest.mark.parametrize("seed", range(40))
def test_fuzz_query(seed):
    """compute_new_worker_query on random cores: fake workers (partial or not, time limits, min_utilization) through hqtick_query vs
    the canonical oracle."""
    from hyperqueue_amd.core import WorkerTypeQuery as WQ
    from hyperqueue_amd.tick import Tick
    from oracle.oracle import Oracle

    cfg, envs, rng = build(5000 + seed)
    if any(v[0]["n_nodes"] for v in envs[0].requests):
        pytest.skip("multi-node queue with fake workers: the reference panics (solver.rs:104-106)")
    g, o = Tick(cfg), Oracle(cfg, canonical=True)
    envs[0].schedule(g); envs[1].schedule(o)
    names = [n for n in envs[0].resource_names if n != "cpus"]
    queries = []
    for _ in range(int(rng.integers(1, 4))):
        res = [("cpus", int(rng.integers(1, 9)))] if rng.random() < 0.8 else []
        for n in names:
            if rng.random() < 0.5:
                res.append((n, int(rng.integers(1, 4))))
        queries.append(WQ(resources=res, partial=bool(rng.random() < 0.4), time_limit_s=(None if rng.random() < 0.6 else float(rng.integers(20, 300))),
                          max_sn_workers=int(rng.integers(1, 5)), max_workers_per_allocation=int(rng.integers(1, 4)),
                          min_utilization=float([0.0, 0.0, 0.5, 1.0][int(rng.integers(0, 4))])))
    assert envs[0].new_worker_query(g, queries) == envs[1].new_worker_query(o, queries)
