"""The coupled placement solver on half-full clusters: where the reference-configured HiGHS proves optimality, the product
must come back `is_optimal = 1` too — with the reference's meaning of the word.  `solve_bounded` (solver/highs.rs:65-88) sets `time_limit` and
nothing else, so HiGHS stops at its default mip_rel_gap = 1e-4; the product certifies the same gap (root LP bound + window search, then the tree)
and spends a bounded extra effort on the exact optimum its canonical answer needs.  Host stages only: runs without a GPU."""
import time

import numpy as np
import pytest

from host_stages import HostStages
from hyperqueue_amd import abi, workloads
from test_host_stages import _objective


@pytest.fixture(scope="module")
def dag_sources():
    ids, prio, rq, off, dep = workloads.make_dag(1_000_000, seed=0)
    src = np.nonzero((off[1:] - off[:-1]) == 0)[0]
    return ids, prio, rq, src


def _unsaturated(dag_sources, W, fill, ncls=8):
    """tools/unsat_probe.py's instance: the first k source tasks of the BASELINE DAG on W idle c3 workers, k ~ fill/0.45 of what the cluster holds"""
    ids, prio, rq, src = dag_sources
    k = min(len(src), int(len(src) * W / 1024 * fill / 0.45))
    sel = src[:k]
    drv = workloads.DagChurn(n_workers=W, churn=0.1, seed=0)
    return drv.snapshot(ids[sel], prio[sel], (rq[sel] % ncls).astype(np.uint32))


# (workers, fill, seconds the reference-configured HiGHS 1.8 needs on the build container): every one of these was `is_optimal = 0` after 5 s in round 1
CASES = [(5, 0.45, 0.06), (32, 0.20, 1.2), (32, 0.45, 0.10), (64, 0.20, 2.1), (128, 0.20, 0.63), (128, 0.45, 0.39)]
# the whole BASELINE cluster, unsaturated: an 8 192 x 3 080 model on which HiGHS and the round-1 solver both hit the 5 s limit.  The bound comes from the
# Lagrangian over the batch-size rows (milp.cpp::lagrangian_bound), the incumbent from the window search: certified in a fraction of a second
LARGE = [(256, 0.20), (512, 0.20), (1024, 0.20), (256, 0.45), (512, 0.45), (1024, 0.45)]


@pytest.mark.parametrize("W,fill", LARGE)
def test_unsaturated_tick_of_a_large_cluster_is_certified(dag_sources, W, fill):
    snap = _unsaturated(dag_sources, W, fill)
    hs = HostStages(abi.make_config(time_limit_s=5.0))
    t0 = time.perf_counter()
    got = hs.stages(snap)
    took = time.perf_counter() - t0
    assert got.status == abi.HQTICK_DONE and got.is_optimal, (W, fill, took)
    assert took < 4.0, took
    n_ready = len(snap.task_id)
    if fill <= 0.2:  # everything fits: every ready task is placed
        assert sum(c for *_, c in got.counts) == n_ready



@pytest.mark.parametrize("W,fill,highs_s", CASES)
def test_unsaturated_tick_is_certified_like_the_reference(dag_sources, W, fill, highs_s):
    from oracle.oracle import Oracle

    snap = _unsaturated(dag_sources, W, fill)
    hs = HostStages(abi.make_config(time_limit_s=5.0))
    t0 = time.perf_counter()
    got = hs.stages(snap)
    took = time.perf_counter() - t0
    assert got.status == abi.HQTICK_DONE and got.is_optimal, (W, fill, took)
    assert took < 4.0, f"certificate took {took:.2f} s (HiGHS: {highs_s} s)"  # loose (the suite runs 8 tests at a time): the measured figures are in DESIGN.md §4
    o = Oracle(abi.make_config(time_limit_s=20.0), reference_solver_options=True)  # HiGHS as the reference configures it
    want = o.tick(snap)
    if not want.is_optimal:
        pytest.skip("HiGHS hit its limit here")
    model = o.last_model()
    zg, zw = _objective(model, got), float(model["objective"])
    assert abs(zg - zw) <= 1.0e-4 * zw, (zg, zw)  # both are within 1e-4 of the true optimum from below


def test_c3p_reduced_is_certified():
    """three priority levels (cuts + blockers: the general model, not the compact one): 64 workers"""
    from oracle.oracle import Oracle

    snap = workloads.make("c3p", n_tasks=160_000, n_workers=64)
    got = HostStages(abi.make_config(time_limit_s=5.0)).stages(snap)
    assert got.is_optimal
    o = Oracle(abi.make_config(time_limit_s=20.0), reference_solver_options=True)
    want = o.tick(snap)
    assert want.is_optimal
    model = o.last_model()
    zg, zw = _objective(model, got), float(model["objective"])
    assert abs(zg - zw) <= 1.0e-4 * zw, (zg, zw)


@pytest.mark.slow
def test_c3p_at_baseline_size_is_certified():
    """BASELINE.md's C3 with three priority levels at full size (1024 workers, 1 M tasks; 8 205 columns x 37 958 rows): HiGHS holds 1.367 after 5 s and
    1.5014 after 60 s without a proof; the product certifies its incumbent against the root LP bound — in 1.6 s on the MI355X box's host and 3.5 s on an
    idle build container, i.e. inside the reference's 5 s limit.  The test gives it 45 s so that a loaded CI machine (this suite runs 8
    tests at a time) cannot turn a timing figure into a failure: what is asserted is the certificate."""
    snap = workloads.make("c3p", n_tasks=1_000_000, n_workers=1024)
    got = HostStages(abi.make_config(time_limit_s=45.0)).stages(snap)
    assert got.status == abi.HQTICK_DONE and got.is_optimal


SMALL = [(1024, 30), (1024, 60), (1024, 89), (1024, 150), (64, 40), (16, 25)]


@pytest.mark.parametrize("W,n_ready", SMALL)
def test_certificate_only_stops_where_the_reference_stops(W, n_ready):
    """HQTICK_FLAG_CERTIFICATE_ONLY (include/hqtick.h): the small coupled ticks of the DAG loop — a few dozen ready tasks on a mostly idle cluster — end at the 1e-4
    certificate as HiGHS does under solve_bounded (solver/highs.rs:65-88): is_optimal = 1, is_canonical = 0, the same batches, an objective within 1e-4 of the one the
    default (exact + canonical) solve reaches and of the oracle's, and a placement that fits every worker."""
    from oracle.oracle import Oracle

    ids, prio, rq, off, dep = workloads.make_dag(200_000, seed=0)
    drv = workloads.DagChurn(n_workers=W, churn=0.1, seed=0)
    snap = drv.snapshot(ids[:n_ready], prio[:n_ready], (rq[:n_ready] % 8).astype(np.uint32))
    exact = HostStages(abi.make_config(time_limit_s=20.0)).stages(snap)
    quick = HostStages(abi.make_config(time_limit_s=20.0, flags=abi.HQTICK_FLAG_CERTIFICATE_ONLY)).stages(snap)
    assert exact.status == quick.status == abi.HQTICK_DONE and exact.is_optimal and quick.is_optimal
    assert quick.batches == exact.batches
    o = Oracle(abi.make_config(time_limit_s=20.0))
    want = o.tick(snap)
    model = o.last_model()
    ze, zq, zw = _objective(model, exact), _objective(model, quick), float(model["objective"])
    assert abs(ze - zw) <= 1.0e-4 * zw and abs(zq - zw) <= 1.0e-4 * zw, (ze, zq, zw)  # (the oracle's HiGHS stops at 1e-4 too)
    assert zq <= ze * (1.0 + 1e-9) and ze - zq <= 1.0e-4 * ze, (zq, ze)
    # the certified point is a placement: every worker's resources hold, no batch hands out more than it has
    R = snap.n_resources
    free = np.asarray(snap.worker_free, np.int64).reshape(-1, R).copy()
    total = np.asarray(snap.worker_total, np.int64).reshape(-1, R)
    placed = {}
    for (q, v, w, c) in quick.counts:
        placed[q] = placed.get(q, 0) + c
        for (r, kind, a) in snap.requests[q][v]["entries"]:
            free[w, r] -= c * (int(total[w, r]) if kind == abi.HQ_ENTRY_ALL else int(a))
    assert (free >= 0).all()
    ready_per_rq = np.bincount(np.asarray(snap.task_rq), minlength=len(snap.requests))
    for q, c in placed.items():
        assert c <= ready_per_rq[q], (q, c)


def test_a_certificate_survives_on_the_safe_dual_bound():
    """csrc/milp.cpp: root_cuts_pass confirms the cut rounds' bound by a cold solve of the final rows; when that tableau fails its consistency test the bound is taken from
    its multipliers instead (safe_dual_bound: the Lagrangian over the boxes, evaluated from the rows — valid whatever the tableau's state).  price_fuzz seed 2317 is the tick
    that lost its certificate there; with the pivot tolerance of 1e-7 its cold solve is consistent again, so the fallback is forced: same status, same
    objective, and the trace names the bound it used."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import price_fuzz\nfrom test_price import stages\nfrom test_host_stages import _objective\nfrom hyperqueue_amd import abi\nfrom oracle.oracle import Oracle\n"
        "snap = price_fuzz.scenario(2317)[0]\ngot, sweeps, rounds = stages(snap, True, tl=20.0)\n"
        "o = Oracle(abi.make_config(time_limit_s=0.05), reference_solver_options=True)\n"
        "try:\n    o.tick(snap)\nexcept Exception:\n    pass\n"
        "print('RESULT', int(got.status), int(got.is_optimal), repr(_objective(o.last_model(), got)))\n"
    ) % (root, os.path.join(root, "tests"), os.path.join(root, "tools"))
    outs = []
    for forced in (False, True):
        env = dict(os.environ, HQMILP_TRACE="1")
        env.pop("HQMILP_FORCE_SAFE_BOUND", None)
        if forced:
            env["HQMILP_FORCE_SAFE_BOUND"] = "1"
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-2000:]
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT")][-1].split()
        outs.append((int(line[1]), int(line[2]), float(line[3]), "safe dual bound" in p.stderr))
    (st0, opt0, z0, named0), (st1, opt1, z1, named1) = outs
    assert (st0, opt0) == (0, 1) and (st1, opt1) == (0, 1)
    assert not named0 and named1   # the fallback is taken only where the cold solve does not confirm (here: because it was told so)
    assert abs(z0 - z1) <= 1e-9 * abs(z0)
