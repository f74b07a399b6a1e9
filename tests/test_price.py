"""The coupled placement by price sweeps (csrc/price.cpp + csrc/price_core.h, the algorithm of k_price_sweep) with the wavefront emulated on the CPU:
run_scheduling_solver's model (scheduler/solver.rs:95-430) decomposed along its worker blocks, the wide rows (batch sizes :264-271, blocker flags
:233-253, priority cuts :274-429) priced out.  What is claimed is what the reference claims of HiGHS (solver/highs.rs:65-88): an incumbent proven
within mip_rel_gap = 1e-4 — checked here against the reference-configured HiGHS (oracle) and against the host-only search of csrc/milp.cpp.
Host stages only: runs without a GPU (the same ticks run through k_price_sweep in tests/test_gpu_price.py)."""
import ctypes as C

import numpy as np
import pytest

from host_stages import HostStages
from hyperqueue_amd import abi, workloads
from hyperqueue_amd.core import priority_from_user
from test_host_stages import _completed_objective, _objective


def stages(snap, emulate: bool, min_cols: int = 0, tl: float = 5.0):
    hs = HostStages(abi.make_config(time_limit_s=tl))
    hs.lib.hqtick_debug_set_price_emulation.argtypes = [C.c_int, C.c_uint32]
    hs.lib.hqtick_debug_last_price.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    hs.lib.hqtick_debug_set_price_emulation(1 if emulate else 0, min_cols)
    try:
        got = hs.stages(snap)
    finally:
        hs.lib.hqtick_debug_set_price_emulation(0, 0)
    sw, rd = C.c_uint32(), C.c_uint32()
    hs.lib.hqtick_debug_last_price(C.byref(sw), C.byref(rd))
    return got, sw.value, rd.value


def highs(snap, tl=30.0):
    from oracle.oracle import Oracle

    o = Oracle(abi.make_config(time_limit_s=tl), reference_solver_options=True)  # HiGHS as the reference configures it
    want = o.tick(snap)
    return want, o.last_model()


@pytest.fixture(scope="module")
def dag_sources():
    ids, prio, rq, off, dep = workloads.make_dag(1_000_000, seed=0)
    return ids, prio, rq, np.nonzero((off[1:] - off[:-1]) == 0)[0]


def unsaturated(dag_sources, W, fill, ncls=8):
    ids, prio, rq, src = dag_sources
    sel = src[: min(len(src), int(len(src) * W / 1024 * fill / 0.45))]
    drv = workloads.DagChurn(n_workers=W, churn=0.1, seed=0)
    return drv.snapshot(ids[sel], prio[sel], (rq[sel] % ncls).astype(np.uint32))


@pytest.mark.parametrize("W,n_tasks", [(64, 160_000), (128, 300_000)])
def test_c3p_reduced_by_price_sweeps(W, n_tasks):
    """three priority levels: blocker flags + cut rows (the general model)"""
    snap = workloads.make("c3p", n_tasks=n_tasks, n_workers=W)
    got, sweeps, rounds = stages(snap, True, min_cols=64)
    assert sweeps > 0 and rounds >= 1
    assert got.status == abi.HQTICK_DONE and got.is_optimal
    want, model = highs(snap)
    zg = _objective(model, got)
    if want.is_optimal:
        assert zg >= float(model["objective"]) * (1.0 - 1e-4) - 1e-12, (zg, model["objective"])
    host, s0, _ = stages(snap, False)
    assert s0 == 0
    assert zg >= _objective(model, host) * (1.0 - 1e-4)


def test_c3p_at_baseline_size_by_price_sweeps():
    """BASELINE.md's C3 with three priority levels at full size (1024 workers, 1 M tasks; 8 205 columns x 37 958 rows).  The host search certifies
    1.50132 in 1.6 s on the MI355X box's host (HiGHS: 1.484 after its 5 s, not optimal); the sweeps reach the trivial bound 1.5014648 to 1e-7."""
    snap = workloads.make("c3p", n_tasks=1_000_000, n_workers=1024)
    got, sweeps, rounds = stages(snap, True)
    assert sweeps > 0 and got.status == abi.HQTICK_DONE and got.is_optimal
    # upper bound that needs no solver: every worker packed completely, sum_w 3 (W - w) / W^2
    W = 1024
    bound = sum(3.0 * (W - w) / W / W for w in range(W))
    from oracle.oracle import Oracle

    o = Oracle(abi.make_config(time_limit_s=0.5), reference_solver_options=True)
    o.tick(snap)  # (for the model only: HiGHS does not finish it)
    zg = _objective(o.last_model(), got)
    assert bound * (1.0 - 1e-4) <= zg <= bound * (1.0 + 1e-9), (zg, bound)


def test_c4p_at_baseline_size_by_price_sweeps():
    """BASELINE configs[3] as BASELINE.md §3 / SURVEY §8(d) write it — "as C3" with 4096 workers and every class a 2-variant OR-list: three priority levels at
    80/15/5 % over 1 M tasks.  The largest model the contract names: 65 536 placement columns + 16 flags, 12 422 rows (cut / blocker rows over 4096 blocks,
    scheduler/solver.rs:233-253,274-429).  Certified against the bound that needs no solver (every worker packed completely); every row of the oracle's model holds."""
    from limits import model_point, rows_hold
    from oracle.oracle import Oracle

    W = 4096
    snap = workloads.make("c4p", n_tasks=1_000_000, n_workers=W)
    got, sweeps, rounds = stages(snap, True, tl=60.0)
    assert sweeps > 0 and got.status == abi.HQTICK_DONE and got.is_optimal
    o = Oracle(abi.make_config(time_limit_s=0.5), reference_solver_options=True)
    o.tick(snap)  # (for the model only: HiGHS does not finish it)
    model = o.last_model()
    assert len(model["obj"]) == 65552 and len(np.unique(snap.task_priority)) == 3
    x = model_point(model, got.counts)
    assert rows_hold(model, x)
    zg = _objective(model, got)
    bound = sum(3.0 * (W - w) / W / W for w in range(W))
    assert bound * (1.0 - 1e-4) <= zg <= bound * (1.0 + 1e-9), (zg, bound)


@pytest.mark.parametrize("W,fill", [(128, 0.45), (256, 0.20), (256, 0.45), (512, 0.20), (1024, 0.20), (1024, 0.45)])
def test_unsaturated_cluster_by_price_sweeps(dag_sources, W, fill):
    """fewer ready tasks than the cluster holds: batch-size rows across all workers (every DAG tick)"""
    snap = unsaturated(dag_sources, W, fill)
    got, sweeps, _ = stages(snap, True, min_cols=512)
    assert sweeps > 0 and got.status == abi.HQTICK_DONE and got.is_optimal
    if fill <= 0.2:
        assert sum(c for *_, c in got.counts) == len(snap.task_id)  # everything fits: every ready task is placed
    host, _, _ = stages(snap, False, tl=5.0)
    from oracle.oracle import Oracle

    o = Oracle(abi.make_config(time_limit_s=0.2), reference_solver_options=True)
    o.tick(snap)
    model = o.last_model()
    zg, zh = _objective(model, got), _objective(model, host)
    assert zg >= zh * (1.0 - 1e-4), (zg, zh)  # both claim 1e-4 of the optimum: neither can be more than that below the other
    if host.is_optimal:
        assert zh >= zg * (1.0 - 1e-4), (zg, zh)


def steady_coupled(W, seed, n_ready, levels):
    """a cluster mid-run (every worker its own free vector), a ready set that does not saturate it, several priority levels"""
    snap = workloads.make_steady("c3", seed=seed, n_workers=W, n_tasks=max(n_ready, 1))
    rng = np.random.default_rng(seed)
    snap.task_id, snap.task_rq = snap.task_id[:n_ready], snap.task_rq[:n_ready]
    snap.task_priority = np.asarray([priority_from_user(int(p)) for p in rng.integers(0, levels, n_ready)], np.uint64)
    return snap


@pytest.mark.parametrize("W,seed,n_ready,levels", [(16, 1, 120, 1), (24, 2, 200, 2), (32, 3, 150, 3), (48, 4, 400, 1), (64, 5, 300, 2), (64, 6, 900, 3), (96, 7, 500, 1), (128, 8, 700, 2)])
def test_heterogeneous_coupled_ticks_by_price_sweeps(W, seed, n_ready, levels):
    snap = steady_coupled(W, seed, n_ready, levels)
    got, sweeps, _ = stages(snap, True, min_cols=16)
    assert got.status in (abi.HQTICK_DONE, abi.HQTICK_NO_PROGRESS, abi.HQTICK_NEED_MORE_COMPUTE)
    host, _, _ = stages(snap, False)
    want, model = highs(snap)
    if got.is_optimal and host.is_optimal:  # (the WHOLE objective on both sides: two certificates are each within 1e-4 of the optimum of THAT, and the flag columns of blocked workers carry part of it)
        zg, zh = _completed_objective(model, got), _completed_objective(model, host)
        assert abs(zg - zh) <= 1e-4 * max(zg, zh) + 1e-12, (zg, zh, sweeps)
    if got.is_optimal and want.is_optimal:  # the whole objective on both sides: the flag columns of blocked workers carry part of it
        zg_all, zw_all = _completed_objective(model, got), float(model["objective"])
        assert zg_all >= zw_all * (1.0 - 1e-4) - 1e-12, (zg_all, zw_all, sweeps)


@pytest.mark.parametrize("variant", ["unsaturated", "priorities"])
def test_c4_shaped_coupled_ticks_by_price_sweeps(variant):
    """BASELINE configs[3]'s shape (every class a 2-variant OR-list: 16 columns per worker block), reduced to 512 workers.  Unsaturated: the one-component
    model HiGHS cannot close at full size; with priorities: flags + cut rows on top — the flag configuration comes from the solver's own
    greedy incumbent."""
    from oracle.oracle import Oracle

    snap = workloads.make("c4", n_tasks=20_000 if variant == "unsaturated" else 150_000, n_workers=512)
    if variant == "priorities":
        rng = np.random.default_rng(1)
        snap.task_priority = np.asarray([priority_from_user(int(p)) for p in rng.choice([0, 1, 2], len(snap.task_id), p=[0.8, 0.15, 0.05])], np.uint64)
    got, sweeps, rounds = stages(snap, True, tl=10.0)
    assert sweeps > 0 and got.status == abi.HQTICK_DONE and got.is_optimal
    host, _, _ = stages(snap, False, tl=5.0)
    o = Oracle(abi.make_config(time_limit_s=0.2), reference_solver_options=True)
    o.tick(snap)
    model = o.last_model()
    zg, zh = _objective(model, got), _objective(model, host)
    assert zg >= zh * (1.0 - 1e-4), (zg, zh)


# ------------------------------------------------------------------------------------------------ the coupled tick's fast path
def _host_stages_path(snap, fast, tl=20.0):
    import ctypes as C

    from host_stages import HostStages
    from hyperqueue_amd import _testhooks, abi

    lib = _testhooks.load()
    lib.hqtick_debug_set_price_emulation.argtypes = [C.c_int, C.c_uint32]
    lib.hqtick_debug_set_fast_path.argtypes = [C.c_int]
    lib.hqtick_debug_last_price.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.hqtick_debug_set_price_emulation(1, 0)
    lib.hqtick_debug_set_fast_path(1 if fast else 0)
    lib.hqtick_debug_check_model_hints.argtypes = [C.c_int]
    lib.hqtick_debug_check_model_hints(1)  # the builder's column bounds recomputed from the rows they stand for
    try:
        got = HostStages(abi.make_config(time_limit_s=tl)).stages(snap)
        assert lib.hqtick_debug_model_hint_mismatches() == 0
    finally:
        lib.hqtick_debug_set_price_emulation(0, 0)
        lib.hqtick_debug_set_fast_path(-1)
        lib.hqtick_debug_check_model_hints(0)
    sw, rd = C.c_uint32(), C.c_uint32()
    lib.hqtick_debug_last_price(C.byref(sw), C.byref(rd))
    return got, sw.value, rd.value


def _busy(name, **kw):
    from hyperqueue_amd import workloads

    return workloads.make_steady(name, **kw)


FAST_CASES = {
    "c3p-256": lambda: __import__("hyperqueue_amd.workloads", fromlist=["make"]).make("c3p", n_tasks=400_000, n_workers=256),
    "c3p-300-seed3": lambda: __import__("hyperqueue_amd.workloads", fromlist=["make"]).make("c3p", seed=3, n_tasks=500_000, n_workers=300),
    "c3p-busy-288": lambda: _busy("c3p", seed=1, n_workers=288, n_tasks=300_000),
    "c4-unsat-160": lambda: __import__("hyperqueue_amd.workloads", fromlist=["make"]).make("c4", seed=8, n_workers=160, n_tasks=2_300),
    "c3-unsat-320": lambda: __import__("hyperqueue_amd.workloads", fromlist=["make"]).make("c3", seed=5, n_workers=320, n_tasks=9_000),
}


@pytest.mark.parametrize("name", sorted(FAST_CASES))
def test_the_fast_path_walks_the_classic_paths_sweeps(name):
    """A large coupled model with the builder's structure hints (Model::col_group / row_lhs / row_block / col_ub) goes to the price sweeps straight from the model
    (csrc/milp.cpp solve(), hqprice::solve_model): no presolve, no components, no scaled row copy, shared left-hand sides read once.  The flattened blocks and wide
    rows must be the ones the classic path (component copy -> hqprice::solve) arrives at: same sweeps, same flag configurations, same status, same counts."""
    snap = FAST_CASES[name]()
    fast, sw_f, rd_f = _host_stages_path(snap, True)
    classic, sw_c, rd_c = _host_stages_path(snap, False)
    assert sw_f > 0 and sw_c > 0  # both went through the sweeps
    assert (fast.status, fast.is_optimal) == (classic.status, classic.is_optimal)
    assert fast.batches == classic.batches
    assert (sw_f, rd_f) == (sw_c, rd_c)
    assert fast.counts == classic.counts
