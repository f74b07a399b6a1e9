"""The per-worker-class block solver of the separable placement (csrc/block_core.h = the algorithm of k_block_solve) on a machine without a GPU:
the wavefront is emulated by a loop over its 64 lanes (libhqtick_test.so, hqtick_debug_block_solve_host / hqtick_debug_set_block_emulation), so
the CPU suite executes the code the GPU runs.  Checked against the exact host solver (csrc/milp.cpp, canonical optimum) block by block, and
against the canonical oracle (HiGHS) on whole ticks of the steady-state shapes.  The GPU tests of the kernel itself: tests/test_gpu_blocks.py."""
import ctypes as C
import dataclasses

import numpy as np
import pytest

from host_stages import HostStages
from hyperqueue_amd import _testhooks, abi, workloads

C3 = [[(0, 1)], [(0, 4)], [(0, 2), (1, 1)], [(0, 1), (1, 0.5)], [(0, 1), (1, 0.25)], [(0, 8), (2, 64)], [(0, 16), (1, 2), (2, 128)], [(0, 1), (2, 1)]]


def block_solve_host(cols, weight, pool, free, total, elig, budget=20000):
    """cols: per column a list of (resource, kind, amount); free/total [n_classes, R]; elig [n_classes] masks -> (x, status, steps)"""
    lib = _testhooks.load()
    off = np.zeros(len(cols) + 1, np.uint32); off[1:] = np.cumsum([len(c) for c in cols])
    res = np.ascontiguousarray([e[0] for c in cols for e in c], np.uint32); kind = np.ascontiguousarray([e[1] for c in cols for e in c], np.uint8)
    amt = np.ascontiguousarray([e[2] for c in cols for e in c], np.uint64)
    weight = np.ascontiguousarray(weight, np.uint32); pool = np.ascontiguousarray(pool, np.float64)
    free = np.ascontiguousarray(free, np.uint64); total = np.ascontiguousarray(total, np.uint64); elig = np.ascontiguousarray(elig, np.uint64)
    n_cls, R = free.shape
    x = np.zeros((n_cls, len(cols)), np.uint32); status = np.zeros(n_cls, np.uint32); steps = np.zeros(n_cls, np.uint32)
    dp = C.POINTER(C.c_double)
    lib.hqtick_debug_block_solve_host.argtypes = [C.c_uint32, C.c_uint32, abi.u32p, abi.u32p, abi.u8p, abi.u64p, abi.u32p, dp, C.c_uint32, abi.u64p, abi.u64p, abi.u64p,
                                                  C.c_uint32, abi.u32p, abi.u32p, abi.u32p]
    rc = lib.hqtick_debug_block_solve_host(len(cols), R, off.ctypes.data_as(abi.u32p), res.ctypes.data_as(abi.u32p), kind.ctypes.data_as(abi.u8p), amt.ctypes.data_as(abi.u64p),
                                           weight.ctypes.data_as(abi.u32p), pool.ctypes.data_as(dp), n_cls, free.ctypes.data_as(abi.u64p), total.ctypes.data_as(abi.u64p),
                                           elig.ctypes.data_as(abi.u64p), budget, x.ctypes.data_as(abi.u32p), status.ctypes.data_as(abi.u32p), steps.ctypes.data_as(abi.u32p))
    assert rc == 0
    return x, status, steps


def milp_block(cols, weight, pool, free, total, elig):
    """the same block through the exact host solver (hqtick_debug_milp_solve, canonical), built as host_model.cpp builds it"""
    from test_host_logic import product_milp

    R = len(free)
    obj, rows, colmap = [], [[] for _ in range(R)], {}
    for g, c in enumerate(cols):
        if not (int(elig) >> g) & 1:
            continue
        sc = 0.0
        for (r, k, a) in c:
            amt = int(total[r]) if k else int(a)
            sc += 0.0 if pool[r] < 0.000001 else (amt / 10000.0) / pool[r]
        j = len(obj); colmap[g] = j
        obj.append(sc * (weight[g] / 10000.0))
        for (r, k, a) in c:
            rows[r].append((j, (int(total[r]) if k else int(a)) / 10000.0))
    if not obj:
        return np.zeros(len(cols), np.uint32)
    rtype, rhs, roff, rcol, rcoef = [], [], [0], [], []
    for r in range(R):
        if rows[r]:
            rtype.append(1); rhs.append(int(free[r]) / 10000.0)
            for (j, a) in rows[r]:
                rcol.append(j); rcoef.append(a)
            roff.append(len(rcol))
    got = product_milp(obj, [0] * len(obj), rtype, rhs, roff, rcol, rcoef, canonical=True)
    assert got is not None and got[2]
    x = np.zeros(len(cols), np.uint32)
    for g, j in colmap.items():
        x[g] = int(round(got[0][j]))
    return x


def c3_block_case(rng):
    cols = [[(r, 0, int(round(a * 10000))) for r, a in c] for c in C3]
    weight = [10000 if rng.integers(4) else int(rng.integers(5000, 25000)) for _ in cols]
    total = np.asarray([1280000, 80000, 5120000], np.uint64)
    free = np.asarray([int(rng.integers(0, 129)) * 10000, int(rng.integers(0, 33)) * 2500, int(rng.integers(0, 513)) * 10000], np.uint64)
    pool = np.asarray([1024 * 128.0, 1024 * 8.0, 1024 * 512.0]) * rng.uniform(0.05, 0.95, 3)
    elig = 0xFF if rng.integers(8) else int(rng.integers(1, 256))
    return cols, weight, pool, free, total, elig


def random_block_case(rng):
    R = int(rng.integers(1, 5)); n = int(rng.integers(1, 11))
    total = (rng.integers(1, 65, R) * 10000).astype(np.uint64)
    free = np.asarray([t if rng.integers(5) == 0 else int(rng.integers(0, t // 100 + 1)) * 100 for t in total], np.uint64)
    pool = rng.integers(1, 50000, R) / 7.0
    grid = [10000, 20000, 40000, 5000, 2500, 80000, 30000, 15000, 70000]
    cols = []
    for _ in range(n):
        ent = [(r, 1 if rng.integers(16) == 0 else 0, grid[int(rng.integers(len(grid)))]) for r in range(R) if rng.integers(2)]
        if not ent:
            ent = [(int(rng.integers(R)), 0, grid[int(rng.integers(len(grid)))])]
        cols.append(ent)
    weight = [10000 if rng.integers(3) else int(rng.integers(1000, 31000)) for _ in cols]
    elig = (1 << n) - 1
    if rng.integers(4) == 0:
        elig &= int(rng.integers(0, 1 << n))
    return cols, weight, pool, free, total, elig


@pytest.mark.parametrize("seed", range(40))
def test_emulated_block_equals_exact_solver(seed):
    """single blocks: the emulated wavefront returns the canonical optimum of csrc/milp.cpp, column for column"""
    rng = np.random.default_rng(seed)
    for k in range(6):
        cols, weight, pool, free, total, elig = (c3_block_case if k % 2 == 0 else random_block_case)(rng)
        x, status, steps = block_solve_host(cols, weight, pool, free[None, :], total[None, :], [elig])
        assert status[0] in (0, 1)
        if status[0] == 1:
            continue  # step budget exhausted: the tick hands such a class to the host solver (covered by test_budget_exhaustion_falls_back)
        want = milp_block(cols, weight, pool, free, total, elig)
        assert x[0].tolist() == want.tolist(), (seed, k, steps[0])
        # exact feasibility in ResourceAmount arithmetic
        used = np.zeros(len(free), object)
        for g, c in enumerate(cols):
            for (r, kd, a) in c:
                used[r] += (int(total[r]) if kd else int(a)) * int(x[0][g])
        assert all(int(used[r]) <= int(free[r]) for r in range(len(free)))


def test_block_shapes_outside_the_kernel_are_refused():
    cols = [[(0, 0, 10000)]] * 40  # 40 eligible columns > 32 per block
    x, status, _ = block_solve_host(cols, [10000] * 40, [100.0], np.asarray([[1280000]], np.uint64), np.asarray([[1280000]], np.uint64), [(1 << 40) - 1])
    assert status[0] == 2 and not x.any()
    cols = [[(r, 0, 10000)] for r in range(5)]  # five resource rows > 4
    x, status, _ = block_solve_host(cols, [10000] * 5, [100.0] * 5, np.full((1, 5), 50000, np.uint64), np.full((1, 5), 50000, np.uint64), [31])
    assert status[0] == 2
    x, status, _ = block_solve_host([[(0, 0, 1)]], [10000], [100.0], np.asarray([[10 ** 9]], np.uint64), np.asarray([[10 ** 9]], np.uint64), [1])
    assert status[0] == 2  # a column that fits 10^9 times: beyond the kernel's value range
    x, status, _ = block_solve_host([[(0, 0, 10000)]], [10000], [100.0], np.asarray([[2 ** 64 - 1]], np.uint64), np.asarray([[50000]], np.uint64), [1])
    assert status[0] == 2  # HQ_AMOUNT_MAX free: the reference's carry-over row, a host matter
    x, status, _ = block_solve_host([[(0, 0, 10000)]], [10000], [100.0], np.asarray([[50000]], np.uint64), np.asarray([[50000]], np.uint64), [0])
    assert status[0] == 0 and not x.any()  # nothing eligible


def _emulated(cfg, snap, budget=4096):
    lib = _testhooks.load()
    lib.hqtick_debug_set_block_emulation.argtypes = [C.c_int, C.c_uint32]
    lib.hqtick_debug_set_block_emulation(1, budget)
    try:
        res = HostStages(cfg).stages(snap)
        a, b = C.c_uint32(), C.c_uint32()
        lib.hqtick_debug_last_blocks(C.byref(a), C.byref(b))
        return res, a.value, b.value
    finally:
        lib.hqtick_debug_set_block_emulation(0, 0)


@pytest.mark.parametrize("name,n_workers,seed", [("c3", 48, 0), ("c3", 64, 1), ("c4", 40, 2), ("c3", 96, 3)])
def test_steady_state_tick_emulated_blocks_vs_oracle(name, n_workers, seed):
    """SURVEY §8(d) steady state, reduced: heterogeneous free vectors (one class per worker), every class saturated -> the separable path with one
    block per worker; blocks through the emulated kernel == blocks through the host solver == the canonical oracle"""
    from oracle.oracle import Oracle

    snap = workloads.make_steady(name, seed=seed, n_tasks=60_000, n_workers=n_workers)
    cfg = abi.make_config(time_limit_s=30.0)
    got, n_emu, n_host = _emulated(cfg, snap)
    plain = HostStages(cfg).stages(snap)
    assert n_emu >= n_workers // 2 and n_host == 0, (n_emu, n_host)
    assert got.is_optimal and got.is_canonical and plain.is_optimal
    assert got.batches == plain.batches and got.counts == plain.counts
    want = Oracle(cfg, canonical=True).tick(snap)
    if want.is_optimal:
        assert got.batches == want.batches
        assert got.counts == want.counts


@pytest.mark.parametrize("name", ["c4", "c3s"])
def test_full_size_host_stages_vs_oracle(name):
    """BASELINE SIZE on the CPU: config 4 cold (4096 workers x 2-variant OR-lists, 1 M tasks: the lazy size rows make it one 16-column block) and
    config 3 in the steady state (1024 workers, 929 distinct classes) — the product's placement (host blocks and the emulated kernel) against the
    canonical oracle (HiGHS per independent component, counting rows checked afterwards: oracle.py::_solve_canonical)"""
    from oracle.oracle import Oracle

    snap = workloads.make("c4") if name == "c4" else workloads.make_steady("c3", seed=0)
    cfg = abi.make_config(time_limit_s=60.0)
    plain = HostStages(cfg).stages(snap)
    emu, n_emu, n_host = _emulated(cfg, snap)
    want = Oracle(cfg, canonical=True).tick(snap)
    assert want.is_optimal and plain.is_optimal and plain.is_canonical and emu.is_canonical
    assert n_emu >= (1 if name == "c4" else 900) and n_host == 0
    assert plain.batches == want.batches and plain.counts == want.counts
    assert emu.counts == want.counts


def test_budget_exhaustion_falls_back_to_host_solver():
    """with a step budget of 1 every searched class is handed back (status 1) and the host solver gives the same counts"""
    snap = workloads.make_steady("c3", seed=5, n_tasks=40_000, n_workers=24)
    cfg = abi.make_config(time_limit_s=30.0)
    got, n_emu, n_host = _emulated(cfg, snap, budget=1)
    plain = HostStages(cfg).stages(snap)
    assert n_host > 0
    assert got.counts == plain.counts and got.batches == plain.batches


@pytest.mark.parametrize("seed", range(60))
def test_fuzz_scenarios_emulated_blocks_equal_host_blocks(seed):
    """the randomised scenario family of the GPU fuzz suite (blocked requests, min_utilization, time limits, `All` entries, weights, MAX amounts):
    wherever the tick is separable, emulated blocks and host blocks give the same counts; elsewhere the switch changes nothing"""
    import test_gpu_fuzz as f

    cfg, envs, _rng = f.build(seed)
    snap = envs[1].snapshot()
    got, _, _ = _emulated(cfg, snap)
    plain = HostStages(cfg).stages(snap)
    assert got.status == plain.status and got.batches == plain.batches
    if plain.is_canonical and got.is_canonical:
        assert got.counts == plain.counts


# ------------------------------------------------------------------------------------------------ the guard on the device's answers
def _guarded(cfg, snap, verify, tick_seq=0, mode=0, cls=0, fill=0):
    """host stages with the emulated blocks under fault injection -> (result, classes taken from the device, solved by the host, (verified, mismatch, rejected))"""
    lib = _testhooks.load()
    lib.hqtick_debug_set_block_guard.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32]
    lib.hqtick_debug_last_block_guard.argtypes = [C.POINTER(C.c_uint32)] * 3
    lib.hqtick_debug_set_block_guard(verify, tick_seq, mode, cls, fill)
    try:
        res, n_emu, n_host = _emulated(cfg, snap)
    finally:
        lib.hqtick_debug_set_block_guard(2, 0, 0, 0, 0)
    v, m, r = C.c_uint32(), C.c_uint32(), C.c_uint32()
    lib.hqtick_debug_last_block_guard(C.byref(v), C.byref(m), C.byref(r))
    return res, n_emu, n_host, (v.value, m.value, r.value)


def test_device_block_answers_are_not_taken_on_trust():
    """VERDICT r03 next 3 / ADVICE r02 #5: a k_block_solve answer used to be accepted once it FITS the rows.  Now every answer must also be MAXIMAL (all costs are
    positive: room for one more task = not an optimum), and a sample of the launch is re-solved by the host's exact solver while the kernel runs and compared column
    by column — one difference and the whole launch is re-solved on the host.  Faults are injected behind the emulated wavefront (what a wrong kernel would hand
    back); the tick's counts must come out as if nothing had happened, and the guard's counters must say who caught what."""
    snap = workloads.make_steady("c3", seed=1, n_tasks=60_000, n_workers=64)
    cfg = abi.make_config(time_limit_s=30.0)
    clean, n_emu, n_host, (ver, mis, rej) = _guarded(cfg, snap, verify=2)
    assert clean.is_optimal and clean.is_canonical and n_host == 0 and (ver, mis, rej) == (2, 0, 0) and n_emu >= 32
    n_cls = n_emu
    # 1. one task short on one class: feasible, so the old check let it through; not maximal -> thrown out, that class re-solved by the host
    for cls in (0, 5, n_cls - 1):
        got, n_emu, n_host, (ver, mis, rej) = _guarded(cfg, snap, verify=0, mode=1, cls=cls)
        assert (mis, rej) == (0, 1) and n_host == 1 and n_emu == n_cls - 1
        assert got.counts == clean.counts and got.is_canonical
    # 2. an answer that does not fit the rows: as before
    got, n_emu, n_host, (ver, mis, rej) = _guarded(cfg, snap, verify=0, mode=2, cls=3)
    assert rej == 1 and n_host == 1 and got.counts == clean.counts
    # 3. feasible AND maximal but not optimal — everything on one column, exactly filling a row: no O(columns) check can see it.  The sample does, when its
    #    window covers the class: the window moves with the tick counter, so over n_cls / verify ticks every class of a steady cluster is looked at once.
    #    (fill: how often the class's first used column fits — found by running the fault with verify = all and growing counts until the reject check lets it pass)
    caught_at = None
    for fill in range(1, 129):  # column 0 = the 1-cpu request: as many of them as the class has free cpus leave no room for anything (every c3 request needs a cpu)
        got, n_emu, n_host, (ver, mis, rej) = _guarded(cfg, snap, verify=0, mode=3, cls=7, fill=(0 << 16) | fill)
        if rej == 0:  # this count fills the cpu row exactly: the per-class checks accept it ...
            assert got.counts != clean.counts  # ... and WITHOUT the sample the wrong answer is placed
            caught_at = (0 << 16) | fill
            break
    assert caught_at is not None
    seen = []
    for tick_seq in range((n_cls + 1) // 2):
        got, n_emu, n_host, (ver, mis, rej) = _guarded(cfg, snap, verify=2, tick_seq=tick_seq, mode=3, cls=7, fill=caught_at)
        seen.append(mis)
        if mis:
            assert n_emu == 0 and got.counts == clean.counts and got.is_canonical  # the launch was distrusted as a whole and re-solved
        else:
            assert got.counts != clean.counts
    assert sum(seen) == 1  # exactly the tick whose window held class 7
    got, n_emu, n_host, (ver, mis, rej) = _guarded(cfg, snap, verify=0xFFFFFFFF, mode=3, cls=7, fill=caught_at)
    assert mis == 1 and got.counts == clean.counts


@pytest.mark.parametrize("seed", range(12))
def test_block_memo_answers_are_the_solvers(seed):
    """The table of earlier class-block answers (HQTICK_FLAG_NO_BLOCK_MEMO switches it off in a context): the second pass over a snapshot answers every host
    block from it and returns what the first returned; a snapshot whose free amounts moved builds other blocks, misses, and gets the solver's answer — the
    same as a run without the table."""
    lib = _testhooks.load()
    lib.hqtick_debug_set_block_memo.argtypes = [C.c_int]
    lib.hqtick_debug_last_block_memo.restype = C.c_uint32
    snap = workloads.make_steady("c3" if seed % 2 == 0 else "c4", seed=seed, n_tasks=30_000, n_workers=12 + seed)
    cfg = abi.make_config(time_limit_s=30.0)
    plain = HostStages(cfg).stages(snap)
    a, b = C.c_uint32(), C.c_uint32()
    lib.hqtick_debug_last_blocks(C.byref(a), C.byref(b))
    n_host = b.value
    assert n_host > 0 and lib.hqtick_debug_last_block_memo() == 0
    lib.hqtick_debug_set_block_memo(1)
    try:
        first = HostStages(cfg).stages(snap)
        n_first = lib.hqtick_debug_last_block_memo()  # (equal classes cannot occur inside one tick — a class IS its model — so the first pass finds nothing)
        second = HostStages(cfg).stages(snap)
        lib.hqtick_debug_last_blocks(C.byref(a), C.byref(b))
        assert n_first == 0 and lib.hqtick_debug_last_block_memo() == b.value == n_host
        assert first.counts == plain.counts and second.counts == plain.counts and second.batches == plain.batches
        assert second.is_optimal == plain.is_optimal and second.is_canonical == plain.is_canonical
        moved = dataclasses.replace(snap, _keep=[], worker_free=snap.worker_free.copy())
        moved.worker_free[: len(moved.worker_free) // 2, 0] = np.maximum(moved.worker_free[: len(moved.worker_free) // 2, 0], 10_000) - 10_000 // 2
        lib.hqtick_debug_set_block_memo(0)
        want = HostStages(cfg).stages(moved)
        lib.hqtick_debug_set_block_memo(1)
        HostStages(cfg).stages(snap)  # (the table holds the unmoved blocks)
        got = HostStages(cfg).stages(moved)
        hits = lib.hqtick_debug_last_block_memo()
        lib.hqtick_debug_last_blocks(C.byref(a), C.byref(b))
        assert hits < b.value  # the moved workers' blocks are new
        assert got.counts == want.counts and got.batches == want.batches
    finally:
        lib.hqtick_debug_set_block_memo(0)
