"""The placement solve itself over the ranks of a sharded scheduler (include/hqtick.h: hqtick_set_exchange): every rank
sweeps only its worker range of a coupled tick (price.h: ShardedSweeper) / solves only every world-th class block of a separable tick (ShardedBlocks) and one small
all-gather per sweep / per launch completes the answer.  Claim: the sharded solve walks EXACTLY the unsharded solve's path — same sweeps, same configurations, same
status, same counts, bit for bit, on every rank.

CPU: the host stages with the emulated wavefront (hqtick_debug_host_stages), one THREAD per rank in this process (the hook's state is thread-local; the exchange is
a barrier + shared buffer) and, below, two PROCESSES over gloo.  On the MI355X the same wrappers run around k_price_sweep / k_block_solve: tests/test_gpu_multi.py."""
import ctypes as C
import dataclasses
import threading

import numpy as np
import pytest

from host_stages import HostStages
from hyperqueue_amd import abi, workloads

XFN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)


def _lib():
    from hyperqueue_amd import _testhooks

    lib = _testhooks.load()
    lib.hqtick_debug_set_exchange.argtypes = [XFN, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    lib.hqtick_debug_set_price_emulation.argtypes = [C.c_int, C.c_uint32]
    lib.hqtick_debug_set_block_emulation.argtypes = [C.c_int, C.c_uint32]
    lib.hqtick_debug_last_price.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.hqtick_debug_last_exchange_calls.restype = C.c_uint32
    return lib


class ThreadExchange:
    """all-gather between `world` threads: every rank copies its part into a shared buffer, a barrier, every rank copies the whole out, a barrier"""

    def __init__(self, world):
        self.world, self.bar, self.buf, self.calls = world, threading.Barrier(world), None, 0

    def fn(self, rank):
        def f(_user, send, recv, n):
            if rank == 0:
                self.buf = (C.c_ubyte * (n * self.world))()
                self.calls += 1
            self.bar.wait()
            C.memmove(C.addressof(self.buf) + rank * n, send, n)
            self.bar.wait()
            C.memmove(recv, self.buf, n * self.world)
            self.bar.wait()
            return 0

        return XFN(f)


def _stages(snap, tl, price_min_cols, blocks, rank=None, world=1, xfn=None, min_blocks=1, min_classes=1):
    lib = _lib()
    hs = HostStages(abi.make_config(time_limit_s=tl))
    lib.hqtick_debug_set_price_emulation(1, price_min_cols)
    lib.hqtick_debug_set_block_emulation(1 if blocks else 0, 0)
    if xfn is not None:
        lib.hqtick_debug_set_exchange(xfn, None, rank, world, min_blocks, min_classes)
    try:
        got = hs.stages(snap)
    finally:
        lib.hqtick_debug_set_price_emulation(0, 0); lib.hqtick_debug_set_block_emulation(0, 0)
        lib.hqtick_debug_set_exchange(XFN(0), None, 0, 1, 1, 1)
    sw, rd = C.c_uint32(), C.c_uint32()
    lib.hqtick_debug_last_price(C.byref(sw), C.byref(rd))
    return got, sw.value, rd.value, int(lib.hqtick_debug_last_exchange_calls()) if xfn is not None else 0


def run_ranks(snap, world, tl=20.0, price_min_cols=16, blocks=True, min_blocks=1, min_classes=1):
    ex = ThreadExchange(world)
    out, errs = [None] * world, []

    def main(r):
        try:
            mine = dataclasses.replace(snap, _keep=[])  # (Snapshot.to_c keeps its arrays alive in the snapshot object: one per thread)
            out[r] = _stages(mine, tl, price_min_cols, blocks, r, world, ex.fn(r), min_blocks, min_classes)
        except BaseException as e:  # noqa: BLE001 — a rank that dies must not leave the others in the barrier
            errs.append((r, e)); ex.bar.abort()

    ts = [threading.Thread(target=main, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(600)
    assert not errs, errs
    return out


def _same(a, b):
    assert a.status == b.status and a.is_optimal == b.is_optimal and a.is_canonical == b.is_canonical
    assert a.batches == b.batches and a.counts == b.counts


def _fuzz(seed):
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from price_fuzz import scenario

    return scenario(seed)[0]


COUPLED = {
    "c3p-64": lambda: workloads.make("c3p", n_tasks=160_000, n_workers=64),
    "c3p-100": lambda: workloads.make("c3p", n_tasks=250_000, n_workers=100),   # 100 blocks over 16 parts of 7: the last parts are short / empty
    "fuzz-2001": lambda: _fuzz(2001),   # tools/price_fuzz.py's family (clusters mid-run, every worker its own block): 12 workers, three flag configurations, branch-and-price
    "fuzz-2003": lambda: _fuzz(2003),   # 128 workers
    "fuzz-2005": lambda: _fuzz(2005),
    "c4-unsat-96": lambda: workloads.make("c4", seed=8, n_workers=96, n_tasks=1_400),
    "c4p-96": lambda: workloads.make("c4p", n_tasks=24_000, n_workers=96),   # configs[3] as written, reduced: sixteen-column blocks + priority cuts and flags
}


@pytest.mark.parametrize("name", sorted(COUPLED))
@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_sweeps_walk_the_unsharded_path(name, world):
    snap = COUPLED[name]()
    plain, sweeps, rounds, _ = _stages(snap, 20.0, 16, False)
    assert sweeps > 0  # a coupled tick: the sweeps did run
    res = run_ranks(snap, world, blocks=False)
    for (got, sw, rd, calls) in res:
        assert (sw, rd) == (sweeps, rounds)
        _same(got, plain)
        assert calls >= sweeps  # one exchange per sweep (+ the pattern fetches)


def test_more_ranks_than_parts():
    """17+ ranks: the master has 16 worker ranges, the ranks beyond them own no block and still take part in every exchange"""
    snap = COUPLED["c3p-64"]()
    plain, sweeps, rounds, _ = _stages(snap, 20.0, 16, False)
    for (got, sw, rd, _calls) in run_ranks(snap, 20, blocks=False):
        assert (sw, rd) == (sweeps, rounds)
        _same(got, plain)


@pytest.mark.parametrize("name,n_workers,seed", [("c3", 48, 0), ("c4", 40, 2), ("c3", 96, 3)])
@pytest.mark.parametrize("world", [2, 5])
def test_sharded_class_blocks_equal_the_unsharded_launch(name, n_workers, seed, world):
    """a separable steady-state tick (about one worker class per worker): every rank solves every world-th class block, one exchange completes the launch"""
    snap = workloads.make_steady(name, seed=seed, n_workers=n_workers, n_tasks=60_000)
    plain, sweeps, _, _ = _stages(snap, 20.0, 1 << 30, True)
    assert sweeps == 0
    for (got, sw, _rd, calls) in run_ranks(snap, world, price_min_cols=1 << 30, blocks=True):
        _same(got, plain)
        assert sw == 0 and calls == 1


def test_below_the_thresholds_every_rank_sweeps_the_whole_model_and_only_the_clock_is_exchanged():
    """a small model: every rank sweeps all of it; what still crosses is ONE word per rank and sweep — the clock reading (and a 'failed' bit), so that replicas near
    the time guard leave the sweeps at the same sweep"""
    snap = COUPLED["c3p-64"]()
    plain, sweeps, rounds, _ = _stages(snap, 20.0, 16, False)
    for (got, sw, rd, calls) in run_ranks(snap, 2, blocks=False, min_blocks=1025):
        assert (sw, rd, calls) == (sweeps, rounds, sweeps)
        _same(got, plain)


def test_the_clock_is_read_collectively():
    """a tick whose limit runs out inside the sweeps: the ranks OR their readings of the clock inside the sweep's exchange, so every replica leaves at the same
    sweep (a rank leaving alone would leave the others in the collective: this test would hang into its join timeout)"""
    snap = workloads.make("c3p", n_tasks=400_000, n_workers=256)
    res = run_ranks(snap, 2, tl=0.02, blocks=False)
    (a, sa, ra, _), (b, sb, rb, _) = res
    assert (sa, ra) == (sb, rb)
    assert a.status == b.status and a.batches == b.batches


# ------------------------------------------------------------------------------------------------ two processes over gloo
def _gloo_rank(rank, world, port, q):
    import os

    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        def f(_user, send, recv, n):
            mine = torch.frombuffer((C.c_ubyte * n).from_address(send), dtype=torch.uint8).clone()
            parts = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(parts, mine)
            whole = torch.cat(parts).contiguous().numpy()
            C.memmove(recv, whole.ctypes.data, n * world)
            return 0

        xfn = XFN(f)
        out = {}
        for name in ("c3p-64", "c4-unsat-96"):
            snap = COUPLED[name]()
            plain, sweeps, rounds, _ = _stages(snap, 20.0, 16, False)
            got, sw, rd, calls = _stages(snap, 20.0, 16, False, rank, world, xfn)
            out[name] = bool((sw, rd) == (sweeps, rounds) and got.counts == plain.counts and got.status == plain.status and got.batches == plain.batches and calls >= sweeps > 0)
            if not out[name]:
                print(f"rank {rank} {name}: sweeps/rounds {(sw, rd)} vs plain {(sweeps, rounds)}, exchanges {calls}, status {got.status} vs {plain.status}, counts equal {got.counts == plain.counts}", flush=True)
        snap = workloads.make_steady("c3", seed=0, n_workers=48, n_tasks=60_000)
        plain, _, _, _ = _stages(snap, 20.0, 1 << 30, True)
        got, _, _, calls = _stages(snap, 20.0, 1 << 30, True, rank, world, xfn)
        out["blocks"] = bool(got.counts == plain.counts and got.batches == plain.batches and calls == 1)
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_two_processes_over_gloo():
    import torch.multiprocessing as mp
    from test_sharded import _free_port

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gloo_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for (_, out) in res:
        assert all(out.values()), res


@pytest.mark.parametrize("fail_at", [0, 1, 4])
def test_a_rank_whose_sweeper_fails_does_not_leave_the_others_in_the_exchange(fail_at):
    """rank 1's sweeper refuses the model at begin() (0) or fails its 1st / 4th sweep.  It still takes part in the exchange the other ranks wait in, carries
    a 'failed' flag, and ALL ranks hand the model to the host search at the same point — same answer on every rank, no rank blocked (the join would time out)."""
    lib = _lib()
    lib.hqtick_debug_set_price_fault.argtypes = [C.c_int]
    snap = COUPLED["c3p-64"]()
    world = 3
    ex = ThreadExchange(world)
    out, errs = [None] * world, []

    def main(r):
        try:
            lib.hqtick_debug_set_price_fault(fail_at if r == 1 else -1)  # (thread-local)
            mine = dataclasses.replace(snap, _keep=[])
            out[r] = _stages(mine, 20.0, 16, False, r, world, ex.fn(r), 1, 1)
        except BaseException as e:  # noqa: BLE001
            errs.append((r, e)); ex.bar.abort()
        finally:
            lib.hqtick_debug_set_price_fault(-1)

    ts = [threading.Thread(target=main, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(300)
        assert not t.is_alive(), "a rank is still waiting in the exchange"
    assert not errs, errs
    for (got, sw, rd, _calls) in out[1:]:
        _same(got, out[0][0])
        assert (sw, rd) == (out[0][1], out[0][2])
    assert out[0][0].status in (abi.HQTICK_DONE, abi.HQTICK_NEED_MORE_COMPUTE)
