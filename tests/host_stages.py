"""CPU-side driver of hqtick_debug_host_stages (include/hqtick_debug.h): the tick's HOST stages — create_task_batches +
run_scheduling_solver, incl. the separable path, the lazy batch-size rows, the elimination of provably empty workers and the exact solver —
on scan outputs computed here in numpy.  The numpy part restates the CONTRACT of the scan kernels (what K2 writes per (worker, variant),
what K0/K1/K1b produce per (level, rq)); it is test infrastructure, the product never runs it."""
import ctypes as C

import numpy as np

from hyperqueue_amd import abi, tick

INT64_MAX = (1 << 63) - 1


def scan_outputs(sc: abi.SnapshotC):
    """(vflags u8[W * NV], vtmc u32[W * NV], levels u64[L] descending, hist u32[L * Q]) — csrc/kernels.hip: worker_eval_block, k_level_hist"""
    W, R, Q = sc.n_workers, sc.n_resources, sc.n_requests
    nv, voff, res, kind, amt, mint = request_tables(sc)
    total = abi._np(sc.worker_total, W * R, np.uint64).reshape(W, R) if W * R else np.zeros((W, R), np.uint64)
    free = abi._np(sc.worker_free, W * R, np.uint64).reshape(W, R) if W * R else np.zeros((W, R), np.uint64)
    rem = abi._np(sc.worker_remaining_ns, W, np.int64) if (W and sc.worker_remaining_ns) else np.full(W, INT64_MAX, np.int64)
    flags, tmc = worker_eval(W, R, nv, voff, res, kind, amt, mint, total, free, rem)
    n = int(sc.n_ready)
    prio = abi._np(sc.task_priority, n, np.uint64) if n else np.zeros(0, np.uint64)
    rq = abi._np(sc.task_rq, n, np.uint32) if n else np.zeros(0, np.uint32)
    levels = np.unique(prio)[::-1].copy()
    hist = np.zeros(len(levels) * max(Q, 1), np.uint32)
    if n:
        asc = levels[::-1]
        li = (len(levels) - 1 - np.searchsorted(asc, prio)).astype(np.int64)  # index in the descending table
        np.add.at(hist, li * Q + rq.astype(np.int64), 1)
    return flags, tmc, levels, hist[: len(levels) * Q]


def request_tables(sc: abi.SnapshotC):
    Q = sc.n_requests
    nv = int(abi._np(sc.rq_variant_off, Q + 1, np.uint32)[Q]) if Q else 0
    voff = abi._np(sc.variant_entry_off, nv + 1, np.uint32) if nv else np.zeros(1, np.uint32)
    ne = int(voff[nv]) if nv else 0
    res, kind, amt = (abi._np(sc.entry_resource, ne, np.uint32), abi._np(sc.entry_kind, ne, np.uint8), abi._np(sc.entry_amount, ne, np.uint64)) if ne else (np.zeros(0, np.uint32),) * 3
    mint = abi._np(sc.variant_min_time_ns, nv, np.uint64) if (nv and sc.variant_min_time_ns) else np.zeros(nv, np.uint64)
    return nv, voff, res, kind, amt, mint


def worker_eval(W, R, nv, voff, res, kind, amt, mint, total, free, rem):
    """csrc/kernels.hip worker_eval_block, per (worker, variant slot): flags (bit 0 fits free, bit 1 fits total, bit 2 time) and task_max_count"""
    flags, tmc = np.zeros(W * nv, np.uint8), np.zeros(W * nv, np.uint32)
    for w in range(W):
        for v in range(nv):
            imm = cap = True
            best = None
            for e in range(int(voff[v]), int(voff[v + 1])):
                r = int(res[e])
                f, t = (int(free[w, r]), int(total[w, r])) if r < R else (0, 0)
                if kind[e] == 0:
                    a = int(amt[e])
                    imm, cap = imm and a <= f, cap and a <= t
                    c = min(f // a, 1024)
                else:
                    imm, cap = imm and f >= 1, cap and t >= 1
                    c = 0 if f == 0 else 1
                best = c if best is None else min(best, c)
            time_ok = int(rem[w]) == INT64_MAX or (int(rem[w]) >= 0 and int(rem[w]) >= int(mint[v]))
            flags[w * nv + v] = (1 if imm else 0) | (2 if cap else 0) | (4 if time_ok else 0)
            tmc[w * nv + v] = 0 if best is None else best
    return flags, tmc


class HostStages:
    """`stages(snapshot)` -> abi.Result with status / is_optimal / is_canonical / batches / counts filled (records empty)."""

    def __init__(self, config=None):
        self.cfg = config or abi.make_config()
        from hyperqueue_amd import _testhooks

        self.lib = _testhooks.load()  # libhqtick_test.so: the product library exports no CPU hook
        self.lib.hqtick_debug_host_stages.argtypes = [C.POINTER(abi.Config), C.POINTER(abi.SnapshotC), abi.u8p, abi.u32p, C.c_uint32, abi.u64p, abi.u32p, C.POINTER(abi.ResultC)]

    def stages(self, snap: abi.Snapshot) -> abi.Result:
        sc = snap.to_c()
        flags, tmc, levels, hist = scan_outputs(sc)
        out = abi.ResultC()
        z8, z32, z64 = np.zeros(1, np.uint8), np.zeros(1, np.uint32), np.zeros(1, np.uint64)
        f, t, lv, h = (flags if len(flags) else z8), (tmc if len(tmc) else z32), (levels if len(levels) else z64), (hist if len(hist) else z32)
        rc = self.lib.hqtick_debug_host_stages(C.byref(self.cfg), C.byref(sc), f.ctypes.data_as(abi.u8p), t.ctypes.data_as(abi.u32p), len(levels),
                                               lv.ctypes.data_as(abi.u64p), h.ctypes.data_as(abi.u32p), C.byref(out))
        if rc < 0:
            raise tick.HqTickError(rc, "hqtick_debug_host_stages")
        return abi.parse_result(out, len(snap.worker_id), snap.n_resources)

    def query(self, snap: abi.Snapshot, fake_ids, fake_total, fake_remaining=None, fake_min_util=None):
        """same contract as Tick.query, through hqtick_debug_host_query: (is_loaded per fake worker, is_optimal)"""
        sc = snap.to_c()
        flags, tmc, levels, hist = scan_outputs(sc)
        n, R = len(fake_ids), snap.n_resources
        ids = np.ascontiguousarray(fake_ids, np.uint32)
        tot = np.ascontiguousarray(np.asarray(fake_total, np.uint64).reshape(n, R))
        rem = np.ascontiguousarray(fake_remaining if fake_remaining is not None else np.full(n, abi.HQ_NO_TIME_LIMIT), np.int64)
        mu = np.ascontiguousarray(fake_min_util if fake_min_util is not None else np.zeros(n), np.float32)
        nv, voff, res, kind, amt, mint = request_tables(sc)
        fflags, ftmc = worker_eval(n, R, nv, voff, res, kind, amt, mint, tot, tot, rem)  # fresh fake workers: free == total
        q = abi.QueryWorkersC(n, ids.ctypes.data_as(abi.u32p), tot.ctypes.data_as(abi.u64p), rem.ctypes.data_as(abi.i64p), mu.ctypes.data_as(abi.f32p))
        out = abi.QueryResultC()
        z8, z32, z64 = np.zeros(1, np.uint8), np.zeros(1, np.uint32), np.zeros(1, np.uint64)
        pick = lambda a, z: a if len(a) else z
        self.lib.hqtick_debug_host_query.argtypes = [C.POINTER(abi.Config), C.POINTER(abi.SnapshotC), C.POINTER(abi.QueryWorkersC), abi.u8p, abi.u32p, abi.u8p, abi.u32p,
                                                     C.c_uint32, abi.u64p, abi.u32p, C.POINTER(abi.QueryResultC)]
        rc = self.lib.hqtick_debug_host_query(C.byref(self.cfg), C.byref(sc), C.byref(q), pick(flags, z8).ctypes.data_as(abi.u8p), pick(tmc, z32).ctypes.data_as(abi.u32p),
                                              pick(fflags, z8).ctypes.data_as(abi.u8p), pick(ftmc, z32).ctypes.data_as(abi.u32p), len(levels),
                                              pick(levels, z64).ctypes.data_as(abi.u64p), pick(hist, z32).ctypes.data_as(abi.u32p), C.byref(out))
        if rc < 0:
            raise tick.HqTickError(rc, "hqtick_debug_host_query")
        return abi._np(out.is_loaded, n, np.uint8).astype(bool), bool(out.is_optimal)

