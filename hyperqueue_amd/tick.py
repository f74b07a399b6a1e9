"""ctypes binding of libhqtick.so (include/hqtick.h).  The library is the product: there is no Python or CPU
implementation of the tick behind this module, and loading/creating a context fails loudly when the HIP library or
a gfx950 device is missing."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from . import abi

_LIB = None
_MLIB = None
LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libhqtick.so")
MEASURE_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libhqtick_test.so")


class HqTickError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"hqtick error {code}: {msg}")
        self.code = code


def load(measure: bool = False):
    """the product library; measure=True: libhqtick_test.so — the same objects plus the measurement hooks of include/hqtick_debug.h (hqtick_time_kernel,
    hqtick_timeline, hqtick_block_profile_last) that the tools under tools/ use.  bench.py, the tests and smoke() run the product library."""
    global _LIB, _MLIB
    if measure:
        if _MLIB is None:
            _MLIB = _bind(C.CDLL(MEASURE_LIB_PATH))
        return _MLIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950)")
        _LIB = _bind(C.CDLL(LIB_PATH))
    return _LIB


def _bind(lib):
    P = C.POINTER
    lib.hqtick_create.argtypes = [P(abi.Config), P(C.c_void_p)]
    lib.hqtick_destroy.argtypes = [C.c_void_p]
    lib.hqtick_run.argtypes = [C.c_void_p, P(abi.SnapshotC), P(abi.ResultC)]
    lib.hqtick_upload_ready.argtypes = [C.c_void_p, C.c_uint64, abi.u64p, abi.u64p, abi.u32p, C.c_int]
    lib.hqtick_run_resident.argtypes = [C.c_void_p, P(abi.SnapshotC), P(abi.ResultC)]
    lib.hqtick_query.argtypes = [C.c_void_p, P(abi.SnapshotC), P(abi.QueryWorkersC), P(abi.QueryResultC)]
    lib.hqtick_last_error.restype = C.c_char_p
    lib.hqtick_last_error.argtypes = [C.c_void_p]
    lib.hqtick_abi_version.restype = C.c_uint32
    lib.hqtick_build_arch.restype = C.c_char_p
    lib.hqtick_kernel_stats_last.argtypes = [C.c_void_p, P(abi.KernelStatsC)]
    return lib


class Tick:
    """One hqtick_ctx (one HIP device, one stream).  `tick(snapshot)` == run_scheduling_inner through the C ABI."""

    def __init__(self, config: Optional[abi.Config] = None, measure: bool = False):
        self.cfg = config or abi.make_config()
        extra = int(os.environ.get("HQTICK_TEST_FLAGS", "0") or 0)  # campaigns (tools/fuzz_more.py): the same scenarios through another emission form
        if extra:
            c = abi.Config.from_buffer_copy(self.cfg)
            c.flags |= extra
            self.cfg = c
        self._lib = load(measure)
        ctx = C.c_void_p()
        rc = self._lib.hqtick_create(C.byref(self.cfg), C.byref(ctx))
        if rc != 0:
            raise HqTickError(rc, "hqtick_create failed (no gfx950 device / HIP runtime?)")
        self._ctx = ctx
        self._add_packed = None

    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.hqtick_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _err(self) -> str:
        return self._lib.hqtick_last_error(self._ctx).decode()

    def tick_raw(self, sc: abi.SnapshotC, resident: bool = False) -> abi.ResultC:
        rc_ = abi.ResultC()
        fn = self._lib.hqtick_run_resident if resident else self._lib.hqtick_run
        rc = fn(self._ctx, C.byref(sc), C.byref(rc_))
        if rc < 0:
            raise HqTickError(rc, self._err())
        return rc_

    def tick(self, snap: abi.Snapshot, resident: bool = False, resident_workers: bool = False, resident_retracting: bool = False) -> abi.Result:
        """resident_workers: the snapshot travels without its worker side; the library completes it from its own worker set (hqtick_cluster_*, ABI 7).
        resident_retracting: the Retracting tasks of the queues come from the library's own table (hqtick_retracting_*, ABI 7)"""
        sc = snap.to_c(resident_workers=resident_workers)
        if resident_retracting:
            sc.n_retracting = abi.HQ_RETRACTING_RESIDENT; sc.retracting_task = None; sc.retracting_worker = None; sc.retracting_redirect_worker = None; sc.retracting_redirect_variant = None
        return abi.parse_result(self.tick_raw(sc, resident), len(snap.worker_id), snap.n_resources)

    def batches(self, snap: abi.Snapshot):
        return abi.parse_batches(self.tick_raw(snap.to_c()))

    def upload_ready(self, task_id: np.ndarray, task_priority: np.ndarray, task_rq: np.ndarray, sorted_: bool = True):
        a, b, c = (np.ascontiguousarray(task_id, np.uint64), np.ascontiguousarray(task_priority, np.uint64), np.ascontiguousarray(task_rq, np.uint32))
        rc = self._lib.hqtick_upload_ready(self._ctx, len(a), a.ctypes.data_as(abi.u64p), b.ctypes.data_as(abi.u64p), c.ctypes.data_as(abi.u32p), 1 if sorted_ else 0)
        if rc < 0:
            raise HqTickError(rc, self._err())

    # -- resident ready-set deltas (include/hqtick.h) -------------------------------------------------------
    def _chk(self, rc: int) -> int:
        if rc < 0:
            raise HqTickError(rc, self._err())
        return rc

    def ready_consume_last(self):
        self._lib.hqtick_ready_consume_last.argtypes = [C.c_void_p]
        self._chk(self._lib.hqtick_ready_consume_last(self._ctx))

    def ready_remove(self, task_id) -> int:
        a = np.ascontiguousarray(task_id, np.uint64)
        self._lib.hqtick_ready_remove.argtypes = [C.c_void_p, C.c_uint64, abi.u64p]
        return self._chk(self._lib.hqtick_ready_remove(self._ctx, len(a), a.ctypes.data_as(abi.u64p)))

    def ready_add(self, task_id, task_priority, task_rq):
        a, b, c = (np.ascontiguousarray(task_id, np.uint64), np.ascontiguousarray(task_priority, np.uint64), np.ascontiguousarray(task_rq, np.uint32))
        self._lib.hqtick_ready_add.argtypes = [C.c_void_p, C.c_uint64, abi.u64p, abi.u64p, abi.u32p]
        self._chk(self._lib.hqtick_ready_add(self._ctx, len(a), a.ctypes.data_as(abi.u64p), b.ctypes.data_as(abi.u64p), c.ctypes.data_as(abi.u32p)))

    def ready_add_stage(self, n: int):
        """(ids u64[n], priorities u64[n], rqs u32[n]) as numpy views of the library's pinned staging buffer: fill them, then ready_add_staged(n)"""
        pi, pp, pq = abi.u64p(), abi.u64p(), abi.u32p()
        self._lib.hqtick_ready_add_stage.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(abi.u64p), C.POINTER(abi.u64p), C.POINTER(abi.u32p)]
        self._chk(self._lib.hqtick_ready_add_stage(self._ctx, n, C.byref(pi), C.byref(pp), C.byref(pq)))
        if n == 0:
            return np.zeros(0, np.uint64), np.zeros(0, np.uint64), np.zeros(0, np.uint32)
        return (np.ctypeslib.as_array(pi, shape=(n,)), np.ctypeslib.as_array(pp, shape=(n,)), np.ctypeslib.as_array(pq, shape=(n,)))

    def ready_add_staged(self, n: int):
        self._lib.hqtick_ready_add_staged.argtypes = [C.c_void_p, C.c_uint64]
        self._chk(self._lib.hqtick_ready_add_staged(self._ctx, n))

    def ready_compact(self):
        self._lib.hqtick_ready_compact.argtypes = [C.c_void_p]
        self._chk(self._lib.hqtick_ready_compact(self._ctx))

    def ready_count(self) -> int:
        self._lib.hqtick_ready_count.restype = C.c_uint64
        self._lib.hqtick_ready_count.argtypes = [C.c_void_p]
        return int(self._lib.hqtick_ready_count(self._ctx))

    # -- cluster tables resident in HBM (include/hqtick.h, ABI 5) ------------------------------------------------------------
    def cluster_upload(self, snap):
        """worker rows + request tables of `snap` (abi.Snapshot or SnapshotC) -> HBM; later ticks read them there"""
        sc = snap.to_c() if hasattr(snap, "to_c") else snap
        self._lib.hqtick_cluster_upload.argtypes = [C.c_void_p, C.POINTER(abi.SnapshotC)]
        self._chk(self._lib.hqtick_cluster_upload(self._ctx, C.byref(sc)))

    def cluster_update_workers(self, worker_index, free_rows, remaining_ns=None):
        """rows whose free resources (and optionally remaining lifetime) changed since the last call"""
        idx = np.ascontiguousarray(worker_index, np.uint32)
        rows = np.ascontiguousarray(free_rows, np.uint64).reshape(-1)
        rem = None if remaining_ns is None else np.ascontiguousarray(remaining_ns, np.int64)
        self._lib.hqtick_cluster_update_workers.argtypes = [C.c_void_p, C.c_uint32, abi.u32p, abi.u64p, C.POINTER(C.c_int64)]
        self._chk(self._lib.hqtick_cluster_update_workers(self._ctx, len(idx), idx.ctypes.data_as(abi.u32p), rows.ctypes.data_as(abi.u64p),
                                                          None if rem is None else rem.ctypes.data_as(C.POINTER(C.c_int64))))

    def cluster_add_workers(self, worker_id, total_rows, free_rows=None, remaining_ns=None, min_utilization=None, flags=None, group=None):
        """on_new_worker (ABI 7): ids ascending above every id present; they take the row indices W .. W + n - 1"""
        ids = np.ascontiguousarray(worker_id, np.uint32)
        tot = np.ascontiguousarray(total_rows, np.uint64).reshape(-1)
        fre = tot if free_rows is None else np.ascontiguousarray(free_rows, np.uint64).reshape(-1)
        opt = lambda a, dt, ct: (None, None) if a is None else (lambda v: (v, v.ctypes.data_as(C.POINTER(ct))))(np.ascontiguousarray(a, dt))
        rem, prem = opt(remaining_ns, np.int64, C.c_int64); mu, pmu = opt(min_utilization, np.float32, C.c_float); fl, pfl = opt(flags, np.uint8, C.c_uint8); gr, pgr = opt(group, np.uint32, C.c_uint32)
        self._lib.hqtick_cluster_add_workers.argtypes = [C.c_void_p, C.c_uint32, abi.u32p, abi.u64p, abi.u64p, C.POINTER(C.c_int64), C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.POINTER(C.c_uint32)]
        self._chk(self._lib.hqtick_cluster_add_workers(self._ctx, len(ids), ids.ctypes.data_as(abi.u32p), tot.ctypes.data_as(abi.u64p), fre.ctypes.data_as(abi.u64p), prem, pmu, pfl, pgr))

    def set_exchange(self, fn):
        """hqtick_set_exchange (ABI 8): fn(send_ptr, recv_ptr, bytes_per_rank) -> 0, an all-gather of host memory between the ranks of a sharded scheduler (the
        placement's price sweeps and class blocks are then split over the ranks); None removes it"""
        XFN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)
        self._lib.hqtick_set_exchange.argtypes = [C.c_void_p, XFN, C.c_void_p]
        self._xfn = XFN((lambda _u, s, r, n: fn(s, r, n))) if fn is not None else XFN(0)  # kept alive with the Tick
        self._chk(self._lib.hqtick_set_exchange(self._ctx, self._xfn, None))

    def ready_add_packed(self, id_runs, prio_runs, task_rq, id_off=None):
        """hqtick_ready_add_packed (ABI 8): id_runs = [(first id, length)], prio_runs = [(priority, length)], task_rq u16 per task, id_off = None (consecutive ids inside a
        run) or u32 offsets from the run's first id — 2-6 bytes per task over PCIe instead of hqtick_ready_add's 20"""
        rq = np.ascontiguousarray(task_rq, np.uint16)
        ist = np.array([r[0] for r in id_runs], np.uint64); iln = np.array([r[1] for r in id_runs], np.uint32)
        pv = np.array([r[0] for r in prio_runs], np.uint64); pln = np.array([r[1] for r in prio_runs], np.uint32)
        off = None if id_off is None else np.ascontiguousarray(id_off, np.uint32)
        f = self._add_packed
        if f is None:  # (plain addresses: building five typed ctypes pointers per call costs more than the library spends preparing the launch)
            f = self._add_packed = self._lib.hqtick_ready_add_packed
            f.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        self._chk(f(self._ctx, len(rq), len(ist), ist.ctypes.data, iln.ctypes.data, off.ctypes.data if off is not None else None, len(pv), pv.ctypes.data, pln.ctypes.data, rq.ctypes.data))

    def cluster_remove_workers(self, worker_id):
        """on_remove_worker (ABI 7): by id; later rows move up.  Returns [(task, target worker id, variant)]: Retracting tasks of the removed workers that carried a
        redirect and are Assigned to its target from now on — the host sends their ComputeTasks messages (ABI 8, hqtick_cluster_last_reassigned)"""
        ids = np.ascontiguousarray(worker_id, np.uint32)
        self._lib.hqtick_cluster_remove_workers.argtypes = [C.c_void_p, C.c_uint32, abi.u32p]
        self._chk(self._lib.hqtick_cluster_remove_workers(self._ctx, len(ids), ids.ctypes.data_as(abi.u32p)))
        n = C.c_uint32(); pt, pw, pv = abi.u64p(), abi.u32p(), abi.u8p()
        self._lib.hqtick_cluster_last_reassigned.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(abi.u64p), C.POINTER(abi.u32p), C.POINTER(abi.u8p)]
        self._chk(self._lib.hqtick_cluster_last_reassigned(self._ctx, C.byref(n), C.byref(pt), C.byref(pw), C.byref(pv)))
        k = n.value
        return list(zip(abi._np(pt, k, np.uint64).tolist(), abi._np(pw, k, np.uint32).tolist(), abi._np(pv, k, np.uint8).tolist())) if k else []

    def cluster_set_blocked(self, worker_id: int, pairs):
        """Worker::blocked_requests of one worker := pairs of (rq, variant) (ABI 7)"""
        rq = np.ascontiguousarray([p[0] for p in pairs], np.uint32); v = np.ascontiguousarray([p[1] for p in pairs], np.uint8)
        self._lib.hqtick_cluster_set_blocked.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, abi.u32p, abi.u8p]
        self._chk(self._lib.hqtick_cluster_set_blocked(self._ctx, int(worker_id), len(pairs), rq.ctypes.data_as(abi.u32p) if len(pairs) else None, v.ctypes.data_as(abi.u8p) if len(pairs) else None))

    def cluster_workers(self) -> np.ndarray:
        n = C.c_uint32(); p = abi.u32p()
        self._lib.hqtick_cluster_workers.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(abi.u32p)]
        self._chk(self._lib.hqtick_cluster_workers(self._ctx, C.byref(n), C.byref(p)))
        return abi._np(p, n.value, np.uint32).copy() if n.value else np.zeros(0, np.uint32)

    def retracting_add(self, task_id, worker_id):
        """process_retracted outside a tick (ABI 7): tasks back in their queue as Retracting{worker id}"""
        t = np.ascontiguousarray(task_id, np.uint64); w = np.ascontiguousarray(worker_id, np.uint32)
        self._lib.hqtick_retracting_add.argtypes = [C.c_void_p, C.c_uint32, abi.u64p, abi.u32p]
        self._chk(self._lib.hqtick_retracting_add(self._ctx, len(t), t.ctypes.data_as(abi.u64p), w.ctypes.data_as(abi.u32p)))

    def retract_response(self, worker_id: int, task_id):
        """on_retract_response (ABI 7) -> [(task, target worker id, variant)] of the tasks that are now Assigned to their redirect target"""
        t = np.ascontiguousarray(task_id, np.uint64)
        n = C.c_uint32(); pt, pw, pv = abi.u64p(), abi.u32p(), abi.u8p()
        self._lib.hqtick_retract_response.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, abi.u64p, C.POINTER(C.c_uint32), C.POINTER(abi.u64p), C.POINTER(abi.u32p), C.POINTER(abi.u8p)]
        self._chk(self._lib.hqtick_retract_response(self._ctx, int(worker_id), len(t), t.ctypes.data_as(abi.u64p), C.byref(n), C.byref(pt), C.byref(pw), C.byref(pv)))
        k = n.value
        return list(zip(abi._np(pt, k, np.uint64).tolist(), abi._np(pw, k, np.uint32).tolist(), abi._np(pv, k, np.uint8).tolist())) if k else []

    def retracting_count(self) -> int:
        self._lib.hqtick_retracting_count.restype = C.c_uint32
        self._lib.hqtick_retracting_count.argtypes = [C.c_void_p]
        return int(self._lib.hqtick_retracting_count(self._ctx))

    def cluster_drop(self):
        self._lib.hqtick_cluster_drop.argtypes = [C.c_void_p]
        self._chk(self._lib.hqtick_cluster_drop(self._ctx))

    # -- device-resident dependency graph (include/hqtick.h) --------------------------------------------------
    def _graph_last_ids(self) -> np.ndarray:
        n = C.c_uint64()
        self._lib.hqtick_graph_last_ids.restype = abi.u64p
        self._lib.hqtick_graph_last_ids.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        p = self._lib.hqtick_graph_last_ids(self._ctx, C.byref(n))
        return abi._np(p, n.value, np.uint64).copy() if n.value else np.zeros(0, np.uint64)

    def graph_add_tasks(self, task_id, task_priority, task_rq, deps) -> np.ndarray:
        """on_new_tasks (reactor.rs:188-220).  `deps` = one list of task ids per task, or a (dep_off, dep_task_id) CSR pair.
        Returns the ids that are ready now (ascending); they are already merged into the resident ready set."""
        a, b, c = (np.ascontiguousarray(task_id, np.uint64), np.ascontiguousarray(task_priority, np.uint64), np.ascontiguousarray(task_rq, np.uint32))
        if isinstance(deps, tuple):
            off, dep = np.ascontiguousarray(deps[0], np.uint32), np.ascontiguousarray(deps[1], np.uint64)
        else:
            off = np.zeros(len(a) + 1, np.uint32)
            off[1:] = np.cumsum([len(d) for d in deps])
            dep = np.ascontiguousarray([x for d in deps for x in d], np.uint64)
        self._lib.hqtick_graph_add_tasks.argtypes = [C.c_void_p, C.c_uint64, abi.u64p, abi.u64p, abi.u32p, abi.u32p, abi.u64p]
        self._chk(self._lib.hqtick_graph_add_tasks(self._ctx, len(a), a.ctypes.data_as(abi.u64p), b.ctypes.data_as(abi.u64p), c.ctypes.data_as(abi.u32p),
                                                   off.ctypes.data_as(abi.u32p), dep.ctypes.data_as(abi.u64p)))
        return self._graph_last_ids()

    def graph_finish(self, task_id) -> tuple[np.ndarray, int]:
        """task_finished (reactor.rs:510-590) for a batch.  Returns (released ids ascending, number of unknown ids)."""
        a = np.ascontiguousarray(task_id, np.uint64)
        self._lib.hqtick_graph_finish.argtypes = [C.c_void_p, C.c_uint64, abi.u64p]
        self._chk(self._lib.hqtick_graph_finish(self._ctx, len(a), a.ctypes.data_as(abi.u64p)))
        self._lib.hqtick_graph_last_unknown.restype = C.c_uint64
        self._lib.hqtick_graph_last_unknown.argtypes = [C.c_void_p]
        return self._graph_last_ids(), int(self._lib.hqtick_graph_last_unknown(self._ctx))

    def graph_remove(self, task_id, recursive: bool = False) -> np.ndarray:
        """Core::remove_task (core.rs:222-240), with the transitive consumers if `recursive` (task.rs:235-250).  Returns the removed ids."""
        a = np.ascontiguousarray(task_id, np.uint64)
        self._lib.hqtick_graph_remove.argtypes = [C.c_void_p, C.c_uint64, abi.u64p, C.c_int]
        self._chk(self._lib.hqtick_graph_remove(self._ctx, len(a), a.ctypes.data_as(abi.u64p), 1 if recursive else 0))
        return self._graph_last_ids()

    def graph_blevel(self, update_ready: bool = False) -> dict:
        """EXTENSION (include/hqtick.h: hqtick_graph_blevel; no reference counterpart, parity unpinned): b-levels into the low 32 bits of the graph's priorities."""
        mx, upd = C.c_uint32(0), C.c_uint32(0)
        self._lib.hqtick_graph_blevel.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        rc = self._lib.hqtick_graph_blevel(self._ctx, 1 if update_ready else 0, C.byref(mx), C.byref(upd))
        if rc < 0:
            self._chk(rc)
        return {"sweeps": int(rc), "max_level": int(mx.value), "ready_updated": int(upd.value)}

    def graph_priorities(self, task_id) -> np.ndarray:
        a = np.ascontiguousarray(task_id, np.uint64)
        out = np.zeros(len(a), np.uint64)
        self._lib.hqtick_graph_priorities.argtypes = [C.c_void_p, C.c_uint64, abi.u64p, abi.u64p]
        self._chk(self._lib.hqtick_graph_priorities(self._ctx, len(a), a.ctypes.data_as(abi.u64p), out.ctypes.data_as(abi.u64p)))
        return out

    def graph_unfinished(self, task_id) -> np.ndarray:
        a = np.ascontiguousarray(task_id, np.uint64)
        out = np.zeros(len(a), np.uint32)
        self._lib.hqtick_graph_unfinished.argtypes = [C.c_void_p, C.c_uint64, abi.u64p, abi.u32p]
        self._chk(self._lib.hqtick_graph_unfinished(self._ctx, len(a), a.ctypes.data_as(abi.u64p), out.ctypes.data_as(abi.u32p)))
        return out

    def graph_stats(self) -> dict:
        st = abi.GraphStatsC()
        self._lib.hqtick_graph_get_stats.argtypes = [C.c_void_p, C.POINTER(abi.GraphStatsC)]
        self._chk(self._lib.hqtick_graph_get_stats(self._ctx, C.byref(st)))
        return {k: getattr(st, k) for k, _ in abi.GraphStatsC._fields_}

    def query(self, snap: abi.Snapshot, fake_ids, fake_total, fake_remaining=None, fake_min_util=None):
        sc = snap.to_c()
        n = len(fake_ids)
        ids = np.ascontiguousarray(fake_ids, np.uint32)
        tot = np.ascontiguousarray(np.asarray(fake_total, np.uint64).reshape(-1))
        rem = np.ascontiguousarray(fake_remaining if fake_remaining is not None else np.full(n, abi.HQ_NO_TIME_LIMIT), np.int64)
        mu = np.ascontiguousarray(fake_min_util if fake_min_util is not None else np.zeros(n), np.float32)
        q = abi.QueryWorkersC(n, ids.ctypes.data_as(abi.u32p), tot.ctypes.data_as(abi.u64p), rem.ctypes.data_as(abi.i64p), mu.ctypes.data_as(abi.f32p))
        out = abi.QueryResultC()
        rc = self._lib.hqtick_query(self._ctx, C.byref(sc), C.byref(q), C.byref(out))
        if rc < 0:
            raise HqTickError(rc, self._err())
        return abi._np(out.is_loaded, n, np.uint8).astype(bool), bool(out.is_optimal)

    def time_kernel(self, which: int, iters: int = 100) -> float:
        """hqtick_time_kernel (measurement library only: Tick(..., measure=True)): average duration (us) of `iters` back-to-back launches of K1 (0) / K4 (1)."""
        us = C.c_double()
        self._lib.hqtick_time_kernel.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]
        rc = self._lib.hqtick_time_kernel(self._ctx, which, iters, C.byref(us))
        if rc < 0:
            raise HqTickError(rc, self._err())
        return us.value

    def set_kernel_timing(self, on):
        """False / True: every measured kernel; 2: K1 (k_level_hist) alone"""
        self._lib.hqtick_set_kernel_timing.argtypes = [C.c_void_p, C.c_int]
        self._chk(self._lib.hqtick_set_kernel_timing(self._ctx, 2 if on == 2 else (1 if on else 0)))

    def kernel_stats(self) -> dict:
        ks = abi.KernelStatsC()
        self._lib.hqtick_kernel_stats_last(self._ctx, C.byref(ks))
        return {f: getattr(ks, f) for f, _ in abi.KernelStatsC._fields_}
