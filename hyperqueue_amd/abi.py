"""ctypes mirror of include/hqtick.h (the C ABI of libhqtick.so) plus snapshot/result marshalling.

Pure plumbing: no scheduling logic lives here.  Field order and types MUST match include/hqtick.h
(tests/test_abi.py checks sizes and that the library exports every declared symbol).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

HQTICK_ABI_VERSION = 10
HQTICK_FLAG_NO_KERNEL_TIMING, HQTICK_FLAG_COMPACT_RECORDS, HQTICK_FLAG_COMPACT_DELTA16, HQTICK_FLAG_NO_BLOCK_MEMO, HQTICK_FLAG_CONSUME_IN_TICK, HQTICK_FLAG_CERTIFICATE_ONLY, HQTICK_FLAG_NO_TICK_CACHES = 1, 2, 4, 8, 16, 32, 64
HQ_AMOUNT_MAX = 0xFFFF_FFFF_FFFF_FFFF
HQ_FRACTIONS_PER_UNIT = 10_000
HQ_MAX_TASK_PER_WORKER = 1024
HQ_NO_TIME_LIMIT = 0x7FFF_FFFF_FFFF_FFFF
HQ_BLOCKER_UNBOUNDED = 0xFFFF_FFFF
HQ_NO_WORKER = 0xFFFF_FFFF
HQ_RETRACTING_RESIDENT = HQ_WORKERS_RESIDENT = 0xFFFF_FFFF  # sentinels of n_retracting / n_workers: that side of the snapshot lives in the library
HQ_ENTRY_AMOUNT, HQ_ENTRY_ALL = 0, 1
HQ_WORKER_SN, HQ_WORKER_STOPPING = 1, 2
HQTICK_DONE, HQTICK_NEED_MORE_COMPUTE, HQTICK_NO_PROGRESS = 0, 1, 2
HQTICK_E_INVALID, HQTICK_E_NO_DEVICE, HQTICK_E_DEVICE = -1, -2, -3
HQTICK_E_CAPACITY, HQTICK_E_QUEUE_UNDERFLOW, HQTICK_E_UNSUPPORTED = -4, -5, -6
HQ_REC_PREFILL, HQ_REC_ASSIGN = 0, 1
HQ_REDIRECT_FROM_PREFILL, HQ_REDIRECT_RETARGET, HQ_REDIRECT_SAME_WORKER = 0, 1, 2

u8p, u32p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
i64p, f32p = C.POINTER(C.c_int64), C.POINTER(C.c_float)


class Config(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32),
        ("proactive_filling_reserve", C.c_uint32),
        ("proactive_filling_max", C.c_uint32),
        ("mip_time_limit_s", C.c_double),
        ("device_index", C.c_int32),
        ("flags", C.c_uint32),
    ]


def make_config(reserve: int = 16, fill_max: int = 40, time_limit_s: float = 60.0, device_index: int = 0, flags: int = 0) -> Config:
    """SchedulerConfig defaults (scheduler/state.rs:19-27); 60 s is the reference's cfg(test) limit."""
    return Config(HQTICK_ABI_VERSION, reserve, fill_max, time_limit_s, device_index, flags)


class SnapshotC(C.Structure):
    _fields_ = [
        ("n_resources", C.c_uint32),
        ("n_workers", C.c_uint32),
        ("worker_id", u32p),
        ("worker_total", u64p),
        ("worker_free", u64p),
        ("worker_remaining_ns", i64p),
        ("worker_min_utilization", f32p),
        ("worker_flags", u8p),
        ("worker_group", u32p),
        ("n_groups", C.c_uint32),
        ("worker_map_rank", u32p),
        ("n_blocked", C.c_uint32),
        ("blocked_worker", u32p),
        ("blocked_rq", u32p),
        ("blocked_variant", u8p),
        ("assigned_off", u32p),
        ("assigned_rq", u32p),
        ("assigned_variant", u8p),
        ("prefilled_off", u32p),
        ("prefilled_rq", u32p),
        ("n_requests", C.c_uint32),
        ("rq_variant_off", u32p),
        ("variant_entry_off", u32p),
        ("variant_n_nodes", u32p),
        ("variant_min_time_ns", u64p),
        ("variant_weight", u32p),
        ("entry_resource", u32p),
        ("entry_kind", u8p),
        ("entry_amount", u64p),
        ("n_ready", C.c_uint64),
        ("task_id", u64p),
        ("task_priority", u64p),
        ("task_rq", u32p),
        ("prefill_off", u32p),
        ("prefill_priority", u64p),
        ("prefill_task", u64p),
        ("prefill_worker", u32p),
        ("n_retracting", C.c_uint32),
        ("retracting_task", u64p),
        ("retracting_worker", u32p),
        ("retracting_redirect_worker", u32p),
        ("retracting_redirect_variant", u8p),
    ]


class GraphStatsC(C.Structure):
    _fields_ = [("n_tasks", C.c_uint64), ("n_slots", C.c_uint64), ("n_edges_live", C.c_uint64), ("n_edges_pool", C.c_uint64), ("n_runs", C.c_uint64),
                ("hash_capacity", C.c_uint64), ("hash_tombstones", C.c_uint64), ("bytes_hbm", C.c_uint64), ("last_kernel_us", C.c_double)]


class QueryWorkersC(C.Structure):
    _fields_ = [
        ("n_workers", C.c_uint32),
        ("worker_id", u32p),
        ("worker_total", u64p),
        ("worker_remaining_ns", i64p),
        ("worker_min_utilization", f32p),
    ]


class ResultC(C.Structure):
    _fields_ = [
        ("status", C.c_int32),
        ("is_optimal", C.c_uint8),
        ("is_canonical", C.c_uint8),
        ("n_batches", C.c_uint32),
        ("batch_rq", u32p),
        ("batch_size", u32p),
        ("batch_limit", u32p),
        ("batch_limit_reached", u8p),
        ("batch_is_blocker", u8p),
        ("batch_cut_off", u32p),
        ("cut_size", u32p),
        ("cut_blocker_off", u32p),
        ("blocker_rq", u32p),
        ("blocker_size", u32p),
        ("n_counts", C.c_uint32),
        ("count_rq", u32p),
        ("count_variant", u8p),
        ("count_worker", u32p),
        ("count_value", u32p),
        ("rec_off", u32p),
        ("rec_task", u64p),
        ("rec_variant", u8p),
        ("rec_kind", u8p),
        ("retract_off", u32p),
        ("retract_task", u64p),
        ("rec_task_lo", u32p),
        ("run_span", u32p),  # hqtick_run_span[W]: (start, count) pairs
        ("runs", u32p),      # hqtick_rec_run[]: (first, job, meta) triples
        ("rec_delta16", C.POINTER(C.c_uint16)),  # HQTICK_FLAG_COMPACT_DELTA16: unit streams, worker w's at unit 4 * rec_off[w]
        ("runs16", u32p),    # hqtick_rec_run16[]: (first, job, meta, first_lo)
        ("n_redirects", C.c_uint32),
        ("redirect_task", u64p),
        ("redirect_worker", u32p),
        ("redirect_variant", u8p),
        ("redirect_kind", u8p),
        ("n_mn", C.c_uint32),
        ("mn_task", u64p),
        ("mn_worker_off", u32p),
        ("mn_worker", u32p),
        ("new_free", u64p),
        ("t_total_us", C.c_double),
        ("t_scan_us", C.c_double),
        ("t_batches_us", C.c_double),
        ("t_solve_us", C.c_double),
        ("t_mapping_us", C.c_double),
    ]


class QueryResultC(C.Structure):
    _fields_ = [("n_workers", C.c_uint32), ("is_loaded", u8p), ("is_optimal", C.c_uint8)]


class KernelStatsC(C.Structure):
    _fields_ = [
        ("level_hist_us", C.c_double),
        ("select_us", C.c_double),
        ("distinct_us", C.c_double),
        ("other_us", C.c_double),
        ("scan_us", C.c_double),
        ("sweep_us", C.c_double),
        ("tick_gpu_us", C.c_double),
        ("algorithmic_bytes", C.c_uint64),
        ("n_assigned", C.c_uint64),
        ("n_prefilled", C.c_uint64),
        ("block_solve_us", C.c_double),
        ("n_classes_device", C.c_uint32),
        ("n_classes_host", C.c_uint32),
        ("block_steps_max", C.c_uint32),
        ("n_classes", C.c_uint32),
        ("solve_classify_us", C.c_double),
        ("solve_blocks_us", C.c_double),
        ("solve_decode_us", C.c_double),
        ("price_sweeps", C.c_uint32), ("price_rounds", C.c_uint32), ("milp_cols", C.c_uint32), ("milp_rows", C.c_uint32),
        ("price_us", C.c_double), ("price_sweep_us", C.c_double), ("milp_us", C.c_double), ("model_us", C.c_double), ("solve_pre_us", C.c_double),
        ("n_classes_verified", C.c_uint32), ("n_classes_mismatch", C.c_uint32), ("n_classes_rejected", C.c_uint32), ("n_classes_memo", C.c_uint32),
        ("exchange_calls", C.c_uint32), ("ready_appends", C.c_uint32), ("exchange_bytes", C.c_uint64), ("exchange_us", C.c_double),
    ]


def _arr(a, dtype) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=dtype))


def _ptr(a: np.ndarray, typ):
    return a.ctypes.data_as(typ)


@dataclass
class Snapshot:
    """Flattened `Core` (see include/hqtick.h `hqtick_snapshot`).  Holds numpy columns and builds the C view."""

    n_resources: int
    worker_id: np.ndarray
    worker_total: np.ndarray  # [W, R]
    worker_free: np.ndarray  # [W, R]
    worker_remaining_ns: np.ndarray
    worker_min_utilization: np.ndarray
    worker_flags: np.ndarray
    worker_group: np.ndarray
    n_groups: int
    blocked: List[Tuple[int, int, int]]  # (worker index, rq, variant)
    assigned: List[List[Tuple[int, int]]]  # per worker: (rq, variant)
    prefilled: List[List[int]]  # per worker: rq
    requests: List[List[dict]]  # per rq: variants {entries:[(res, kind, amount)], n_nodes, min_time_ns, weight}
    task_id: np.ndarray
    task_priority: np.ndarray
    task_rq: np.ndarray
    prefill: Dict[int, Tuple[int, List[Tuple[int, int]]]] = field(default_factory=dict)  # rq -> (priority, [(task, worker idx)])
    worker_map_rank: Optional[np.ndarray] = None
    retracting: List[Tuple[int, int, int, int]] = field(default_factory=list)  # (task, old worker idx, redirect worker idx | HQ_NO_WORKER, redirect variant), ascending task id
    _keep: list = field(default_factory=list, repr=False)

    def to_c(self, resident_workers: bool = False) -> SnapshotC:
        """resident_workers: leave the worker side out (worker_id == NULL, n_workers = 0, no blocked triples) — the library completes it from the
        worker set it keeps itself (hqtick_cluster_upload / _add_workers / _remove_workers / _set_blocked / _update_workers, ABI 7); the per-worker
        CSRs of running and prefilled tasks still travel (they belong to the task side of the reactor's state)."""
        keep = self._keep
        keep.clear()
        W = len(self.worker_id)
        R = self.n_resources
        s = SnapshotC()

        def put(name, arr, dtype, typ):
            a = _arr(arr, dtype)
            keep.append(a)
            setattr(s, name, _ptr(a, typ))
            return a

        s.n_resources = R
        if resident_workers:
            s.n_workers = HQ_WORKERS_RESIDENT  # fails loudly if the library holds no worker set (0 would run as a tick of zero workers)
            s.n_groups = self.n_groups
        else:
            s.n_workers = W
            put("worker_id", self.worker_id, np.uint32, u32p)
            put("worker_total", np.asarray(self.worker_total, dtype=np.uint64).reshape(W * R), np.uint64, u64p)
            put("worker_free", np.asarray(self.worker_free, dtype=np.uint64).reshape(W * R), np.uint64, u64p)
            put("worker_remaining_ns", self.worker_remaining_ns, np.int64, i64p)
            put("worker_min_utilization", self.worker_min_utilization, np.float32, f32p)
            put("worker_flags", self.worker_flags, np.uint8, u8p)
            put("worker_group", self.worker_group, np.uint32, u32p)
            s.n_groups = self.n_groups
            if self.worker_map_rank is not None:
                put("worker_map_rank", self.worker_map_rank, np.uint32, u32p)
            s.n_blocked = len(self.blocked)
            put("blocked_worker", [b[0] for b in self.blocked], np.uint32, u32p)
            put("blocked_rq", [b[1] for b in self.blocked], np.uint32, u32p)
            put("blocked_variant", [b[2] for b in self.blocked], np.uint8, u8p)
        off = np.zeros(W + 1, dtype=np.uint32)
        off[1:] = np.cumsum([len(a) for a in self.assigned]) if W else []
        put("assigned_off", off, np.uint32, u32p)
        put("assigned_rq", [x[0] for a in self.assigned for x in a], np.uint32, u32p)
        put("assigned_variant", [x[1] for a in self.assigned for x in a], np.uint8, u8p)
        off = np.zeros(W + 1, dtype=np.uint32)
        off[1:] = np.cumsum([len(a) for a in self.prefilled]) if W else []
        put("prefilled_off", off, np.uint32, u32p)
        put("prefilled_rq", [x for a in self.prefilled for x in a], np.uint32, u32p)
        Q = len(self.requests)
        s.n_requests = Q
        rq_off, v_eoff, v_nodes, v_time, v_weight, e_res, e_kind, e_amount = [0], [0], [], [], [], [], [], []
        for variants in self.requests:
            for v in variants:
                for (res, kind, amount) in v["entries"]:
                    e_res.append(res)
                    e_kind.append(kind)
                    e_amount.append(amount)
                v_eoff.append(len(e_res))
                v_nodes.append(v.get("n_nodes", 0))
                v_time.append(v.get("min_time_ns", 0))
                v_weight.append(v.get("weight", 10_000))
            rq_off.append(len(v_nodes))
        put("rq_variant_off", rq_off, np.uint32, u32p)
        put("variant_entry_off", v_eoff, np.uint32, u32p)
        put("variant_n_nodes", v_nodes, np.uint32, u32p)
        put("variant_min_time_ns", v_time, np.uint64, u64p)
        put("variant_weight", v_weight, np.uint32, u32p)
        put("entry_resource", e_res, np.uint32, u32p)
        put("entry_kind", e_kind, np.uint8, u8p)
        put("entry_amount", e_amount, np.uint64, u64p)
        s.n_ready = len(self.task_id)
        put("task_id", self.task_id, np.uint64, u64p)
        put("task_priority", self.task_priority, np.uint64, u64p)
        put("task_rq", self.task_rq, np.uint32, u32p)
        poff, pprio, ptask, pworker = [0], [], [], []
        for q in range(Q):
            if q in self.prefill and self.prefill[q][1]:
                prio, items = self.prefill[q]
                pprio.append(prio)
                for (t, w) in items:
                    ptask.append(t)
                    pworker.append(w)
            else:
                pprio.append(0)
            poff.append(len(ptask))
        put("prefill_off", poff, np.uint32, u32p)
        put("prefill_priority", pprio, np.uint64, u64p)
        put("prefill_task", ptask, np.uint64, u64p)
        put("prefill_worker", pworker, np.uint32, u32p)
        s.n_retracting = len(self.retracting)
        put("retracting_task", [r[0] for r in self.retracting], np.uint64, u64p)
        put("retracting_worker", [r[1] for r in self.retracting], np.uint32, u32p)
        put("retracting_redirect_worker", [r[2] for r in self.retracting], np.uint32, u32p)
        put("retracting_redirect_variant", [r[3] for r in self.retracting], np.uint8, u8p)
        return s


def _np(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)


@dataclass
class Batch:
    rq: int
    size: int
    limit: int
    limit_reached: bool
    is_blocker: bool
    cuts: List[Tuple[int, List[Tuple[int, Optional[int]]]]]  # (size, [(rq, Some(size)|None)])


@dataclass
class Result:
    status: int
    is_optimal: bool
    batches: List[Batch]
    counts: List[Tuple[int, int, int, int]]  # (rq, variant, worker index, count) in the reference's iteration order
    records: List[List[Tuple[int, int, int]]]  # per worker index: (task, variant|0xFF, kind)
    retracts: List[List[int]]
    redirects: List[Tuple[int, int, int]]  # (task, worker index, variant)  [+ kind in redirect_kinds, same order]
    mn: List[Tuple[int, List[int]]]
    new_free: np.ndarray  # [W, R]
    times_us: Dict[str, float]
    redirect_kinds: List[int] = field(default_factory=list)
    is_canonical: bool = True  # hqtick_result.is_canonical: the tie-break phase completed

    def assigned(self, w: int) -> List[Tuple[int, int]]:
        return [(t, v) for (t, v, k) in self.records[w] if k == HQ_REC_ASSIGN]

    def prefills(self, w: int) -> List[int]:
        return [t for (t, v, k) in self.records[w] if k == HQ_REC_PREFILL]

    def counts_dict(self) -> Dict[Tuple[int, int, int], int]:
        return {(rq, v, w): c for (rq, v, w, c) in self.counts}


def parse_batches(r: ResultC) -> List[Batch]:
    nb = r.n_batches
    rq, size, limit = _np(r.batch_rq, nb, np.uint32), _np(r.batch_size, nb, np.uint32), _np(r.batch_limit, nb, np.uint32)
    lr, blk = _np(r.batch_limit_reached, nb, np.uint8), _np(r.batch_is_blocker, nb, np.uint8)
    coff = _np(r.batch_cut_off, nb + 1, np.uint32) if nb else np.zeros(1, np.uint32)
    ncuts = int(coff[-1]) if nb else 0
    csz = _np(r.cut_size, ncuts, np.uint32)
    cboff = _np(r.cut_blocker_off, ncuts + 1, np.uint32) if ncuts else np.zeros(1, np.uint32)
    nbl = int(cboff[-1]) if ncuts else 0
    brq, bsz = _np(r.blocker_rq, nbl, np.uint32), _np(r.blocker_size, nbl, np.uint32)
    out = []
    for b in range(nb):
        cuts = []
        for c in range(int(coff[b]), int(coff[b + 1])):
            bl = [(int(brq[k]), None if int(bsz[k]) == HQ_BLOCKER_UNBOUNDED else int(bsz[k])) for k in range(int(cboff[c]), int(cboff[c + 1]))]
            cuts.append((int(csz[c]), bl))
        out.append(Batch(int(rq[b]), int(size[b]), int(limit[b]), bool(lr[b]), bool(blk[b]), cuts))
    return out


def expand_compact(r: ResultC, W: int, off: np.ndarray):
    """(task ids, variants, kinds) of every record from the compact emission — what the host shim does while it applies the records"""
    n = int(off[-1])
    lo = _np(r.rec_task_lo, n, np.uint32).astype(np.uint64)
    span = _np(r.run_span, 2 * W, np.uint32).reshape(W, 2)
    rs, rc = span[:, 0], span[:, 1]
    task = np.zeros(n, np.uint64); var = np.zeros(n, np.uint8); kind = np.zeros(n, np.uint8)
    have = np.nonzero(off[1:] > off[:-1])[0]
    if len(have):
        hi_run = int(max(int(rs[w]) + int(rc[w]) for w in have))
        runs = _np(r.runs, 3 * hi_run, np.uint32).reshape(hi_run, 3)
        first, job, meta = runs[:, 0], runs[:, 1], runs[:, 2].astype(np.uint16)
        for w in have:
            a, b, s0, c = int(off[w]), int(off[w + 1]), int(rs[w]), int(rc[w])
            assert c >= 1 and first[s0] == 0, (w, c)
            starts = first[s0:s0 + c].astype(np.int64)
            assert (np.diff(starts) > 0).all() and starts[-1] < b - a
            lens = np.diff(np.append(starts, b - a))
            task[a:b] = (np.repeat(job[s0:s0 + c].astype(np.uint64), lens) << np.uint64(32)) | lo[a:b]
            var[a:b] = np.repeat((meta[s0:s0 + c] & 0xFF).astype(np.uint8), lens)
            kind[a:b] = np.repeat((meta[s0:s0 + c] >> 8).astype(np.uint8), lens)
    return task.tolist(), var.tolist(), kind.tolist()


def expand_delta16(r: ResultC, W: int, off: np.ndarray):
    """(task ids, variants, kinds) of every record from the 16-bit-difference emission (HQTICK_FLAG_COMPACT_DELTA16) — the decoder of include/hqtick.h"""
    n = int(off[-1])
    span = _np(r.run_span, 2 * W, np.uint32).reshape(W, 2)
    task = np.zeros(n, np.uint64); var = np.zeros(n, np.uint8); kind = np.zeros(n, np.uint8)
    have = np.nonzero(off[1:] > off[:-1])[0]
    if not len(have):
        return [], [], []
    hi_run = int(max(int(span[w, 0]) + int(span[w, 1]) for w in have))
    runs = _np(r.runs16, 4 * hi_run, np.uint32).reshape(hi_run, 4)
    units_all = _np(r.rec_delta16, 4 * n, np.uint16)
    for w in have:
        a, b, s0, c = int(off[w]), int(off[w + 1]), int(span[w, 0]), int(span[w, 1])
        assert c >= 1 and runs[s0, 0] == 0 and s0 == a, (w, c)
        tot = b - a
        starts = runs[s0:s0 + c, 0].astype(np.int64)
        assert (np.diff(starts) > 0).all() and starts[-1] < tot
        lens = np.diff(np.append(starts, tot))
        units = units_all[4 * a:4 * a + 3 * tot].astype(np.int64)
        opens = np.zeros(tot, bool); opens[starts] = True
        lo = np.zeros(tot, np.int64)
        lo[starts] = runs[s0:s0 + c, 3].astype(np.int64)
        n_plain = tot - c  # records that read units
        if n_plain and not (units[:n_plain] == 0xFFFF).any():  # no escape: one unit per record, running sums inside every run
            d = np.zeros(tot, np.int64); d[~opens] = units[:n_plain]
            seg = np.repeat(np.arange(c), lens)
            cs = np.cumsum(d); base = cs[starts]  # cs at a run's first record (its own d is 0)
            lo = (lo[starts][seg] + cs - base[seg]) & 0xFFFFFFFF
        elif n_plain:
            up = 0; prev = 0
            for i in range(tot):
                if opens[i]:
                    prev = int(lo[i]); continue
                u = int(units[up])
                if u != 0xFFFF:
                    prev = (prev + u) & 0xFFFFFFFF; up += 1
                else:
                    prev = int(units[up + 1]) | (int(units[up + 2]) << 16); up += 3
                lo[i] = prev
        task[a:b] = (np.repeat(runs[s0:s0 + c, 1].astype(np.uint64), lens) << np.uint64(32)) | lo.astype(np.uint64)
        meta = runs[s0:s0 + c, 2].astype(np.uint16)
        var[a:b] = np.repeat((meta & 0xFF).astype(np.uint8), lens)
        kind[a:b] = np.repeat((meta >> 8).astype(np.uint8), lens)
    return task.tolist(), var.tolist(), kind.tolist()


def record_task_ids(r: ResultC, W: int) -> np.ndarray:
    """task ids of every record of a tick in CSR order (rec_off), from either emission format"""
    off = _np(r.rec_off, W + 1, np.uint32)
    n = int(off[-1])
    if n == 0:
        return np.zeros(0, np.uint64)
    if r.rec_delta16:
        return np.asarray(expand_delta16(r, W, off)[0], np.uint64)
    if r.rec_task_lo:
        return np.asarray(expand_compact(r, W, off)[0], np.uint64)
    return _np(r.rec_task, n, np.uint64).copy()


def parse_result(r: ResultC, n_workers: int, n_resources: int, full: bool = True) -> Result:
    batches = parse_batches(r)
    nc = r.n_counts
    counts = list(
        zip(
            _np(r.count_rq, nc, np.uint32).tolist(),
            _np(r.count_variant, nc, np.uint8).tolist(),
            _np(r.count_worker, nc, np.uint32).tolist(),
            _np(r.count_value, nc, np.uint32).tolist(),
        )
    )
    W = n_workers
    records: List[List[Tuple[int, int, int]]] = [[] for _ in range(W)]
    retracts: List[List[int]] = [[] for _ in range(W)]
    if r.rec_off and W:
        off = _np(r.rec_off, W + 1, np.uint32)
        n = int(off[-1])
        if n and r.rec_delta16:  # HQTICK_FLAG_COMPACT_DELTA16: 16-bit differences + runs that carry their first low id
            t, v, k = expand_delta16(r, W, off)
        elif n and r.rec_task_lo:  # compact emission (HQTICK_FLAG_COMPACT_RECORDS): u32 low halves + runs of (job, variant, kind)
            t, v, k = expand_compact(r, W, off)
        else:
            t, v, k = _np(r.rec_task, n, np.uint64).tolist(), _np(r.rec_variant, n, np.uint8).tolist(), _np(r.rec_kind, n, np.uint8).tolist()
        for w in range(W):
            a, b = int(off[w]), int(off[w + 1])
            records[w] = list(zip(t[a:b], v[a:b], k[a:b]))
    if r.retract_off and W:
        off = _np(r.retract_off, W + 1, np.uint32)
        t = _np(r.retract_task, int(off[-1]), np.uint64).tolist()
        for w in range(W):
            retracts[w] = t[int(off[w]) : int(off[w + 1])]
    nr = r.n_redirects
    redirects = list(zip(_np(r.redirect_task, nr, np.uint64).tolist(), _np(r.redirect_worker, nr, np.uint32).tolist(), _np(r.redirect_variant, nr, np.uint8).tolist()))
    mn = []
    if r.n_mn:
        mt = _np(r.mn_task, r.n_mn, np.uint64).tolist()
        mo = _np(r.mn_worker_off, r.n_mn + 1, np.uint32)
        mw = _np(r.mn_worker, int(mo[-1]), np.uint32).tolist()
        for i in range(r.n_mn):
            mn.append((mt[i], mw[int(mo[i]) : int(mo[i + 1])]))
    nf = _np(r.new_free, W * n_resources, np.uint64).reshape(W, n_resources) if r.new_free else np.zeros((W, n_resources), np.uint64)
    times = dict(total=r.t_total_us, scan=r.t_scan_us, batches=r.t_batches_us, solve=r.t_solve_us, mapping=r.t_mapping_us)
    kinds = _np(r.redirect_kind, nr, np.uint8).tolist() if r.redirect_kind else [HQ_REDIRECT_FROM_PREFILL] * nr
    return Result(r.status, bool(r.is_optimal), batches, counts, records, retracts, redirects, mn, nf, times, kinds, bool(r.is_canonical))
