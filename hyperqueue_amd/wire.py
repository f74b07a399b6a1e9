"""ctypes binding of include/hqwire.h: the worker-message wire encoding of a tick.

`WireTables` / `WireRecords` flatten the host's view -- task attributes, interned `TaskConfiguration`s, the tick's mapping -- into the SoA
arrays of the ABI.  `encode_device` places them in HBM (torch tensors, plumbing only) and runs the three kernels of
`hqwire_encode_device`; `encode_host_debug` hands host arrays to `hqwire_debug_encode_host`, which executes the same phase functions on
the CPU (tests only).  There is no Python encoder here: the oracle lives in `oracle/wire_oracle.py` and is imported by tests only.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import tick

HQWIRE_ABI_VERSION = 2
HQWIRE_MAX_FRAGMENTS = 16
HQWIRE_MAX_RECORDS = 2048
SLOT_OK, SLOT_OVERSIZE, SLOT_UNKNOWN, SLOT_TOO_MANY = 0, 1, 2, 3
HQWIRE_OK, HQWIRE_CAPACITY = 0, 1

_vp = C.c_void_p


class TablesC(C.Structure):
    _fields_ = [("n_tasks", C.c_uint64), ("task_id", _vp), ("task_rq", _vp), ("task_instance", _vp), ("task_priority", _vp), ("task_config", _vp),
                ("entry_some", _vp), ("entry_off", _vp), ("entry_blob", _vp), ("n_configs", C.c_uint32), ("config_time_some", _vp),
                ("config_time_secs", _vp), ("config_time_nanos", _vp), ("body_off", _vp), ("body_blob", _vp)]


class RecordsC(C.Structure):
    _fields_ = [("n_workers", C.c_uint32), ("n_records", C.c_uint32), ("worker_id", _vp), ("rec_off", _vp), ("rec_task", _vp), ("rec_variant", _vp),
                ("rec_kind", _vp), ("retract_off", _vp), ("retract_task", _vp), ("n_mn", C.c_uint32), ("mn_task", _vp), ("mn_worker_off", _vp),
                ("mn_worker", _vp)]


class OutputC(C.Structure):
    _fields_ = [("bytes", _vp), ("capacity", C.c_uint64), ("slot_off", _vp), ("slot_status", _vp), ("header", _vp), ("scratch", _vp),
                ("scratch_bytes", C.c_uint64), ("slot_nfrag", _vp), ("frag_end", _vp), ("msg_size_limit", C.c_uint64)]


SYMBOLS = ["hqwire_scratch_bytes", "hqwire_encode_device", "hqwire_abi_version"]


def load() -> C.CDLL:
    lib = tick.load()
    lib.hqwire_scratch_bytes.argtypes = [C.c_uint64, C.c_uint64]
    lib.hqwire_scratch_bytes.restype = C.c_uint64
    lib.hqwire_encode_device.argtypes = [C.POINTER(TablesC), C.POINTER(RecordsC), C.POINTER(OutputC), _vp]
    lib.hqwire_abi_version.restype = C.c_uint32
    return lib


@dataclass
class WireTables:
    """Arrays of `hqwire_tables` (numpy, host)."""
    task_id: np.ndarray
    task_rq: np.ndarray
    task_instance: np.ndarray
    task_priority: np.ndarray
    task_config: np.ndarray
    entry_some: np.ndarray
    entry_off: np.ndarray
    entry_blob: np.ndarray
    config_time_some: np.ndarray
    config_time_secs: np.ndarray
    config_time_nanos: np.ndarray
    body_off: np.ndarray
    body_blob: np.ndarray

    @staticmethod
    def build(attrs: Dict[int, Tuple[int, int, int, int, Optional[bytes]]], configs: Sequence[Tuple[Optional[Tuple[int, int]], bytes]]) -> "WireTables":
        """attrs: task id -> (rq, instance_id, priority, config index, entry or None); configs: [(time_limit (secs, nanos) or None, body)]."""
        ids = sorted(attrs)
        entry_off, blob = [0], bytearray()
        for t in ids:
            e = attrs[t][4]
            if e is not None:
                blob += e
            entry_off.append(len(blob))
        body_off, bodies = [0], bytearray()
        for (_, body) in configs:
            bodies += body
            body_off.append(len(bodies))
        u8 = lambda b: np.frombuffer(bytes(b) or b"\0", np.uint8).copy()
        return WireTables(
            np.array(ids, np.uint64), np.array([attrs[t][0] for t in ids], np.uint32), np.array([attrs[t][1] for t in ids], np.uint32),
            np.array([attrs[t][2] for t in ids], np.uint64), np.array([attrs[t][3] for t in ids], np.uint32),
            np.array([attrs[t][4] is not None for t in ids], np.uint8), np.array(entry_off, np.uint64), u8(blob),
            np.array([c[0] is not None for c in configs], np.uint8), np.array([c[0][0] if c[0] else 0 for c in configs], np.uint64),
            np.array([c[0][1] if c[0] else 0 for c in configs], np.uint32), np.array(body_off, np.uint64), u8(bodies))

    @property
    def n_tasks(self) -> int:
        return len(self.task_id)

    @property
    def n_configs(self) -> int:
        return len(self.config_time_some)

    def arrays(self) -> List[np.ndarray]:
        return [self.task_id, self.task_rq, self.task_instance, self.task_priority, self.task_config, self.entry_some, self.entry_off, self.entry_blob,
                self.config_time_some, self.config_time_secs, self.config_time_nanos, self.body_off, self.body_blob]


@dataclass
class WireRecords:
    """Arrays of `hqwire_records` (numpy, host) -- the layout of `hqtick_result` / the record sink."""
    worker_id: np.ndarray
    rec_off: np.ndarray
    rec_task: np.ndarray
    rec_variant: np.ndarray
    rec_kind: np.ndarray
    retract_off: np.ndarray
    retract_task: np.ndarray
    mn_task: np.ndarray
    mn_worker_off: np.ndarray
    mn_worker: np.ndarray

    @staticmethod
    def build(worker_ids: Sequence[int], records: Sequence[Sequence[Tuple[int, int, int]]], retracts: Sequence[Sequence[int]],
              mn: Sequence[Tuple[int, Sequence[int]]] = ()) -> "WireRecords":
        rec_off, ret_off, mn_off = [0], [0], [0]
        for r in records:
            rec_off.append(rec_off[-1] + len(r))
        for r in retracts:
            ret_off.append(ret_off[-1] + len(r))
        for (_, ws) in mn:
            mn_off.append(mn_off[-1] + len(ws))
        flat = [x for r in records for x in r]
        return WireRecords(
            np.array(worker_ids, np.uint32), np.array(rec_off, np.uint32), np.array([x[0] for x in flat], np.uint64),
            np.array([x[1] & 0xFF for x in flat], np.uint8), np.array([x[2] for x in flat], np.uint8), np.array(ret_off, np.uint32),
            np.array([t for r in retracts for t in r], np.uint64), np.array([t for (t, _) in mn], np.uint64), np.array(mn_off, np.uint32),
            np.array([w for (_, ws) in mn for w in ws], np.uint32))

    @property
    def n_workers(self) -> int:
        return len(self.worker_id)

    @property
    def n_records(self) -> int:
        return int(self.rec_off[-1])

    @property
    def n_mn(self) -> int:
        return len(self.mn_task)

    def arrays(self) -> List[np.ndarray]:
        return [self.worker_id, self.rec_off, self.rec_task, self.rec_variant, self.rec_kind, self.retract_off, self.retract_task, self.mn_task,
                self.mn_worker_off, self.mn_worker]


@dataclass
class WireResult:
    status: int                    # HQWIRE_OK / HQWIRE_CAPACITY
    total_bytes: int
    slot_status: np.ndarray        # [n_slots]
    slot_off: np.ndarray           # [2 * n_slots + 1]
    data: bytes
    slot_nfrag: Optional[np.ndarray] = None  # [n_slots] ComputeTasks messages per slot (fragmentation, hqwire ABI 2)
    frag_end: Optional[np.ndarray] = None    # [n_slots * HQWIRE_MAX_FRAGMENTS]

    def messages(self, records: WireRecords) -> List[Tuple[int, bytes]]:
        """[(worker id, message bytes)] in the order `send_messages` emits them (mapping.rs:259-292); slots the device did not build
        (slot_status != 0) contribute their RetractTasks message only."""
        out = []
        W = records.n_workers
        for s in range(len(self.slot_status)):
            wid = int(records.worker_id[s]) if s < W else int(records.worker_id[records.mn_worker[records.mn_worker_off[s - W]]])
            lo, hi = int(self.slot_off[2 * s]), int(self.slot_off[2 * s + 1])
            if hi > lo:
                out.append((wid, self.data[lo:hi]))  # RetractTasks
            lo, hi = int(self.slot_off[2 * s + 1]), int(self.slot_off[2 * s + 2])
            if hi > lo:
                nf = int(self.slot_nfrag[s]) if self.slot_nfrag is not None else 1
                for f in range(max(nf, 1)):  # a worker's ComputeTasks part: one message, or the fragments the builder's size limit cuts it into
                    end = int(self.frag_end[s * HQWIRE_MAX_FRAGMENTS + f]) if (self.frag_end is not None and nf >= 1) else hi
                    out.append((wid, self.data[lo:end]))
                    lo = end
                assert lo == hi, (s, lo, hi)
        return out


def _padded(a: np.ndarray) -> np.ndarray:
    return a if a.size else np.zeros(1, a.dtype)  # never hand a NULL pointer for an empty array


def _structs(t: WireTables, r: WireRecords, ptrs_t: List[int], ptrs_r: List[int]):
    tc = TablesC(t.n_tasks, *ptrs_t[:8], t.n_configs, *ptrs_t[8:])
    rc = RecordsC(r.n_workers, r.n_records, *ptrs_r[:7], r.n_mn, *ptrs_r[7:])
    return tc, rc


def encode_host_debug(t: WireTables, r: WireRecords, capacity: int, order: int = 0, limit: int = 0, fragments: bool = True) -> WireResult:
    """`hqwire_debug_encode_host_order` of libhqtick_test.so: the kernels' phase functions on the CPU (tests only; the product library has no such
    entry point); `order` = sequence of the emulated threads."""
    from . import _testhooks

    lib = _testhooks.load()
    lib.hqwire_scratch_bytes.argtypes = [C.c_uint64, C.c_uint64]
    lib.hqwire_scratch_bytes.restype = C.c_uint64
    lib.hqwire_debug_encode_host_order.argtypes = [C.POINTER(TablesC), C.POINTER(RecordsC), C.POINTER(OutputC), C.c_int]
    ta, ra = [_padded(np.ascontiguousarray(a)) for a in t.arrays()], [_padded(np.ascontiguousarray(a)) for a in r.arrays()]
    tc, rc = _structs(t, r, [a.ctypes.data for a in ta], [a.ctypes.data for a in ra])
    S = r.n_workers + r.n_mn
    data, slot_off, status, header = np.zeros(max(1, capacity), np.uint8), np.zeros(2 * S + 1, np.uint64), np.zeros(max(1, S), np.uint8), np.zeros(4, np.uint32)
    scratch = np.zeros(int(lib.hqwire_scratch_bytes(r.n_records + r.n_mn, S)) // 8 + 1, np.uint64)
    nfrag, frag_end = np.zeros(max(1, S), np.uint32), np.zeros(max(1, S) * HQWIRE_MAX_FRAGMENTS, np.uint64)
    oc = OutputC(data.ctypes.data, capacity, slot_off.ctypes.data, status.ctypes.data, header.ctypes.data, scratch.ctypes.data, scratch.nbytes,
                 nfrag.ctypes.data if fragments else None, frag_end.ctypes.data if fragments else None, limit)
    rc_ = lib.hqwire_debug_encode_host_order(C.byref(tc), C.byref(rc), C.byref(oc), order)
    if rc_ != 0:
        raise tick.HqTickError(rc_, "hqwire_debug_encode_host_order")
    total = int(header[2]) | int(header[3]) << 32
    return WireResult(int(header[0]), total, status[:S].copy(), slot_off, data[:total].tobytes() if header[0] == HQWIRE_OK else b"",
                      nfrag[:S].copy() if fragments else None, frag_end if fragments else None)


def _run_device(lib, torch, dev, tc: TablesC, rc: RecordsC, n_slots: int, n_rec_incl_mn: int, capacity: int, limit: int = 0) -> WireResult:
    zeros = lambda n: torch.zeros(max(8, int(n)), dtype=torch.uint8, device=dev)
    data, slot_off, status, header = zeros(capacity), zeros(8 * (2 * n_slots + 1)), zeros(n_slots), zeros(16)
    scratch = zeros(int(lib.hqwire_scratch_bytes(n_rec_incl_mn, n_slots)) + 8)
    nfrag, frag_end = zeros(4 * max(1, n_slots)), zeros(8 * max(1, n_slots) * HQWIRE_MAX_FRAGMENTS)
    oc = OutputC(data.data_ptr(), capacity, slot_off.data_ptr(), status.data_ptr(), header.data_ptr(), scratch.data_ptr(), scratch.numel(),
                 nfrag.data_ptr(), frag_end.data_ptr(), limit)
    stream = torch.cuda.current_stream(dev).cuda_stream
    rc_ = lib.hqwire_encode_device(C.byref(tc), C.byref(rc), C.byref(oc), _vp(stream))
    if rc_ != 0:
        raise tick.HqTickError(rc_, "hqwire_encode_device")
    torch.cuda.synchronize(dev)
    h = header.cpu().numpy().view(np.uint32)
    total = int(h[2]) | int(h[3]) << 32
    return WireResult(int(h[0]), total, status.cpu().numpy()[:n_slots].copy(), slot_off.cpu().numpy().view(np.uint64)[: 2 * n_slots + 1].copy(),
                      data[:total].cpu().numpy().tobytes() if h[0] == HQWIRE_OK else b"",
                      nfrag.cpu().numpy().view(np.uint32)[:n_slots].copy(), frag_end.cpu().numpy().view(np.uint64).copy())


def _upload(torch, dev, arrays: List[np.ndarray]):
    return [torch.from_numpy(_padded(np.ascontiguousarray(a)).view(np.uint8).copy()).to(dev) for a in arrays]


def encode_device(t: WireTables, r: WireRecords, capacity: int, device: str = "cuda:0", limit: int = 0) -> WireResult:
    """`hqwire_encode_device`: tables and records in HBM, three kernels on torch's current stream, result copied back for inspection."""
    import torch

    lib, dev = load(), torch.device(device)
    tt, rt = _upload(torch, dev, t.arrays()), _upload(torch, dev, r.arrays())
    tc, rc = _structs(t, r, [x.data_ptr() for x in tt], [x.data_ptr() for x in rt])
    return _run_device(lib, torch, dev, tc, rc, r.n_workers + r.n_mn, r.n_records + r.n_mn, capacity, limit)


def encode_from_sink(t: WireTables, sink, n_workers: int, sink_cap_records: int, n_records: int, side: WireRecords, capacity: int) -> WireResult:
    """The chained path of DESIGN.md 8d: the tick left its records in a device record sink (`hqtick_set_record_sink`; `sink` = that torch
    tensor), the encoder reads `rec_off / task / variant / kind` right there.  `side` carries what the sink does not hold -- worker ids,
    retract and multi-node CSRs of the tick's result (a few entries, uploaded here); its own rec_* arrays are ignored."""
    import torch

    from .sharded import sink_layout

    lib, dev = load(), sink.device
    # the library derives the record capacity from the sink's byte size (hqtick.cpp: hqtick_sink_capacity_records); the array offsets follow from it
    lib.hqtick_sink_capacity_records.restype = C.c_uint32
    lib.hqtick_sink_capacity_records.argtypes = [C.c_uint32, C.c_size_t]
    cap = int(lib.hqtick_sink_capacity_records(n_workers, sink.numel()))
    if sink_cap_records and cap != sink_cap_records:
        raise ValueError(f"sink of {sink.numel()} bytes holds {cap} records, caller assumed {sink_cap_records}")
    o_off, o_task, o_var, o_kind, _total = sink_layout(n_workers, cap)
    tt, st = _upload(torch, dev, t.arrays()), _upload(torch, dev, side.arrays())
    base = sink.data_ptr()
    ptrs_r = [st[0].data_ptr(), base + o_off, base + o_task, base + o_var, base + o_kind, st[5].data_ptr(), st[6].data_ptr(), st[7].data_ptr(),
              st[8].data_ptr(), st[9].data_ptr()]
    tc = TablesC(t.n_tasks, *[x.data_ptr() for x in tt[:8]], t.n_configs, *[x.data_ptr() for x in tt[8:]])
    rc = RecordsC(n_workers, n_records, *ptrs_r[:7], side.n_mn, *ptrs_r[7:])
    return _run_device(lib, torch, dev, tc, rc, n_workers + side.n_mn, n_records + side.n_mn, capacity)


# ---- worker-sharded ticks: every rank encodes the messages of ITS workers, one all-gather merges the byte buffers -----------------
def shard_records(r: WireRecords, rank: int, world: int) -> WireRecords:
    """The mapping as rank `rank` of a worker-sharded tick holds it: records and retracts of its own workers only
    (owner = FxHash(worker_id) % world, hqtick_set_shard); multi-node messages are rank 0's."""
    from .sharded import owner_of

    mine = [owner_of(int(w), world) == rank for w in r.worker_id]
    recs = [[(int(r.rec_task[i]), int(r.rec_variant[i]), int(r.rec_kind[i])) for i in range(int(r.rec_off[w]), int(r.rec_off[w + 1]))] if mine[w] else []
            for w in range(r.n_workers)]
    rets = [[int(t) for t in r.retract_task[int(r.retract_off[w]):int(r.retract_off[w + 1])]] if mine[w] else [] for w in range(r.n_workers)]
    mn = [(int(r.mn_task[k]), [int(x) for x in r.mn_worker[int(r.mn_worker_off[k]):int(r.mn_worker_off[k + 1])]]) for k in range(r.n_mn)] if rank == 0 else []
    out = WireRecords.build([int(w) for w in r.worker_id], recs, rets, mn)
    return out


def pack_wire_shard(res: WireResult, n_slots: int, capacity: int) -> np.ndarray:
    """Fixed-size shard buffer for the all-gather: u64 total | u64 slot_off[2 * n_slots + 1] | bytes[capacity]."""
    if res.status != HQWIRE_OK or res.total_bytes > capacity:
        raise ValueError(f"wire shard does not fit: {res.total_bytes} bytes, capacity {capacity}")
    if res.slot_nfrag is not None and (res.slot_nfrag > 1).any():
        raise ValueError("a fragmented slot (> 32 MiB for one worker): this exchange format carries one ComputeTasks message per slot")
    head = np.full(2 * n_slots + 2, res.total_bytes, np.uint64)  # slots this rank does not have (padding up to the widest rank) stay empty ranges
    head[0] = res.total_bytes
    head[1:1 + len(res.slot_off)] = res.slot_off
    body = np.zeros(capacity, np.uint8)
    body[: res.total_bytes] = np.frombuffer(res.data, np.uint8)
    return np.concatenate([head.view(np.uint8), body])


def merge_wire_shards(merged: np.ndarray, world: int, n_slots_per_rank: Sequence[int], capacity: int, worker_id_of_slot) -> List[Tuple[int, bytes]]:
    """[(worker id, message bytes)] from the all-gathered shard buffers, slot by slot (worker-index order, multi-node slots last; of every
    slot only its owner holds bytes).  `worker_id_of_slot(rank, slot)` names the receiver."""
    out: List[Tuple[int, bytes]] = []
    views, pos = [], 0
    for rnk in range(world):
        S = n_slots_per_rank[rnk]
        size = 8 * (2 * S + 2) + capacity
        buf = merged[pos:pos + size]
        pos += size
        head = buf[: 8 * (2 * S + 2)].view(np.uint64)
        views.append((S, head[1:], buf[8 * (2 * S + 2):]))
    for s in range(max(n_slots_per_rank)):
        for rnk in range(world):
            S, off, body = views[rnk]
            if s >= S:
                continue
            for j in (2 * s, 2 * s + 1):
                lo, hi = int(off[j]), int(off[j + 1])
                if hi > lo:
                    out.append((worker_id_of_slot(rnk, s), body[lo:hi].tobytes()))
    return out


def all_gather_messages(res: WireResult, records: WireRecords, world: int, capacity: int, group=None) -> List[Tuple[int, bytes]]:
    """One `all_gather` of the ranks' shard buffers (RCCL on the GPUs; gloo in the CPU tests), then the merge: every rank ends up with the
    tick's full message list.  Ranks agree on n_workers; rank 0 may carry extra multi-node slots."""
    import torch
    import torch.distributed as dist

    W = records.n_workers
    n_slots = len(res.slot_status)
    counts = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([n_slots], dtype=torch.int64), group=group)
    slots = [max(int(c.item()) for c in counts)] * world  # equal-sized buffers: pad to the widest rank (rank 0 carries the multi-node slots)
    mine = torch.from_numpy(pack_wire_shard(res, slots[0], capacity))
    bufs = [torch.zeros(8 * (2 * S + 2) + capacity, dtype=torch.uint8) for S in slots]
    dist.all_gather(bufs, mine, group=group)
    merged = np.concatenate([b.numpy() for b in bufs])
    mn_root = {}  # multi-node slots exist on rank 0 only; every rank learns their receivers from rank 0's records
    obj = [[int(records.worker_id[records.mn_worker[records.mn_worker_off[k]]]) for k in range(records.n_mn)] if dist.get_rank(group) == 0 else None]
    dist.broadcast_object_list(obj, src=0, group=group)
    roots = obj[0]

    def receiver(rnk, s):
        return int(records.worker_id[s]) if s < W else roots[s - W]

    return merge_wire_shards(merged, world, slots, capacity, receiver)
