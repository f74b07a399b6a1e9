"""Worker-sharded tick across the GPUs of one node (BASELINE.json north_star).

One process per GPU (`torch.distributed`, backend "nccl" == RCCL over xGMI).  Every rank runs the tick on the SAME
snapshot: the ready-set scans, the batches and the placement are replicated and deterministic, so no data has to be
exchanged before the mapping stage.  Each rank expands and emits the records of the workers it owns,

    owner(worker) = FxHash(worker_id) % world_size                      (hbmap.hash_u32, the hash tako's Map uses)

into a fixed-capacity device buffer (the "record sink", include/hqtick.h), and ONE all-gather of those buffers merges the
shards' assignment vectors on every rank.  `ShardedTick.tick()` returns the merged result in the same `abi.Result` shape as
the single-GPU path.

The collective plumbing is backend-agnostic so that it can be tested on CPU (gloo, world_size 2) with a stand-in tick:
`tick_fn(snapshot) -> abi.Result` produces the FULL result, of which this rank packs only its own shard.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, List, Optional, Tuple

import numpy as np

from . import abi
from .hbmap import hash_u32

SINK_MAGIC = 0x48515354


def owner_of(worker_id: int, world: int) -> int:
    return hash_u32(int(worker_id)) % world if world > 1 else 0


def sink_layout(n_workers: int, cap: int) -> Tuple[int, int, int, int, int]:
    """(off_off, off_task, off_variant, off_kind, total bytes) — must match hqtick_sink_bytes() in csrc/hqtick.cpp."""
    o_off = 16
    o_task = (o_off + (n_workers + 1) * 4 + 7) & ~7
    o_var = o_task + cap * 8
    o_kind = o_var + cap
    total = (o_task + cap * 10 + 15) & ~15
    return o_off, o_task, o_var, o_kind, total


class ShardDivergence(RuntimeError):
    """the ranks' replicated placements differ (placement_checksum in the sink headers): a tick that ran into its time limit"""


def placement_checksum(res: abi.Result) -> int:
    """FNV-1a over the counts, the multi-node sets and is_optimal — hqhost::Counts::checksum (csrc/host_model.h)."""
    h = 2166136261

    def mix(v):
        nonlocal h
        for i in range(4):
            h = ((h ^ ((v >> (8 * i)) & 0xFF)) * 16777619) & 0xFFFFFFFF

    mix(1 if res.is_optimal else 0)
    last = None
    for (rq, variant, worker, count) in res.counts:  # grouped by (rq, variant) in iteration order
        if (rq, variant) != last:
            mix(rq); mix(variant); last = (rq, variant)
        mix(worker); mix(count)
    for task, workers in res.mn:  # stand-in for the library's per-request multi-node sets: any deterministic function of them will do here
        mix(0xFFFFFFFF); mix(task & 0xFFFFFFFF)
        for w in workers:
            mix(w)
    return h


def pack_shard(res: abi.Result, worker_ids, rank: int, world: int, cap: int, checksum: Optional[int] = None) -> np.ndarray:
    """CPU stand-in for what K5b + the sink header copy produce on the GPU: this rank's records in sink layout."""
    W = len(worker_ids)
    o_off, o_task, o_var, o_kind, total = sink_layout(W, cap)
    buf = np.zeros(total, np.uint8)
    off = np.zeros(W + 1, np.uint32)
    tasks, variants, kinds = [], [], []
    for w in range(W):
        if owner_of(worker_ids[w], world) == rank:
            for (t, v, k) in res.records[w]:
                tasks.append(t); variants.append(v); kinds.append(k)
        off[w + 1] = len(tasks)
    n = len(tasks)
    if n > cap:
        raise ValueError(f"record sink too small: {n} records, capacity {cap}")
    buf[0:16].view(np.uint32)[:] = [n, placement_checksum(res) if checksum is None else checksum, SINK_MAGIC, cap]
    buf[o_off:o_off + (W + 1) * 4].view(np.uint32)[:] = off
    buf[o_task:o_task + n * 8].view(np.uint64)[:] = np.asarray(tasks, np.uint64)
    buf[o_var:o_var + n] = np.asarray(variants, np.uint8)
    buf[o_kind:o_kind + n] = np.asarray(kinds, np.uint8)
    return buf


def merge_shards(merged: np.ndarray, world: int, n_workers: int, cap: int) -> List[List[Tuple[int, int, int]]]:
    """Per-worker record lists from the all-gathered sinks (rank-major)."""
    o_off, o_task, o_var, o_kind, total = sink_layout(n_workers, cap)
    assert merged.size == world * total, (merged.size, world, total)
    records: List[List[Tuple[int, int, int]]] = [[] for _ in range(n_workers)]
    for r in range(world):
        b = merged[r * total:(r + 1) * total]
        n, chk, magic, c = b[0:16].view(np.uint32).tolist()
        W = n_workers
        if magic != SINK_MAGIC or c != cap:
            raise ValueError(f"shard {r}: bad sink header {(n, chk, hex(magic), c)}")
        if r == 0:
            chk0 = chk
        elif chk != chk0:
            raise ShardDivergence(f"shard {r}: placement checksum {chk:#x} != {chk0:#x} of shard 0")
        off = b[o_off:o_off + (W + 1) * 4].view(np.uint32)
        t = b[o_task:o_task + n * 8].view(np.uint64).tolist()
        v = b[o_var:o_var + n].tolist()
        k = b[o_kind:o_kind + n].tolist()
        for w in range(W):
            a, e = int(off[w]), int(off[w + 1])
            if e > a:
                assert not records[w], f"worker {w} has records in two shards"
                records[w] = list(zip(t[a:e], v[a:e], k[a:e]))
    return records


class ShardedTick:
    """`tick(snapshot)` on every rank -> merged `abi.Result` on every rank.

    backend "hip": libhqtick.so with hqtick_set_shard + a device record sink, merged by one RCCL all-gather.
    backend callable: CPU stand-in (tests): `tick_fn(snapshot) -> abi.Result` (full result), merged by one gloo all-gather.
    """

    def __init__(self, config: Optional[abi.Config] = None, rank: int = 0, world: int = 1, records_per_shard: int = 1 << 18,
                 backend="hip", group=None, collective: str = "auto"):
        import torch

        self.torch = torch
        self.rank, self.world, self.cap, self.group = rank, world, int(records_per_shard), group
        self.cfg = config or abi.make_config()
        self.backend = backend
        self._sink = self._merged = None
        self._sink_workers = -1
        self.n_divergent = 0  # ticks on which the replicas disagreed and rank 0's placement was broadcast
        self.collective = "torch"
        self._fallback_ids = None
        if backend == "hip":
            from .tick import Tick

            self.t = Tick(self.cfg)
            lib = self.t._lib
            lib.hqtick_set_shard.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
            lib.hqtick_set_record_sink.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
            lib.hqtick_sink_bytes.restype = C.c_size_t
            lib.hqtick_sink_bytes.argtypes = [C.c_uint32, C.c_uint32]
            rc = lib.hqtick_set_shard(self.t._ctx, rank, world)
            if rc:
                raise RuntimeError(f"hqtick_set_shard failed: {rc}")
            # The merge collective lives INSIDE the C ABI (hqtick_comm_init / hqtick_shard_allgather: librccl, loaded by the library): a Rust host
            # needs nothing else.  torch.distributed only carries the 128-byte communicator id from rank 0 to the other ranks here — the job any
            # out-of-band channel of the host (its own TCP connections) would do.  collective="torch" keeps the all_gather_into_tensor path.
            # collective="host": the shards are staged through host memory and merged by whatever backend the process group has (gloo) — no RCCL; this is
            # what lets two real ranks share ONE GPU in tests/test_gpu_multi.py (RCCL refuses two ranks on one device).
            if collective == "auto":  # the library's collective whenever this process really is one rank of `world` (else: shards simulated in one process)
                d = torch.distributed
                collective = "library" if (world > 1 and d.is_available() and d.is_initialized() and d.get_world_size(group) == world) else "torch"
            self.collective = collective
            self.comm_world = world if collective == "library" else 0  # ranks of the library's RCCL communicator (0: the collective is not the library's)
            if collective == "library":
                lib.hqtick_comm_unique_id.argtypes = [C.c_void_p]
                lib.hqtick_comm_init.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
                lib.hqtick_shard_allgather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
                uid = (C.c_ubyte * 128)()
                if rank == 0:
                    rc = lib.hqtick_comm_unique_id(uid)
                    if rc:
                        raise RuntimeError(f"hqtick_comm_unique_id failed: {rc}")
                if world > 1:
                    box = [bytes(uid)]
                    torch.distributed.broadcast_object_list(box, src=0, group=group)
                    uid = (C.c_ubyte * 128).from_buffer_copy(box[0])
                rc = lib.hqtick_comm_init(self.t._ctx, uid, rank, world)
                ok = torch.tensor([0 if rc else 1], dtype=torch.int32, device=torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else "cpu")
                if world > 1:  # every rank takes the same path: if the library's communicator did not come up SOMEWHERE, all of them merge through torch.distributed
                    torch.distributed.all_reduce(ok, op=torch.distributed.ReduceOp.MIN, group=group)
                if int(ok.item()) == 0:
                    if world == 1:
                        raise RuntimeError(f"hqtick_comm_init failed: {rc}: {self.t._err()}")
                    import warnings

                    warnings.warn(f"hqtick_comm_init failed on some rank (here: {rc}): the shards are merged by torch.distributed.all_gather_into_tensor instead of the library's RCCL call")
                    self.collective = "torch"; self.comm_world = 0

            if world > 1 and self.collective in ("host", "torch") and torch.distributed.is_available() and torch.distributed.is_initialized():
                self._install_exchange()  # the placement solve is split over the ranks too (include/hqtick.h: hqtick_set_exchange); "library": the RCCL communicator does it

    def _install_exchange(self):
        """the ranks' exchange of small host buffers inside the tick, through the process group this ShardedTick merges with (gloo: CPU tensors; nccl: staged through
        the device) — what a Rust host would do over its own channel"""
        torch, world, group = self.torch, self.world, self.group
        dev = None if self.collective == "host" else torch.device("cuda", self.cfg.device_index)

        def exchange(send, recv, n):
            mine = torch.frombuffer((C.c_ubyte * n).from_address(send), dtype=torch.uint8).clone()
            if dev is not None:
                mine = mine.to(dev)
            parts = [torch.empty_like(mine) for _ in range(world)]
            torch.distributed.all_gather(parts, mine, group=group)
            whole = torch.cat(parts).cpu().contiguous().numpy()
            C.memmove(recv, whole.ctypes.data, n * world)
            return 0

        self.t.set_exchange(exchange)

    def _buffers(self, n_workers: int):
        if self._sink_workers != n_workers:
            total = sink_layout(n_workers, self.cap)[4]
            dev = f"cuda:{self.cfg.device_index}" if self.backend == "hip" else "cpu"
            self._sink = self.torch.zeros(total, dtype=self.torch.uint8, device=dev)
            self._merged = self.torch.zeros(total * self.world, dtype=self.torch.uint8, device=dev)
            self._sink_workers = n_workers
            if self.backend == "hip":
                assert self.t._lib.hqtick_sink_bytes(n_workers, self.cap) == total
                rc = self.t._lib.hqtick_set_record_sink(self.t._ctx, C.c_void_p(self._sink.data_ptr()), C.c_size_t(total))
                if rc:
                    raise RuntimeError(f"hqtick_set_record_sink failed: {rc}")
        return self._sink, self._merged

    def set_capacity(self, records_per_shard: int):
        """another sink capacity (every rank must choose the same: the all-gather moves fixed-size blocks); the buffers are re-made on the next tick"""
        self.cap = int(records_per_shard)
        self._sink_workers = -1

    def upload_ready(self, task_id, task_priority, task_rq):
        self.t.upload_ready(task_id, task_priority, task_rq)

    def tick_local(self, sc: abi.SnapshotC, n_workers: int, resident: bool = False):
        """This rank's shard only, no collective: (local ResultC, this rank's sink as a device tensor)."""
        sink, _ = self._buffers(n_workers)
        return self.t.tick_raw(sc, resident=resident), sink

    def tick_device(self, sc: abi.SnapshotC, n_workers: int, resident: bool = False):
        """The timed part on the GPU: sharded tick + the one all-gather.  Returns (local ResultC, merged device tensor)."""
        sink, merged = self._buffers(n_workers)
        res = self.t.tick_raw(sc, resident=resident)  # returns after this shard's kernels have finished (stream-synchronised)
        if getattr(self, "_corrupt_next_checksum", False):  # test hook (tests/test_gpu_multi.py): this replica pretends it placed differently
            sink[4:8] = sink[4:8] ^ 0xFF
            self._corrupt_next_checksum = False
        if self.collective == "library":  # ncclAllGather on the ctx's stream, inside libhqtick.so; returns when the merged vector is there
            rc = self.t._lib.hqtick_shard_allgather(self.t._ctx, C.c_void_p(merged.data_ptr()), C.c_size_t(merged.numel()))
            if rc:
                raise RuntimeError(f"hqtick_shard_allgather failed: {rc}: {self.t._err()}")
        elif self.collective == "host" and self.world > 1:
            mine = sink.cpu()
            parts = [self.torch.empty_like(mine) for _ in range(self.world)]
            self.torch.distributed.all_gather(parts, mine, group=self.group)
            merged = self.torch.cat(parts)
        elif self.world > 1:
            self.torch.distributed.all_gather_into_tensor(merged, sink, group=self.group)
        else:
            merged.copy_(sink)
        return res, merged

    def tick(self, snap: abi.Snapshot, resident: bool = False) -> abi.Result:
        W = len(snap.worker_id)
        if self.backend == "hip":
            res_c, merged = self.tick_device(snap.to_c(), W, resident)
            out = abi.parse_result(res_c, W, snap.n_resources)  # records empty here: they live in the sink
            host = merged.cpu().numpy()
        else:
            full = self.backend(snap)
            sink, merged = self._buffers(W)
            sink.copy_(self.torch.from_numpy(pack_shard(full, snap.worker_id, self.rank, self.world, self.cap)))
            if self.world > 1:
                self.torch.distributed.all_gather_into_tensor(merged, sink, group=self.group)
            else:
                merged.copy_(sink)
            out = full
            host = merged.numpy()
        self._fallback_ids = None
        try:
            out.records = merge_shards(host, self.world, W, self.cap)
        except ShardDivergence:
            out = self._fallback_from_rank0(snap, resident)
            self.n_divergent += 1
            # what the applied placement handed out: the ranks other than 0 still hold their own (divergent) selection as "last tick"
            self._fallback_ids = np.asarray(sorted([t for recs in out.records for (t, _v, _k) in recs] + [t for (t, _ws) in out.mn]), np.uint64)
        return out

    def consume_last(self):
        """hqtick_ready_consume_last for the sharded scheduler: every replica's resident ready set loses what the APPLIED placement handed out.
        After a divergence that is rank 0's placement, which the other ranks remove by id (their own last selection is not what was applied)."""
        if getattr(self, "_fallback_ids", None) is not None and self.rank != 0:
            self.t.ready_remove(self._fallback_ids)
        else:
            self.t.ready_consume_last()
        self._fallback_ids = None

    def _fallback_from_rank0(self, snap: abi.Snapshot, resident: bool) -> abi.Result:
        """The replicas disagree (a time-limited tick): every rank takes rank 0's placement.  Rank 0 repeats the tick unsharded into a sink that
        holds every worker's records and broadcasts it — one more collective, only on this path."""
        import pickle

        dist = self.torch.distributed
        W = len(snap.worker_id)
        if self.backend == "hip":
            lib = self.t._lib
            full_cap = self.cap * self.world
            total = sink_layout(W, full_cap)[4]
            dev = f"cuda:{self.cfg.device_index}"
            big = self.torch.zeros(total, dtype=self.torch.uint8, device=dev)
            meta = [None]
            if self.rank == 0:
                lib.hqtick_set_shard(self.t._ctx, 0, 1)
                lib.hqtick_set_record_sink(self.t._ctx, C.c_void_p(big.data_ptr()), C.c_size_t(total))
                try:
                    res_c = self.t.tick_raw(snap.to_c(), resident=resident)
                    out = abi.parse_result(res_c, W, snap.n_resources)
                finally:
                    lib.hqtick_set_shard(self.t._ctx, self.rank, self.world)
                    self._sink_workers = -1  # the per-shard sink is re-attached by the next tick
                meta[0] = pickle.dumps((out.status, out.is_optimal, out.batches, out.counts, out.retracts, out.redirects, out.mn, out.new_free, out.times_us, out.redirect_kinds, out.is_canonical))
            if self.collective == "host":
                big_h = big.cpu()
                dist.broadcast(big_h, src=0, group=self.group)
                big = big_h
            else:
                dist.broadcast(big, src=0, group=self.group)
            dist.broadcast_object_list(meta, src=0, group=self.group)
            st, opt, batches, counts, retracts, redirects, mn, nf, times, kinds, canon = pickle.loads(meta[0])
            out = abi.Result(st, opt, batches, counts, [[] for _ in range(W)], retracts, redirects, mn, nf, times, kinds, canon)
            out.records = merge_shards(big.cpu().numpy(), 1, W, full_cap)
            return out
        box = [self.backend(snap) if self.rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=self.group)
        return box[0]
