"""CPU oracle of the worker-message wire encoding -- TEST INFRASTRUCTURE ONLY.

What the reference does between `WorkerTaskMapping::send_messages` and the socket, restated in plain Python
(paths relative to /root/reference/crates/tako/src/internal/):

    scheduler/mapping.rs:259-292   per worker: RetractTasks, then ComputeTasks built from prefills (variant None) and assigned
                                   tasks (Some(v)); per multi-node task one single-task message to the root worker
    server/task.rs:315-445         ComputeTasksBuilder: shared data (time_limit, body) deduplicated per message by configuration,
                                   `estimated_size` bookkeeping, a message is cut as soon as the estimate exceeds MAX_FRAME_SIZE / 4
    messages/worker.rs:27-52,76-88 ComputeTaskSeparateData / ComputeTaskSharedData / ComputeTasksMsg / TaskIdsMsg / ToWorkerMessage
    transfer/auth.rs:253-263       bincode `DefaultOptions::new().with_fixint_encoding()`

Third-party: bincode 1.3.3 and serde derive are not under /root/reference.  Their published format, as used here: little endian;
fixed-width integers; `usize` as u64; sequences (Vec, ThinVec, Rc<[u8]>) as a u64 length followed by the elements; `Option` as one
tag byte 0/1 followed by the value; enum variants as a u32 index in declaration order; structs as their fields in declaration order;
newtype structs (`JobId(u32)`, `Priority(u64)`, ...) as the inner value; `Duration` as `secs: u64, nanos: u32`.  No reference test holds
wire bytes for these messages, so the byte layout is **parity unpinned** beyond that specification; the message *structure*
(which tasks, which order, which variants) is pinned by the tick's golden tests.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

MAX_FRAME_SIZE = 128 * 1024 * 1024      # crates/tako/src/lib.rs:31
MAX_TASK_MSG_SIZE = MAX_FRAME_SIZE // 4  # server/task.rs:315
TAG_COMPUTE_TASKS, TAG_RETRACT_TASKS = 0, 1  # ToWorkerMessage variant indices (messages/worker.rs:76-79)


@dataclass
class Config:
    """TaskConfiguration as far as the message sees it (server/task.rs:95-101, :347-352)."""
    time_limit: Optional[Tuple[int, int]]  # (secs, nanos)
    body: bytes


@dataclass
class TaskAttr:
    rq: int
    instance_id: int
    priority: int           # Task::priority() = Priority::from_user_priority (server/task.rs:175-177)
    config: int             # index into the configuration table; equal configurations share an index (the builder's Map key)
    entry: Optional[bytes]  # Task::entry


def u32(v: int) -> bytes:
    return struct.pack("<I", v)


def u64(v: int) -> bytes:
    return struct.pack("<Q", v)


def enc_task_id(packed: int) -> bytes:
    """TaskId { job_id: u32, job_task_id: u32 } (common/ids.rs:17-21); packed = job << 32 | task."""
    return u32(packed >> 32) + u32(packed & 0xFFFFFFFF)


def enc_separate(shared_index: int, task: int, a: TaskAttr, variant: Optional[int], node_list: Sequence[int]) -> bytes:
    """ComputeTaskSeparateData (messages/worker.rs:27-39), field by field."""
    out = u64(shared_index) + enc_task_id(task) + u32(a.rq)
    out += b"\x00" if variant is None else b"\x01" + bytes([variant])
    out += u32(a.instance_id) + u64(a.priority)
    out += u64(len(node_list)) + b"".join(u32(w) for w in node_list)
    out += b"\x00" if a.entry is None else b"\x01" + u64(len(a.entry)) + a.entry
    return out


def enc_shared(c: Config) -> bytes:
    """ComputeTaskSharedData (messages/worker.rs:41-45)."""
    out = b"\x00" if c.time_limit is None else b"\x01" + u64(c.time_limit[0]) + u32(c.time_limit[1])
    return out + u64(len(c.body)) + c.body


def estimate_task(a: TaskAttr, node_list: Sequence[int]) -> int:
    """estimate_task_data_size (server/task.rs:405-427): size_of_val of the fields, Option<ResourceVariantId> = 2 bytes."""
    return 8 + 8 + 4 + 2 + 4 + 8 + 4 * len(node_list) + (len(a.entry) if a.entry is not None else 0)


def estimate_shared(c: Config) -> int:
    """estimate_shared_data_size (server/task.rs:430-433): size_of::<Option<Duration>>() = 16."""
    return 16 + len(c.body)


class ComputeTasksBuilder:
    """server/task.rs:321-402."""

    def __init__(self, configs: Sequence[Config], limit: int = MAX_TASK_MSG_SIZE):
        self.configs, self.limit = configs, limit
        self.tasks: List[bytes] = []
        self.index: Dict[int, int] = {}
        self.shared: List[bytes] = []
        self.estimated = 0

    def _message(self) -> bytes:
        return u32(TAG_COMPUTE_TASKS) + u64(len(self.tasks)) + b"".join(self.tasks) + u64(len(self.shared)) + b"".join(self.shared)

    def add_task(self, task: int, a: TaskAttr, variant: Optional[int], node_list: Sequence[int]) -> Optional[bytes]:
        if a.config not in self.index:
            self.index[a.config] = len(self.shared)
            self.estimated += estimate_shared(self.configs[a.config])
            self.shared.append(enc_shared(self.configs[a.config]))
        self.tasks.append(enc_separate(self.index[a.config], task, a, variant, node_list))
        self.estimated += estimate_task(a, node_list)
        if self.estimated > self.limit:  # create_message_on_overflow
            msg = self._message()
            self.tasks, self.shared, self.index, self.estimated = [], [], {}, 0
            return msg
        return None

    def into_last_message(self) -> Optional[bytes]:
        return self._message() if self.tasks else None


def retract_message(ids: Sequence[int]) -> bytes:
    """ToWorkerMessage::RetractTasks(TaskIdsMsg { ids }) (messages/worker.rs:54-57,78)."""
    return u32(TAG_RETRACT_TASKS) + u64(len(ids)) + b"".join(enc_task_id(t) for t in ids)


def send_messages(attrs: Dict[int, TaskAttr], configs: Sequence[Config], worker_ids: Sequence[int],
                  records: Sequence[Sequence[Tuple[int, int, int]]], retracts: Sequence[Sequence[int]],
                  mn: Sequence[Tuple[int, Sequence[int]]] = (), limit: int = MAX_TASK_MSG_SIZE) -> List[Tuple[int, bytes]]:
    """scheduler/mapping.rs:259-292 over the tick's result: `records[w]` = [(task, variant, kind)] in send order (kind 0 = prefill ->
    variant None), `retracts[w]` = task ids, `mn` = [(task, [worker index, root first])].  Returns [(worker id, message bytes)] in the
    order of the calls to `send_worker_message`, workers in the order given (the reference walks its worker map)."""
    out: List[Tuple[int, bytes]] = []
    for w, wid in enumerate(worker_ids):
        if retracts[w]:
            out.append((wid, retract_message(retracts[w])))
        b = ComputeTasksBuilder(configs, limit)
        for (task, variant, kind) in records[w]:
            msg = b.add_task(task, attrs[task], None if kind == 0 else variant, [])
            if msg is not None:
                out.append((wid, msg))
        msg = b.into_last_message()
        if msg is not None:
            out.append((wid, msg))
    for (task, ws) in mn:
        b = ComputeTasksBuilder(configs, limit)
        nodes = [worker_ids[i] for i in ws]
        msg = b.add_task(task, attrs[task], 0, nodes) or b.into_last_message()
        out.append((nodes[0], msg))
    return out
