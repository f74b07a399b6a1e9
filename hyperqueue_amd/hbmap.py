"""Host-side emulation of the iteration order of tako's `Map`/`Set` (hashbrown + FxHash).

The reference keeps `worker_map: Map<WorkerId, Worker>` and `TaskQueue::prefill: Set<TaskId>`
(crates/tako/src/internal/server/workermap.rs:9, scheduler/taskqueue.rs:118) and the tick iterates both
(scheduler/mapping.rs:179, scheduler/taskqueue.rs:381-388).  A Rust host hands the real order over the ABI
(`worker_map_rank`, `prefill_task`); this Python mirror of the host has to reproduce it, so it restates the
published SwissTable algorithm (hashbrown 0.17: 16-wide groups on x86-64, triangular probing, tombstones,
4->8->16->next_pow2(cap*8/7) growth) and fxhash 0.2.1 (`h = (rotl(h,5) ^ w) * 0x517cc1b727220a95`).
Small sets only (pure Python).
"""
from __future__ import annotations

from typing import Callable, Iterable, Iterator, List

MASK64 = (1 << 64) - 1
FX_SEED = 0x517CC1B727220A95
WIDTH = 16
EMPTY, DELETED = 0xFF, 0x80


def _fx(h: int, w: int) -> int:
    h = ((h << 5) | (h >> 59)) & MASK64
    return ((h ^ w) * FX_SEED) & MASK64


def hash_u32(v: int) -> int:
    return _fx(0, v & 0xFFFFFFFF)


def hash_task_id(packed: int) -> int:
    return _fx(_fx(0, (packed >> 32) & 0xFFFFFFFF), packed & 0xFFFFFFFF)


class HbSet:
    """Insertion/removal-order faithful SwissTable of integer keys."""

    def __init__(self, hasher: Callable[[int], int], keys: Iterable[int] = ()):
        self.hasher = hasher
        self.nb = 0
        self.items = 0
        self.growth_left = 0
        self.ctrl: List[int] = []
        self.slot: List[int] = []
        for k in keys:
            self.insert(k)

    @staticmethod
    def _cap(mask: int) -> int:
        return mask if mask < 8 else ((mask + 1) // 8) * 7

    @staticmethod
    def _buckets_for(cap: int) -> int:
        if cap < 15:
            return 4 if cap < 4 else (8 if cap < 8 else 16)
        adj, p = cap * 8 // 7, 1
        while p < adj:
            p <<= 1
        return p

    def _alloc(self, nb: int):
        self.nb, self.items = nb, 0
        self.ctrl = [EMPTY] * (nb + WIDTH)
        self.slot = [0] * nb
        self.growth_left = self._cap(nb - 1)

    def _set_ctrl(self, i: int, c: int):
        m = self.nb - 1
        self.ctrl[i] = c
        self.ctrl[((i - WIDTH) & m) + WIDTH] = c

    def _first_special(self, pos: int) -> int:
        for b in range(WIDTH):
            if self.ctrl[pos + b] & 0x80:
                return b
        return -1

    def _insert_slot(self, h: int) -> int:
        m = self.nb - 1
        pos, stride = h & m, 0
        while True:
            b = self._first_special(pos)
            if b >= 0:
                idx = (pos + b) & m
                if not (self.ctrl[idx] & 0x80):
                    idx = self._first_special(0)
                return idx
            stride += WIDTH
            pos = (pos + stride) & m

    def _find(self, key: int) -> int:
        if self.nb == 0:
            return -1
        h = self.hasher(key)
        h2, m = h >> 57, self.nb - 1
        pos, stride = h & m, 0
        while True:
            any_empty = False
            for b in range(WIDTH):
                c = self.ctrl[pos + b]
                if c == h2:
                    idx = (pos + b) & m
                    if not (self.ctrl[idx] & 0x80) and self.slot[idx] == key:
                        return idx
                if c == EMPTY:
                    any_empty = True
            if any_empty:
                return -1
            stride += WIDTH
            pos = (pos + stride) & m

    def _resize(self, cap: int):
        old = [self.slot[i] for i in range(self.nb) if not (self.ctrl[i] & 0x80)]
        self._alloc(self._buckets_for(cap))
        for k in old:
            h = self.hasher(k)
            idx = self._insert_slot(h)
            self._set_ctrl(idx, h >> 57)
            self.slot[idx] = k
        self.items = len(old)
        self.growth_left = self._cap(self.nb - 1) - self.items

    def _rehash_in_place(self):
        nb, m = self.nb, self.nb - 1
        for i in range(nb):
            self.ctrl[i] = EMPTY if (self.ctrl[i] & 0x80) else DELETED
        if nb < WIDTH:
            for i in range(nb, WIDTH):
                self.ctrl[i] = EMPTY
            for i in range(nb):
                self.ctrl[WIDTH + i] = self.ctrl[i]
        else:
            for i in range(WIDTH):
                self.ctrl[nb + i] = self.ctrl[i]
        for i in range(nb):
            if self.ctrl[i] != DELETED:
                continue
            while True:
                h = self.hasher(self.slot[i])
                ni, p0 = self._insert_slot(h), h & m
                if ((i - p0) & m) // WIDTH == ((ni - p0) & m) // WIDTH:
                    self._set_ctrl(i, h >> 57)
                    break
                prev = self.ctrl[ni]
                self._set_ctrl(ni, h >> 57)
                if prev == EMPTY:
                    self._set_ctrl(i, EMPTY)
                    self.slot[ni] = self.slot[i]
                    break
                self.slot[i], self.slot[ni] = self.slot[ni], self.slot[i]
        self.growth_left = self._cap(m) - self.items

    def _reserve_one(self):
        if self.growth_left >= 1:
            return
        if self.nb == 0:
            self._alloc(self._buckets_for(1))
            return
        new_items, full = self.items + 1, self._cap(self.nb - 1)
        if new_items <= full // 2:
            self._rehash_in_place()
        else:
            self._resize(max(new_items, full + 1))

    def insert(self, key: int) -> bool:
        self._reserve_one()
        if self._find(key) >= 0:
            return False
        h = self.hasher(key)
        idx = self._insert_slot(h)
        if self.ctrl[idx] == EMPTY:
            self.growth_left -= 1
        self._set_ctrl(idx, h >> 57)
        self.slot[idx] = key
        self.items += 1
        return True

    def remove(self, key: int) -> bool:
        i = self._find(key)
        if i < 0:
            return False
        m = self.nb - 1
        before = (i - WIDTH) & m
        lead = 0
        for b in range(WIDTH - 1, -1, -1):
            if self.ctrl[before + b] == EMPTY:
                break
            lead += 1
        trail = 0
        for b in range(WIDTH):
            if self.ctrl[i + b] == EMPTY:
                break
            trail += 1
        if lead + trail >= WIDTH:
            self._set_ctrl(i, DELETED)
        else:
            self._set_ctrl(i, EMPTY)
            self.growth_left += 1
        self.items -= 1
        return True

    def __contains__(self, key: int) -> bool:
        return self._find(key) >= 0

    def __len__(self) -> int:
        return self.items

    def __iter__(self) -> Iterator[int]:
        for i in range(self.nb):
            if not (self.ctrl[i] & 0x80):
                yield self.slot[i]


def worker_id_set(keys: Iterable[int] = ()) -> HbSet:
    return HbSet(hash_u32, keys)


def task_id_set(keys: Iterable[int] = ()) -> HbSet:
    return HbSet(hash_task_id, keys)
