// Wire encoding of the tick's worker messages (include/hqwire.h): the phases of the three kernels.
//
// Each kernel is a fixed sequence of PHASES separated by workgroup barriers; inside a phase a thread depends on no other thread of the
// same phase (LDS atomics apart).  The phases are plain functions of (arguments, LDS block, slot, tid), compiled for the device (kernels in
// wire.hip) and for the host, where the debug hook of include/hqtick_debug.h runs them for tid = 0..255 in turn -- so the CPU test suite
// executes the very same code the GPU does, minus the hardware.  Byte layout: bincode 1.3.3 fixint, little endian
// (/root/reference/crates/tako/src/internal/transfer/auth.rs:253-263); struct field order: messages/worker.rs:27-57.
#pragma once
#include <cstdint>
#include <new>

#include "../../include/hqwire.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define HQW_HD __host__ __device__ inline
#else
#define HQW_HD inline
#endif

namespace hqwire {

constexpr int BLOCK = 256, GROUPS = 16, GSIZE = BLOCK / GROUPS;
constexpr uint32_t HT = 4096, NONE = 0xFFFFFFFFu, MAXREC = HQWIRE_MAX_RECORDS, FMAX = HQWIRE_MAX_FRAGMENTS;
static_assert(HT >= 2 * MAXREC, "dedup table load factor <= 0.5");

struct Args {
    hqwire_tables t;
    hqwire_records r;
    hqwire_output o;
    uint32_t n_slots;
    // views into o.scratch
    uint64_t *slot_len;    // [2 * n_slots] bytes of the RetractTasks / ComputeTasks message of every slot
    uint32_t *slot_ncfg;   // [n_slots]     distinct configurations of the slot's ComputeTasks message
    uint32_t *rec_row;     // [n_rec]       row of the record's task in the attribute table
    uint32_t *rec_shared;  // [n_rec]       shared_index of the record
    uint32_t *cfg_list;    // [n_rec]       slot's configurations in first-occurrence order, at [slot's first record + k]
                           //               (a fragmented slot: every fragment's list at [slot's first record + fragment's first record + k])
    uint32_t *frag_rec;    // [n_slots * FMAX] record index (inside the slot) at which fragment f ends
    uint32_t *frag_ncfg;   // [n_slots * FMAX] distinct configurations of fragment f
    uint64_t limit;        // the builder's estimate limit
};

HQW_HD uint64_t scratch_bytes(uint64_t n_rec, uint64_t n_slots) { return 16 * n_slots + 4 * ((n_slots + 1) & ~1ull) + 12 * n_rec + 8ull * FMAX * n_slots + 16; }

HQW_HD void bind_scratch(Args &a) {
    uint8_t *p = (uint8_t *)a.o.scratch;
    const uint64_t n_rec = (uint64_t)a.r.n_records + a.r.n_mn;
    a.slot_len = (uint64_t *)p;
    p += 16 * (uint64_t)a.n_slots;
    a.slot_ncfg = (uint32_t *)p;
    p += 4 * (((uint64_t)a.n_slots + 1) & ~1ull);
    a.rec_row = (uint32_t *)p;
    a.rec_shared = a.rec_row + n_rec;
    a.cfg_list = a.rec_shared + n_rec;
    a.frag_rec = a.cfg_list + n_rec;
    a.frag_ncfg = a.frag_rec + (uint64_t)FMAX * a.n_slots;
    a.limit = a.o.msg_size_limit ? a.o.msg_size_limit : HQWIRE_MAX_TASK_MSG_SIZE;
}

// ---- what one message slot covers -------------------------------------------------------------------------------------------------------
struct Slot {
    uint32_t rec0, n;   // records [rec0, rec0 + n) in the scratch numbering (multi-node tasks follow the workers' records)
    uint32_t n_retract, retract0;
    bool mn;
    uint32_t mn_k;
};
HQW_HD Slot slot_of(const Args &a, uint32_t s) {
    Slot v{};
    if (s < a.r.n_workers) {
        v.rec0 = a.r.rec_off[s];
        v.n = a.r.rec_off[s + 1] - v.rec0;
        if (a.r.retract_off) {
            v.retract0 = a.r.retract_off[s];
            v.n_retract = a.r.retract_off[s + 1] - v.retract0;
        }
    } else {
        v.mn = true;
        v.mn_k = s - a.r.n_workers;
        v.rec0 = a.r.n_records + v.mn_k;
        v.n = 1;
    }
    return v;
}
struct Rec {
    uint64_t task;
    bool variant_some;
    uint8_t variant;
    uint32_t n_nodes, node0;
};
HQW_HD Rec rec_of(const Args &a, const Slot &s, uint32_t i) {  // i: index inside the slot
    Rec r{};
    if (s.mn) {  // ComputeTasksBuilder::single_task(task, 0.into(), worker_ids)   mapping.rs:284-291
        r.task = a.r.mn_task[s.mn_k];
        r.variant_some = true;
        r.variant = 0;
        r.node0 = a.r.mn_worker_off[s.mn_k];
        r.n_nodes = a.r.mn_worker_off[s.mn_k + 1] - r.node0;
    } else {
        r.task = a.r.rec_task[s.rec0 + i];
        r.variant_some = a.r.rec_kind[s.rec0 + i] != 0;  // prefills travel with variant None   mapping.rs:267-272
        r.variant = a.r.rec_variant[s.rec0 + i];
    }
    return r;
}
HQW_HD void run_of(uint32_t n, int tid, uint32_t &lo, uint32_t &hi) {  // contiguous share of thread tid
    const uint32_t per = (n + BLOCK - 1) / BLOCK;
    lo = (uint32_t)tid * per < n ? (uint32_t)tid * per : n;
    hi = lo + per < n ? lo + per : n;
}

// ---- sizes (messages/worker.rs:27-45; estimates: server/task.rs:405-433) -------------------------------------------------------------------
HQW_HD uint64_t entry_len(const Args &a, uint32_t row) { return a.t.entry_some[row] ? a.t.entry_off[row + 1] - a.t.entry_off[row] : 0; }
HQW_HD uint64_t rec_bytes(const Args &a, const Rec &r, uint32_t row) {
    return 42 + (r.variant_some ? 1 : 0) + 4ull * r.n_nodes + (a.t.entry_some[row] ? 8 + entry_len(a, row) : 0);
}
HQW_HD uint64_t rec_estimate(const Args &a, const Rec &r, uint32_t row) { return 34 + 4ull * r.n_nodes + entry_len(a, row); }
HQW_HD uint64_t body_len(const Args &a, uint32_t cfg) { return a.t.body_off[cfg + 1] - a.t.body_off[cfg]; }
HQW_HD uint64_t shared_bytes(const Args &a, uint32_t cfg) { return 1 + (a.t.config_time_some[cfg] ? 12 : 0) + 8 + body_len(a, cfg); }
HQW_HD uint64_t shared_estimate(const Args &a, uint32_t cfg) { return 16 + body_len(a, cfg); }

// ---- little-endian stores at any alignment -------------------------------------------------------------------------------------------------
// On the device a 4- or 8-byte value leaves as ONE store instruction whatever its alignment (gfx950 global memory takes unaligned dword / qword
// accesses): a 43-byte record is 9 store instructions instead of 43 single bytes.  On the host (debug hook) memcpy says the same thing.
HQW_HD uint8_t *put8(uint8_t *p, uint8_t v) {
    *p = v;
    return p + 1;
}
HQW_HD uint8_t *put32(uint8_t *p, uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    *reinterpret_cast<uint32_t __attribute__((aligned(1))) *>(p) = v;
#else
    for (int b = 0; b < 4; b++) p[b] = (uint8_t)(v >> (8 * b));
#endif
    return p + 4;
}
HQW_HD uint8_t *put64(uint8_t *p, uint64_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    *reinterpret_cast<uint64_t __attribute__((aligned(1))) *>(p) = v;
#else
    for (int b = 0; b < 8; b++) p[b] = (uint8_t)(v >> (8 * b));
#endif
    return p + 8;
}
// n bytes from src to dst, both at any alignment, by the threads tid, tid + stride, ...: single bytes up to the first 16-byte boundary of dst, 16-byte
// stores (unaligned 16-byte loads) through the middle, single bytes at the end
HQW_HD void copy_bytes(uint8_t *dst, const uint8_t *src, uint64_t n, int tid, int stride) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint64_t head = n < 16 ? n : ((16 - (reinterpret_cast<uintptr_t>(dst) & 15)) & 15);
    for (uint64_t b = (uint64_t)tid; b < head; b += stride) dst[b] = src[b];
    const uint64_t n16 = (n - head) / 16;
    typedef uint32_t __attribute__((ext_vector_type(4), aligned(1))) u4_unaligned;
    typedef uint32_t __attribute__((ext_vector_type(4))) u4;
    for (uint64_t q = (uint64_t)tid; q < n16; q += stride)
        *reinterpret_cast<u4 *>(dst + head + 16 * q) = *reinterpret_cast<const u4_unaligned *>(src + head + 16 * q);
    for (uint64_t b = head + 16 * n16 + (uint64_t)tid; b < n; b += stride) dst[b] = src[b];
#else
    for (uint64_t b = (uint64_t)tid; b < n; b += stride) dst[b] = src[b];
#endif
}

// ---- LDS atomics (plain on the host, where the phases run one thread at a time) ------------------------------------------------------------
HQW_HD uint32_t lds_cas(uint32_t *p, uint32_t expect, uint32_t val) {
#if defined(__HIP_DEVICE_COMPILE__)
    return atomicCAS(p, expect, val);
#else
    const uint32_t old = *p;
    if (old == expect) *p = val;
    return old;
#endif
}
HQW_HD void lds_min(uint32_t *p, uint32_t val) {
#if defined(__HIP_DEVICE_COMPILE__)
    atomicMin(p, val);
#else
    if (val < *p) *p = val;
#endif
}

// Lower bound in the ascending id column.  A binary search over a million rows is a chain of 20 dependent global loads (~14 us of k_wire_plan's 41):
// here every step loads 15 pivots at once and keeps one sixteenth of the range, 5 rounds instead of 20.
// First an interpolated guess: the tasks of a job are an id range (job << 32 | 1..n), so in the common table — a few large jobs — the row of an id is where its
// offset into the id span says, and ONE window of 16 loads around that point holds it (2 rounds instead of 5 + 1; k_wire_plan: 23 -> 14 us on the C3 tick).  Where it does
// not (many jobs: the span is mostly holes) the window still halves the range for the search below.
HQW_HD uint32_t find_row(const Args &a, uint64_t task) {
    uint64_t lo = 0, hi = a.t.n_tasks;  // answer in [lo, hi]: rows below lo are < task, rows from hi on are >= task
    if (hi >= 64) {
        const uint64_t first = a.t.task_id[0], last = a.t.task_id[hi - 1];
        if (task < first || task > last) return NONE;
        if (last > first) {
            const double frac = (double)(task - first) / (double)(last - first);
            uint64_t g = (uint64_t)(frac * (double)(hi - 1));
            if (g > hi - 1) g = hi - 1;
            const uint64_t w0 = g >= 8 ? g - 8 : 0, w1 = w0 + 16 <= hi ? w0 + 16 : hi;  // window [w0, w1)
            uint64_t v[16];
            for (int j = 0; j < 16; j++) v[j] = w0 + j < w1 ? a.t.task_id[w0 + j] : ~0ull;
            if (task < v[0]) hi = w0;                               // everything from w0 on is > task
            else if (w1 - w0 == 16 && task > v[15]) lo = w1;        // everything below w1 is < task
            else {  // inside the window
                for (int j = 0; j < 16; j++) if (w0 + j < w1 && v[j] == task) return (uint32_t)(w0 + j);
                return NONE;
            }
        }
    }
    while (hi - lo > 16) {
        const uint64_t step = (hi - lo + 15) / 16;
        uint64_t piv[15];
        for (int j = 0; j < 15; j++) { const uint64_t i = lo + (uint64_t)(j + 1) * step; piv[j] = i < hi ? a.t.task_id[i] : ~0ull; }  // independent loads: one round trip
        int below = 0;  // pivots < task (they ascend; positions past hi count as +inf)
        for (int j = 0; j < 15; j++) below += (lo + (uint64_t)(j + 1) * step < hi && piv[j] < task) ? 1 : 0;
        const uint64_t nlo = below ? lo + (uint64_t)below * step + 1 : lo;  // the last pivot below the task is itself excluded
        const uint64_t nhi = below < 15 && lo + (uint64_t)(below + 1) * step < hi ? lo + (uint64_t)(below + 1) * step : hi;
        lo = nlo; hi = nhi;
    }
    uint64_t v[16];
    for (int j = 0; j < 16; j++) v[j] = lo + j < hi ? a.t.task_id[lo + j] : ~0ull;
    uint32_t c = 0;
    for (int j = 0; j < 16; j++) c += (lo + j < hi && v[j] < task) ? 1u : 0u;
    lo += c;
    return lo < a.t.n_tasks && a.t.task_id[lo] == task ? (uint32_t)lo : NONE;
}
HQW_HD uint32_t hash_cfg(uint32_t cfg) { return (cfg * 2654435761u) >> 20; }  // 12 bits = HT

// =========================================================================================================================================
// Kernel 1 "plan": one workgroup per slot.  Row lookup, per-message configuration dedup (ComputeTasksBuilder::configuration_index,
// task.rs:327,346-359: shared_index = rank of the configuration's first occurrence in the message), byte lengths.
// =========================================================================================================================================
struct PlanLds {
    // 39.9 KB: four workgroups per CU (160 KB), i.e. the 1024 slots of a BASELINE tick are resident at once (at 49 KB — with an LDS copy of the slot's
    // configuration list and 32-bit counters — three fitted, and the launch took two rounds)
    uint32_t key[HT], val[HT];  // configuration -> index (inside the slot) of its first record; later -> its rank
    uint8_t first[MAXREC];
    uint64_t part_bytes[BLOCK], part_est[BLOCK];
    uint16_t part_cnt[BLOCK], base_cnt[BLOCK];  // <= MAXREC
    uint64_t grp_bytes[GROUPS], grp_est[GROUPS];  // sums over 16 consecutive threads' parts (two-level sums: a single thread walking 256 LDS entries
    uint32_t grp_cnt[GROUPS], grp_base[GROUPS];   // costs ~10 us of dependent LDS latency per walk; 16 + 16 + 15 steps cost 2)
    uint64_t rec_total, est_total;
    uint32_t n_cfg, bad, n_frag;
};

HQW_HD uint32_t ht_find(const PlanLds &l, uint32_t cfg) {
    uint32_t h = hash_cfg(cfg);
    while (l.key[h] != cfg) h = (h + 1) & (HT - 1);
    return h;
}

HQW_HD void plan_p0(const Args &, PlanLds &l, uint32_t, int tid) {
    for (uint32_t j = (uint32_t)tid; j < HT; j += BLOCK) l.key[j] = l.val[j] = NONE;
    if (tid == 0) {
        l.bad = HQWIRE_SLOT_OK;
        l.n_cfg = 0;
        l.rec_total = l.est_total = 0;
    }
}
HQW_HD void plan_p1(const Args &a, PlanLds &l, uint32_t s, int tid) {
    const Slot sv = slot_of(a, s);
    l.part_bytes[tid] = l.part_est[tid] = 0;
    if (sv.n > MAXREC) {
        if (tid == 0) l.bad = HQWIRE_SLOT_TOO_MANY;
        return;
    }
    uint32_t lo, hi;
    run_of(sv.n, tid, lo, hi);
    uint64_t bytes = 0, est = 0;
    for (uint32_t i = lo; i < hi; i++) {
        const Rec r = rec_of(a, sv, i);
        uint32_t row = find_row(a, r.task);
        if (row != NONE && a.t.task_config[row] >= a.t.n_configs) row = NONE;
        a.rec_row[sv.rec0 + i] = row;
        // (rec_shared holds the record's configuration until phase 5 turns it into the shared index: phases 2, 4 and 5 read it back with one coalesced load
        // instead of the chain rec_row -> task_config, two dependent gathers each)
        a.rec_shared[sv.rec0 + i] = row == NONE ? NONE : a.t.task_config[row];
        if (row == NONE) {
            l.bad = HQWIRE_SLOT_UNKNOWN;  // same value from every thread that stores it
            continue;
        }
        bytes += rec_bytes(a, r, row);
        est += rec_estimate(a, r, row);
        const uint32_t cfg = a.t.task_config[row];
        for (uint32_t h = hash_cfg(cfg);; h = (h + 1) & (HT - 1)) {
            const uint32_t prev = lds_cas(&l.key[h], NONE, cfg);
            if (prev == NONE || prev == cfg) {
                lds_min(&l.val[h], i);
                break;
            }
        }
    }
    l.part_bytes[tid] = bytes;
    l.part_est[tid] = est;
}
HQW_HD void plan_p2(const Args &a, PlanLds &l, uint32_t s, int tid) {
    const Slot sv = slot_of(a, s);
    l.part_cnt[tid] = 0;
    if (sv.n > MAXREC) return;
    uint32_t lo, hi, cnt = 0;
    run_of(sv.n, tid, lo, hi);
    for (uint32_t i = lo; i < hi; i++) {
        const uint32_t cfg = a.rec_shared[sv.rec0 + i];  // written by this thread in p1
        const bool f = cfg != NONE && l.val[ht_find(l, cfg)] == i;
        l.first[i] = f ? 1 : 0;
        cnt += f ? 1 : 0;
    }
    l.part_cnt[tid] = (uint16_t)cnt;
}
HQW_HD void plan_p3a(const Args &, PlanLds &l, uint32_t, int tid) {  // group sums of the per-thread record counts / bytes / estimates
    if (tid >= GROUPS) return;
    uint32_t c = 0;
    uint64_t b = 0, e = 0;
    for (int t = tid * GSIZE; t < (tid + 1) * GSIZE; t++) { c += l.part_cnt[t]; b += l.part_bytes[t]; e += l.part_est[t]; }
    l.grp_cnt[tid] = c; l.grp_bytes[tid] = b; l.grp_est[tid] = e;
}
HQW_HD void plan_p3b(const Args &, PlanLds &l, uint32_t, int tid) {
    if (tid != 0) return;
    uint32_t c = 0;
    uint64_t b = 0, e = 0;
    for (int g = 0; g < GROUPS; g++) { l.grp_base[g] = c; c += l.grp_cnt[g]; b += l.grp_bytes[g]; e += l.grp_est[g]; }
    l.n_cfg = c;
    l.rec_total = b;
    l.est_total = e;
}
HQW_HD void plan_p3c(const Args &, PlanLds &l, uint32_t, int tid) {  // every thread: distinct configurations first seen by the threads before it
    uint32_t c = l.grp_base[tid / GSIZE];
    for (int t = (tid / GSIZE) * GSIZE; t < tid; t++) c += l.part_cnt[t];
    l.base_cnt[tid] = (uint16_t)c;
}
HQW_HD void plan_p4(const Args &a, PlanLds &l, uint32_t s, int tid) {
    const Slot sv = slot_of(a, s);
    if (sv.n > MAXREC) return;
    uint32_t lo, hi, k = l.base_cnt[tid];
    run_of(sv.n, tid, lo, hi);
    for (uint32_t i = lo; i < hi; i++) {
        if (!l.first[i]) continue;
        const uint32_t cfg = a.rec_shared[sv.rec0 + i];
        l.val[ht_find(l, cfg)] = k;  // from here on the table maps configuration -> shared_index (nobody reads first-indices any more)
        a.cfg_list[sv.rec0 + k] = cfg;  // (read back in phase 5 by other threads of this workgroup: behind the barrier)
        k++;
    }
}
HQW_HD void plan_p5(const Args &a, PlanLds &l, uint32_t s, int tid) {
    const Slot sv = slot_of(a, s);
    l.part_bytes[tid] = l.part_est[tid] = 0;
    if (sv.n > MAXREC) return;
    uint32_t lo, hi;
    run_of(sv.n, tid, lo, hi);
    for (uint32_t i = lo; i < hi; i++) {
        const uint32_t cfg = a.rec_shared[sv.rec0 + i];  // (this thread's own record: phases 1, 2, 4 and 5 walk the same runs)
        a.rec_shared[sv.rec0 + i] = cfg == NONE ? NONE : l.val[ht_find(l, cfg)];
    }
    uint64_t bytes = 0, est = 0;
    for (uint32_t k = (uint32_t)tid; k < l.n_cfg; k += BLOCK) {
        const uint32_t cfg = a.cfg_list[sv.rec0 + k];
        bytes += shared_bytes(a, cfg);
        est += shared_estimate(a, cfg);
    }
    l.part_bytes[tid] = bytes;
    l.part_est[tid] = est;
}
HQW_HD void plan_p6a(const Args &, PlanLds &l, uint32_t, int tid) {  // group sums of the shared-data bytes / estimates
    if (tid >= GROUPS) return;
    uint64_t b = 0, e = 0;
    for (int t = tid * GSIZE; t < (tid + 1) * GSIZE; t++) { b += l.part_bytes[t]; e += l.part_est[t]; }
    l.grp_bytes[tid] = b; l.grp_est[tid] = e;
}
HQW_HD void plan_p6(const Args &a, PlanLds &l, uint32_t s, int tid) {
    if (tid != 0) return;
    const Slot sv = slot_of(a, s);
    uint64_t sh = 0, est = l.est_total;
    for (int g = 0; g < GROUPS; g++) {
        sh += l.grp_bytes[g];
        est += l.grp_est[g];
    }
    uint32_t status = l.bad;
    l.n_frag = (status == HQWIRE_SLOT_OK && sv.n) ? 1 : 0;
    if (status == HQWIRE_SLOT_OK && est > a.limit) {  // create_message_on_overflow  task.rs:388-400
        if (a.o.slot_nfrag && a.o.frag_end) l.n_frag = FMAX + 1;  // "cut it": phase 7 finds the cuts
        else status = HQWIRE_SLOT_OVERSIZE;
    }
    l.bad = status;
    a.slot_ncfg[s] = l.n_cfg;
    a.slot_len[2 * s] = sv.n_retract ? 12 + 8ull * sv.n_retract : 0;                                  // tag + len + ids
    a.slot_len[2 * s + 1] = (status == HQWIRE_SLOT_OK && sv.n) ? 4 + 8 + l.rec_total + 8 + sh : 0;    // tag + len + tasks + len + shared
    if (l.n_frag == 1) { a.frag_rec[(uint64_t)s * FMAX] = sv.n; a.frag_ncfg[(uint64_t)s * FMAX] = l.n_cfg; }
}
// Fragmentation of an over-limit slot, exactly as the builder does it (task.rs:346-400): records in send order; a configuration's shared
// estimate counts at its first use INSIDE the current message, then the record's own estimate; the message is cut after the record that
// takes the estimate past the limit, and the configuration index starts afresh.  One thread: the cut positions are a sequential
// function of the running estimate (rare path: a worker receiving more than 32 MiB of task data in one tick).
HQW_HD void plan_p7(const Args &a, PlanLds &l, uint32_t s, int tid) {
    if (tid != 0) return;
    const Slot sv = slot_of(a, s);
    if (l.n_frag == FMAX + 1) {
        uint32_t nf = 0, r0 = 0, ncfg = 0;
        uint64_t est = 0, bytes = 0, total = 0;
        for (uint32_t j = 0; j < HT; j++) l.key[j] = l.val[j] = NONE;
        bool too_many = false;
        for (uint32_t i = 0; i < sv.n; i++) {
            const uint32_t row = a.rec_row[sv.rec0 + i], cfg = a.t.task_config[row];
            uint32_t h = hash_cfg(cfg);
            while (l.key[h] != NONE && l.key[h] != cfg) h = (h + 1) & (HT - 1);
            if (l.key[h] == NONE) {
                l.key[h] = cfg; l.val[h] = ncfg;
                a.cfg_list[sv.rec0 + r0 + ncfg] = cfg;
                ncfg++;
                est += shared_estimate(a, cfg); bytes += shared_bytes(a, cfg);
            }
            a.rec_shared[sv.rec0 + i] = l.val[h];
            const Rec r = rec_of(a, sv, i);
            est += rec_estimate(a, r, row); bytes += rec_bytes(a, r, row);
            const bool last = i + 1 == sv.n;
            if (est > a.limit || last) {
                if (nf == FMAX) { too_many = true; break; }
                a.frag_rec[(uint64_t)s * FMAX + nf] = i + 1; a.frag_ncfg[(uint64_t)s * FMAX + nf] = ncfg;
                total += 4 + 8 + 8 + bytes;
                nf++;
                r0 = i + 1; ncfg = 0; est = 0; bytes = 0;
                if (!last) for (uint32_t j = 0; j < HT; j++) l.key[j] = l.val[j] = NONE;
            }
        }
        if (too_many) { l.bad = HQWIRE_SLOT_OVERSIZE; l.n_frag = 0; a.slot_len[2 * s + 1] = 0; }
        else { l.n_frag = nf; a.slot_len[2 * s + 1] = total; }
    }
    a.o.slot_status[s] = (uint8_t)l.bad;
    if (a.o.slot_nfrag) a.o.slot_nfrag[s] = l.bad == HQWIRE_SLOT_OK ? l.n_frag : 0;
    if (!a.o.slot_nfrag) { /* fragment tables live in the scratch only */ }
    a.slot_ncfg[s] = (a.slot_ncfg[s] & 0x00FFFFFFu) | ((l.bad == HQWIRE_SLOT_OK ? l.n_frag : 0) << 24);  // emit reads the fragment count from here (nfrag_of)
}

// =========================================================================================================================================
// Kernel 2 "scan": one workgroup.  Exclusive scan of the 2 * n_slots message lengths -> slot_off, header.
// =========================================================================================================================================
struct ScanLds {
    uint64_t part[BLOCK], base[BLOCK], grp[GROUPS], grp_base[GROUPS];
};
HQW_HD void scan_p1(const Args &a, ScanLds &l, int tid) {
    uint32_t lo, hi;
    run_of(2 * a.n_slots, tid, lo, hi);
    uint64_t sum = 0;
    for (uint32_t j = lo; j < hi; j++) sum += a.slot_len[j];
    l.part[tid] = sum;
}
HQW_HD void scan_p2a(const Args &, ScanLds &l, int tid) {
    if (tid >= GROUPS) return;
    uint64_t sum = 0;
    for (int t = tid * GSIZE; t < (tid + 1) * GSIZE; t++) sum += l.part[t];
    l.grp[tid] = sum;
}
HQW_HD void scan_p2(const Args &a, ScanLds &l, int tid) {
    if (tid != 0) return;
    uint64_t run = 0;
    for (int g = 0; g < GROUPS; g++) {
        l.grp_base[g] = run;
        run += l.grp[g];
    }
    a.o.slot_off[2 * (uint64_t)a.n_slots] = run;
    a.o.header[0] = run > a.o.capacity ? HQWIRE_CAPACITY : HQWIRE_OK;
    a.o.header[1] = a.n_slots;
    a.o.header[2] = (uint32_t)run;
    a.o.header[3] = (uint32_t)(run >> 32);
}
HQW_HD void scan_p2c(const Args &, ScanLds &l, int tid) {
    uint64_t run = l.grp_base[tid / GSIZE];
    for (int t = (tid / GSIZE) * GSIZE; t < tid; t++) run += l.part[t];
    l.base[tid] = run;
}
HQW_HD void scan_p3(const Args &a, ScanLds &l, int tid) {
    uint32_t lo, hi;
    run_of(2 * a.n_slots, tid, lo, hi);
    uint64_t run = l.base[tid];
    for (uint32_t j = lo; j < hi; j++) {
        a.o.slot_off[j] = run;
        run += a.slot_len[j];
    }
}

// =========================================================================================================================================
// Kernel 3 "emit": one workgroup per slot writes its messages.
// =========================================================================================================================================
struct EmitLds {
    uint64_t part_rec[BLOCK], part_sh[BLOCK], base_rec[BLOCK], base_sh[BLOCK];
    uint64_t grp_rec[GROUPS], grp_sh[GROUPS], gbase_rec[GROUPS], gbase_sh[GROUPS];
    uint32_t body_rel[MAXREC];  // offset of shared entry k's body inside the ComputeTasks message
    uint64_t rec_total, msg_base;
};
HQW_HD uint32_t nfrag_of(const Args &a, uint32_t s) { return a.slot_ncfg[s] >> 24; }
HQW_HD bool emit_active(const Args &a, uint32_t s) { return a.o.header[0] == HQWIRE_OK && a.slot_len[2 * s + 1] != 0; }
// fragment f of slot s: its records [r0, r1) (indices inside the slot), its configuration list and count
struct Frag { uint32_t r0, r1, ncfg, cfg0; };
HQW_HD Frag frag_of(const Args &a, const Slot &sv, uint32_t s, uint32_t f) {
    Frag g{};
    g.r0 = f ? a.frag_rec[(uint64_t)s * FMAX + f - 1] : 0;
    g.r1 = a.frag_rec[(uint64_t)s * FMAX + f];
    g.ncfg = a.frag_ncfg[(uint64_t)s * FMAX + f];
    g.cfg0 = sv.rec0 + g.r0;
    return g;
}

HQW_HD void emit_p1(const Args &a, EmitLds &l, uint32_t s, uint32_t f, int tid) {
    const Slot sv = slot_of(a, s);
    l.part_rec[tid] = l.part_sh[tid] = 0;
    if (a.o.header[0] != HQWIRE_OK) return;
    if (f == 0 && a.slot_len[2 * s]) {  // ToWorkerMessage::RetractTasks(TaskIdsMsg { ids })   mapping.rs:261-266
        uint8_t *m = a.o.bytes + a.o.slot_off[2 * s];
        if (tid == 0) put64(put32(m, 1), sv.n_retract);
        for (uint32_t j = (uint32_t)tid; j < sv.n_retract; j += BLOCK) {
            const uint64_t id = a.r.retract_task[sv.retract0 + j];
            put32(put32(m + 12 + 8ull * j, (uint32_t)(id >> 32)), (uint32_t)id);
        }
    }
    if (!emit_active(a, s)) return;
    const Frag g = frag_of(a, sv, s, f);
    uint32_t lo, hi;
    run_of(g.r1 - g.r0, tid, lo, hi);
    uint64_t sum = 0;
    for (uint32_t i = g.r0 + lo; i < g.r0 + hi; i++) sum += rec_bytes(a, rec_of(a, sv, i), a.rec_row[sv.rec0 + i]);
    l.part_rec[tid] = sum;
    run_of(g.ncfg, tid, lo, hi);
    sum = 0;
    for (uint32_t k = lo; k < hi; k++) sum += shared_bytes(a, a.cfg_list[g.cfg0 + k]);
    l.part_sh[tid] = sum;
}
HQW_HD void emit_p2a(const Args &a, EmitLds &l, uint32_t s, uint32_t, int tid) {
    if (tid >= GROUPS || !emit_active(a, s)) return;
    uint64_t r = 0, h = 0;
    for (int t = tid * GSIZE; t < (tid + 1) * GSIZE; t++) { r += l.part_rec[t]; h += l.part_sh[t]; }
    l.grp_rec[tid] = r; l.grp_sh[tid] = h;
}
HQW_HD void emit_p2(const Args &a, EmitLds &l, uint32_t s, uint32_t f, int tid) {
    if (tid != 0 || !emit_active(a, s)) return;
    uint64_t r = 0, h = 0;
    for (int g = 0; g < GROUPS; g++) {
        l.gbase_rec[g] = r;
        r += l.grp_rec[g];
        l.gbase_sh[g] = h;
        h += l.grp_sh[g];
    }
    l.rec_total = r;
    const Slot sv = slot_of(a, s);
    const Frag g = frag_of(a, sv, s, f);
    l.msg_base = f ? a.o.frag_end[(uint64_t)s * FMAX + f - 1] : a.o.slot_off[2 * s + 1];
    uint8_t *m = a.o.bytes + l.msg_base;
    put64(put32(m, 0), g.r1 - g.r0);               // ToWorkerMessage::ComputeTasks, tasks.len()
    put64(m + 12 + r, g.ncfg);                     // shared_data.len()
    if (a.o.frag_end) a.o.frag_end[(uint64_t)s * FMAX + f] = l.msg_base + 12 + r + 8 + h;
}
HQW_HD void emit_p2c(const Args &a, EmitLds &l, uint32_t s, uint32_t, int tid) {
    if (!emit_active(a, s)) return;
    uint64_t r = l.gbase_rec[tid / GSIZE], h = l.gbase_sh[tid / GSIZE];
    for (int t = (tid / GSIZE) * GSIZE; t < tid; t++) { r += l.part_rec[t]; h += l.part_sh[t]; }
    l.base_rec[tid] = r; l.base_sh[tid] = h;
}
HQW_HD void emit_p3(const Args &a, EmitLds &l, uint32_t s, uint32_t f, int tid) {
    if (!emit_active(a, s)) return;
    const Slot sv = slot_of(a, s);
    const Frag g = frag_of(a, sv, s, f);
    uint8_t *m = a.o.bytes + l.msg_base;
    uint32_t lo, hi;
    run_of(g.r1 - g.r0, tid, lo, hi);
    uint8_t *p = m + 12 + l.base_rec[tid];
    for (uint32_t i = g.r0 + lo; i < g.r0 + hi; i++) {  // ComputeTaskSeparateData   messages/worker.rs:27-39
        const Rec r = rec_of(a, sv, i);
        const uint32_t row = a.rec_row[sv.rec0 + i];
        p = put64(p, a.rec_shared[sv.rec0 + i]);                                    // shared_index: usize
        p = put32(put32(p, (uint32_t)(r.task >> 32)), (uint32_t)r.task);             // id: TaskId { job_id, job_task_id }
        p = put32(p, a.t.task_rq[row]);                                             // resource_rq_id
        p = put8(p, r.variant_some ? 1 : 0);                                        // resource_rq_variant: Option<u8>
        if (r.variant_some) p = put8(p, r.variant);
        p = put32(p, a.t.task_instance[row]);                                       // instance_id
        p = put64(p, a.t.task_priority[row]);                                       // priority
        p = put64(p, r.n_nodes);                                                    // node_list: Vec<WorkerId>
        for (uint32_t j = 0; j < r.n_nodes; j++) p = put32(p, a.r.worker_id[a.r.mn_worker[r.node0 + j]]);
        p = put8(p, a.t.entry_some[row] ? 1 : 0);                                   // entry: Option<ThinVec<u8>>
        if (a.t.entry_some[row]) {
            const uint64_t n = entry_len(a, row), e0 = a.t.entry_off[row];
            p = put64(p, n);
            copy_bytes(p, a.t.entry_blob + e0, n, 0, 1);
            p += n;
        }
    }
    run_of(g.ncfg, tid, lo, hi);
    const uint64_t shared0 = 12 + l.rec_total + 8;
    p = m + shared0 + l.base_sh[tid];
    for (uint32_t k = lo; k < hi; k++) {  // ComputeTaskSharedData   messages/worker.rs:41-45 (the body itself: phase 4)
        const uint32_t cfg = a.cfg_list[g.cfg0 + k];
        p = put8(p, a.t.config_time_some[cfg] ? 1 : 0);
        if (a.t.config_time_some[cfg]) p = put32(put64(p, a.t.config_time_secs[cfg]), a.t.config_time_nanos[cfg]);
        p = put64(p, body_len(a, cfg));
        l.body_rel[k] = (uint32_t)(p - m);
        p += body_len(a, cfg);
    }
}
HQW_HD void emit_p4(const Args &a, EmitLds &l, uint32_t s, uint32_t f, int tid) {
    if (!emit_active(a, s)) return;
    const Slot sv = slot_of(a, s);
    const Frag g = frag_of(a, sv, s, f);
    uint8_t *m = a.o.bytes + l.msg_base;
    // bodies: one wavefront per body, the four wavefronts of the workgroup on four bodies at a time (byte-coalesced 16-byte stores).  The whole workgroup on
    // one body after the other was eight load -> store round trips in a row for a typical message (8 configurations of ~1 KB: a body is 64 stores of 16 bytes, a
    // quarter of the workgroup); this is two.
    for (uint32_t k = (uint32_t)tid / 64; k < g.ncfg; k += BLOCK / 64) {
        const uint32_t cfg = a.cfg_list[g.cfg0 + k];
        const uint64_t n = body_len(a, cfg), b0 = a.t.body_off[cfg];
        uint8_t *dst = m + l.body_rel[k];
        copy_bytes(dst, a.t.body_blob + b0, n, tid % 64, 64);
    }
}

// ---- host execution of the phases (debug hook, sanitizer harness): one emulated thread after the other, a loop end = a workgroup barrier --------
// `order`: sequence of the 256 emulated threads inside every phase (0 ascending, 1 descending, 2 a fixed permutation) -- the result must not
// depend on it.  Returns false when the LDS stand-ins cannot be allocated.
inline bool run_on_host(const Args &a, int order) {
    int seq[BLOCK];
    for (int i = 0; i < BLOCK; i++) seq[i] = order == 1 ? BLOCK - 1 - i : order == 2 ? (i * 77 + 13) % BLOCK : i;  // 77 is coprime to 256
    PlanLds *pl = new (std::nothrow) PlanLds;
    ScanLds *sl = new (std::nothrow) ScanLds;
    EmitLds *el = new (std::nothrow) EmitLds;
    const bool ok = pl && sl && el;
    if (ok) {
#define HQW_PHASE(fn, lds, s) \
    for (int q = 0; q < BLOCK; q++) fn(a, lds, s, seq[q])
        for (uint32_t s = 0; s < a.n_slots; s++) {
            HQW_PHASE(plan_p0, *pl, s);
            HQW_PHASE(plan_p1, *pl, s);
            HQW_PHASE(plan_p2, *pl, s);
            HQW_PHASE(plan_p3a, *pl, s);
            HQW_PHASE(plan_p3b, *pl, s);
            HQW_PHASE(plan_p3c, *pl, s);
            HQW_PHASE(plan_p4, *pl, s);
            HQW_PHASE(plan_p5, *pl, s);
            HQW_PHASE(plan_p6a, *pl, s);
            HQW_PHASE(plan_p6, *pl, s);
            HQW_PHASE(plan_p7, *pl, s);
        }
        for (int q = 0; q < BLOCK; q++) scan_p1(a, *sl, seq[q]);
        for (int q = 0; q < BLOCK; q++) scan_p2a(a, *sl, seq[q]);
        for (int q = 0; q < BLOCK; q++) scan_p2(a, *sl, seq[q]);
        for (int q = 0; q < BLOCK; q++) scan_p2c(a, *sl, seq[q]);
        for (int q = 0; q < BLOCK; q++) scan_p3(a, *sl, seq[q]);
        for (uint32_t s = 0; s < a.n_slots; s++) {
            const uint32_t nf = nfrag_of(a, s) ? nfrag_of(a, s) : 1;  // a slot without a ComputeTasks part still runs fragment 0: its RetractTasks message
            for (uint32_t f = 0; f < nf; f++) {
                for (int q = 0; q < BLOCK; q++) emit_p1(a, *el, s, f, seq[q]);
                for (int q = 0; q < BLOCK; q++) emit_p2a(a, *el, s, f, seq[q]);
                for (int q = 0; q < BLOCK; q++) emit_p2(a, *el, s, f, seq[q]);
                for (int q = 0; q < BLOCK; q++) emit_p2c(a, *el, s, f, seq[q]);
                for (int q = 0; q < BLOCK; q++) emit_p3(a, *el, s, f, seq[q]);
                for (int q = 0; q < BLOCK; q++) emit_p4(a, *el, s, f, seq[q]);
            }
        }
#undef HQW_PHASE
    }
    delete pl;
    delete sl;
    delete el;
    return ok;
}

}  // namespace hqwire
