// Iteration order of the reference's `Map` (hashbrown::HashMap + fxhash::FxBuildHasher,
// /root/reference/crates/tako/src/internal/common/data_structures.rs:7) for the two maps the mapping stage walks:
//   counts: Map<WorkerId,u32>             built at scheduler/solver.rs:467-475, swept at scheduler/mapping.rs:43
//   sn_counts: Map<(RqId,VariantId), _>   built at scheduler/solver.rs:466-478, walked at scheduler/mapping.rs:36
// Both are insert-only within a tick, so only growth + slot placement have to be reproduced (no tombstones).
// hashbrown 0.17 / fxhash 0.2.1 are not under /root/reference: restated from their published algorithms; the
// Rust host can bypass this by passing the orders it observes.
#pragma once
#include <cstddef>
#include <cstdint>
#include <emmintrin.h>
#include <vector>

namespace hqhb {

inline uint64_t fx_step(uint64_t h, uint64_t word) { return (((h << 5) | (h >> 59)) ^ word) * 0x517cc1b727220a95ULL; }
inline uint64_t hash_worker_id(uint32_t id) { return fx_step(0, id); }                       // WorkerId(u32): one write_u32
inline uint64_t hash_rq_variant(uint32_t rq, uint8_t v) { return fx_step(fx_step(0, rq), v); }  // (u32, u8) tuple

// Simulates inserting n distinct keys (given by their 64-bit hashes, in insertion order) into an empty table and
// returns, in iteration order, the insertion index of every element.
// The table is kept the way hashbrown keeps it: one control byte per bucket (0xFF = EMPTY, anything else = taken) followed by a mirror of the first group, so that
// a group — 16 consecutive control bytes from an arbitrary position — is ONE unaligned SSE2 load and "first vacant bucket of the group" one movemask + count of
// trailing zeros (round 6: the scalar form looked at up to 16 buckets per probe with a wrap each, ~10 ns per placement; a heterogeneous tick simulates ~8 maps of
// 600-1000 workers, each of them placed about twice over its growth history).  Tables smaller than a group (4 / 8 buckets) see EMPTY padding between their
// buckets and the mirror, and an index that lands on the padding is looked up again in the aligned group at 0 — as the scalar form did.
inline void insertion_order(const uint64_t *hashes, uint32_t n, std::vector<uint32_t> &order) {
    const size_t GROUP = 16;                 // SSE2 group width on x86-64
    using std::size_t;
    // two (control bytes, bucket -> insertion index) pairs that swap roles at every growth: no allocation per growth step
    static thread_local std::vector<uint8_t> ctrl_a, ctrl_b;
    static thread_local std::vector<uint32_t> val_a, val_b;
    std::vector<uint8_t> *ctrl = &ctrl_a, *ctrl_spare = &ctrl_b;
    std::vector<uint32_t> *val = &val_a, *val_spare = &val_b;
    size_t nbuckets = 0, used = 0, room = 0;
    auto capacity_of = [](size_t nb) { return nb <= 8 ? nb - 1 : nb / 8 * 7; };
    auto buckets_for = [](size_t cap) -> size_t {
        if (cap < 4) return 4;
        if (cap < 8) return 8;
        if (cap < 15) return 16;
        size_t want = cap * 8 / 7, p = 1;
        while (p < want) p <<= 1;
        return p;
    };
    auto empties = [](const uint8_t *c) -> unsigned { return (unsigned)_mm_movemask_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i *>(c))); };
    // first vacant bucket along the triangular probe sequence, marked taken (in place and in the mirror)
    auto take = [&](uint8_t *c, size_t nb, uint64_t h) -> size_t {
        const size_t mask = nb - 1;
        size_t pos = (size_t)h & mask, stride = 0, idx;
        for (;;) {
            const unsigned m = empties(c + pos);
            if (m) {
                idx = (pos + (size_t)__builtin_ctz(m)) & mask;
                if (nb < GROUP && c[idx] != 0xFF) idx = (size_t)__builtin_ctz(empties(c));   // landed on the padding of a tiny table: the aligned group at 0
                break;
            }
            stride += GROUP;
            pos = (pos + stride) & mask;
        }
        c[idx] = 0;
        c[((idx - GROUP) & mask) + GROUP] = 0;
        return idx;
    };
    for (uint32_t i = 0; i < n; i++) {
        if (room == 0) {  // reserve(1): grow to hold max(items + 1, full_capacity + 1) and re-insert in iteration order
            size_t want = nbuckets == 0 ? 1 : (used + 1 > capacity_of(nbuckets) + 1 ? used + 1 : capacity_of(nbuckets) + 1);
            size_t nb = buckets_for(want);
            ctrl_spare->assign(nb + GROUP, 0xFF);
            if (val_spare->size() < nb) val_spare->resize(nb);
            uint8_t *big = ctrl_spare->data(); uint32_t *bigv = val_spare->data();
            const uint8_t *old = ctrl->data(); const uint32_t *oldv = val->data();
            for (size_t b0 = 0; b0 < nbuckets; b0 += GROUP) {
                unsigned full = ~empties(old + b0) & 0xFFFFu;
                if (nbuckets < GROUP) full &= (1u << nbuckets) - 1u;
                while (full) { const size_t b = b0 + (size_t)__builtin_ctz(full); full &= full - 1; bigv[take(big, nb, hashes[oldv[b]])] = oldv[b]; }
            }
            std::swap(ctrl, ctrl_spare); std::swap(val, val_spare);
            nbuckets = nb;
            room = capacity_of(nb) - used;
        }
        (*val)[take(ctrl->data(), nbuckets, hashes[i])] = i;
        used++;
        room--;
    }
    order.resize(n);
    uint32_t *out = order.data();
    const uint8_t *c = ctrl->data(); const uint32_t *v = val->data();
    for (size_t b0 = 0; b0 < nbuckets; b0 += GROUP) {
        unsigned full = ~empties(c + b0) & 0xFFFFu;
        if (nbuckets < GROUP) full &= (1u << nbuckets) - 1u;
        while (full) { *out++ = v[b0 + (size_t)__builtin_ctz(full)]; full &= full - 1; }
    }
}

inline void insertion_order_u32(const uint32_t *keys, uint32_t n, std::vector<uint32_t> &order) {
    std::vector<uint64_t> h(n);
    for (uint32_t i = 0; i < n; i++) h[i] = hash_worker_id(keys[i]);
    insertion_order(h.data(), n, order);
}

}  // namespace hqhb
