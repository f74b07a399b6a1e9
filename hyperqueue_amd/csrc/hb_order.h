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
#include <vector>

namespace hqhb {

inline uint64_t fx_step(uint64_t h, uint64_t word) { return (((h << 5) | (h >> 59)) ^ word) * 0x517cc1b727220a95ULL; }
inline uint64_t hash_worker_id(uint32_t id) { return fx_step(0, id); }                       // WorkerId(u32): one write_u32
inline uint64_t hash_rq_variant(uint32_t rq, uint8_t v) { return fx_step(fx_step(0, rq), v); }  // (u32, u8) tuple

// Simulates inserting n distinct keys (given by their 64-bit hashes, in insertion order) into an empty table and
// returns, in iteration order, the insertion index of every element.
inline void insertion_order(const uint64_t *hashes, uint32_t n, std::vector<uint32_t> &order) {
    const int GROUP = 16;                 // SSE2 group width on x86-64
    const uint32_t VACANT = 0xFFFFFFFFu;
    using std::size_t;
    // two bucket arrays (bucket -> insertion index, VACANT = empty control byte) that swap roles at every growth: no allocation per growth step
    static thread_local std::vector<uint32_t> buf_a, buf_b;
    std::vector<uint32_t> *owner = &buf_a, *spare = &buf_b;
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
    // first vacant bucket along the triangular probe sequence; groups are GROUP consecutive control bytes starting
    // at an arbitrary position, and tables smaller than a group see vacant padding before their mirrored bytes.
    auto place = [&](uint32_t *tab, size_t nb, uint64_t h) -> size_t {
        size_t mask = nb - 1, pos = (size_t)h & mask, stride = 0;
        if (nb >= (size_t)GROUP) {  // the common case: a group is 16 consecutive buckets (wrapping)
            for (;;) {
                for (int b = 0; b < GROUP; b++) { const size_t idx = (pos + b) & mask; if (tab[idx] == VACANT) return idx; }
                stride += GROUP;
                pos = (pos + stride) & mask;
            }
        }
        for (;;) {
            for (int b = 0; b < GROUP; b++) {
                size_t lane = pos + b;
                const bool vacant = lane < nb ? tab[lane] == VACANT : (lane < (size_t)GROUP ? true : tab[lane - GROUP] == VACANT);
                if (vacant) {
                    size_t idx = lane & mask;
                    if (tab[idx] != VACANT) {  // landed on padding of a tiny table: rescan from bucket 0
                        for (size_t i = 0; i < nb; i++) if (tab[i] == VACANT) return i;
                    }
                    return idx;
                }
            }
            stride += GROUP;
            pos = (pos + stride) & mask;
        }
    };
    for (uint32_t i = 0; i < n; i++) {
        if (room == 0) {  // reserve(1): grow to hold max(items + 1, full_capacity + 1) and re-insert in iteration order
            size_t want = nbuckets == 0 ? 1 : (used + 1 > capacity_of(nbuckets) + 1 ? used + 1 : capacity_of(nbuckets) + 1);
            size_t nb = buckets_for(want);
            spare->assign(nb, VACANT);
            uint32_t *big = spare->data(); const uint32_t *old = owner->data();
            for (size_t b = 0; b < nbuckets; b++) if (old[b] != VACANT) big[place(big, nb, hashes[old[b]])] = old[b];
            std::swap(owner, spare);
            nbuckets = nb;
            room = capacity_of(nb) - used;
        }
        uint32_t *tab = owner->data();
        tab[place(tab, nbuckets, hashes[i])] = i;
        used++;
        room--;
    }
    order.clear();
    order.reserve(n);
    const uint32_t *tab = owner->data();
    for (size_t b = 0; b < nbuckets; b++) if (tab[b] != VACANT) order.push_back(tab[b]);
}

inline void insertion_order_u32(const uint32_t *keys, uint32_t n, std::vector<uint32_t> &order) {
    static thread_local std::vector<uint64_t> h;   // (no allocation per simulated map)
    if (h.size() < n) h.resize(n);
    for (uint32_t i = 0; i < n; i++) h[i] = hash_worker_id(keys[i]);
    insertion_order(h.data(), n, order);
}

}  // namespace hqhb
