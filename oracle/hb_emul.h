// TEST INFRASTRUCTURE — part of the CPU oracle (see oracle/hq_oracle.cpp header). Never linked into the product.
//
// Emulation of the iteration order of `crate::Map`/`crate::Set`
//   = hashbrown::HashMap/HashSet<_, _, fxhash::FxBuildHasher>
//   (/root/reference/crates/tako/src/internal/common/data_structures.rs:7,81;
//    Cargo.lock: hashbrown 0.17.1, fxhash 0.2.1 — sources NOT under /root/reference).
//
// The tick observes that order at scheduler/mapping.rs:36 (sn_counts.into_iter()), :43 (counts.iter_mut()),
// :179 (worker_map.values_mut()) and scheduler/taskqueue.rs:381-388 (prefill Set drain).
//
// Restated from the published algorithms of those crates:
//   fxhash 0.2.1  FxHasher64: h = (rotl(h,5) ^ word) * 0x517cc1b727220a95, one word per write_u8/u32/u64
//   hashbrown     SwissTable, Group::WIDTH = 16 (x86-64 SSE2), h1 = hash, h2 = hash >> 57,
//                 triangular group probing, EMPTY=0xFF / DELETED=0x80 control bytes mirrored in the
//                 trailing group, capacity policy 4 -> 8 -> 16 -> next_pow2(cap*8/7), growth by
//                 reserve_rehash (rehash in place when items+1 <= full_capacity/2, else resize to
//                 max(items+1, full_capacity+1)), iteration = ascending bucket index.
// "parity unpinned" beyond the reference's small pinned tests: no Rust toolchain here to run the real crates.
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

namespace hb {

static const uint64_t FX_SEED = 0x517cc1b727220a95ULL;
inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
inline uint64_t fx_word(uint64_t h, uint64_t w) { return (rotl64(h, 5) ^ w) * FX_SEED; }
inline uint64_t fx_u32(uint32_t v) { return fx_word(0, v); }                      // WorkerId(u32), ResourceRqId(u32)
inline uint64_t fx_taskid(uint64_t packed) {                                       // TaskId{job_id:u32, job_task_id:u32}
    return fx_word(fx_word(0, (uint32_t)(packed >> 32)), (uint32_t)packed);
}
inline uint64_t fx_rqv(uint32_t rq, uint8_t v) { return fx_word(fx_word(0, rq), v); }  // (ResourceRqId, ResourceVariantId)

static const int WIDTH = 16;
static const uint8_t EMPTY = 0xFF, DELETED = 0x80;

// A hashbrown RawTable holding 64-bit payloads ("keys"); the caller supplies the hash of a payload.
template <typename HashFn> struct Table {
    HashFn hash_of;
    size_t buckets = 0;  // 0 = unallocated singleton
    size_t items = 0, growth_left = 0;
    std::vector<uint8_t> ctrl;  // buckets + WIDTH
    std::vector<uint64_t> slot;

    explicit Table(HashFn h = HashFn()) : hash_of(h) {}

    static size_t bucket_mask_to_capacity(size_t mask) { return mask < 8 ? mask : ((mask + 1) / 8) * 7; }
    static size_t capacity_to_buckets(size_t cap) {
        if (cap < 15) return cap < 4 ? 4 : (cap < 8 ? 8 : 16);
        size_t adj = cap * 8 / 7, p = 1;
        while (p < adj) p <<= 1;
        return p;
    }
    size_t mask() const { return buckets - 1; }
    static bool is_full(uint8_t c) { return (c & 0x80) == 0; }

    void alloc(size_t nb) {
        buckets = nb;
        ctrl.assign(nb + WIDTH, EMPTY);
        slot.assign(nb, 0);
        items = 0;
        growth_left = bucket_mask_to_capacity(nb - 1);
    }
    void set_ctrl(size_t i, uint8_t c) {
        size_t i2 = ((i - WIDTH) & mask()) + WIDTH;
        ctrl[i] = c;
        ctrl[i2] = c;
    }
    // lowest lane in group at pos whose ctrl is EMPTY or DELETED (high bit set), -1 if none
    int group_first_special(size_t pos) const {
        for (int b = 0; b < WIDTH; b++)
            if (ctrl[pos + b] & 0x80) return b;
        return -1;
    }
    size_t find_insert_slot(uint64_t hash) const {
        size_t pos = (size_t)hash & mask(), stride = 0;
        for (;;) {
            int b = group_first_special(pos);
            if (b >= 0) {
                size_t idx = (pos + b) & mask();
                if (is_full(ctrl[idx])) {  // small table: hit a mirrored byte
                    int b0 = group_first_special(0);
                    idx = (size_t)b0;
                }
                return idx;
            }
            stride += WIDTH;
            pos = (pos + stride) & mask();
        }
    }
    // returns bucket index or -1
    long find(uint64_t key) const {
        if (buckets == 0) return -1;
        uint64_t hash = hash_of(key);
        uint8_t h2 = (uint8_t)(hash >> 57);
        size_t pos = (size_t)hash & mask(), stride = 0;
        for (;;) {
            bool any_empty = false;
            for (int b = 0; b < WIDTH; b++) {
                uint8_t c = ctrl[pos + b];
                if (c == h2) {
                    size_t idx = (pos + b) & mask();
                    if (is_full(ctrl[idx]) && slot[idx] == key) return (long)idx;
                }
                if (c == EMPTY) any_empty = true;
            }
            if (any_empty) return -1;
            stride += WIDTH;
            pos = (pos + stride) & mask();
        }
    }
    void resize(size_t capacity) {
        Table nt(hash_of);
        nt.alloc(capacity_to_buckets(capacity));
        for (size_t i = 0; i < buckets; i++)
            if (is_full(ctrl[i])) {
                uint64_t h = hash_of(slot[i]);
                size_t idx = nt.find_insert_slot(h);
                nt.set_ctrl(idx, (uint8_t)(h >> 57));
                nt.slot[idx] = slot[i];
            }
        nt.items = items;
        nt.growth_left = bucket_mask_to_capacity(nt.buckets - 1) - items;
        *this = nt;
    }
    void rehash_in_place() {
        // prepare: FULL -> DELETED, DELETED -> EMPTY
        for (size_t i = 0; i < buckets; i++) ctrl[i] = is_full(ctrl[i]) ? DELETED : EMPTY;
        if (buckets < (size_t)WIDTH) {
            for (size_t i = buckets; i < (size_t)WIDTH; i++) ctrl[i] = EMPTY;
            for (size_t i = 0; i < buckets; i++) ctrl[WIDTH + i] = ctrl[i];
        } else {
            for (int i = 0; i < WIDTH; i++) ctrl[buckets + i] = ctrl[i];
        }
        for (size_t i = 0; i < buckets; i++) {
            if (ctrl[i] != DELETED) continue;
            for (;;) {
                uint64_t h = hash_of(slot[i]);
                size_t ni = find_insert_slot(h);
                size_t p0 = (size_t)h & mask();
                auto pidx = [&](size_t p) { return ((p - p0) & mask()) / WIDTH; };
                if (pidx(i) == pidx(ni)) {
                    set_ctrl(i, (uint8_t)(h >> 57));
                    break;
                }
                uint8_t prev = ctrl[ni];
                set_ctrl(ni, (uint8_t)(h >> 57));
                if (prev == EMPTY) {
                    set_ctrl(i, EMPTY);
                    slot[ni] = slot[i];
                    break;
                } else {
                    std::swap(slot[i], slot[ni]);
                }
            }
        }
        growth_left = bucket_mask_to_capacity(mask()) - items;
    }
    void reserve_one() {
        if (growth_left >= 1) return;
        if (buckets == 0) {
            alloc(capacity_to_buckets(1));
            return;
        }
        size_t new_items = items + 1, full = bucket_mask_to_capacity(mask());
        if (new_items <= full / 2)
            rehash_in_place();
        else
            resize(new_items > full + 1 ? new_items : full + 1);
    }
    // HashMap::insert / HashSet::insert of a key not (necessarily) present
    bool insert(uint64_t key) {
        reserve_one();  // hashbrown reserves before probing (find_or_find_insert_slot)
        if (find(key) >= 0) return false;
        uint64_t h = hash_of(key);
        size_t idx = find_insert_slot(h);
        uint8_t old = ctrl[idx];
        if (old == EMPTY) growth_left--;
        set_ctrl(idx, (uint8_t)(h >> 57));
        slot[idx] = key;
        items++;
        return true;
    }
    bool remove(uint64_t key) {
        long fi = find(key);
        if (fi < 0) return false;
        size_t i = (size_t)fi, before = (i - WIDTH) & mask();
        // leading empties of group before + trailing empties of group at i
        int lead = 0, trail = 0;
        for (int b = WIDTH - 1; b >= 0 && ctrl[before + b] != EMPTY; b--) lead++;   // "leading_zeros" of match_empty
        for (int b = 0; b < WIDTH && ctrl[i + b] != EMPTY; b++) trail++;            // "trailing_zeros" of match_empty
        if (lead + trail >= WIDTH)
            set_ctrl(i, DELETED);
        else {
            set_ctrl(i, EMPTY);
            growth_left++;
        }
        items--;
        return true;
    }
    // iteration order: ascending bucket index
    template <typename F> void for_each(F f) const {
        for (size_t i = 0; i < buckets; i++)
            if (is_full(ctrl[i])) f(slot[i]);
    }
    std::vector<uint64_t> order() const {
        std::vector<uint64_t> r;
        r.reserve(items);
        for_each([&](uint64_t k) { r.push_back(k); });
        return r;
    }
};

struct HashU32 {
    uint64_t operator()(uint64_t k) const { return fx_u32((uint32_t)k); }
};
struct HashTaskId {
    uint64_t operator()(uint64_t k) const { return fx_taskid(k); }
};
struct HashRqV {  // key packed as rq << 8 | v
    uint64_t operator()(uint64_t k) const { return fx_rqv((uint32_t)(k >> 8), (uint8_t)k); }
};
using WorkerIdTable = Table<HashU32>;
using TaskIdTable = Table<HashTaskId>;
using RqVTable = Table<HashRqV>;

}  // namespace hb
