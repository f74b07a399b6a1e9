// `Map<ResourceIndex, ResourceFractions>` of the worker-side pools with the iteration order of the reference's map type
// (hashbrown::HashMap + fxhash::FxBuildHasher, /root/reference/crates/tako/src/internal/common/data_structures.rs:7).
// The order is observable: `best_fraction_match` (worker/resources/pool.rs:372-380) returns the FIRST entry, in iteration
// order, among those with the smallest sufficient remainder.  Unlike the tick's maps (hb_order.h) these see removals
// (pool.rs:471,495), so control bytes, tombstones, in-place rehash and growth are all modelled.
// hashbrown 0.17.1 / fxhash 0.2.1 (Cargo.lock) are not under /root/reference: restated from their published algorithm
// (16-byte SSE2 groups, triangular probing, 7/8 load factor, EMPTY = 0xFF, DELETED = 0x80, tag = top 7 hash bits).
#pragma once
#include <cstddef>
#include <cstdint>
#include <utility>
#include <vector>

namespace hqhb {

class U32Map {
  public:
    // HashMap::insert: reserve(1) first, then find-or-insert (an existing key keeps its bucket)
    void insert(uint32_t key, uint32_t val) {
        reserve_one();
        long i = find(key);
        if (i >= 0) {
            vals_[i] = val;
            return;
        }
        put(key, val);
    }
    uint32_t *get(uint32_t key) {
        long i = find(key);
        return i < 0 ? nullptr : &vals_[i];
    }
    bool contains(uint32_t key) const { return find(key) >= 0; }
    bool remove(uint32_t key) {
        long i = find(key);
        if (i < 0) return false;
        const size_t mask = nb_ - 1, before = ((size_t)i - W) & mask;
        // erase(): the bucket may become EMPTY only if no probe sequence can have passed over a full group here
        size_t lead = 0, trail = 0;
        for (int b = W - 1; b >= 0 && ctrl_[before + b] != EMPTY; b--) lead++;
        for (int b = 0; b < W && ctrl_[i + b] != EMPTY; b++) trail++;
        if (lead + trail >= (size_t)W) set_ctrl(i, DELETED);
        else {
            set_ctrl(i, EMPTY);
            growth_left_++;
        }
        items_--;
        return true;
    }
    size_t size() const { return items_; }
    // iteration: ascending bucket index
    template <class F> void for_each(F f) const {
        for (size_t i = 0; i < nb_; i++)
            if (!(ctrl_[i] & 0x80)) f(keys_[i], vals_[i]);
    }
    template <class F> void for_each_mut(F f) {
        for (size_t i = 0; i < nb_; i++)
            if (!(ctrl_[i] & 0x80)) f(keys_[i], vals_[i]);
    }
    uint32_t max_value() const {
        uint32_t m = 0;
        for_each([&](uint32_t, uint32_t v) { if (v > m) m = v; });
        return m;
    }

  private:
    static constexpr int W = 16;
    static constexpr uint8_t EMPTY = 0xFF, DELETED = 0x80;
    size_t nb_ = 0, items_ = 0, growth_left_ = 0;
    std::vector<uint8_t> ctrl_;
    std::vector<uint32_t> keys_, vals_;

    static uint64_t hash(uint32_t key) { return (uint64_t)key * 0x517cc1b727220a95ULL; }  // FxHasher::write_u32 from 0
    static size_t capacity(size_t nb) { return nb <= 8 ? nb - 1 : nb / 8 * 7; }
    static size_t buckets_for(size_t cap) {
        if (cap < 4) return 4;
        if (cap < 8) return 8;
        if (cap < 15) return 16;
        size_t want = cap * 8 / 7, p = 1;
        while (p < want) p <<= 1;
        return p;
    }
    void alloc(size_t nb) {
        nb_ = nb;
        ctrl_.assign(nb + W, EMPTY);
        keys_.assign(nb, 0);
        vals_.assign(nb, 0);
        growth_left_ = capacity(nb);
        items_ = 0;
    }
    void set_ctrl(size_t i, uint8_t c) {
        ctrl_[i] = c;
        ctrl_[((i - W) & (nb_ - 1)) + W] = c;  // mirrored tail bytes
    }
    int special_in_group(size_t pos) const {
        for (int b = 0; b < W; b++)
            if (ctrl_[pos + b] & 0x80) return b;
        return -1;
    }
    size_t insert_slot(uint64_t h) const {
        const size_t mask = nb_ - 1;
        size_t pos = (size_t)h & mask, stride = 0;
        for (;;) {
            int b = special_in_group(pos);
            if (b >= 0) {
                size_t idx = (pos + b) & mask;
                if (!(ctrl_[idx] & 0x80)) idx = (size_t)special_in_group(0);  // hit the mirror of a table smaller than a group
                return idx;
            }
            stride += W;
            pos = (pos + stride) & mask;
        }
    }
    long find(uint32_t key) const {
        if (nb_ == 0) return -1;
        const uint64_t h = hash(key);
        const uint8_t tag = (uint8_t)(h >> 57);
        const size_t mask = nb_ - 1;
        size_t pos = (size_t)h & mask, stride = 0;
        for (;;) {
            bool saw_empty = false;
            for (int b = 0; b < W; b++) {
                uint8_t c = ctrl_[pos + b];
                if (c == tag) {
                    size_t idx = (pos + b) & mask;
                    if (!(ctrl_[idx] & 0x80) && keys_[idx] == key) return (long)idx;
                } else if (c == EMPTY) saw_empty = true;
            }
            if (saw_empty) return -1;
            stride += W;
            pos = (pos + stride) & mask;
        }
    }
    void put(uint32_t key, uint32_t val) {
        const uint64_t h = hash(key);
        size_t idx = insert_slot(h);
        if (ctrl_[idx] == EMPTY) growth_left_--;
        set_ctrl(idx, (uint8_t)(h >> 57));
        keys_[idx] = key;
        vals_[idx] = val;
        items_++;
    }
    void resize(size_t cap) {
        std::vector<uint32_t> k, v;
        for_each([&](uint32_t a, uint32_t b) { k.push_back(a); v.push_back(b); });
        alloc(buckets_for(cap));
        for (size_t i = 0; i < k.size(); i++) put(k[i], v[i]);
    }
    void rehash_in_place() {
        const size_t nb = nb_, mask = nb - 1;
        for (size_t i = 0; i < nb; i++) ctrl_[i] = (ctrl_[i] & 0x80) ? EMPTY : DELETED;  // FULL -> DELETED, rest -> EMPTY
        if (nb < (size_t)W) {
            for (size_t i = nb; i < (size_t)W; i++) ctrl_[i] = EMPTY;
            for (size_t i = 0; i < nb; i++) ctrl_[W + i] = ctrl_[i];
        } else {
            for (int i = 0; i < W; i++) ctrl_[nb + i] = ctrl_[i];
        }
        for (size_t i = 0; i < nb; i++) {
            if (ctrl_[i] != DELETED) continue;
            for (;;) {
                const uint64_t h = hash(keys_[i]);
                const size_t ni = insert_slot(h), home = (size_t)h & mask;
                if ((((i - home) & mask) / W) == (((ni - home) & mask) / W)) {
                    set_ctrl(i, (uint8_t)(h >> 57));
                    break;
                }
                const uint8_t prev = ctrl_[ni];
                set_ctrl(ni, (uint8_t)(h >> 57));
                if (prev == EMPTY) {
                    set_ctrl(i, EMPTY);
                    keys_[ni] = keys_[i];
                    vals_[ni] = vals_[i];
                    break;
                }
                std::swap(keys_[i], keys_[ni]);
                std::swap(vals_[i], vals_[ni]);
            }
        }
        growth_left_ = capacity(nb) - items_;
    }
    void reserve_one() {
        if (growth_left_ >= 1) return;
        if (nb_ == 0) {
            alloc(buckets_for(1));
            return;
        }
        const size_t full = capacity(nb_);
        if (items_ + 1 <= full / 2) rehash_in_place();
        else resize(items_ + 1 > full + 1 ? items_ + 1 : full + 1);
    }
};

}  // namespace hqhb
