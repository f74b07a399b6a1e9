// gfx950 (MI355X / CDNA4) kernels of the tako scheduling tick.  Hand-written HIP, wave64 throughout.
//
// The ready set lives in HBM as three columns sorted by task id:  task_id u64 | priority u64 | rq u32  (20 B/task).
// Streaming kernels over it:
//   K0  distinct_priorities  8 B/task   which Priority values exist                (taskqueue.rs:115-119 BTreeMap keys)
//   K1  level_hist          12 B/task   tasks per (priority level, request) group  (taskqueue.rs:273-302 iter_priority_sizes);
//                                        leaves a 2 B/task group key behind for K4
//   K4  select_scatter       2 B/task (+8 B per taken task)  the first take[g] ids of every group, in id order
//                                        (taskqueue.rs:320-355 take_tasks / :304-318 take_tasks_for_prefill)
// All are HBM/L2-bound integer scans: no MFMA.  Stable ranks inside a group come from wave-private LDS counters
// plus a wave-level "match-any" built from __ballot (one ballot per key bit), so no sort of the ready set is needed.
// K5a/K5b expand per-(request, variant, worker) counts into per-worker records (mapping.rs:36-131): K5a one wavefront per
// (key, sweep) builds ballot bit rows of the round-robin, K5b one workgroup per worker gathers its records.
#include "kernels.h"

namespace hqk {

namespace {

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

// lanes (among `active`) holding the same key as this lane: one __ballot per key bit
__device__ __forceinline__ uint64_t match_any(uint32_t key, int nbits, bool active) {
    uint64_t m = __ballot(active);
    for (int b = 0; b < nbits; b++) {
        bool bit = (key >> b) & 1u;
        uint64_t bal = __ballot(active && bit);
        m &= bit ? bal : ~bal;
    }
    return m;
}

// index of `p` in the descending table lv[0..L)  (L >= 1, p is known to be present)
__device__ __forceinline__ uint32_t level_of(const uint64_t *lv, uint32_t L, uint64_t p) {
    uint32_t lo = 0, hi = L;  // first index with lv[idx] <= p
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if (lv[mid] > p) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// ------------------------------------------------------------------------------------------------ K0
__global__ void __launch_bounds__(256) k_distinct_priorities(const uint64_t *__restrict__ prio, uint64_t n,
                                                             uint64_t *__restrict__ set, uint32_t *__restrict__ flags) {
    __shared__ unsigned long long cache[256];  // block-local claim table: one wave per block publishes a given value
    for (int i = threadIdx.x; i < 256; i += blockDim.x) cache[i] = PRIO_EMPTY;
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t rounds = (n + 4 * stride - 1) / (4 * stride);
    uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (uint64_t r = 0; r < rounds; r++, i0 += 4 * stride) {
        uint64_t pv[4];
        bool av[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { uint64_t i = i0 + u * stride; av[u] = i < n; pv[u] = av[u] ? prio[i] : 0; }  // 4 independent loads in flight
#pragma unroll
        for (int u = 0; u < 4; u++) {
            bool active = av[u];
            uint64_t p = pv[u];
            if (active && p == PRIO_EMPTY) { atomicOr(&flags[0], 1u); active = false; }
            if (active && cache[mix64(p) & 255u] == p) active = false;
            uint64_t todo = __ballot(active);
            while (todo) {  // one lane per distinct value of the wave
                int first = __ffsll((long long)todo) - 1;
                uint64_t lead = __shfl((unsigned long long)p, first, 64);
                bool same = active && p == lead;
                if ((int)lane_id() == first) {
                    uint64_t h = mix64(lead);
                    unsigned long long claimed = atomicCAS(&cache[h & 255u], (unsigned long long)PRIO_EMPTY, (unsigned long long)lead);
                    if (claimed != lead) {  // first wave of this block to see the value (or a cache collision): publish globally
                        uint32_t slot = (uint32_t)h & (PRIO_SET_CAP - 1);
                        bool done = false;
                        for (uint32_t probe = 0; probe < PRIO_SET_CAP; probe++) {
                            uint64_t cur = set[slot];
                            if (cur == lead) { done = true; break; }
                            if (cur == PRIO_EMPTY) {
                                uint64_t old = atomicCAS((unsigned long long *)&set[slot], (unsigned long long)PRIO_EMPTY, (unsigned long long)lead);
                                if (old == PRIO_EMPTY || old == lead) { done = true; break; }
                            }
                            slot = (slot + 1) & (PRIO_SET_CAP - 1);
                        }
                        if (!done) atomicExch(&flags[1], 1u);
                    }
                }
                todo &= ~__ballot(same);
                active = active && !same;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ K0b
static const uint32_t LEVEL_CAP = hqk::MAX_LEVELS;  // distinct priority levels one tick can carry (32 KiB of LDS)

__global__ void __launch_bounds__(1024) k_sort_levels(const uint64_t *__restrict__ set, const uint32_t *__restrict__ flags,
                                                      uint64_t *__restrict__ levels, uint32_t *__restrict__ n_levels) {
    extern __shared__ uint64_t lv[];  // LEVEL_CAP entries
    __shared__ uint32_t cnt;
    if (threadIdx.x == 0) cnt = 0;
    for (uint32_t i = threadIdx.x; i < LEVEL_CAP; i += blockDim.x) lv[i] = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < PRIO_SET_CAP; i += blockDim.x) {
        uint64_t v = set[i];
        if (v != PRIO_EMPTY) {
            uint32_t k = atomicAdd(&cnt, 1u);
            if (k < LEVEL_CAP) lv[k] = v;
        }
    }
    __syncthreads();
    uint32_t n = cnt;
    if (n > LEVEL_CAP) { if (threadIdx.x == 0) n_levels[0] = 0xFFFFFFFFu; return; }
    uint32_t P = 1; while (P < n) P <<= 1;
    // bitonic sort, descending; the zero padding sinks to the end
    for (uint32_t k = 2; k <= P; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = threadIdx.x; i < P; i += blockDim.x) {
                uint32_t ixj = i ^ j;
                if (ixj > i) {
                    uint64_t a = lv[i], b = lv[ixj];
                    bool desc = (i & k) == 0;
                    if (desc ? (a < b) : (a > b)) { lv[i] = b; lv[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
    uint32_t shift = (flags[0] & 1u) ? 1u : 0u;  // Priority == u64::MAX present: it is the top level
    if (threadIdx.x == 0) { if (shift) levels[0] = PRIO_EMPTY; n_levels[0] = n + shift; }
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) levels[i + shift] = lv[i];
}

// ------------------------------------------------------------------------------------------------ K1
// LDS layout: [levels: lds_levels u64][counters: WPB*G u32].  Every wavefront owns a contiguous slice of the ready set and a
// private counter row, so no block-level synchronisation is needed after the level table is staged.
template <int WPB>
__global__ void __launch_bounds__(WPB * 64) k_level_hist(const uint64_t *__restrict__ prio, const uint32_t *__restrict__ rq, uint64_t n,
                                                         const uint64_t *__restrict__ levels, uint32_t L, uint32_t Q,
                                                         uint32_t tasks_per_wave, uint32_t n_waves, uint32_t stride, uint32_t lds_levels,
                                                         uint32_t *__restrict__ wave_tab, uint16_t *__restrict__ gkey,
                                                         uint32_t *__restrict__ err_flag) {
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t G = L * Q;
    uint64_t *s_levels = reinterpret_cast<uint64_t *>(smem);
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(smem + (size_t)lds_levels * 8) + (threadIdx.x >> 6) * G;
    const uint32_t lane = lane_id();
    const uint32_t wave = blockIdx.x * WPB + (threadIdx.x >> 6);
    for (uint32_t i = threadIdx.x; i < lds_levels; i += blockDim.x) s_levels[i] = levels[i];
    for (uint32_t g = lane; g < G; g += 64) s_cnt[g] = 0;
    __syncthreads();
    if (wave >= n_waves) return;
    const uint64_t *lvp = lds_levels ? s_levels : levels;
    const uint64_t lv0 = lvp[0];
    const uint64_t begin = (uint64_t)wave * tasks_per_wave;
    const uint64_t end = begin + tasks_per_wave < n ? begin + tasks_per_wave : n;
    uint32_t err = 0;
    for (uint64_t b = begin; b < end; b += 256) {
        uint64_t pv[4];
        uint32_t qv[4];
        bool av[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {  // 8 independent loads in flight per lane before the first use
            uint64_t i = b + (uint64_t)u * 64 + lane;
            av[u] = i < end;
            pv[u] = av[u] ? prio[i] : 0;
            qv[u] = av[u] ? rq[i] : 0;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (!av[u]) continue;
            uint64_t i = b + (uint64_t)u * 64 + lane;
            uint32_t lv = L == 1 ? (pv[u] == lv0 ? 0u : 1u) : level_of(lvp, L, pv[u]);
            uint16_t key = GKEY_INVALID;
            if (lv >= L || lvp[lv] != pv[u]) err |= 1u;       // priority missing from the level table
            else if (qv[u] >= Q) err |= 2u;                   // request id out of range
            else { uint32_t g = lv * Q + qv[u]; key = (uint16_t)g; atomicAdd(&s_cnt[g], 1u); }
            gkey[i] = key;
        }
    }
    if (err) atomicOr(err_flag, err);
    // publish this slice's counts, transposed to [G][stride] so the scan and K4 read rows contiguously
    for (uint32_t g = lane; g < G; g += 64) wave_tab[(size_t)g * stride + wave] = s_cnt[g];
}

// ------------------------------------------------------------------------------------------------ K1b
// One wavefront per group row: 16 consecutive entries per lane (4 x dwordx4), local prefix + one wave scan per 1024 entries.
__global__ void __launch_bounds__(256) k_scan_rows(uint32_t *__restrict__ wave_tab, uint32_t n_waves, uint32_t stride, uint32_t G,
                                                   uint32_t *__restrict__ hist) {
    const uint32_t row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= G) return;
    const uint32_t lane = lane_id();
    uint32_t *r = wave_tab + (size_t)row * stride;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < n_waves; base += 1024) {
        const uint32_t i0 = base + lane * 16;
        uint32_t v[16];
        if (i0 < stride) {  // stride is a multiple of 16: the whole 16-entry run is inside the row
            const uint4 *src = reinterpret_cast<const uint4 *>(r + i0);
#pragma unroll
            for (int k = 0; k < 4; k++) { uint4 t = src[k]; v[4 * k] = t.x; v[4 * k + 1] = t.y; v[4 * k + 2] = t.z; v[4 * k + 3] = t.w; }
        } else {
#pragma unroll
            for (int k = 0; k < 16; k++) v[k] = 0;
        }
        uint32_t run = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) { uint32_t t = (i0 + k < n_waves) ? v[k] : 0u; v[k] = run; run += t; }  // padding entries count as 0
        uint32_t incl = run;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { uint32_t t = __shfl_up(incl, off, 64); if ((int)lane >= off) incl += t; }
        const uint32_t excl = carry + incl - run;
        const uint32_t total = __shfl(incl, 63, 64);
        if (i0 < stride) {
            uint4 *dst = reinterpret_cast<uint4 *>(r + i0);
#pragma unroll
            for (int k = 0; k < 4; k++) dst[k] = make_uint4(v[4 * k] + excl, v[4 * k + 1] + excl, v[4 * k + 2] + excl, v[4 * k + 3] + excl);
        }
        carry += total;
    }
    if (lane == 0) hist[row] = carry;
}

// ------------------------------------------------------------------------------------------------ K4
// LDS layout: [counters: WPB*G u32]([take G][base G] when LDS_PLAN).
template <int WPB, bool LDS_PLAN>
__global__ void __launch_bounds__(WPB * 64) k_select(const uint64_t *__restrict__ task_id, const uint16_t *__restrict__ gkey, uint64_t n,
                                                     uint32_t Q, uint32_t G, uint32_t tasks_per_wave, uint32_t n_waves, uint32_t stride,
                                                     const uint32_t *__restrict__ wave_off, const uint32_t *__restrict__ take,
                                                     const uint32_t *__restrict__ base, uint64_t *__restrict__ sel_task,
                                                     uint16_t *__restrict__ sel_level) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t *s_all = reinterpret_cast<uint32_t *>(smem);
    uint32_t *s_cnt = s_all + (threadIdx.x >> 6) * G;
    const uint32_t *tk = take, *bs = base;
    if (LDS_PLAN) {
        uint32_t *s_take = s_all + WPB * G, *s_base = s_take + G;
        for (uint32_t g = threadIdx.x; g < G; g += blockDim.x) { s_take[g] = take[g]; s_base[g] = base[g]; }
        tk = s_take; bs = s_base;
    }
    const uint32_t lane = lane_id();
    const uint32_t wave = blockIdx.x * WPB + (threadIdx.x >> 6);
    bool need = false;
    if (wave < n_waves) {
        for (uint32_t g = lane; g < G; g += 64) { uint32_t o = wave_off[(size_t)g * stride + wave]; s_cnt[g] = o; need = need || o < take[g]; }
    }
    __syncthreads();
    if (wave >= n_waves || !__ballot(need)) return;  // every group this slice could feed is already exhausted by earlier slices
    int nbits = 0; while ((1u << nbits) < G) nbits++;
    const uint64_t begin = (uint64_t)wave * tasks_per_wave;
    const uint64_t end = begin + tasks_per_wave < n ? begin + tasks_per_wave : n;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    for (uint64_t b = begin; b < end; b += 256) {
        uint16_t kv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { uint64_t i = b + (uint64_t)u * 64 + lane; kv[u] = i < end ? gkey[i] : GKEY_INVALID; }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (b + (uint64_t)u * 64 >= end) break;  // wave-uniform
            const uint64_t i = b + (uint64_t)u * 64 + lane;
            const uint32_t g = kv[u];
            const bool active = g != GKEY_INVALID;
            const uint64_t peers = match_any(g, nbits, active);
            if (active) {
                const uint32_t before = (uint32_t)__popcll(peers & lt_mask);
                const uint32_t cur = s_cnt[g];  // wave-private counter: leaders of distinct groups write distinct words
                const uint32_t rank = cur + before;
                if (rank < tk[g]) {
                    const uint32_t dst = bs[g] + rank;
                    sel_task[dst] = task_id[i];
                    sel_level[dst] = (uint16_t)(g / Q);
                }
                if (before == 0) s_cnt[g] = cur + (uint32_t)__popcll(peers);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ K2
__global__ void __launch_bounds__(256) k_worker_eval(const uint64_t *__restrict__ total, const uint64_t *__restrict__ free_,
                                                     const int64_t *__restrict__ remaining_ns, uint32_t W, uint32_t R, RequestTable rt,
                                                     uint8_t *__restrict__ flags, uint32_t *__restrict__ tmc) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= W * rt.n_variants) return;
    uint32_t w = t / rt.n_variants, v = t % rt.n_variants;
    bool imm = true, cap = true;
    uint64_t best = 0xFFFFFFFFFFFFFFFFull;
    bool any = false;
    for (uint32_t e = rt.variant_entry_off[v]; e < rt.variant_entry_off[v + 1]; e++) {
        uint32_t r = rt.entry_resource[e];
        uint64_t f = r < R ? free_[(size_t)w * R + r] : 0, tt = r < R ? total[(size_t)w * R + r] : 0;
        uint64_t c;
        if (rt.entry_kind[e] == 0) {  // amount
            uint64_t a = rt.entry_amount[e];
            imm = imm && a <= f; cap = cap && a <= tt;
            c = f / a; if (c > 1024) c = 1024;            // MAX_TASK_PER_WORKER  workerload.rs:12,131
        } else {                                           // All: min_amount = 1 fraction  request.rs:34-36
            imm = imm && f >= 1; cap = cap && tt >= 1;
            c = f == 0 ? 0 : 1;                            // workerload.rs:133-141
        }
        if (!any || c < best) best = c;
        any = true;
    }
    int64_t rem = remaining_ns[w];
    bool time_ok = rem == INT64_MAX || (rem >= 0 && (uint64_t)rem >= rt.variant_min_time_ns[v]);  // worker.rs:320-326
    flags[t] = (imm ? 1 : 0) | (cap ? 2 : 0) | (time_ok ? 4 : 0);
    tmc[t] = any ? (uint32_t)best : 0;
}

// ------------------------------------------------------------------------------------------------ K5a
// One wavefront per (key, sweep) unit: bit row [c_j > s] over the key's workers in Map iteration order (one __ballot per 64
// workers), exclusive prefix popcount per word, and T_k(s) = sum_j min(c_j, s).
__global__ void __launch_bounds__(256) k_sweep_bits(MapKeys mk, uint32_t n_units) {
    const uint32_t unit = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (unit >= n_units) return;
    const uint32_t lane = lane_id();
    uint32_t lo = 0, hi = mk.n_keys;  // last key with key_t_off[k] <= unit
    while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (mk.key_t_off[mid] <= unit) lo = mid; else hi = mid; }
    const uint32_t k = lo, s = unit - mk.key_t_off[k];
    const uint32_t nk = mk.key_ord_off[k + 1] - mk.key_ord_off[k];
    const uint32_t *cnts = mk.ord_cnt + mk.key_ord_off[k];
    const uint32_t words = (nk + 63) >> 6;
    const size_t row = (size_t)mk.key_bits_off[k] + (size_t)s * words;
    uint32_t running = 0, summin = 0;
    for (uint32_t wd = 0; wd < words; wd++) {
        const uint32_t j = wd * 64 + lane;
        const uint32_t c = j < nk ? cnts[j] : 0u;
        const uint64_t m = __ballot(c > s);
        summin += c < s ? c : s;
        if (lane == 0) { mk.bits[row + wd] = m; mk.pre[row + wd] = running; }
        running += (uint32_t)__popcll(m);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) summin += __shfl_xor(summin, off, 64);
    if (lane == 0) mk.t_sweep[unit] = summin;
}

// ------------------------------------------------------------------------------------------------ K5b
// One workgroup per worker.  LDS: e_task u64[max_items] | e_lvl u16[max_items] | e_meta u16[max_items] | k_start u32[n_keys+1]
// | k_pos u32[n_keys] | k_cnt u32[n_keys] | misc u32[4]
__global__ void __launch_bounds__(256) k_expand_mapping(MapKeys mk, uint32_t W, const uint64_t *__restrict__ sel_task,
                                                        const uint16_t *__restrict__ sel_level, uint32_t max_items,
                                                        uint64_t *__restrict__ rec_task, uint8_t *__restrict__ rec_variant,
                                                        uint8_t *__restrict__ rec_kind, uint32_t *__restrict__ err_flag) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint64_t *e_task = reinterpret_cast<uint64_t *>(smem);
    uint16_t *e_lvl = reinterpret_cast<uint16_t *>(e_task + max_items);
    uint16_t *e_meta = e_lvl + max_items;  // variant | valid << 8
    uint32_t *k_start = reinterpret_cast<uint32_t *>(e_meta + max_items);  // 12 B per item: stays 4-byte aligned
    uint32_t *k_pos = k_start + mk.n_keys + 1;
    uint32_t *k_cnt = k_pos + mk.n_keys;
    uint32_t *misc = k_cnt + mk.n_keys;  // [0] min level, [1] max level, [2] holes
    const uint32_t w = blockIdx.x, nkeys = mk.n_keys, lane = lane_id();
    const uint32_t out0 = mk.out_off[w];
    for (uint32_t k = threadIdx.x; k < nkeys; k += blockDim.x) {
        const uint32_t pos = mk.wpos[(size_t)k * W + w];
        k_pos[k] = pos;
        k_cnt[k] = pos == 0xFFFFFFFFu ? 0u : mk.ord_cnt[mk.key_ord_off[k] + pos];
    }
    if (threadIdx.x == 0) { misc[0] = 0xFFFFu; misc[1] = 0; misc[2] = 0; }
    __syncthreads();
    if (threadIdx.x < 64) {  // exclusive scan of the per-key counts
        uint32_t carry = 0;
        for (uint32_t base = 0; base < nkeys; base += 64) {
            const uint32_t k = base + lane;
            const uint32_t c = k < nkeys ? k_cnt[k] : 0u;
            uint32_t incl = c;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { uint32_t t = __shfl_up(incl, off, 64); if ((int)lane >= off) incl += t; }
            if (k < nkeys) k_start[k] = carry + incl - c;
            carry += __shfl(incl, 63, 64);
        }
        if (lane == 0) k_start[nkeys] = carry;
    }
    __syncthreads();
    const uint32_t n = k_start[nkeys];
    if (n > max_items) { if (threadIdx.x == 0) atomicExch(err_flag, 2u); return; }
    // new prefills first, in queue (request id) order (mapping.rs:266-272)
    uint32_t npf = 0;
    for (uint32_t pi = 0; pi < mk.n_pfq; pi++) {
        const uint32_t j = mk.pfl_j[(size_t)pi * W + w];
        if (j == 0xFFFFFFFFu) continue;
        const uint32_t cnt = mk.pfq_size[pi], src = mk.pfq_src[pi] + j * cnt;
        for (uint32_t t = threadIdx.x; t < cnt; t += blockDim.x) {
            rec_task[out0 + npf + t] = sel_task[src + t];
            rec_variant[out0 + npf + t] = 0xFF;
            rec_kind[out0 + npf + t] = 0;  // HQ_REC_PREFILL
        }
        npf += cnt;
    }
    // gather: item e = (key k, sweep s) in key order then sweep order
    for (uint32_t e = threadIdx.x; e < n; e += blockDim.x) {
        uint32_t lo = 0, hi = nkeys;  // last key with k_start[k] <= e (it is the non-empty one)
        while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (k_start[mid] <= e) lo = mid; else hi = mid; }
        const uint32_t k = lo, s = e - k_start[k], pos = k_pos[k];
        const uint32_t nk = mk.key_ord_off[k + 1] - mk.key_ord_off[k];
        const uint32_t words = (nk + 63) >> 6;
        const size_t cell = (size_t)mk.key_bits_off[k] + (size_t)s * words + (pos >> 6);
        const uint32_t rank = mk.pre[cell] + (uint32_t)__popcll(mk.bits[cell] & ((1ull << (pos & 63)) - 1ull));
        const uint32_t idx = mk.t_sweep[mk.key_t_off[k] + s] + rank;  // index inside the key's take_tasks() vector
        const uint32_t q = mk.key_rq[k];
        const uint32_t pfs = mk.rq_pf_start[q], pfn = mk.rq_pf_n[q];
        const uint32_t p = mk.key_seg_start[k] + idx;                 // position in the queue's logical sequence
        if (p >= pfs && p < pfs + pfn) {                               // an already-prefilled task: retract/redirect is host work
            e_meta[e] = 0; e_task[e] = 0; e_lvl[e] = 0;
            misc[2] = 1;
        } else {
            const uint32_t src = mk.rq_sel_base[q] + (p >= pfs + pfn ? p - pfn : p);
            const uint16_t lv = sel_level[src];
            e_task[e] = sel_task[src];
            e_lvl[e] = lv;
            e_meta[e] = (uint16_t)(mk.key_variant[k] | 0x100u);
            atomicMin(&misc[0], (uint32_t)lv);
            atomicMax(&misc[1], (uint32_t)lv);
        }
    }
    __syncthreads();
    const bool trivial = misc[2] == 0 && misc[0] >= misc[1];  // one priority level, no holes: already in final order
    // stable sort by priority descending == level ascending (mapping.rs:128-131) by rank counting
    for (uint32_t e = threadIdx.x; e < n; e += blockDim.x) {
        const uint16_t meta = e_meta[e];
        if (!(meta & 0x100u)) continue;
        uint32_t pos = e;
        if (!trivial) {
            const uint16_t lv = e_lvl[e];
            pos = 0;
            for (uint32_t o = 0; o < n; o++) {
                if (!(e_meta[o] & 0x100u)) continue;
                const uint16_t lo_ = e_lvl[o];
                pos += (lo_ < lv || (lo_ == lv && o < e)) ? 1u : 0u;
            }
        }
        const uint32_t dst = out0 + npf + pos;
        rec_task[dst] = e_task[e];
        rec_variant[dst] = (uint8_t)(meta & 0xFFu);
        rec_kind[dst] = 1;  // HQ_REC_ASSIGN
    }
}

}  // namespace

// ================================================================================================ host wrappers
hipError_t distinct_priorities(const uint64_t *prio, uint64_t n, uint64_t *set, uint32_t *flags, hipStream_t s) {
    if (n == 0) return hipSuccess;
    uint64_t blocks = (n + 1023) / 1024;
    if (blocks > 512) blocks = 512;  // two blocks per CU: every block publishes each distinct value once, so fewer blocks = fewer same-address atomics
    hipLaunchKernelGGL(k_distinct_priorities, dim3((unsigned)blocks), dim3(256), 0, s, prio, n, set, flags);
    return hipGetLastError();
}

hipError_t sort_levels(const uint64_t *set, const uint32_t *flags, uint64_t *levels, uint32_t *n_levels, hipStream_t s) {
    hipLaunchKernelGGL(k_sort_levels, dim3(1), dim3(1024), LEVEL_CAP * 8, s, set, flags, levels, n_levels);
    return hipGetLastError();
}

static uint32_t lds_levels_for(uint32_t L) { return L <= 1024 ? L : 0; }

hipError_t level_hist(const uint64_t *prio, const uint32_t *rq, uint64_t n, const uint64_t *levels, uint32_t L, uint32_t Q, WaveGeom geom,
                uint32_t *wave_tab, uint16_t *gkey, uint32_t *err_flag, hipStream_t s) {
    if (n == 0 || geom.n_waves == 0) return hipSuccess;
    const uint32_t G = L * Q, ll = lds_levels_for(L);
    hipError_t e;
    if (geom.waves_per_block == 4) {
        size_t lds = (size_t)ll * 8 + (size_t)4 * G * 4;
        auto kern = k_level_hist<4>;
        if (lds > 48 * 1024 && (e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3((geom.n_waves + 3) / 4), dim3(256), lds, s, prio, rq, n, levels, L, Q, geom.tasks_per_wave, geom.n_waves,
                           geom.tab_stride, ll, wave_tab, gkey, err_flag);
    } else {
        size_t lds = (size_t)ll * 8 + (size_t)G * 4;
        auto kern = k_level_hist<1>;
        if (lds > 48 * 1024 && (e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(geom.n_waves), dim3(64), lds, s, prio, rq, n, levels, L, Q, geom.tasks_per_wave, geom.n_waves,
                           geom.tab_stride, ll, wave_tab, gkey, err_flag);
    }
    return hipGetLastError();
}

hipError_t scan_waves(uint32_t *wave_tab, WaveGeom geom, uint32_t G, uint32_t *hist, hipStream_t s) {
    if (G == 0) return hipSuccess;
    hipLaunchKernelGGL(k_scan_rows, dim3((G + 3) / 4), dim3(256), 0, s, wave_tab, geom.n_waves, geom.tab_stride, G, hist);
    return hipGetLastError();
}

hipError_t select_scatter(const uint64_t *task_id, const uint16_t *gkey, uint64_t n, uint32_t Q, uint32_t G, WaveGeom geom,
                    const uint32_t *wave_off, const uint32_t *take, const uint32_t *base, uint64_t *sel_task, uint16_t *sel_level,
                    hipStream_t s) {
    if (n == 0 || geom.n_waves == 0 || G == 0) return hipSuccess;
    hipError_t e;
    if (geom.waves_per_block == 4) {
        size_t lds = (size_t)6 * G * 4;
        auto kern = k_select<4, true>;
        if (lds > 48 * 1024 && (e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3((geom.n_waves + 3) / 4), dim3(256), lds, s, task_id, gkey, n, Q, G, geom.tasks_per_wave, geom.n_waves,
                           geom.tab_stride, wave_off, take, base, sel_task, sel_level);
    } else {
        size_t lds = (size_t)G * 4;
        auto kern = k_select<1, false>;
        if (lds > 48 * 1024 && (e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(geom.n_waves), dim3(64), lds, s, task_id, gkey, n, Q, G, geom.tasks_per_wave, geom.n_waves,
                           geom.tab_stride, wave_off, take, base, sel_task, sel_level);
    }
    return hipGetLastError();
}

hipError_t worker_eval(const uint64_t *total, const uint64_t *free_, const int64_t *remaining_ns, uint32_t W, uint32_t R, RequestTable rt,
                 uint8_t *flags, uint32_t *tmc, hipStream_t s) {
    uint32_t n = W * rt.n_variants;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_worker_eval, dim3((n + 255) / 256), dim3(256), 0, s, total, free_, remaining_ns, W, R, rt, flags, tmc);
    return hipGetLastError();
}

hipError_t sweep_bits(MapKeys mk, uint32_t n_units, hipStream_t s) {
    if (n_units == 0) return hipSuccess;
    hipLaunchKernelGGL(k_sweep_bits, dim3((n_units + 3) / 4), dim3(256), 0, s, mk, n_units);
    return hipGetLastError();
}

size_t expand_mapping_lds(uint32_t max_items, uint32_t n_keys) {
    return (size_t)max_items * 12 + 4 + ((size_t)3 * n_keys + 1 + 4) * 4;
}

hipError_t expand_mapping(MapKeys mk, uint32_t W, const uint64_t *sel_task, const uint16_t *sel_level, uint32_t max_items,
                    uint64_t *rec_task, uint8_t *rec_variant, uint8_t *rec_kind, uint32_t *err_flag, hipStream_t s) {
    if (W == 0) return hipSuccess;
    size_t lds = expand_mapping_lds(max_items, mk.n_keys);
    hipError_t e;
    if (lds > 48 * 1024 && (e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_expand_mapping), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_expand_mapping, dim3(W), dim3(256), lds, s, mk, W, sel_task, sel_level, max_items, rec_task, rec_variant, rec_kind, err_flag);
    return hipGetLastError();
}

}  // namespace hqk
