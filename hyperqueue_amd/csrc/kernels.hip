// gfx950 (MI355X / CDNA4) kernels of the tako scheduling tick.  Hand-written HIP, wave64 throughout.
//
// The ready set lives in HBM as three columns sorted by task id:  task_id u64 | priority u64 | rq u32  (20 B/task).
// Three streaming kernels walk it:
//   K0  distinct_priorities  8 B/task   which Priority values exist                (taskqueue.rs:115-119 BTreeMap keys)
//   K1  level_hist          12 B/task   tasks per (priority level, request) group  (taskqueue.rs:273-302 iter_priority_sizes)
//   K4  select_scatter      12 B/task (+8 B per taken task)  the first take[g] ids of every group, in id order
//                                        (taskqueue.rs:320-355 take_tasks / :304-318 take_tasks_for_prefill)
// All three are HBM/L2-bound integer scans: no MFMA.  Stable ranks inside a group come from wave-private LDS counters
// plus a wave-level "match-any" built from __ballot (one ballot per key bit), so no sort of the ready set is needed.
// K5 expands per-(request, variant, worker) counts into per-worker records (mapping.rs:36-131) — one workgroup per worker.
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

// ------------------------------------------------------------------------------------------------ K1 / K4 shared pieces
// LDS layout: [levels: Lc u64][counters: WPB*G u32]([take G][base G] for K4 when they fit)
template <int WPB, bool SELECT>
__global__ void __launch_bounds__(WPB * 64) k_group_pass(const uint64_t *__restrict__ task_id, const uint64_t *__restrict__ prio,
                                                         const uint32_t *__restrict__ rq, uint64_t n,
                                                         const uint64_t *__restrict__ levels, uint32_t L, uint32_t Q,
                                                         uint32_t tasks_per_wave, uint32_t n_waves, uint32_t lds_levels,
                                                         uint32_t *__restrict__ wave_tab,  // K1: out counts  K4: in offsets   [G][n_waves]
                                                         const uint32_t *__restrict__ take, const uint32_t *__restrict__ base,
                                                         uint64_t *__restrict__ sel_task, uint16_t *__restrict__ sel_level,
                                                         uint32_t *__restrict__ err_flag) {
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t G = L * Q;
    uint64_t *s_levels = reinterpret_cast<uint64_t *>(smem);
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(smem + (size_t)lds_levels * 8) + (threadIdx.x >> 6) * G;
    const uint32_t lane = lane_id();
    const uint32_t wave = blockIdx.x * WPB + (threadIdx.x >> 6);
    for (uint32_t i = threadIdx.x; i < lds_levels; i += blockDim.x) s_levels[i] = levels[i];
    if (wave < n_waves) {
        if (SELECT) { for (uint32_t g = lane; g < G; g += 64) s_cnt[g] = wave_tab[(size_t)g * n_waves + wave]; }
        else { for (uint32_t g = lane; g < G; g += 64) s_cnt[g] = 0; }
    }
    __syncthreads();
    if (wave >= n_waves) return;
    const uint64_t *lvp = lds_levels ? s_levels : levels;
    int nbits = 0; while ((1u << nbits) < G) nbits++;
    const uint64_t begin = (uint64_t)wave * tasks_per_wave;
    const uint64_t end = begin + tasks_per_wave < n ? begin + tasks_per_wave : n;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    for (uint64_t b = begin; b < end; b += 64) {
        uint64_t i = b + lane;
        bool active = i < end;
        uint64_t p = active ? prio[i] : 0;
        uint32_t q = active ? rq[i] : 0;
        uint32_t g = 0;
        if (active) {
            uint32_t lv = level_of(lvp, L, p);
            if (lv >= L || lvp[lv] != p) { if (!SELECT) atomicOr(err_flag, 1u); active = false; }        // priority missing from the level table
            else if (q >= Q) { if (!SELECT) atomicOr(err_flag, 2u); active = false; }                   // request id out of range
            else g = lv * Q + q;
        }
        uint64_t peers = match_any(g, nbits, active);
        if (active) {
            uint32_t before = (uint32_t)__popcll(peers & lt_mask);
            uint32_t cur = s_cnt[g];  // wave-private counter: only this wave's lanes touch it, leaders of distinct groups write distinct words
            if (SELECT) {
                uint32_t rank = cur + before;
                if (rank < take[g]) {
                    uint32_t dst = base[g] + rank;
                    sel_task[dst] = task_id[i];
                    sel_level[dst] = (uint16_t)(g / Q);
                }
            }
            if (before == 0) s_cnt[g] = cur + (uint32_t)__popcll(peers);
        }
    }
    if (!SELECT) {
        // publish this slice's counts, transposed to [G][n_waves] so the scan and K4 read rows contiguously
        for (uint32_t g = lane; g < G; g += 64) wave_tab[(size_t)g * n_waves + wave] = s_cnt[g];
    }
}

// ------------------------------------------------------------------------------------------------ K1b
__global__ void __launch_bounds__(256) k_scan_waves(uint32_t *__restrict__ wave_tab, uint32_t n_waves, uint32_t *__restrict__ hist) {
    __shared__ uint32_t part[256];
    uint32_t *row = wave_tab + (size_t)blockIdx.x * n_waves;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < n_waves; base += 256) {
        uint32_t i = base + threadIdx.x;
        uint32_t v = i < n_waves ? row[i] : 0;
        part[threadIdx.x] = v;
        __syncthreads();
        for (uint32_t off = 1; off < 256; off <<= 1) {  // Hillis-Steele inclusive scan of the tile
            uint32_t t = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
            __syncthreads();
            part[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < n_waves) row[i] = carry + part[threadIdx.x] - v;
        uint32_t tile_sum = part[255];
        __syncthreads();
        carry += tile_sum;
    }
    if (threadIdx.x == 0) hist[blockIdx.x] = carry;
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

// ------------------------------------------------------------------------------------------------ K5
__global__ void __launch_bounds__(256) k_expand_mapping(MapKeys mk, const uint64_t *__restrict__ sel_task,
                                                        const uint16_t *__restrict__ sel_level, const uint64_t *__restrict__ levels,
                                                        uint32_t max_items, uint32_t max_count, uint64_t *__restrict__ rec_task,
                                                        uint8_t *__restrict__ rec_variant, uint8_t *__restrict__ rec_kind,
                                                        uint32_t *__restrict__ err_flag) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint64_t *e_task = reinterpret_cast<uint64_t *>(smem);
    uint64_t *e_prio = e_task + max_items;
    uint16_t *e_meta = reinterpret_cast<uint16_t *>(e_prio + max_items);  // variant | valid << 8
    const uint32_t w = blockIdx.x;
    const uint32_t k0 = mk.wk_off[w], k1 = mk.wk_off[w + 1];
    const uint32_t out0 = mk.out_off[w];
    // new prefills first, in queue order (mapping.rs:266-272)
    uint32_t npf = 0;
    for (uint32_t c = mk.pfl_off[w]; c < mk.pfl_off[w + 1]; c++) {
        uint32_t src = mk.pfl_src[c], cnt = mk.pfl_cnt[c];
        for (uint32_t t = threadIdx.x; t < cnt; t += blockDim.x) {
            rec_task[out0 + npf + t] = sel_task[src + t];
            rec_variant[out0 + npf + t] = 0xFF;
            rec_kind[out0 + npf + t] = 0;  // HQ_REC_PREFILL
        }
        npf += cnt;
    }
    // gather the tasks of every (request, variant) key this worker takes part in, in key order then sweep order
    uint32_t *h_cnt = reinterpret_cast<uint32_t *>(e_meta + max_items + (max_items & 1));  // [max_count + 2] histogram of the counts ahead of us
    uint32_t n = 0;
    for (uint32_t kk = k0; kk < k1; kk++) {
        const uint32_t key = mk.wk_key[kk], pos = mk.wk_pos[kk];
        const uint32_t *cnts = mk.ord_cnt + mk.key_ord_off[key];
        const uint32_t c = cnts[pos];
        const uint32_t maxc = mk.key_t_off[key + 1] - mk.key_t_off[key] - 1;  // T has maxc + 1 entries
        if (n + c > max_items || maxc > max_count) { if (threadIdx.x == 0) atomicExch(err_flag, 2u); return; }
        for (uint32_t t = threadIdx.x; t <= maxc + 1; t += blockDim.x) h_cnt[t] = 0;
        __syncthreads();
        for (uint32_t jj = threadIdx.x; jj < pos; jj += blockDim.x) atomicAdd(&h_cnt[cnts[jj]], 1u);  // workers ahead of us in the Map's iteration order
        __syncthreads();
        const uint32_t q = mk.key_rq[key];
        const uint32_t pfs = mk.rq_pf_start[q], pfn = mk.rq_pf_n[q], seg = mk.key_seg_start[key], sbase = mk.rq_sel_base[q];
        const uint8_t variant = mk.key_variant[key];
        for (uint32_t s = threadIdx.x; s < c; s += blockDim.x) {
            uint32_t rank = 0;  // workers before this one that still hold a count in sweep s: counts > s
            for (uint32_t cc = s + 1; cc <= maxc; cc++) rank += h_cnt[cc];
            uint32_t k = mk.t_sweep[mk.key_t_off[key] + s] + rank;  // index of the task inside the key's take_tasks() vector
            uint32_t p = seg + k;                                    // position in the queue's logical sequence
            uint32_t e = n + s;
            if (p >= pfs && p < pfs + pfn) {                         // an already-prefilled task: retract/redirect is host work
                e_meta[e] = 0; e_task[e] = 0; e_prio[e] = 0;
            } else {
                uint32_t src = sbase + (p >= pfs + pfn ? p - pfn : p);
                e_task[e] = sel_task[src];
                e_prio[e] = levels[sel_level[src]];
                e_meta[e] = (uint16_t)(variant | 0x100u);
            }
        }
        __syncthreads();
        n += c;
    }
    // stable sort by priority descending (mapping.rs:128-131) by rank counting
    for (uint32_t e = threadIdx.x; e < n; e += blockDim.x) {
        uint16_t meta = e_meta[e];
        if (!(meta & 0x100u)) continue;
        uint64_t pr = e_prio[e];
        uint32_t pos = 0;
        for (uint32_t o = 0; o < n; o++) {
            if (!(e_meta[o] & 0x100u)) continue;
            uint64_t po = e_prio[o];
            pos += (po > pr || (po == pr && o < e)) ? 1u : 0u;
        }
        uint32_t dst = out0 + npf + pos;
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

template <bool SELECT>
static hipError_t launch_group_pass(const uint64_t *task_id, const uint64_t *prio, const uint32_t *rq, uint64_t n, const uint64_t *levels,
                              uint32_t L, uint32_t Q, WaveGeom geom, uint32_t *wave_tab, const uint32_t *take, const uint32_t *base,
                              uint64_t *sel_task, uint16_t *sel_level, uint32_t *err_flag, hipStream_t s) {
    if (n == 0 || geom.n_waves == 0) return hipSuccess;
    uint32_t G = L * Q, ll = lds_levels_for(L);
    hipError_t e;
    if (geom.waves_per_block == 4) {
        size_t lds = (size_t)ll * 8 + (size_t)4 * G * 4;
        auto kern = k_group_pass<4, SELECT>;
        if (lds > 48 * 1024 && (e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3((geom.n_waves + 3) / 4), dim3(256), lds, s, task_id, prio, rq, n, levels, L, Q,
                           geom.tasks_per_wave, geom.n_waves, ll, wave_tab, take, base, sel_task, sel_level, err_flag);
    } else {
        size_t lds = (size_t)ll * 8 + (size_t)G * 4;
        auto kern = k_group_pass<1, SELECT>;
        if (lds > 48 * 1024 && (e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(geom.n_waves), dim3(64), lds, s, task_id, prio, rq, n, levels, L, Q, geom.tasks_per_wave,
                           geom.n_waves, ll, wave_tab, take, base, sel_task, sel_level, err_flag);
    }
    return hipGetLastError();
}

hipError_t level_hist(const uint64_t *prio, const uint32_t *rq, uint64_t n, const uint64_t *levels, uint32_t L, uint32_t Q, WaveGeom geom,
                uint32_t *wave_cnt, uint32_t *err_flag, hipStream_t s) {
    return launch_group_pass<false>(nullptr, prio, rq, n, levels, L, Q, geom, wave_cnt, nullptr, nullptr, nullptr, nullptr, err_flag, s);
}

hipError_t scan_waves(uint32_t *wave_cnt, uint32_t n_waves, uint32_t G, uint32_t *hist, hipStream_t s) {
    if (G == 0) return hipSuccess;
    hipLaunchKernelGGL(k_scan_waves, dim3(G), dim3(256), 0, s, wave_cnt, n_waves, hist);
    return hipGetLastError();
}

hipError_t worker_eval(const uint64_t *total, const uint64_t *free_, const int64_t *remaining_ns, uint32_t W, uint32_t R, RequestTable rt,
                 uint8_t *flags, uint32_t *tmc, hipStream_t s) {
    uint32_t n = W * rt.n_variants;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_worker_eval, dim3((n + 255) / 256), dim3(256), 0, s, total, free_, remaining_ns, W, R, rt, flags, tmc);
    return hipGetLastError();
}

hipError_t select_scatter(const uint64_t *task_id, const uint64_t *prio, const uint32_t *rq, uint64_t n, const uint64_t *levels, uint32_t L,
                    uint32_t Q, WaveGeom geom, const uint32_t *wave_off, const uint32_t *take, const uint32_t *base, uint64_t *sel_task,
                    uint16_t *sel_level, hipStream_t s) {
    return launch_group_pass<true>(task_id, prio, rq, n, levels, L, Q, geom, const_cast<uint32_t *>(wave_off), take, base, sel_task, sel_level,
                            nullptr, s);
}

hipError_t expand_mapping(MapKeys mk, uint32_t W, const uint64_t *sel_task, const uint16_t *sel_level, const uint64_t *levels,
                    uint32_t max_items, uint32_t max_count, uint64_t *rec_task, uint8_t *rec_variant, uint8_t *rec_kind, uint32_t *err_flag, hipStream_t s) {
    if (W == 0) return hipSuccess;
    size_t lds = (size_t)max_items * 18 + 16 + ((size_t)max_count + 4) * 4;
    hipError_t e;
    if (lds > 48 * 1024 && (e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_expand_mapping), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_expand_mapping, dim3(W), dim3(256), lds, s, mk, sel_task, sel_level, levels, max_items, max_count, rec_task, rec_variant,
                       rec_kind, err_flag);
    return hipGetLastError();
}

}  // namespace hqk
