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
#include <hip/hip_ext.h>

namespace hqk {
namespace { thread_local LaunchTimer g_timer; }
void time_next_launch(hipEvent_t start, hipEvent_t stop) { g_timer.start = start; g_timer.stop = stop; }
LaunchTimer take_launch_timer() { LaunchTimer t = g_timer; g_timer = LaunchTimer{}; return t; }
}  // namespace hqk
// a measured launch: bracketed by the pending timer's events (if any) at the dispatch
#define HQK_TIMED_LAUNCH(kern, grid, block, lds, s, ...)                                                  \
    do {                                                                                                  \
        hqk::LaunchTimer t_ = hqk::take_launch_timer();                                                   \
        hipExtLaunchKernelGGL(kern, grid, block, lds, s, t_.start, t_.stop, 0, __VA_ARGS__);              \
    } while (0)

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
static const uint32_t LEVEL_CAP = hqk::MAX_LEVELS;  // distinct priority levels one tick can carry (32 KiB of LDS in k_sort_levels; the compact list behind the set)
__global__ void __launch_bounds__(256) k_distinct_priorities(const uint64_t *__restrict__ prio, const uint32_t *__restrict__ rq, uint64_t n,
                                                             uint64_t *__restrict__ set, uint32_t *__restrict__ flags) {
    __shared__ unsigned long long cache[256];  // block-local claim table: one wave per block publishes a given value
    for (int i = threadIdx.x; i < 256; i += blockDim.x) cache[i] = PRIO_EMPTY;
    __syncthreads();
    // Workgroup b reads the tasks [1024 b, 1024 b + 1024) — the very range workgroup b of K1 (k_level_hist: four wavefronts x 256 tasks) reads next, and workgroups are
    // dealt to the XCDs round-robin by their index: a tick that has to rediscover its levels leaves every slice of the priority / request columns in the L2 of the XCD
    // that scans it a moment later.  (Grid-stride over the whole array, as this kernel was until round 6, spread every XCD's reads over all slices: K1 behind it took
    // 7.6 us against 5.5 behind nothing — profiles/r06.)
    const uint64_t stride = 256;
    const uint64_t rounds = 1;
    uint64_t i0 = (uint64_t)blockIdx.x * 1024 + threadIdx.x;
    for (uint64_t r = 0; r < rounds; r++) {
        uint64_t pv[4];
        bool av[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {  // 4 independent loads in flight; a tombstone (task handed out / removed since the last compaction) has no level any more
            uint64_t i = i0 + u * stride; av[u] = i < n; pv[u] = av[u] ? prio[i] : 0;
            if (rq && av[u] && rq[i] == RQ_TOMBSTONE) av[u] = false;
        }
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
                                if (old == PRIO_EMPTY) {  // this lane put the value into the set: it also goes onto the compact list behind the set (what k_sort_levels reads)
                                    const uint32_t k = atomicAdd(&flags[3], 1u);
                                    if (k < LEVEL_CAP) set[PRIO_SET_CAP + k] = lead;
                                }
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

__global__ void __launch_bounds__(1024) k_sort_levels(uint64_t *__restrict__ set, uint32_t *__restrict__ flags,
                                                      uint64_t *__restrict__ levels, uint32_t *__restrict__ n_levels, uint64_t *__restrict__ host_out, uint32_t seq) {
    // The distinct values arrive as a compact list behind the set (k_distinct_priorities appends a value when it inserts it; flags[3] counts them): a tick with three
    // levels sorts three values instead of scanning the 32 768 slots of the set (11.8 -> ~3 us: round 6 — the cold headline pays for this kernel on every tick).
    // Out: the level table in HBM (K1 reads it) AND, when host_out is given (pinned, device-mapped), [n_levels | flags[0] | flags[1]] + the table straight into host
    // memory — one stream synchronisation instead of three copies and two; the flag words are cleared for the scan that follows.
    extern __shared__ uint64_t lv[];  // LEVEL_CAP entries
    const uint32_t n = flags[3], f0 = flags[0], f1 = flags[1];
    __syncthreads();   // (every thread has read the flag words before thread 0 clears them at the end)
    if (n > LEVEL_CAP) {
        if (threadIdx.x == 0) {
            n_levels[0] = 0xFFFFFFFFu; n_levels[1] = 0; flags[0] = flags[1] = flags[2] = flags[3] = 0;   // (as a u64 head: 0xFFFFFFFF levels — the speculative scan refuses)
            if (host_out) { uint32_t *ho = reinterpret_cast<uint32_t *>(host_out); ho[0] = 0xFFFFFFFFu; ho[1] = f0; ho[2] = f1; __threadfence_system(); __hip_atomic_store(&ho[3], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
        }
        return;
    }
    uint32_t P = 1; while (P < n) P <<= 1;
    for (uint32_t i = threadIdx.x; i < P; i += blockDim.x) lv[i] = i < n ? set[PRIO_SET_CAP + i] : 0;
    __syncthreads();
    {   // leave the set EMPTY again: the slots of the listed values are located first (read-only: a cleared slot would cut another value's probe chain), then cleared —
        // the next discovery needs no 256 KB fill in front of it (two `fillBufferAligned` launches per cold tick until round 6)
        uint32_t my_slot[4]; int ns = 0;   // (LEVEL_CAP / 1024 threads)
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
            const uint64_t v = lv[i];
            uint32_t slot = (uint32_t)mix64(v) & (PRIO_SET_CAP - 1);
            for (uint32_t probe = 0; probe < PRIO_SET_CAP && set[slot] != v; probe++) slot = (slot + 1) & (PRIO_SET_CAP - 1);
            my_slot[ns++] = slot;
        }
        __syncthreads();
        for (int k = 0; k < ns; k++) set[my_slot[k]] = PRIO_EMPTY;
    }
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
    uint32_t shift = (f0 & 1u) ? 1u : 0u;  // Priority == u64::MAX present: it is the top level
    if (threadIdx.x == 0) {
        if (shift) { levels[0] = PRIO_EMPTY; if (host_out) host_out[2] = PRIO_EMPTY; }
        n_levels[0] = n + shift;
        {   // the head K1's speculative launch reads in ONE round trip: [count | level 0 .. 3] as five u64 (kernels.h: level_hist, n_levels_dev)
            uint64_t *head = reinterpret_cast<uint64_t *>(n_levels);
            head[0] = n + shift;
            for (uint32_t i = 0; i < 4; i++) { const uint32_t k = i - shift; head[1 + i] = (shift && i == 0) ? PRIO_EMPTY : (k < n ? lv[k] : 0); }
        }
        if (host_out) { reinterpret_cast<uint32_t *>(host_out)[0] = n + shift; reinterpret_cast<uint32_t *>(host_out)[1] = f0; reinterpret_cast<uint32_t *>(host_out)[2] = f1; }
        flags[0] = flags[1] = flags[2] = flags[3] = 0;
    }
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) { levels[i + shift] = lv[i]; if (host_out) host_out[2 + i + shift] = lv[i]; }
    if (host_out) {   // the completion word last (system-scope release): the host spins on it instead of synchronising the stream (~3 us against ~10: the GPU sits idle meanwhile)
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(&reinterpret_cast<uint32_t *>(host_out)[3], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ------------------------------------------------------------------------------------------------ K2
// One workgroup per 32 workers.  Worker rows and the whole request table are staged in LDS with coalesced loads (the inputs
// may sit in pinned host memory: every byte crosses PCIe once per workgroup), then one thread per (worker, variant).
// LDS: [total 32*R u64][free 32*R u64][rem 32 i64][entry_amount NE u64][variant_min_time NV u64][variant_entry_off NV+1 u32]
//      [entry_resource NE u32][entry_kind NE u8]
struct EvalArgs {
    const uint64_t *total, *free_; const int64_t *remaining_ns; uint32_t W, R; RequestTable rt; uint32_t n_entries; uint8_t *flags; uint32_t *tmc;
};

__device__ __forceinline__ void worker_eval_block(unsigned char *smem, uint32_t block, const EvalArgs &ea) {
    const uint64_t *__restrict__ total = ea.total, *__restrict__ free_ = ea.free_;
    const int64_t *__restrict__ remaining_ns = ea.remaining_ns;
    const RequestTable &rt = ea.rt;
    uint8_t *__restrict__ flags = ea.flags; uint32_t *__restrict__ tmc = ea.tmc;
    const uint32_t W = ea.W, R = ea.R, NV = rt.n_variants, NE = ea.n_entries;
    uint64_t *s_tot = reinterpret_cast<uint64_t *>(smem);
    uint64_t *s_free = s_tot + 32 * R;
    int64_t *s_rem = reinterpret_cast<int64_t *>(s_free + 32 * R);
    uint64_t *s_amt = reinterpret_cast<uint64_t *>(s_rem + 32);
    uint64_t *s_time = s_amt + NE;
    uint32_t *s_off = reinterpret_cast<uint32_t *>(s_time + NV);
    uint32_t *s_res = s_off + NV + 1;
    uint8_t *s_kind = reinterpret_cast<uint8_t *>(s_res + NE);
    const uint32_t w0 = block * 32, nw = W - w0 < 32 ? W - w0 : 32;
    for (uint32_t i = threadIdx.x; i < nw * R; i += blockDim.x) { s_tot[i] = total[(size_t)w0 * R + i]; s_free[i] = free_[(size_t)w0 * R + i]; }
    for (uint32_t i = threadIdx.x; i < nw; i += blockDim.x) s_rem[i] = remaining_ns[w0 + i];
    for (uint32_t i = threadIdx.x; i < NE; i += blockDim.x) { s_amt[i] = rt.entry_amount[i]; s_res[i] = rt.entry_resource[i]; s_kind[i] = rt.entry_kind[i]; }
    for (uint32_t i = threadIdx.x; i < NV; i += blockDim.x) s_time[i] = rt.variant_min_time_ns[i];
    for (uint32_t i = threadIdx.x; i <= NV; i += blockDim.x) s_off[i] = rt.variant_entry_off[i];
    __syncthreads();
    for (uint32_t idx = threadIdx.x; idx < nw * NV; idx += blockDim.x) {
        const uint32_t wl = idx / NV, v = idx % NV;
        bool imm = true, cap = true, any = false;
        uint64_t best = 0xFFFFFFFFFFFFFFFFull;
        for (uint32_t e = s_off[v]; e < s_off[v + 1]; e++) {
            const uint32_t r = s_res[e];
            const uint64_t f = r < R ? s_free[wl * R + r] : 0, tt = r < R ? s_tot[wl * R + r] : 0;
            uint64_t c;
            if (s_kind[e] == 0) {  // amount
                const uint64_t a = s_amt[e];
                imm = imm && a <= f; cap = cap && a <= tt;
                c = f / a; if (c > 1024) c = 1024;            // MAX_TASK_PER_WORKER  workerload.rs:12,131
            } else {                                           // All: min_amount = 1 fraction  request.rs:34-36
                imm = imm && f >= 1; cap = cap && tt >= 1;
                c = f == 0 ? 0 : 1;                            // workerload.rs:133-141
            }
            if (!any || c < best) best = c;
            any = true;
        }
        const int64_t rem = s_rem[wl];
        const bool time_ok = rem == INT64_MAX || (rem >= 0 && (uint64_t)rem >= s_time[v]);  // worker.rs:320-326
        const size_t t = (size_t)(w0 + wl) * NV + v;
        flags[t] = (imm ? 1 : 0) | (cap ? 2 : 0) | (time_ok ? 4 : 0);
        tmc[t] = any ? (uint32_t)best : 0;
    }
}

__global__ void __launch_bounds__(256) k_worker_eval(EvalArgs ea) {
    extern __shared__ __align__(16) unsigned char smem[];
    worker_eval_block(smem, blockIdx.x, ea);
}

// ------------------------------------------------------------------------------------------------ K1
// Every wavefront owns a contiguous slice of the ready set and a private LDS counter row.  A lane handles two consecutive
// tasks per 128-task tile (one dwordx4 of priorities, one dwordx2 of request ids, one dword of group keys), two tiles in
// flight.  With at most 4 priority levels the level table travels in the kernel arguments: no staging load, no barrier.
// LDS layout: [levels: lds_levels u64][counters: WPB*G u32].
struct Levels4 { uint64_t v[4]; };

// NCOPY: copies of a wavefront's counter row (lane & (NCOPY - 1) picks one): a tick has a handful of groups, so the 64 lanes of one ds_add hit a handful of
// addresses and the LDS serialises them per address — with 8 copies at most 8 lanes share a counter.  The copies are summed when the slice is published.
template <int WPB, bool SMALL_L, int NCOPY>
__global__ void __launch_bounds__(WPB * 64) k_level_hist(const uint64_t *__restrict__ prio, const uint32_t *__restrict__ rq, uint64_t n,
                                                         const uint64_t *__restrict__ levels, Levels4 l4, uint32_t L, uint32_t Q,
                                                         uint32_t tasks_per_wave, uint32_t n_waves, uint32_t stride, uint32_t lds_levels,
                                                         uint32_t *__restrict__ wave_tab, uint16_t *__restrict__ gkey,
                                                         uint32_t *__restrict__ err_flag, uint32_t n_eval_blocks, EvalArgs ea,
                                                         const uint32_t *__restrict__ n_levels_dev) {
    extern __shared__ __align__(16) unsigned char smem[];
    if (blockIdx.x < n_eval_blocks) {  // ride-along workgroups (dispatched first): K2, whose PCIe round trips hide under the scan
        worker_eval_block(smem, blockIdx.x, ea);
        return;
    }
    const uint32_t hist_block = blockIdx.x - n_eval_blocks;
    const uint32_t lane = lane_id();
    const uint32_t wave = hist_block * WPB + (threadIdx.x >> 6);
    const uint64_t begin = (uint64_t)wave * tasks_per_wave;
    const uint64_t end = begin + tasks_per_wave < n ? begin + tasks_per_wave : n;
    // The slice's first 256 tasks are asked for before anything that needs the level table: behind a level discovery the table is a round trip to HBM away (written
    // by another XCD a moment ago), and that round trip used to sit in FRONT of the columns' — two memory latencies in a kernel that is four of them long (round 6).
    ulonglong2 pv[2], pn[2];
    uint2 qv[2], qn[2];
    auto load_tiles = [&](uint64_t b, ulonglong2 *pd, uint2 *qd) {
#pragma unroll
        for (int u = 0; u < 2; u++) {  // both tiles' loads in flight before the first use
            const uint64_t i = b + (uint64_t)u * 128 + 2 * lane;
            if (i + 1 < end) { pd[u] = *reinterpret_cast<const ulonglong2 *>(prio + i); qd[u] = *reinterpret_cast<const uint2 *>(rq + i); }
            else if (i < end) { pd[u] = make_ulonglong2(prio[i], 0); qd[u] = make_uint2(rq[i], 0); }
        }
    };
    if (wave < n_waves) load_tiles(begin, pv, qv);
    const uint32_t GL = L * Q;   // the counter rows' layout follows the level count the launch was sized for (what the LDS was sized for, too)
    uint64_t *s_levels = reinterpret_cast<uint64_t *>(smem);
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(smem + (size_t)(SMALL_L ? 0 : lds_levels) * 8) + (threadIdx.x >> 6) * GL * NCOPY;
    uint32_t *my_cnt = s_cnt + (lane & (uint32_t)(NCOPY - 1)) * GL;
    if (!SMALL_L) for (uint32_t i = threadIdx.x; i < lds_levels; i += blockDim.x) s_levels[i] = levels[i];
    for (uint32_t g = lane; g < GL * NCOPY; g += 64) s_cnt[g] = 0;
    if (SMALL_L && n_levels_dev) {
        // Launched right behind the level discovery, before the host has seen its result (a tick that rediscovers its levels: round 6): the table and its length come
        // from HBM — k_sort_levels has just written them — and `L` is only the bound the launch was sized for (4).  More levels than that: the scan refuses (err bit 4),
        // the host reads the table and launches the general variant.  Uniform scalar loads, a few hundred ns; no host round trip between discovery and scan.
        const uint64_t *head = reinterpret_cast<const uint64_t *>(n_levels_dev);   // [count | level 0 .. 3]: five independent loads, one round trip
        uint64_t h0 = head[0], h1 = head[1], h2 = head[2], h3 = head[3], h4 = head[4];
        asm volatile("" : "+s"(h0), "+s"(h1), "+s"(h2), "+s"(h3), "+s"(h4));   // all five asked for before the count is looked at (the compiler sank the levels' loads below the test on the count: a second round trip)
        const uint32_t Ld = (uint32_t)h0;
        if (Ld == 0 || Ld > L) { if (hist_block == 0 && threadIdx.x == 0) atomicOr(err_flag, 4u); return; }
        L = Ld;
        l4.v[0] = h1; l4.v[1] = h2; l4.v[2] = h3; l4.v[3] = h4;
    }
    const uint32_t G = L * Q;
    if (!SMALL_L) __syncthreads();
    if (wave >= n_waves) return;
    const uint64_t *lvp = lds_levels ? s_levels : levels;
    uint32_t err = 0;
    // (priority, rq) -> group key, counting it; GKEY_INVALID for a task the tables do not cover
    auto classify = [&](uint64_t p, uint32_t q) -> uint16_t {
        uint32_t lv;
        if (SMALL_L) lv = p == l4.v[0] ? 0u : (L > 1 && p == l4.v[1]) ? 1u : (L > 2 && p == l4.v[2]) ? 2u : (L > 3 && p == l4.v[3]) ? 3u : L;
        else { lv = level_of(lvp, L, p); if (lv < L && lvp[lv] != p) lv = L; }
        if (q == RQ_TOMBSTONE) return GKEY_INVALID;            // task already handed out / removed (hqtick_ready_*): not an error
        if (lv >= L) { err |= 1u; return GKEY_INVALID; }       // priority missing from the level table
        if (q >= Q) { err |= 2u; return GKEY_INVALID; }        // request id out of range
        const uint32_t g = lv * Q + q;
        atomicAdd(&my_cnt[g], 1u);
        return (uint16_t)g;
    };
    for (uint64_t b = begin; b < end; b += 256) {
        const bool more = b + 256 < end;
        if (more) load_tiles(b + 256, pn, qn);   // the next 256 tasks are on their way while these are classified
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const uint64_t i = b + (uint64_t)u * 128 + 2 * lane;
            if (i + 1 < end) {
                const uint32_t k0 = classify(pv[u].x, qv[u].x), k1 = classify(pv[u].y, qv[u].y);
                *reinterpret_cast<uint32_t *>(gkey + i) = k0 | (k1 << 16);
            } else if (i < end) {
                gkey[i] = classify(pv[u].x, qv[u].x);
            }
        }
        if (more) {
#pragma unroll
            for (int u = 0; u < 2; u++) { pv[u] = pn[u]; qv[u] = qn[u]; }
        }
    }
    if (err) atomicOr(err_flag, err);
    // publish this slice's counts, transposed to [G][stride] so the scan and K4 read rows contiguously
    for (uint32_t g = lane; g < G; g += 64) {
        uint32_t c = s_cnt[g];
#pragma unroll
        for (int k = 1; k < NCOPY; k++) c += s_cnt[(uint32_t)k * GL + g];
        wave_tab[(size_t)g * stride + wave] = c;
    }
}

// ------------------------------------------------------------------------------------------------ K1b
// One wavefront per group row: 64 consecutive entries per lane (16 x dwordx4 in flight), local prefix + one wave scan per
// 4096 entries.  The row total goes to hist[] — which may live in pinned host memory (written once per row).
__global__ void __launch_bounds__(256) k_scan_rows(uint32_t *__restrict__ wave_tab, uint32_t n_waves, uint32_t stride, uint32_t G,
                                                   uint32_t *__restrict__ hist, uint32_t *__restrict__ err_in,
                                                   uint32_t *__restrict__ err_out, uint32_t n_eval_blocks, EvalArgs ea) {
    extern __shared__ __align__(16) unsigned char smem[];
    if (blockIdx.x < n_eval_blocks) {  // ride-along workgroups (dispatched first): K2.  This launch is a latency chain of 6 us on two workgroups; K2's 32
        worker_eval_block(smem, blockIdx.x, ea);  // workgroups run beside it for free, and K1's launch — the one the roofline is priced on — is K1 alone
        return;
    }
    const uint32_t row = (blockIdx.x - n_eval_blocks) * 4 + (threadIdx.x >> 6);
    if (row == 0 && threadIdx.x == 0 && err_out) { err_out[0] = err_in[0]; err_in[0] = 0; }  // K1's validation flags: forwarded to the host-visible word, device word re-armed
    if (row >= G) return;
    const uint32_t lane = lane_id();
    uint32_t *r = wave_tab + (size_t)row * stride;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < n_waves; base += 4096) {
        const uint32_t i0 = base + lane * 64;
        uint4 v[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {  // stride is a multiple of 16 entries: every 4-entry group is inside the row or fully outside
            const uint32_t i = i0 + 4 * k;
            v[k] = i < stride ? *reinterpret_cast<const uint4 *>(r + i) : make_uint4(0, 0, 0, 0);
        }
        uint32_t run = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) {  // padding entries (index >= n_waves) count as 0
            const uint32_t i = i0 + 4 * k;
            uint32_t t0 = i < n_waves ? v[k].x : 0u, t1 = i + 1 < n_waves ? v[k].y : 0u, t2 = i + 2 < n_waves ? v[k].z : 0u, t3 = i + 3 < n_waves ? v[k].w : 0u;
            v[k].x = run; run += t0; v[k].y = run; run += t1; v[k].z = run; run += t2; v[k].w = run; run += t3;
        }
        uint32_t incl = run;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { uint32_t t = __shfl_up(incl, off, 64); if ((int)lane >= off) incl += t; }
        const uint32_t excl = carry + incl - run;
        const uint32_t total = __shfl(incl, 63, 64);
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const uint32_t i = i0 + 4 * k;
            if (i < stride) *reinterpret_cast<uint4 *>(r + i) = make_uint4(v[k].x + excl, v[k].y + excl, v[k].z + excl, v[k].w + excl);
        }
        carry += total;
    }
    if (lane == 0) hist[row] = carry;
}

// ------------------------------------------------------------------------------------------------ K4
// LDS layout: [counters: WPB*G u32]([take G][base G] when the plan is staged).
// PLAN: 0 = take/base arrive in the kernel arguments (G <= 64: no load from pinned or global memory on the critical path),
//       1 = staged into LDS from `take`/`base`,  2 = read in place (large G, one wavefront per workgroup).
// tnc / tsb: the TRANSPOSED scatter of a group (K5b's gather).  A request whose single placement key gives every one of its n workers the same count c
// is taken round-robin: the task at position p of the queue goes to the worker at list position p % n as its (p / n)-th task — worker w's tasks sit n * 8 bytes apart in
// queue order, and K5b's gather touched one 64-byte line per record (6.0 MB fetched for 2.3 MB of algorithmic bytes).  With tnc[g] = n << 16 | c (0: off) and
// tsb[g] = the request's base in sel_* (so that p = base[g] - tsb[g] + rank), K4 writes the task to tsb + (p % n) * c + p / n instead: each worker's ids contiguous.
template <int N> struct SelPlanN { uint32_t take[N], base[N], tnc[N], tsb[N]; };  // the plan in the kernel arguments: 8 bytes per group of the launch call's argument block
                                                                  // (a launch call costs ~1.7 ns per argument byte on the host: 16 groups = 128 B, 64 = 512 B)

template <int WPB, int PLAN, int PN>
__global__ void __launch_bounds__(WPB * 64) k_select(const uint64_t *__restrict__ task_id, const uint16_t *__restrict__ gkey, uint64_t n,
                                                     uint32_t Q, uint32_t G, uint32_t tasks_per_wave, uint32_t n_waves, uint32_t stride,
                                                     const uint32_t *__restrict__ wave_off, const uint32_t *__restrict__ take,
                                                     const uint32_t *__restrict__ base, SelPlanN<PN> pa, uint64_t *__restrict__ sel_task,
                                                     uint16_t *__restrict__ sel_key, uint32_t n_select_blocks,
                                                     const uint4 *__restrict__ copy_src, uint4 *__restrict__ copy_dst, uint32_t copy_n16,
                                                     uint32_t *__restrict__ mark_rq, uint32_t mark_and_select) {
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t n_copy_blocks = gridDim.x - n_select_blocks;
    if (blockIdx.x < n_copy_blocks) {  // ride-along workgroups, dispatched FIRST so their PCIe round trip hides under the selection:
        // bring the mapping plan from pinned host memory into HBM for K5a/K5b
        for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < copy_n16; i += n_copy_blocks * blockDim.x) copy_dst[i] = copy_src[i];
        return;
    }
    const uint32_t sel_block = blockIdx.x - n_copy_blocks;
    uint32_t *s_all = reinterpret_cast<uint32_t *>(smem);
    uint32_t *s_cnt = s_all + (threadIdx.x >> 6) * G;
    const uint32_t *tk = take, *bs = base, *tnc = base + G, *tsb = base + 2 * G;  // the plan in memory: [take G][base G][tnc G][tsb G]
    if (PLAN != 2) {
        uint32_t *s_take = s_all + WPB * G, *s_base = s_take + G;
        uint32_t *s_tnc = s_base + G, *s_tsb = s_tnc + G;
        for (uint32_t g = threadIdx.x; g < G; g += blockDim.x) {
            s_take[g] = PLAN == 0 ? pa.take[g] : take[g]; s_base[g] = PLAN == 0 ? pa.base[g] : base[g];
            s_tnc[g] = PLAN == 0 ? pa.tnc[g] : base[G + g]; s_tsb[g] = PLAN == 0 ? pa.tsb[g] : base[2 * G + g];
        }
        tk = s_take; bs = s_base; tnc = s_tnc; tsb = s_tsb;
    }
    const uint32_t lane = lane_id();
    const uint32_t wave = sel_block * WPB + (threadIdx.x >> 6);
    if (PLAN != 2) __syncthreads();
    if (wave >= n_waves) return;
    const uint64_t begin = (uint64_t)wave * tasks_per_wave;
    const uint64_t end = begin + tasks_per_wave < n ? begin + tasks_per_wave : n;
    uint16_t kv[4];
    uint64_t idv[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {  // first tile's keys (2 B per task) in flight while the slice offsets arrive (wasted only on exhausted slices)
        uint64_t i = begin + (uint64_t)u * 64 + lane;
        kv[u] = i < end ? gkey[i] : GKEY_INVALID;
    }
    bool need = false;  // counters are wave-private: no workgroup barrier from here on
    for (uint32_t g = lane; g < G; g += 64) { uint32_t o = wave_off[(size_t)g * stride + wave]; s_cnt[g] = o; need = need || o < tk[g]; }
    if (!__ballot(need)) return;  // every group this slice could feed is already exhausted by earlier slices
    // The ids (8 B per task) are read by the lanes whose task is TAKEN, once its rank is known from the keys — not by the slice: with three priority levels spread over
    // the whole ready set every 256-task slice feeds some group, and streaming every live slice's 2 KB of ids to take 5 % of them was 8 MB of a 13 MB launch (VERDICT r05:
    // FETCH 4.5 x the algorithmic bytes).  The four tiles' taken ids are in flight together, behind the ranks.
    int nbits = 0; while ((1u << nbits) < G) nbits++;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    for (uint64_t b = begin; b < end; b += 256) {
        if (b != begin) {
#pragma unroll
            for (int u = 0; u < 4; u++) { uint64_t i = b + (uint64_t)u * 64 + lane; kv[u] = i < end ? gkey[i] : GKEY_INVALID; }
        }
        uint32_t dstv[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (b + (uint64_t)u * 64 >= end) break;  // wave-uniform
            const uint32_t g = kv[u];
            const bool active = g != GKEY_INVALID;
            const uint64_t peers = match_any(g, nbits, active);
            if (active) {
                const uint32_t before = (uint32_t)__popcll(peers & lt_mask);
                const uint32_t cur = s_cnt[g];  // wave-private counter: leaders of distinct groups write distinct words
                const uint32_t rank = cur + before;
                if (rank < tk[g]) {
                    if (mark_rq) mark_rq[b + (uint64_t)u * 64 + lane] = RQ_TOMBSTONE;  // the task leaves the ready set: hqtick_ready_consume_last (then nothing else is written), or
                                                                                      // the tick itself under HQTICK_FLAG_CONSUME_IN_TICK (mark_and_select: the selection as well)
                    if (!mark_rq || mark_and_select) {
                        uint32_t dst = bs[g] + rank;
                        const uint32_t nc = tnc[g];
                        if (nc) {  // worker-major: position p of the request's queue -> (worker p % n, its task p / n)
                            const uint32_t sb = tsb[g], p = dst - sb, nw = nc >> 16, c = nc & 0xFFFFu;
                            if (p < nw * c) { const uint32_t sw = p / nw; dst = sb + (p - sw * nw) * c + sw; }
                        }
                        dstv[u] = dst;
                        idv[u] = task_id[b + (uint64_t)u * 64 + lane];
                    }
                }
                if (before == 0) s_cnt[g] = cur + (uint32_t)__popcll(peers);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) if (dstv[u] != 0xFFFFFFFFu) { sel_task[dstv[u]] = idv[u]; sel_key[dstv[u]] = kv[u]; }  // group key; its level is g / Q (K5b)
    }
}

// ------------------------------------------------------------------------------------------------ K5a
// Grid (sweep quad, key): a workgroup stages its key's counts (Map iteration order) in LDS, then each of its 4 wavefronts
// builds the bit row of ONE sweep s: lane l assembles word l = [c_j > s] for j in [64 l, 64 l + 64) from LDS (row-padded,
// conflict-free), a wave scan of the popcounts gives the per-word exclusive prefix, and T_k(s) = sum_j min(c_j, s).
static const uint32_t SWEEP_LDS_COUNTS = 24576;  // workers per key whose counts fit the LDS staging (96 KiB + padding)

__global__ void __launch_bounds__(256) k_sweep_bits(MapKeys mk) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t *s_c = reinterpret_cast<uint32_t *>(smem);  // count of position j at s_c[j + (j >> 6)]
    const uint32_t k = blockIdx.y, lane = lane_id();
    if (mk.key_tr[k]) return;  // a key stored worker-major by K4 (every worker the same count): nobody reads its bit rows
    const uint32_t t0 = mk.key_t_off[k], n_sweeps = mk.key_t_off[k + 1] - t0;  // sweeps 0..maxc
    if (blockIdx.x * 4 >= n_sweeps) return;
    const uint32_t nk = mk.key_ord_off[k + 1] - mk.key_ord_off[k];
    const uint32_t *cnts = mk.ord_cnt + mk.key_ord_off[k];
    const uint32_t words = (nk + 63) >> 6;
    // the counts may sit in pinned host memory (the early launch reads its tables in place): eight loads in flight per thread and round trip — one
    // load per trip cost the launch four PCIe latencies at 1024 workers per key
    for (uint32_t j0 = threadIdx.x; j0 < words * 64; j0 += 8 * blockDim.x) {
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const uint32_t j = j0 + u * blockDim.x; v[u] = j < nk ? cnts[j] : 0u; }
#pragma unroll
        for (int u = 0; u < 8; u++) { const uint32_t j = j0 + u * blockDim.x; if (j < words * 64) s_c[j + (j >> 6)] = v[u]; }
    }
    __syncthreads();
    const uint32_t s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= n_sweeps) return;
    const size_t row = (size_t)mk.key_bits_off[k] + (size_t)s * words;
    uint32_t carry = 0, summin = 0;
    for (uint32_t w0 = 0; w0 < words; w0 += 64) {
        const uint32_t wd = w0 + lane;
        uint64_t m = 0;
        if (wd < words) {
            const uint32_t *c = s_c + (size_t)wd * 65;
#pragma unroll 8
            for (uint32_t i = 0; i < 64; i++) { const uint32_t cv = c[i]; m |= (uint64_t)(cv > s) << i; summin += cv < s ? cv : s; }
        }
        const uint32_t pc = (uint32_t)__popcll(m);
        uint32_t incl = pc;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { uint32_t t = __shfl_up(incl, off, 64); if ((int)lane >= off) incl += t; }
        if (wd < words) { mk.bits[row + wd] = m; mk.pre[row + wd] = carry + incl - pc; }
        carry += __shfl(incl, 63, 64);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) summin += __shfl_xor(summin, off, 64);
    if (lane == 0) mk.t_sweep[t0 + s] = summin;
}

static const uint32_t PF_STAGE = 64;    // prefilling requests whose chunk descriptors K5b stages in LDS (one wavefront scans them)
static const uint32_t RUN_STAGE = 512;  // runs of one worker staged in LDS before they are written (6 KB)

// ------------------------------------------------------------------------------------------------ K5b
// One workgroup per worker.  LDS: e_task u64[max_items] | e_lvl u16[max_items] | e_meta u16[max_items] | k_start u32[n_keys+1]
// | k_pos | k_cnt | k_rq | k_seg | k_toff | k_boff | k_words  (u32[n_keys] each) | k_var u8[n_keys] (padded) | misc u32[4]
// Compact emission (HQTICK_FLAG_COMPACT_RECORDS): the records cross PCIe as 4 bytes each — the low half of the task id — plus one 10-byte RUN per
// maximal stretch of a worker's records that share (job id, variant, kind).  LDS then also holds the worker's final sequence:
// f_task u64[max_out] | f_meta u16[max_out] (variant | kind << 8) behind the tables above.
__global__ void __launch_bounds__(256) k_expand_mapping(MapKeys mk, uint32_t W, const uint64_t *__restrict__ sel_task,
                                                        const uint16_t *__restrict__ sel_key, uint32_t Q, uint32_t max_items,
                                                        uint64_t *__restrict__ rec_task, uint8_t *__restrict__ rec_variant,
                                                        uint8_t *__restrict__ rec_kind, uint32_t *__restrict__ err_flag, CompactOut co, uint32_t max_out,
                                                        uint32_t sort_cap) {
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t nkeys = mk.n_keys;
    uint64_t *e_task = reinterpret_cast<uint64_t *>(smem);
    uint64_t *f_task = e_task + max_items;  // compact mode only (max_out == 0 otherwise)
    uint16_t *e_lvl = reinterpret_cast<uint16_t *>(f_task + max_out);
    uint16_t *e_meta = e_lvl + max_items;  // variant | valid << 8
    uint16_t *f_meta = e_meta + max_items;
    uint32_t *k_start = reinterpret_cast<uint32_t *>(f_meta + max_out + ((max_items * 2 + max_out) & 1u));  // keeps 4-byte alignment
    const bool compact = co.rec_lo != nullptr;
    uint32_t *k_pos = k_start + nkeys + 1;
    uint32_t *k_cnt = k_pos + nkeys, *k_rq = k_cnt + nkeys, *k_seg = k_rq + nkeys, *k_toff = k_seg + nkeys, *k_boff = k_toff + nkeys, *k_words = k_boff + nkeys;
    uint32_t *k_trc = k_words + nkeys;
    uint32_t *misc = k_trc + nkeys;  // [0] min level, [1] max level, [2] holes
    uint32_t *s_key = misc + 4;        // [sort_cap] (level << 16 | item) keys of the stable sort; sort_cap = 0 on ticks that cannot reorder
    uint8_t *k_var = reinterpret_cast<uint8_t *>(s_key + sort_cap);
    const uint32_t w = blockIdx.x, lane = lane_id();
    // Three rounds of global loads, each issued as one batch: (1) the worker's output range, its row of every per-key table and its prefill chunks;
    // (2) the round-robin cells of its items + the ids of its prefill records; (3) the ids and group keys of its items.  (Five dependent rounds —
    // range, tables, chunk index, chunk ids, cells, ids — cost the launch ~4 us more.)
    __shared__ uint32_t p_j[PF_STAGE], p_cnt[PF_STAGE], p_src[PF_STAGE], p_start[PF_STAGE + 1];  // the worker's chunk of every prefilling request
    const bool pf_staged = mk.n_pfq <= PF_STAGE;
    const uint32_t out0 = mk.out_off[w], out1 = mk.out_off[w + 1];
    for (uint32_t k = threadIdx.x; k < nkeys; k += blockDim.x) {  // every per-key table in one round of independent loads
        k_pos[k] = mk.wpos[(size_t)k * W + w];
        k_cnt[k] = mk.wcnt[(size_t)k * W + w];
        k_rq[k] = mk.key_rq[k]; k_seg[k] = mk.key_seg_start[k]; k_toff[k] = mk.key_t_off[k]; k_boff[k] = mk.key_bits_off[k];
        k_words[k] = (mk.key_ord_off[k + 1] - mk.key_ord_off[k] + 63) >> 6;
        k_trc[k] = mk.key_tr[k];
        k_var[k] = mk.key_variant[k];
    }
    if (pf_staged && threadIdx.x < mk.n_pfq) {
        const uint32_t pi = threadIdx.x, j = mk.pfl_j[(size_t)pi * W + w], c = mk.pfq_size[pi];
        p_j[pi] = j; p_cnt[pi] = j == 0xFFFFFFFFu ? 0u : c; p_src[pi] = mk.pfq_src[pi] + (j == 0xFFFFFFFFu ? 0u : j * c);
    }
    if (out1 == out0) return;  // no record for this worker (or the worker belongs to another rank's shard)
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
    } else if (threadIdx.x < 128 && pf_staged) {  // second wavefront: exclusive scan of the prefill chunk lengths (PF_STAGE == 64: one step)
        const uint32_t c = lane < mk.n_pfq ? p_cnt[lane] : 0u;
        uint32_t incl = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { uint32_t t = __shfl_up(incl, off, 64); if ((int)lane >= off) incl += t; }
        if (lane < mk.n_pfq) p_start[lane] = incl - c;
        if (lane == 63) p_start[mk.n_pfq] = incl;
    }
    __syncthreads();
    const uint32_t n = k_start[nkeys];
    if (n > max_items) { if (threadIdx.x == 0) err_flag[0] = 2u; return; }
    // new prefills first, in queue (request id) order (mapping.rs:266-272)
    uint32_t npf = 0;
    if (pf_staged) npf = mk.n_pfq ? p_start[mk.n_pfq] : 0u;
    else for (uint32_t pi = 0; pi < mk.n_pfq; pi++) {  // more prefilling requests than the stage holds: chunk by chunk
        const uint32_t j = mk.pfl_j[(size_t)pi * W + w];
        if (j == 0xFFFFFFFFu) continue;
        const uint32_t cnt = mk.pfq_size[pi], src = mk.pfq_src[pi] + j * cnt;
        for (uint32_t t = threadIdx.x; t < cnt; t += blockDim.x) {
            if (compact) { if (npf + t < max_out) { f_task[npf + t] = sel_task[src + t]; f_meta[npf + t] = 0x00FFu; } continue; }
            rec_task[out0 + npf + t] = sel_task[src + t];
            rec_variant[out0 + npf + t] = 0xFF;
            rec_kind[out0 + npf + t] = 0;  // HQ_REC_PREFILL
        }
        npf += cnt;
    }
    // prefill records and gathered items side by side: thread t handles prefill record u = base + t and item e = base + t of every 256-wide step, so that
    // the loads of both are in flight together.  Item e = (key k, sweep s) in key order then sweep order.
    const uint32_t n_pf_staged = pf_staged ? npf : 0u;
    for (uint32_t base = 0; base < (n > n_pf_staged ? n : n_pf_staged); base += blockDim.x) {
        const uint32_t u = base + threadIdx.x, e = u;
        const bool do_pf = u < n_pf_staged, do_item = e < n;
        uint64_t pf_id = 0;
        if (do_pf) {
            uint32_t lo = 0, hi = mk.n_pfq;  // last chunk with p_start[pi] <= u (empty chunks have equal starts: the last one is the non-empty one)
            while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (p_start[mid] <= u) lo = mid; else hi = mid; }
            pf_id = sel_task[p_src[lo] + (u - p_start[lo])];
        }
        if (do_item) {
            uint32_t lo = 0, hi = nkeys;  // last key with k_start[k] <= e (it is the non-empty one)
            while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (k_start[mid] <= e) lo = mid; else hi = mid; }
            const uint32_t k = lo, s = e - k_start[k], pos = k_pos[k];
            if (k_trc[k]) {  // worker-major key: this worker's tasks are the k_trc[k] ids from its own offset on (no cell, no hole: the host transposes only plain queues)
                const uint32_t src = mk.rq_sel_base[k_rq[k]] + pos * k_trc[k] + s;
                const uint16_t lv = sort_cap ? (uint16_t)(sel_key[src] / Q) : (uint16_t)0;
                e_task[e] = sel_task[src];
                e_lvl[e] = lv;
                e_meta[e] = (uint16_t)(k_var[k] | 0x100u);
                if (sort_cap) { atomicMin(&misc[0], (uint32_t)lv); atomicMax(&misc[1], (uint32_t)lv); }
                if (do_pf) {
                    if (compact) { if (u < max_out) { f_task[u] = pf_id; f_meta[u] = 0x00FFu; } }
                    else { rec_task[out0 + u] = pf_id; rec_variant[out0 + u] = 0xFF; rec_kind[out0 + u] = 0; }
                }
                continue;
            }
            const size_t cell = (size_t)k_boff[k] + (size_t)s * k_words[k] + (pos >> 6);
            const uint32_t q = k_rq[k];
            const uint32_t pre = mk.pre[cell], tsw = mk.t_sweep[k_toff[k] + s];
            const uint64_t bw = mk.bits[cell];
            const uint32_t pfs = mk.rq_pf_start[q], pfn = mk.rq_pf_n[q], sbase = mk.rq_sel_base[q];
            const uint32_t idx = tsw + pre + (uint32_t)__popcll(bw & ((1ull << (pos & 63)) - 1ull));  // index inside the key's take_tasks() vector
            const uint32_t p = k_seg[k] + idx;                              // position in the queue's logical sequence
            bool hole = p >= pfs && p < pfs + pfn;                         // an already-prefilled task: retract/redirect is host work
            if (!hole && mk.n_holes) {                                     // a Retracting task of the queue: redirect, no record (host work too)
                const uint64_t hk = ((uint64_t)q << 32) | p;
                uint32_t hlo = 0, hhi = mk.n_holes;
                while (hlo < hhi) { uint32_t mid = (hlo + hhi) >> 1; if (mk.holes[mid] < hk) hlo = mid + 1; else hhi = mid; }
                hole = hlo < mk.n_holes && mk.holes[hlo] == hk;
            }
            if (hole) {
                e_meta[e] = 0; e_task[e] = 0; e_lvl[e] = 0;
                misc[2] = 1;
            } else {
                const uint32_t src = sbase + (p >= pfs + pfn ? p - pfn : p);
                // priority level of the task's group — only a tick that can reorder needs it (sort_cap != 0: several levels, holes or prefilled tasks): the cold
                // tick skips one strided 2-byte gather per record (a 64-byte line each)
                const uint16_t lv = sort_cap ? (uint16_t)(sel_key[src] / Q) : (uint16_t)0;
                e_task[e] = sel_task[src];
                e_lvl[e] = lv;
                e_meta[e] = (uint16_t)(k_var[k] | 0x100u);
                if (sort_cap) { atomicMin(&misc[0], (uint32_t)lv); atomicMax(&misc[1], (uint32_t)lv); }
            }
        }
        if (do_pf) {
            if (compact) { if (u < max_out) { f_task[u] = pf_id; f_meta[u] = 0x00FFu; } }
            else { rec_task[out0 + u] = pf_id; rec_variant[out0 + u] = 0xFF; rec_kind[out0 + u] = 0; }  // HQ_REC_PREFILL
        }
    }
    __syncthreads();
    const bool trivial = misc[2] == 0 && misc[0] >= misc[1];  // one priority level, no holes: already in final order
    // stable sort by priority descending == level ascending (mapping.rs:128-131).  Sorting (level << 16 | item) is a stable sort by level; a bitonic
    // network over the next power of two in LDS: O(n log^2 n / 256) per thread (n = 1024 records, 3 levels: 55 steps of 4 compare-exchanges, where
    // counting ranks costs 4096 LDS reads per record).  Holes (no record) sort to the end.
    uint32_t P = 0;
    if (!trivial && sort_cap) {
        P = 1; while (P < n) P <<= 1;
        if (P > sort_cap || P > 65536u) P = 0;  // cannot happen (the host sizes sort_cap from max_items); rank counting below stays correct
    }
    if (P) {
        for (uint32_t i = threadIdx.x; i < P; i += blockDim.x) s_key[i] = (i < n && (e_meta[i] & 0x100u)) ? ((uint32_t)e_lvl[i] << 16 | i) : 0xFFFFFFFFu;
        __syncthreads();
        for (uint32_t k = 2; k <= P; k <<= 1) {
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                for (uint32_t i = threadIdx.x; i < P; i += blockDim.x) {
                    const uint32_t o = i ^ j;
                    if (o > i) {
                        const uint32_t a = s_key[i], b = s_key[o];
                        if ((a > b) == ((i & k) == 0)) { s_key[i] = b; s_key[o] = a; }
                    }
                }
                __syncthreads();
            }
        }
    }
    for (uint32_t t = threadIdx.x; t < n; t += blockDim.x) {
        uint32_t e = t, pos = t;  // trivial: already in final order
        if (P) {                  // position t takes the item the network put there
            const uint32_t key = s_key[t];
            if (key == 0xFFFFFFFFu) continue;
            e = key & 0xFFFFu;
        }
        const uint16_t meta = e_meta[e];
        if (!(meta & 0x100u)) continue;
        if (!trivial && !P) {     // fallback: rank counting
            const uint16_t lv = e_lvl[e];
            pos = 0;
            for (uint32_t o = 0; o < n; o++) {
                if (!(e_meta[o] & 0x100u)) continue;
                const uint16_t lo_ = e_lvl[o];
                pos += (lo_ < lv || (lo_ == lv && o < e)) ? 1u : 0u;
            }
        }
        if (compact) { if (npf + pos < max_out) { f_task[npf + pos] = e_task[e]; f_meta[npf + pos] = (uint16_t)((meta & 0xFFu) | 0x100u); } continue; }
        const uint32_t dst = out0 + npf + pos;
        rec_task[dst] = e_task[e];
        rec_variant[dst] = (uint8_t)(meta & 0xFFu);
        rec_kind[dst] = 1;  // HQ_REC_ASSIGN
    }
    if (!compact) return;
    // ---- compact emission: runs of equal (job, variant, kind) over the worker's final sequence ----
    __syncthreads();
    const uint32_t tot = mk.out_off[w + 1] - out0;
    if (tot > max_out) { if (threadIdx.x == 0) err_flag[0] = 2u; return; }
    __shared__ uint32_t s_wave[4], s_wave_u[4];
    const bool d16 = co.units != nullptr;  // HQTICK_FLAG_COMPACT_DELTA16: 16-bit differences instead of 32-bit low halves
    const uint32_t RW = d16 ? 4u : 3u;     // words per run record
    uint32_t *s_run = reinterpret_cast<uint32_t *>(smem + ((reinterpret_cast<uintptr_t>(k_var + nkeys) - reinterpret_cast<uintptr_t>(smem) + 3) & ~(uintptr_t)3));  // [RUN_STAGE * RW], compact
                                                                                           // launches only: the worker's runs as they go out
    uint16_t *s_units = reinterpret_cast<uint16_t *>(s_run + RUN_STAGE * 4);                // [3 * max_out + 1] delta mode: the worker's unit stream
    const uint32_t chunk = (tot + blockDim.x - 1) / blockDim.x, lo = threadIdx.x * chunk, hi = lo + chunk < tot ? lo + chunk : tot;
    auto boundary = [&](uint32_t i) { return i == 0 || (uint32_t)(f_task[i] >> 32) != (uint32_t)(f_task[i - 1] >> 32) || f_meta[i] != f_meta[i - 1]; };
    // delta mode: a record that does not open a run travels as ONE 16-bit unit = its low id minus its predecessor's (records of one (worker, key) come in
    // ascending id order, a few thousand ids apart), or as THREE units (0xFFFF, low 16 bits, high 16 bits) when the difference does not fit; the record
    // that opens a run has its low id in the run record
    auto usize = [&](uint32_t i) { return boundary(i) ? 0u : (((uint32_t)f_task[i] - (uint32_t)f_task[i - 1]) < 0xFFFFu ? 1u : 3u); };
    uint32_t mine = 0, mine_u = 0;
    for (uint32_t i = lo; i < hi; i++) { mine += boundary(i) ? 1u : 0u; if (d16) mine_u += usize(i); }
    uint32_t incl = mine, incl_u = mine_u;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t t = __shfl_up(incl, off, 64), tu = __shfl_up(incl_u, off, 64);
        if ((int)lane >= off) { incl += t; incl_u += tu; }
    }
    if (lane == 63) { s_wave[threadIdx.x >> 6] = incl; s_wave_u[threadIdx.x >> 6] = incl_u; }
    __syncthreads();
    uint32_t before = incl - mine, before_u = incl_u - mine_u;
    for (uint32_t wv = 0; wv < (threadIdx.x >> 6); wv++) { before += s_wave[wv]; before_u += s_wave_u[wv]; }
    const uint32_t n_runs = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3], n_units = s_wave_u[0] + s_wave_u[1] + s_wave_u[2] + s_wave_u[3];
    // The worker's runs go to the run slots out0 .. out0 + n_runs (a worker has at most one run per record, so the record offsets double as run
    // offsets): no allocation.  A shared counter here — one atomicAdd per workgroup on one address — serialised the 1024 workgroups of the launch
    // for 8.5 us of its 35 (measured by replacing it with a constant).
    if (threadIdx.x == 0) co.run_span[w] = make_uint2(out0, n_runs);  // one 8-byte store
    // Every store below crosses PCIe as a write of its own: the runs are 12- / 16-byte records staged in LDS and written by consecutive lanes.
    const bool staged = n_runs <= RUN_STAGE;
    uint32_t r = before;
    if (staged) for (uint32_t i = lo; i < hi; i++) if (boundary(i)) {
        s_run[RW * r] = i; s_run[RW * r + 1] = (uint32_t)(f_task[i] >> 32); s_run[RW * r + 2] = f_meta[i];
        if (d16) s_run[RW * r + 3] = (uint32_t)f_task[i];
        r++;
    }
    if (d16) {
        uint32_t up = before_u;
        for (uint32_t i = lo; i < hi; i++) {
            if (boundary(i)) continue;
            const uint32_t cur = (uint32_t)f_task[i], d = cur - (uint32_t)f_task[i - 1];
            if (d < 0xFFFFu) s_units[up++] = (uint16_t)d;
            else { s_units[up] = 0xFFFFu; s_units[up + 1] = (uint16_t)cur; s_units[up + 2] = (uint16_t)(cur >> 16); up += 3; }
        }
    }
    __syncthreads();
    uint32_t *runs = co.runs + (size_t)out0 * RW;
    if (staged) { for (uint32_t i = threadIdx.x; i < n_runs * RW; i += blockDim.x) runs[i] = s_run[i]; }
    else for (uint32_t i = lo; i < hi; i++) if (boundary(i)) {  // more runs than the stage holds: each thread writes its own
        runs[RW * r] = i; runs[RW * r + 1] = (uint32_t)(f_task[i] >> 32); runs[RW * r + 2] = f_meta[i];
        if (d16) runs[RW * r + 3] = (uint32_t)f_task[i];
        r++;
    }
    if (d16) {  // the unit stream of worker w starts at unit 4 * out0 (8-byte aligned; at most 3 units per record): pairs of units leave as 4-byte stores
        if (threadIdx.x == 0 && (n_units & 1u)) s_units[n_units] = 0;
        __syncthreads();
        uint32_t *dst = reinterpret_cast<uint32_t *>(co.units + (size_t)out0 * 4);
        const uint32_t *src = reinterpret_cast<const uint32_t *>(s_units);
        // system-scope (write-through) stores: the units start crossing PCIe as they are issued instead of when the launch's final release writes the L2 back
        // (19.4 instead of 20.0 us)
        for (uint32_t i = threadIdx.x; i < (n_units + 1) / 2; i += blockDim.x) __hip_atomic_store(dst + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    } else for (uint32_t i = threadIdx.x; i < tot; i += blockDim.x) __hip_atomic_store(co.rec_lo + out0 + i, (uint32_t)f_task[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ------------------------------------------------------------------------------------------------ resident cluster tables (f1)
// Rows of the worker table that changed since the last tick (tasks finished or started, time limits running down): one thread per (row, resource).
__global__ void __launch_bounds__(256) k_scatter_worker_rows(uint64_t *__restrict__ free_, int64_t *__restrict__ rem, uint32_t R, uint32_t n,
                                                             const uint32_t *__restrict__ idx, const uint64_t *__restrict__ rows, const int64_t *__restrict__ new_rem) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * R) return;
    const uint32_t i = t / R, r = t % R, w = idx[i];
    free_[(size_t)w * R + r] = rows[t];
    if (r == 0 && new_rem) rem[w] = new_rem[i];
}

// Workers joined or left (on_new_worker / on_remove_worker, server/reactor.rs:20-186): the three row blocks are re-packed into a fresh table.
// Row i of the new table comes from old row src[i] (src[i] < W_old) or from row src[i] - W_old of the staged new workers (pinned memory).
__global__ void __launch_bounds__(256) k_repack_worker_rows(const uint64_t *__restrict__ old_total, const uint64_t *__restrict__ old_free, const int64_t *__restrict__ old_rem,
                                                            uint32_t W_old, uint32_t R, uint32_t W_new, const uint32_t *__restrict__ src,
                                                            const uint64_t *__restrict__ add_total, const uint64_t *__restrict__ add_free, const int64_t *__restrict__ add_rem,
                                                            uint64_t *__restrict__ new_total, uint64_t *__restrict__ new_free, int64_t *__restrict__ new_rem) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= W_new * R) return;
    const uint32_t i = t / R, r = t % R, sidx = src[i];
    if (sidx < W_old) {
        new_total[t] = old_total[(size_t)sidx * R + r]; new_free[t] = old_free[(size_t)sidx * R + r];
        if (r == 0) new_rem[i] = old_rem[sidx];
    } else {
        const uint32_t a = sidx - W_old;
        new_total[t] = add_total[(size_t)a * R + r]; new_free[t] = add_free[(size_t)a * R + r];
        if (r == 0) new_rem[i] = add_rem[a];
    }
}

// ------------------------------------------------------------------------------------------------ resident ready-set deltas (f1)
// TaskQueues::add_ready_task / take_tasks / remove on the HBM-resident columns (scheduler/taskqueue.rs:37-43,146-217,304-355).
// Removal is a tombstone in the rq column (RQ_TOMBSTONE); k_rebuild drops tombstones and merges a sorted batch of new tasks.

// one thread per id to remove: binary search in the sorted id column
__global__ void __launch_bounds__(256) k_mark_removed(const uint64_t *__restrict__ ids, uint32_t *__restrict__ rq, uint64_t n,
                                                      const uint64_t *__restrict__ rm, uint32_t n_rm, uint32_t *__restrict__ n_done) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_rm) return;
    const uint64_t want = rm[t];
    uint64_t lo = 0, hi = n;
    while (lo < hi) { uint64_t mid = (lo + hi) >> 1; if (ids[mid] < want) lo = mid + 1; else hi = mid; }
    // a removed-and-re-added id can sit next to its own tombstone: take the live one.  The claim is an atomic exchange, so an id listed twice
    // in one call (or removed concurrently) is counted once — TaskQueue::remove is idempotent (scheduler/taskqueue.rs:196-217).
    for (; lo < n && ids[lo] == want; lo++) {
        if (rq[lo] == RQ_TOMBSTONE) continue;
        if (atomicExch(&rq[lo], RQ_TOMBSTONE) != RQ_TOMBSTONE) { atomicAdd(n_done, 1u); break; }
    }
}

// HQTICK_FLAG_CONSUME_IN_TICK, a tick that failed after its selection was launched: put back what it took.  K1 of that tick wrote a valid group key for every task
// that was live when it ran (tombstones got GKEY_INVALID) and nothing has overwritten the column since, so "group key valid, request id a tombstone" is exactly
// "tombstoned by this tick's K4" — and the request id it had is key % Q.  One streaming pass (6 B per task), no search, no list of what was selected.
__global__ void __launch_bounds__(256) k_restore_consumed(const uint16_t *__restrict__ gkey, uint32_t *__restrict__ rq, uint64_t n, uint32_t Q, uint32_t *__restrict__ n_done) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint32_t mine = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint16_t g = gkey[i];
        if (g != GKEY_INVALID && rq[i] == RQ_TOMBSTONE) { rq[i] = (uint32_t)g % Q; mine++; }
    }
    if (mine) atomicAdd(n_done, mine);
}

// live tasks per 256-task slice
__global__ void __launch_bounds__(256) k_live_count(const uint32_t *__restrict__ rq, uint64_t n, uint32_t n_slices, uint32_t *__restrict__ slice_cnt) {
    const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6), lane = lane_id();
    if (wave >= n_slices) return;
    const uint64_t begin = (uint64_t)wave * 256;
    uint32_t c = 0;
#pragma unroll
    for (int u = 0; u < 4; u++) { const uint64_t i = begin + (uint64_t)u * 64 + lane; c += (uint32_t)__popcll(__ballot(i < n && rq[i] != RQ_TOMBSTONE)); }
    if (lane == 0) slice_cnt[wave] = c;
}

// Rebuild, pass 1 — one wavefront per 256-task slice of the old columns: every surviving task moves to
//     slice_off[slice] + (live tasks before it in the slice) + (new tasks with a smaller id),
// the last term found by a binary search inside the slice's share [lb0, lb1) of the sorted batch (two searches per slice).
// Also records, per old position, the live count before it inside its slice (u8) for pass 2.
__global__ void __launch_bounds__(256) k_rebuild(const uint64_t *__restrict__ oid, const uint64_t *__restrict__ oprio, const uint32_t *__restrict__ orq,
                                                 uint64_t n, uint32_t n_slices, const uint32_t *__restrict__ slice_off,
                                                 const uint64_t *__restrict__ aid, uint32_t n_add, uint64_t *__restrict__ nid, uint64_t *__restrict__ nprio,
                                                 uint32_t *__restrict__ nrq, uint8_t *__restrict__ pre8, uint32_t *__restrict__ err_flag) {
    const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6), lane = lane_id();
    if (wave >= n_slices) return;
    const uint64_t begin = (uint64_t)wave * 256;
    const uint32_t len = (uint32_t)(n - begin < 256 ? n - begin : 256);
    uint64_t idv[4], pv[4];
    uint32_t qv[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const uint32_t e = (uint32_t)u * 64 + lane;
        const bool ok = e < len;
        idv[u] = ok ? oid[begin + e] : 0; pv[u] = ok ? oprio[begin + e] : 0; qv[u] = ok ? orq[begin + e] : RQ_TOMBSTONE;
    }
    auto lower_bound_add = [&](uint64_t key) -> uint32_t { uint32_t lo = 0, hi = n_add; while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (aid[mid] < key) lo = mid + 1; else hi = mid; } return lo; };
    const uint32_t lb0 = (wave == 0 || n_add == 0) ? 0u : lower_bound_add(oid[begin]);
    const uint32_t lb1 = (wave + 1 >= n_slices || n_add == 0) ? n_add : lower_bound_add(oid[begin + 256]);
    const uint32_t base = slice_off[wave];
    const uint64_t lt = (1ull << lane) - 1ull;
    uint32_t run = 0;
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const uint32_t e = (uint32_t)u * 64 + lane;
        const bool live = qv[u] != RQ_TOMBSTONE;
        const uint64_t b = __ballot(live);
        const uint32_t pre = run + (uint32_t)__popcll(b & lt);
        run += (uint32_t)__popcll(b);
        if (e < len) pre8[begin + e] = (uint8_t)pre;  // <= 255: at most e live tasks precede position e
        if (!live) continue;
        uint32_t lo = lb0, hi = lb1;  // new tasks of this slice with a smaller id
        while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (aid[mid] < idv[u]) lo = mid + 1; else hi = mid; }
        if (lo < lb1 && aid[lo] == idv[u]) atomicOr(err_flag, 4u);  // the id is already in the ready set
        const uint64_t dst = (uint64_t)base + pre + lo;
        nid[dst] = idv[u]; nprio[dst] = pv[u]; nrq[dst] = qv[u];
    }
}

// Rebuild, pass 2 — one thread per new task: it lands behind the live old tasks with a smaller id and the new tasks before it.
__global__ void __launch_bounds__(256) k_merge_adds(const uint64_t *__restrict__ oid, uint64_t n, const uint32_t *__restrict__ slice_off,
                                                    const uint8_t *__restrict__ pre8, uint32_t n_live, const uint64_t *__restrict__ aid,
                                                    const uint64_t *__restrict__ aprio, const uint32_t *__restrict__ arq, uint32_t n_add,
                                                    uint64_t *__restrict__ nid, uint64_t *__restrict__ nprio, uint32_t *__restrict__ nrq, uint32_t *__restrict__ err_flag) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_add) return;
    const uint64_t key = aid[j];
    // the batch is validated here, not in a host loop over it: ids strictly ascending (8), request id 0xFFFFFFFF reserved for the tombstone (16)
    if (j > 0 && aid[j - 1] >= key) atomicOr(err_flag, 8u);
    if (arq[j] == RQ_TOMBSTONE) atomicOr(err_flag, 16u);
    uint64_t lo = 0, hi = n;  // first old position with id >= key
    if (n && oid[n - 1] < key) lo = n;  // the usual case: fresh ids sort behind everything resident
    else while (lo < hi) { uint64_t mid = (lo + hi) >> 1; if (oid[mid] < key) lo = mid + 1; else hi = mid; }
    const uint32_t live_before = lo >= n ? n_live : slice_off[lo >> 8] + pre8[lo];
    const uint64_t dst = (uint64_t)live_before + j;
    nid[dst] = key; nprio[dst] = aprio[j]; nrq[dst] = arq[j];
}

// A PACKED batch of new ready tasks (hqtick_ready_add_packed) expanded into the three columns the merge works on: what crossed PCIe is 2-6 bytes per task — the
// tasks a submit or a finished wave makes ready have ids in a few consecutive runs, one priority per run, and a request id that fits 16 bits — instead of 20.
// One thread per task: its id run and its priority run by binary search over the runs' first positions (a handful of entries, read from pinned memory in place).
struct PackedAdds {
    uint32_t n, n_id_runs, n_prio_runs;
    const uint64_t *id_start; const uint32_t *id_first;      // [n_id_runs] first id / first position of the run;  id_first[n_id_runs] = n
    const uint32_t *id_off;                                   // [n] offset from the run's first id, or nullptr: position inside the run (consecutive ids)
    const uint64_t *prio_value; const uint32_t *prio_first;   // [n_prio_runs], prio_first[n_prio_runs] = n
    const uint16_t *rq;                                       // [n]
};
// err_flag != nullptr: the batch goes straight to the tail of the resident columns (the append path) and is validated here — ascending ids, behind the resident set,
// no reserved request id; nullptr: the merge kernels behind this one validate it.
__global__ void __launch_bounds__(256) k_unpack_adds(PackedAdds pa, uint64_t *__restrict__ aid, uint64_t *__restrict__ aprio, uint32_t *__restrict__ arq,
                                                     uint64_t last_resident_id, uint32_t *__restrict__ err_flag) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= pa.n) return;
    uint32_t lo = 0, hi = pa.n_id_runs;  // last run with id_first <= j
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (pa.id_first[mid] <= j) lo = mid; else hi = mid; }
    const uint32_t off = pa.id_off ? pa.id_off[j] : j - pa.id_first[lo];
    aid[j] = pa.id_start[lo] + off;
    if (err_flag) {
        const uint64_t key = pa.id_start[lo] + off;
        uint64_t prev = last_resident_id;
        if (j > 0) { const uint32_t pl = pa.id_first[lo] == j ? lo - 1 : lo; prev = pa.id_start[pl] + (pa.id_off ? pa.id_off[j - 1] : (j - 1) - pa.id_first[pl]); }
        if (prev >= key) atomicOr(err_flag, j > 0 ? 8u : 4u);
        if (pa.rq[j] == 0xFFFFu) atomicOr(err_flag, 16u);
    }
    uint32_t plo = 0, phi = pa.n_prio_runs;
    while (phi - plo > 1) { const uint32_t mid = (plo + phi) >> 1; if (pa.prio_first[mid] <= j) plo = mid; else phi = mid; }
    aprio[j] = pa.prio_value[plo];
    const uint32_t q = pa.rq[j];
    arq[j] = q == 0xFFFFu ? RQ_TOMBSTONE : q;  // (0xFFFF is not a request id of the packed form: it trips the merge's reserved-id check)
}

// Fresh ids sort behind everything resident (TaskIds are minted ascending: the usual batch): the batch is APPENDED behind the resident columns — tombstones stay where
// they are until the next compaction — instead of re-writing all three columns (k_rebuild: 40 MB at 1 M tasks).  Validates the batch like k_merge_adds does.
__global__ void __launch_bounds__(256) k_append_adds(const uint64_t *__restrict__ aid, const uint64_t *__restrict__ aprio, const uint32_t *__restrict__ arq, uint32_t n_add,
                                                     uint64_t last_resident_id, uint64_t *__restrict__ nid, uint64_t *__restrict__ nprio, uint32_t *__restrict__ nrq,
                                                     uint32_t *__restrict__ err_flag) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_add) return;
    const uint64_t key = aid[j];
    // not ascending / not behind the resident set (the host checked the latter; last_resident_id = UINT64_MAX: nothing is resident, the first id is free)
    if (j > 0 ? aid[j - 1] >= key : (last_resident_id != 0xFFFFFFFFFFFFFFFFull && key <= last_resident_id)) atomicOr(err_flag, j > 0 ? 8u : 4u);
    const uint32_t q = arq[j];
    if (q == RQ_TOMBSTONE) atomicOr(err_flag, 16u);
    nid[j] = key; nprio[j] = aprio[j]; nrq[j] = q;
}

// ------------------------------------------------------------------------------------------------ position of given tasks in their queues
// One wavefront per wanted id: its group key and its rank inside the group (= how many tasks of the same (level, rq) precede it in
// id order), from the sorted id column, the key column and the scanned slice table.  Used for the few ready tasks that are in
// state Retracting (scheduler/mapping.rs:66-80): the host needs to know which of them this tick takes.  key = 0xFFFFFFFF: not found.
__global__ void __launch_bounds__(256) k_rank_of(const uint64_t *__restrict__ ids, const uint16_t *__restrict__ gkey, uint64_t n,
                                                 const uint32_t *__restrict__ wave_off, uint32_t stride, uint32_t tasks_per_wave,
                                                 const uint64_t *__restrict__ want, uint32_t n_want, uint32_t *__restrict__ out_key,
                                                 uint32_t *__restrict__ out_rank) {
    const uint32_t t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = lane_id();
    if (t >= n_want) return;
    const uint64_t key = want[t];
    uint64_t lo = 0, hi = n;
    while (lo < hi) { uint64_t mid = (lo + hi) >> 1; if (ids[mid] < key) lo = mid + 1; else hi = mid; }
    while (lo < n && ids[lo] == key && gkey[lo] == GKEY_INVALID) lo++;  // a removed-and-re-added id sits behind its tombstone
    if (lo >= n || ids[lo] != key || gkey[lo] == GKEY_INVALID) { if (lane == 0) { out_key[t] = 0xFFFFFFFFu; out_rank[t] = 0; } return; }
    const uint32_t g = gkey[lo];
    const uint64_t slice = lo / tasks_per_wave, begin = slice * tasks_per_wave;
    uint32_t before = 0;
    for (uint64_t b = begin; b < lo; b += 64) { const uint64_t i = b + lane; before += (uint32_t)__popcll(__ballot(i < lo && gkey[i] == g)); }
    if (lane == 0) { out_key[t] = g; out_rank[t] = wave_off[(size_t)g * stride + slice] + before; }
}

// ------------------------------------------------------------------------------------------------ upload of an unsorted ready set
// Bitonic sort of (id, priority, rq) triples by id in global memory — one launch per (k, j) stage.  Used once per
// hqtick_upload_ready(sorted = 0); n is padded to a power of two with id = 2^64 - 1 sentinels that sort to the end.
__global__ void __launch_bounds__(256) k_bitonic_step(uint64_t *__restrict__ id, uint64_t *__restrict__ prio, uint32_t *__restrict__ rq,
                                                      uint64_t n_pow2, uint64_t j, uint64_t k) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pow2) return;
    const uint64_t l = i ^ j;
    if (l <= i) return;
    const uint64_t a = id[i], b = id[l];
    const bool up = (i & k) == 0;
    if (up ? a > b : a < b) {
        id[i] = b; id[l] = a;
        const uint64_t pa = prio[i]; prio[i] = prio[l]; prio[l] = pa;
        const uint32_t qa = rq[i]; rq[i] = rq[l]; rq[l] = qa;
    }
}

__global__ void __launch_bounds__(256) k_fill_sentinel(uint64_t *__restrict__ id, uint64_t from, uint64_t to) {
    const uint64_t i = from + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < to) id[i] = 0xFFFFFFFFFFFFFFFFull;
}

// first adjacent pair that is not strictly ascending (duplicates / sentinel collisions) -> flag
__global__ void __launch_bounds__(256) k_check_sorted(const uint64_t *__restrict__ id, uint64_t n, uint32_t *__restrict__ flag) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i + 1 < n && id[i] >= id[i + 1]) atomicOr(flag, 1u);
}

}  // namespace

// ================================================================================================ host wrappers
hipError_t distinct_priorities(const uint64_t *prio, const uint32_t *rq, uint64_t n, uint64_t *set, uint32_t *flags, hipStream_t s) {
    if (n == 0) return hipSuccess;
    uint64_t blocks = (n + 1023) / 1024;  // K1's geometry (see the kernel): one workgroup per 1024 tasks; every workgroup publishes each distinct value once (a read of the set's slot, mostly)
    hipLaunchKernelGGL(k_distinct_priorities, dim3((unsigned)blocks), dim3(256), 0, s, prio, rq, n, set, flags);
    return hipGetLastError();
}

hipError_t sort_levels(uint64_t *set, uint32_t *flags, uint64_t *levels, uint32_t *n_levels, uint64_t *host_out, uint32_t seq, hipStream_t s) {
    hipLaunchKernelGGL(k_sort_levels, dim3(1), dim3(1024), LEVEL_CAP * 8, s, set, flags, levels, n_levels, host_out, seq);
    return hipGetLastError();
}

// K1b for long rows (more than 1024 slices: the BASELINE sizes): one WORKGROUP per group row.  A thread takes 16 entries of a 4096-entry step as four
// coalesced dwordx4 (consecutive lanes, consecutive 16 bytes: 8 cache lines per load instruction where the one-wavefront-per-row kernel above touches
// 64), the four 1024-entry quarters are scanned side by side (wave scan + 16 wave totals through LDS, one barrier per step).  3 907 entries per row at
// C3: 4.2 us against 6.3.
__global__ void __launch_bounds__(256) k_scan_rows_wg(uint32_t *__restrict__ wave_tab, uint32_t n_waves, uint32_t stride, uint32_t G,
                                                      uint32_t *__restrict__ hist, uint32_t *__restrict__ err_in,
                                                      uint32_t *__restrict__ err_out, uint32_t n_eval_blocks, EvalArgs ea) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ uint32_t s_wt[2][4][4];  // [step parity][quarter][wavefront]: inclusive totals
    if (blockIdx.x < n_eval_blocks) {   // ride-along workgroups (dispatched first): K2, as in k_scan_rows
        worker_eval_block(smem, blockIdx.x, ea);
        return;
    }
    const uint32_t row = blockIdx.x - n_eval_blocks;
    if (row == 0 && threadIdx.x == 0 && err_out) { err_out[0] = err_in[0]; err_in[0] = 0; }
    if (row >= G) return;
    const uint32_t lane = lane_id(), wv = threadIdx.x >> 6;
    uint32_t *r = wave_tab + (size_t)row * stride;
    uint32_t carry = 0;
    for (uint32_t base = 0, par = 0; base < n_waves; base += 4096, par ^= 1) {
        uint4 v[4];
        uint32_t sum[4], incl[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {  // stride is a multiple of 16 entries: every 4-entry group is inside the row or fully outside
            const uint32_t i = base + k * 1024 + threadIdx.x * 4;
            v[k] = i < stride ? *reinterpret_cast<const uint4 *>(r + i) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {  // padding entries (index >= n_waves) count as 0
            const uint32_t i = base + k * 1024 + threadIdx.x * 4;
            const uint32_t t0 = i < n_waves ? v[k].x : 0u, t1 = i + 1 < n_waves ? v[k].y : 0u, t2 = i + 2 < n_waves ? v[k].z : 0u, t3 = i + 3 < n_waves ? v[k].w : 0u;
            v[k].x = 0; v[k].y = t0; v[k].z = t0 + t1; v[k].w = t0 + t1 + t2;
            sum[k] = t0 + t1 + t2 + t3; incl[k] = sum[k];
        }
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
#pragma unroll
            for (int k = 0; k < 4; k++) { const uint32_t t = __shfl_up(incl[k], off, 64); if ((int)lane >= off) incl[k] += t; }
        }
        if (lane == 63) {
#pragma unroll
            for (int k = 0; k < 4; k++) s_wt[par][k][wv] = incl[k];
        }
        __syncthreads();  // (the other parity's totals are rewritten only after every thread has passed the NEXT barrier: no second one needed)
        uint32_t off = carry;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t before = 0, all = 0;
#pragma unroll
            for (uint32_t w = 0; w < 4; w++) { const uint32_t t = s_wt[par][k][w]; all += t; if (w < wv) before += t; }
            const uint32_t excl = off + before + incl[k] - sum[k];
            const uint32_t i = base + k * 1024 + threadIdx.x * 4;
            if (i < stride) *reinterpret_cast<uint4 *>(r + i) = make_uint4(v[k].x + excl, v[k].y + excl, v[k].z + excl, v[k].w + excl);
            off += all;
        }
        carry = off;
    }
    if (threadIdx.x == 0) hist[row] = carry;
}

static uint32_t lds_levels_for(uint32_t L) { return L <= 1024 ? L : 0; }

// An empty kernel of K1's grid, launched the way the measured kernels are: what a dispatch bracketed by its own start / stop events records when the kernel does
// nothing (bench.py: `roofline.empty_launch_us` — the floor of the duration `roofline.frac` is priced on).
static __global__ void __launch_bounds__(256) k_empty_grid(uint32_t) {}
hipError_t empty_like_level_hist(WaveGeom geom, hipStream_t s) {
    if (geom.n_waves == 0) return hipSuccess;
    if (geom.waves_per_block == 4) HQK_TIMED_LAUNCH(k_empty_grid, dim3((geom.n_waves + 3) / 4), dim3(256), 0, s, geom.n_waves);
    else HQK_TIMED_LAUNCH(k_empty_grid, dim3(geom.n_waves), dim3(64), 0, s, geom.n_waves);
    return hipGetLastError();
}

hipError_t level_hist(const uint64_t *prio, const uint32_t *rq, uint64_t n, const uint64_t *levels, const uint64_t *levels_host, uint32_t L, uint32_t Q,
                WaveGeom geom, uint32_t *wave_tab, uint16_t *gkey, uint32_t *err_flag, const WorkerEvalArgs *ride_along, hipStream_t s, const uint32_t *n_levels_dev) {
    if (n == 0 || geom.n_waves == 0) return hipSuccess;
    const uint32_t G = L * Q;
    if (n_levels_dev && (L > 4 || G > 64 || geom.waves_per_block != 4)) return hipErrorInvalidValue;   // (the speculative launch exists for the small variant only)
    const bool small = L <= 4 && (levels_host != nullptr || n_levels_dev != nullptr);
    const uint32_t ll = small ? 0 : lds_levels_for(L);
    Levels4 l4{};
    if (small && !n_levels_dev) for (uint32_t i = 0; i < L; i++) l4.v[i] = levels_host[i];
    EvalArgs ea{};
    uint32_t neb = 0; size_t eval_lds = 0;
    if (ride_along && ride_along->W && ride_along->rt.n_variants) {
        ea = EvalArgs{ride_along->total, ride_along->free_, ride_along->remaining_ns, ride_along->W, ride_along->R, ride_along->rt, ride_along->n_entries, ride_along->flags, ride_along->tmc};
        neb = (ride_along->W + 31) / 32; eval_lds = worker_eval_lds(ride_along->R, ride_along->rt.n_variants, ride_along->n_entries);
    }
    hipError_t e;
#define HQK_LAUNCH_HIST(WPB, SMALL, NCOPY, GRID, BLOCK)                                                                                                     \
    do {                                                                                                                                             \
        size_t lds = (size_t)ll * 8 + (size_t)(WPB) * G * 4 * (NCOPY);                                                                                \
        if (eval_lds > lds) lds = eval_lds;                                                                                                          \
        auto kern = k_level_hist<WPB, SMALL, NCOPY>;                                                                                                        \
        if (lds > 48 * 1024 && (e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return e; \
        HQK_TIMED_LAUNCH(kern, dim3((GRID) + neb), dim3(BLOCK), lds, s, prio, rq, n, levels, l4, L, Q, geom.tasks_per_wave, geom.n_waves, geom.tab_stride, ll, wave_tab, gkey, \
                           err_flag, neb, ea, n_levels_dev);                                                                                                       \
    } while (0)
    if (geom.waves_per_block == 4) {
        if (small && G <= 64) HQK_LAUNCH_HIST(4, true, 8, (geom.n_waves + 3) / 4, 256);  // a handful of groups: eight copies of the counter row (8 KB of LDS at most)
        else if (small) HQK_LAUNCH_HIST(4, true, 1, (geom.n_waves + 3) / 4, 256);
        else HQK_LAUNCH_HIST(4, false, 1, (geom.n_waves + 3) / 4, 256);
    } else { if (small) HQK_LAUNCH_HIST(1, true, 1, geom.n_waves, 64); else HQK_LAUNCH_HIST(1, false, 1, geom.n_waves, 64); }
#undef HQK_LAUNCH_HIST
    return hipGetLastError();
}

hipError_t scan_waves(uint32_t *wave_tab, WaveGeom geom, uint32_t G, uint32_t *hist, uint32_t *err_in, uint32_t *err_out, hipStream_t s, const WorkerEvalArgs *ride_along) {
    if (G == 0) return hipSuccess;
    EvalArgs ea{};
    uint32_t neb = 0; size_t lds = 0;
    if (ride_along && ride_along->W && ride_along->rt.n_variants) {
        ea = EvalArgs{ride_along->total, ride_along->free_, ride_along->remaining_ns, ride_along->W, ride_along->R, ride_along->rt, ride_along->n_entries, ride_along->flags, ride_along->tmc};
        neb = (ride_along->W + 31) / 32; lds = worker_eval_lds(ride_along->R, ride_along->rt.n_variants, ride_along->n_entries);
        hipError_t e;
        if (lds > 48 * 1024 && (e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_scan_rows), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return e;
    }
    if (geom.n_waves > 1024) {  // long rows: a workgroup each
        hipError_t e;
        if (lds > 48 * 1024 && (e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_scan_rows_wg), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return e;
        HQK_TIMED_LAUNCH(k_scan_rows_wg, dim3(G + neb), dim3(256), lds, s, wave_tab, geom.n_waves, geom.tab_stride, G, hist, err_in, err_out, neb, ea);
        return hipGetLastError();
    }
    HQK_TIMED_LAUNCH(k_scan_rows, dim3((G + 3) / 4 + neb), dim3(256), lds, s, wave_tab, geom.n_waves, geom.tab_stride, G, hist, err_in, err_out, neb, ea);
    return hipGetLastError();
}

static __global__ void __launch_bounds__(256) k_copy16(const uint4 *__restrict__ src, uint4 *__restrict__ dst, uint32_t n16) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += gridDim.x * blockDim.x) dst[i] = src[i];  // (launched with one element per thread)
}

hipError_t copy_pinned_to_hbm(const void *src, void *dst, size_t bytes, hipStream_t s) {
    const uint32_t n16 = (uint32_t)((bytes + 15) / 16);
    if (!n16) return hipSuccess;
    hipLaunchKernelGGL(k_copy16, dim3((n16 + 255) / 256), dim3(256), 0, s, reinterpret_cast<const uint4 *>(src), reinterpret_cast<uint4 *>(dst), n16);  // one PCIe round trip
    return hipGetLastError();
}

hipError_t select_scatter(const uint64_t *task_id, const uint16_t *gkey, uint64_t n, uint32_t Q, uint32_t G, WaveGeom geom,
                    const uint32_t *wave_off, const uint32_t *take_host, const uint32_t *take_dev, uint64_t *sel_task, uint16_t *sel_key,
                    const void *plan_src, void *plan_dst, size_t plan_bytes, uint32_t *mark_rq, hipStream_t s, uint32_t mark_and_select) {
    const uint32_t n16 = (uint32_t)((plan_bytes + 15) / 16);
    const bool sel = n != 0 && geom.n_waves != 0 && G != 0;
    hipError_t e;
    if (sel && geom.waves_per_block == 4 && G <= 64) {
        // small plan: take/base travel in the kernel arguments; ride-along workgroups copy the whole plan into HBM for K5a/K5b
        size_t lds = (size_t)8 * G * 4;
        const uint32_t nsb = (geom.n_waves + 3) / 4, ncb = n16 ? (n16 + 255) / 256 : 0;  // one 16-byte PCIe read per thread: a single round trip (four per thread cost the launch 2.7 us)
        if (G <= 16) {
            SelPlanN<16> pa{};
            for (uint32_t g = 0; g < G; g++) { pa.take[g] = take_host[g]; pa.base[g] = take_host[G + g]; pa.tnc[g] = take_host[2 * G + g]; pa.tsb[g] = take_host[3 * G + g]; }
            HQK_TIMED_LAUNCH((k_select<4, 0, 16>), dim3(nsb + ncb), dim3(256), lds, s, task_id, gkey, n, Q, G, geom.tasks_per_wave, geom.n_waves, geom.tab_stride, wave_off,
                               (const uint32_t *)nullptr, (const uint32_t *)nullptr, pa, sel_task, sel_key, nsb, reinterpret_cast<const uint4 *>(plan_src),
                               reinterpret_cast<uint4 *>(plan_dst), n16, mark_rq, mark_and_select);
        } else {
            SelPlanN<64> pa{};
            for (uint32_t g = 0; g < G; g++) { pa.take[g] = take_host[g]; pa.base[g] = take_host[G + g]; pa.tnc[g] = take_host[2 * G + g]; pa.tsb[g] = take_host[3 * G + g]; }
            HQK_TIMED_LAUNCH((k_select<4, 0, 64>), dim3(nsb + ncb), dim3(256), lds, s, task_id, gkey, n, Q, G, geom.tasks_per_wave, geom.n_waves, geom.tab_stride, wave_off,
                               (const uint32_t *)nullptr, (const uint32_t *)nullptr, pa, sel_task, sel_key, nsb, reinterpret_cast<const uint4 *>(plan_src),
                               reinterpret_cast<uint4 *>(plan_dst), n16, mark_rq, mark_and_select);
        }
        return hipGetLastError();
    }
    if (n16) {  // larger plans: copy first (own launch), then select from the HBM copy
        hipLaunchKernelGGL(k_copy16, dim3((n16 + 255) / 256), dim3(256), 0, s, reinterpret_cast<const uint4 *>(plan_src), reinterpret_cast<uint4 *>(plan_dst), n16);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    if (!sel) return hipSuccess;
    SelPlanN<1> none{};
    if (geom.waves_per_block == 4) {
        size_t lds = (size_t)8 * G * 4;
        auto kern = k_select<4, 1, 1>;
        if (lds > 48 * 1024 && (e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return e;
        const uint32_t nsb = (geom.n_waves + 3) / 4;
        HQK_TIMED_LAUNCH(kern, dim3(nsb), dim3(256), lds, s, task_id, gkey, n, Q, G, geom.tasks_per_wave, geom.n_waves, geom.tab_stride, wave_off, take_dev, take_dev + G, none,
                           sel_task, sel_key, nsb, (const uint4 *)nullptr, (uint4 *)nullptr, 0u, mark_rq, mark_and_select);
        return hipGetLastError();
    }
    size_t lds = (size_t)G * 4;
    auto kern = k_select<1, 2, 1>;
    if (lds > 48 * 1024 && (e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return e;
    HQK_TIMED_LAUNCH(kern, dim3(geom.n_waves), dim3(64), lds, s, task_id, gkey, n, Q, G, geom.tasks_per_wave, geom.n_waves, geom.tab_stride, wave_off, take_dev, take_dev + G, none,
                       sel_task, sel_key, geom.n_waves, (const uint4 *)nullptr, (uint4 *)nullptr, 0u, mark_rq, mark_and_select);
    return hipGetLastError();
}

size_t worker_eval_lds(uint32_t R, uint32_t n_variants, uint32_t n_entries) {
    return (size_t)32 * R * 16 + 32 * 8 + (size_t)n_entries * 13 + (size_t)n_variants * 12 + 4 + 16;
}

hipError_t worker_eval(const uint64_t *total, const uint64_t *free_, const int64_t *remaining_ns, uint32_t W, uint32_t R, RequestTable rt,
                 uint32_t n_entries, uint8_t *flags, uint32_t *tmc, hipStream_t s) {
    if (W == 0 || rt.n_variants == 0) return hipSuccess;
    size_t lds = worker_eval_lds(R, rt.n_variants, n_entries);
    hipError_t e;
    if (lds > 48 * 1024 && (e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_worker_eval), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_worker_eval, dim3((W + 31) / 32), dim3(256), lds, s, EvalArgs{total, free_, remaining_ns, W, R, rt, n_entries, flags, tmc});
    return hipGetLastError();
}

size_t sweep_bits_lds(uint32_t max_workers_per_key) { return ((size_t)((max_workers_per_key + 63) / 64) * 65 + 1) * 4; }

hipError_t sweep_bits(MapKeys mk, uint32_t max_count, uint32_t max_workers_per_key, hipStream_t s) {
    if (mk.n_keys == 0) return hipSuccess;
    if (max_workers_per_key > SWEEP_LDS_COUNTS) return hipErrorInvalidValue;
    size_t lds = sweep_bits_lds(max_workers_per_key);
    hipError_t e;
    if (lds > 48 * 1024 && (e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_sweep_bits), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return e;
    HQK_TIMED_LAUNCH(k_sweep_bits, dim3((max_count + 1 + 3) / 4, mk.n_keys), dim3(256), lds, s, mk);
    return hipGetLastError();
}

uint32_t expand_mapping_sort_cap(uint32_t max_items) { uint32_t p = 1; while (p < max_items) p <<= 1; return p; }

size_t expand_mapping_lds(uint32_t max_items, uint32_t n_keys, uint32_t max_out, bool may_reorder) {
    return (size_t)max_items * 12 + (size_t)max_out * 10 + 2 + ((size_t)9 * n_keys + 1 + 4) * 4 + n_keys + 16 + (may_reorder ? (size_t)expand_mapping_sort_cap(max_items) * 4 : 0) +
           (max_out ? (size_t)RUN_STAGE * 16 + 4 + (size_t)max_out * 6 + 8 : 0);  // max_out != 0 = compact emission: the run stage + the 16-bit unit stream (delta mode)
}

hipError_t expand_mapping(MapKeys mk, uint32_t W, const uint64_t *sel_task, const uint16_t *sel_key, uint32_t Q, uint32_t max_items,
                    uint64_t *rec_task, uint8_t *rec_variant, uint8_t *rec_kind, uint32_t *err_flag, CompactOut co, uint32_t max_out, bool may_reorder, hipStream_t s) {
    if (W == 0) return hipSuccess;
    if (!co.rec_lo) max_out = 0;
    size_t lds = expand_mapping_lds(max_items, mk.n_keys, max_out, may_reorder);
    const uint32_t sort_cap = may_reorder ? expand_mapping_sort_cap(max_items) : 0;
    hipError_t e;
    if (lds > 48 * 1024 && (e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_expand_mapping), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return e;
    HQK_TIMED_LAUNCH(k_expand_mapping, dim3(W), dim3(256), lds, s, mk, W, sel_task, sel_key, Q, max_items, rec_task, rec_variant, rec_kind, err_flag, co, max_out, sort_cap);
    return hipGetLastError();
}

hipError_t scatter_worker_rows(uint64_t *free_, int64_t *rem, uint32_t R, uint32_t n, const uint32_t *idx, const uint64_t *rows, const int64_t *new_rem, hipStream_t s) {
    if (n == 0 || R == 0) return hipSuccess;
    hipLaunchKernelGGL(k_scatter_worker_rows, dim3((n * R + 255) / 256), dim3(256), 0, s, free_, rem, R, n, idx, rows, new_rem);
    return hipGetLastError();
}

hipError_t repack_worker_rows(const uint64_t *old_total, const uint64_t *old_free, const int64_t *old_rem, uint32_t W_old, uint32_t R, uint32_t W_new, const uint32_t *src,
                              const uint64_t *add_total, const uint64_t *add_free, const int64_t *add_rem, uint64_t *new_total, uint64_t *new_free, int64_t *new_rem, hipStream_t s) {
    if (W_new == 0 || R == 0) return hipSuccess;
    hipLaunchKernelGGL(k_repack_worker_rows, dim3((W_new * R + 255) / 256), dim3(256), 0, s, old_total, old_free, old_rem, W_old, R, W_new, src, add_total, add_free, add_rem, new_total, new_free, new_rem);
    return hipGetLastError();
}

hipError_t ready_restore_consumed(const uint16_t *gkey, uint32_t *rq, uint64_t n, uint32_t Q, uint32_t *n_done, hipStream_t s) {
    if (n == 0 || Q == 0) return hipSuccess;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((n + 255) / 256, 2048);
    hipLaunchKernelGGL(k_restore_consumed, dim3(blocks), dim3(256), 0, s, gkey, rq, n, Q, n_done);
    return hipGetLastError();
}

hipError_t ready_mark_removed(const uint64_t *ids, uint32_t *rq, uint64_t n, const uint64_t *rm, uint32_t n_rm, uint32_t *n_done, hipStream_t s) {
    if (n == 0 || n_rm == 0) return hipSuccess;
    hipLaunchKernelGGL(k_mark_removed, dim3((n_rm + 255) / 256), dim3(256), 0, s, ids, rq, n, rm, n_rm, n_done);
    return hipGetLastError();
}

hipError_t ready_live_count(const uint32_t *rq, uint64_t n, uint32_t *slice_cnt, hipStream_t s) {
    const uint32_t n_slices = (uint32_t)((n + 255) / 256);
    if (n_slices == 0) return hipSuccess;
    hipLaunchKernelGGL(k_live_count, dim3((n_slices + 3) / 4), dim3(256), 0, s, rq, n, n_slices, slice_cnt);
    return hipGetLastError();
}

hipError_t ready_rebuild(const uint64_t *oid, const uint64_t *oprio, const uint32_t *orq, uint64_t n, uint32_t n_live, const uint32_t *slice_off, const uint64_t *aid,
                   const uint64_t *aprio, const uint32_t *arq, uint32_t n_add, uint64_t *nid, uint64_t *nprio, uint32_t *nrq, uint8_t *pre8, uint32_t *err_flag, hipStream_t s) {
    const uint32_t n_slices = (uint32_t)((n + 255) / 256);
    if (n_slices == 0) return hipSuccess;
    hipLaunchKernelGGL(k_rebuild, dim3((n_slices + 3) / 4), dim3(256), 0, s, oid, oprio, orq, n, n_slices, slice_off, aid, n_add, nid, nprio, nrq, pre8, err_flag);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || n_add == 0) return e;
    hipLaunchKernelGGL(k_merge_adds, dim3((n_add + 255) / 256), dim3(256), 0, s, oid, n, slice_off, pre8, n_live, aid, aprio, arq, n_add, nid, nprio, nrq, err_flag);
    return hipGetLastError();
}

hipError_t ready_unpack_adds(uint32_t n, uint32_t n_id_runs, const uint64_t *id_start, const uint32_t *id_first, const uint32_t *id_off, uint32_t n_prio_runs, const uint64_t *prio_value,
                             const uint32_t *prio_first, const uint16_t *rq, uint64_t *aid, uint64_t *aprio, uint32_t *arq, uint64_t last_resident_id, uint32_t *err_flag, hipStream_t s) {
    if (n == 0) return hipSuccess;
    PackedAdds pa{n, n_id_runs, n_prio_runs, id_start, id_first, id_off, prio_value, prio_first, rq};
    HQK_TIMED_LAUNCH(k_unpack_adds, dim3((n + 255) / 256), dim3(256), 0, s, pa, aid, aprio, arq, last_resident_id, err_flag);  // (the caller may wait on the dispatch's own completion signal: time_next_launch)
    return hipGetLastError();
}

hipError_t ready_append(const uint64_t *aid, const uint64_t *aprio, const uint32_t *arq, uint32_t n_add, uint64_t last_resident_id, uint64_t *nid, uint64_t *nprio, uint32_t *nrq,
                        uint32_t *err_flag, hipStream_t s) {
    if (n_add == 0) return hipSuccess;
    HQK_TIMED_LAUNCH(k_append_adds, dim3((n_add + 255) / 256), dim3(256), 0, s, aid, aprio, arq, n_add, last_resident_id, nid, nprio, nrq, err_flag);
    return hipGetLastError();
}

hipError_t sort_ready(uint64_t *id, uint64_t *prio, uint32_t *rq, uint64_t n, uint64_t n_pow2, uint32_t *dup_flag, hipStream_t s) {
    if (n <= 1) return hipSuccess;
    if (n_pow2 > n) hipLaunchKernelGGL(k_fill_sentinel, dim3((unsigned)((n_pow2 - n + 255) / 256)), dim3(256), 0, s, id, n, n_pow2);
    const unsigned blocks = (unsigned)((n_pow2 + 255) / 256);
    for (uint64_t k = 2; k <= n_pow2; k <<= 1)
        for (uint64_t j = k >> 1; j > 0; j >>= 1) hipLaunchKernelGGL(k_bitonic_step, dim3(blocks), dim3(256), 0, s, id, prio, rq, n_pow2, j, k);
    hipLaunchKernelGGL(k_check_sorted, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, id, n, dup_flag);
    return hipGetLastError();
}

hipError_t rank_of(const uint64_t *ids, const uint16_t *gkey, uint64_t n, const uint32_t *wave_off, WaveGeom geom, const uint64_t *want, uint32_t n_want,
             uint32_t *out_key, uint32_t *out_rank, hipStream_t s) {
    if (n_want == 0) return hipSuccess;
    hipLaunchKernelGGL(k_rank_of, dim3((n_want + 3) / 4), dim3(256), 0, s, ids, gkey, n, wave_off, geom.tab_stride, geom.tasks_per_wave, want, n_want, out_key, out_rank);
    return hipGetLastError();
}

}  // namespace hqk
