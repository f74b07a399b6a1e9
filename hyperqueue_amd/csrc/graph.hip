// Kernels and host driver of the device-resident dependency graph (graph.h).  gfx950 only.
#include "graph.h"

#include <algorithm>
#include <cstring>

#include "../../include/hqtick.h"

namespace hqgraph {

namespace {

constexpr uint32_t BIG_GRID = 1024;
constexpr uint32_t SORT_CH = 4096;     // elements per workgroup in the LDS stages of the sort: 256 threads x 16, 48 KB
constexpr uint32_t BFS_LEVELS_PER_SYNC = 8;

// ------------------------------------------------------------------------------------------------------------ hash table
__device__ __forceinline__ uint32_t ht_hash(uint64_t k) {  // murmur3 finaliser
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return (uint32_t)k;
}
__device__ __forceinline__ uint32_t ht_find_pos(const View &v, uint64_t id) {
    uint32_t h = ht_hash(id) & v.ht_mask;
    for (uint32_t probe = 0; probe <= v.ht_mask; probe++) {
        uint64_t k = __atomic_load_n(&v.ht_key[h], __ATOMIC_RELAXED);
        if (k == id) return h;
        if (k == HT_EMPTY) return NONE;
        h = (h + 1) & v.ht_mask;
    }
    return NONE;
}
__device__ __forceinline__ uint32_t ht_find(const View &v, uint64_t id) {
    uint32_t p = ht_find_pos(v, id);
    return p == NONE ? NONE : v.ht_val[p];
}
// 0 = inserted, 1 = the key is already there, 2 = table full
__device__ __forceinline__ int ht_insert(const View &v, uint64_t id, uint32_t slot) {
    uint32_t h = ht_hash(id) & v.ht_mask;
    for (uint32_t probe = 0; probe <= v.ht_mask; probe++) {
        unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long *>(&v.ht_key[h]), (unsigned long long)HT_EMPTY, (unsigned long long)id);
        if (old == HT_EMPTY) { v.ht_val[h] = slot; return 0; }
        if (old == id) return 1;
        h = (h + 1) & v.ht_mask;
    }
    return 2;
}

// position of the calling lane among the lanes that want a slot; one atomic per wavefront
__device__ __forceinline__ uint32_t wave_append(uint32_t *counter, bool want) {
    const uint64_t m = __ballot(want);
    if (m == 0) return 0;
    const uint32_t lane = __lane_id(), leader = (uint32_t)__ffsll((long long)m) - 1u;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
    base = __shfl(base, (int)leader);
    return base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
}

// sum over the wavefront (every lane must call)
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ uint32_t new_slot(const View &v, uint32_t i, uint32_t free_top0, uint32_t n_slots0) {
    return i < free_top0 ? v.free_slot[free_top0 - 1u - i] : n_slots0 + (i - free_top0);
}

// ------------------------------------------------------------------------------------------------------------------- add
__global__ void __launch_bounds__(256) k_g_precheck(View v, const uint64_t *__restrict__ ids, uint32_t n) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (ht_find_pos(v, ids[i]) != NONE) atomicOr(&v.ctl->err, (uint32_t)ERR_EXISTS);
}

__global__ void __launch_bounds__(256) k_g_insert(View v, const uint64_t *__restrict__ ids, const uint64_t *__restrict__ prio, const uint32_t *__restrict__ rq,
                                                  uint32_t n, uint32_t free_top0, uint32_t n_slots0, uint32_t batch_no) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n || v.ctl->err) return;
    const uint32_t sl = new_slot(v, i, free_top0, n_slots0);
    v.id[sl] = ids[i]; v.prio[sl] = prio[i]; v.rq[sl] = rq[i];
    v.unfinished[sl] = 0; v.head[sl] = NONE; v.order[sl] = ((uint64_t)batch_no << 32) | i;
    int r = ht_insert(v, ids[i], sl);
    if (r == 1) atomicOr(&v.ctl->err, (uint32_t)ERR_DUP_IN_BATCH);
    if (r == 2) atomicOr(&v.ctl->err, (uint32_t)ERR_CAPACITY);
}

// undo of k_g_insert after ERR_DUP_IN_BATCH: the batch's keys leave the table, the slots stay free (host counters unchanged)
__global__ void __launch_bounds__(256) k_g_rollback(View v, const uint64_t *__restrict__ ids, uint32_t n, uint32_t free_top0, uint32_t n_slots0) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t sl = new_slot(v, i, free_top0, n_slots0);
    v.unfinished[sl] = ST_FREE;
    uint32_t p = ht_find_pos(v, ids[i]);
    if (p != NONE && v.ht_val[p] == sl) v.ht_key[p] = HT_TOMB;
}

// one thread per dependency entry: resolve it, count it on the consumer, rank it among the batch's edges of its producer
__global__ void __launch_bounds__(256) k_g_link_count(View v, const uint32_t *__restrict__ dep_off, const uint64_t *__restrict__ dep_id, uint32_t n, uint32_t E,
                                                      uint32_t free_top0, uint32_t n_slots0, uint32_t batch_no, uint32_t *__restrict__ edge_ds,
                                                      uint32_t *__restrict__ edge_rank) {
    uint32_t e = blockIdx.x * 256 + threadIdx.x;
    if (e >= E) return;
    if (v.ctl->err) { edge_ds[e] = NONE; return; }
    uint32_t lo = 0, hi = n;  // consumer i: dep_off[i] <= e < dep_off[i + 1]
    while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (dep_off[mid] <= e) lo = mid; else hi = mid; }
    const uint32_t i = lo, cs = new_slot(v, i, free_top0, n_slots0);
    uint32_t ds = ht_find(v, dep_id[e]);
    if (ds != NONE) {
        const uint64_t o = v.order[ds];
        // reactor.rs:193-203: a dependency is kept only if the task map holds it when the consumer is processed; tasks later in the same
        // batch (and the task itself) are not there yet
        if ((uint32_t)(o >> 32) == batch_no && (uint32_t)o >= i) ds = NONE;
    }
    uint32_t rank = 0;
    if (ds != NONE) {
        atomicAdd(&v.unfinished[cs], 1u);
        rank = atomicAdd(&v.tmp_cnt[ds], 1u);
    }
    edge_ds[e] = ds; edge_rank[e] = rank;
}

__global__ void __launch_bounds__(256) k_g_link_alloc(View v, uint32_t E, const uint32_t *__restrict__ edge_ds, const uint32_t *__restrict__ edge_rank, uint32_t cap_edges) {
    __shared__ uint32_t w_runs[4], w_len[4], b_run[4], b_edge[4];
    const uint32_t e = blockIdx.x * 256 + threadIdx.x, wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t ds = NONE, len = 0;
    if (e < E && !v.ctl->err) { ds = edge_ds[e]; if (ds != NONE && edge_rank[e] == 0) len = v.tmp_cnt[ds]; else ds = NONE; }
    // one run record and `len` edge slots per producer of the batch.  Both pool counters sit in one 8-byte word and are bumped by ONE 64-bit
    // atomic per workgroup: same-address atomics serialise at ~10 ns each, one per wavefront and counter was 1 ms at 3 M edges
    const uint64_t m = __ballot(ds != NONE);
    const uint32_t my_run = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    uint32_t incl = len;
    for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(incl, o); if (lane >= (uint32_t)o) incl += t; }
    if (lane == 63) { w_runs[wid] = (uint32_t)__popcll(m); w_len[wid] = incl; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t tr = w_runs[0] + w_runs[1] + w_runs[2] + w_runs[3], tl = w_len[0] + w_len[1] + w_len[2] + w_len[3];
        unsigned long long old = 0;
        if (tr) old = atomicAdd(reinterpret_cast<unsigned long long *>(&v.ctl->run_top), ((unsigned long long)tl << 32) | tr);
        uint32_t br = (uint32_t)old, be = (uint32_t)(old >> 32);
        for (int w = 0; w < 4; w++) { b_run[w] = br; b_edge[w] = be; br += w_runs[w]; be += w_len[w]; }
    }
    __syncthreads();
    if (ds == NONE) return;
    const uint32_t r = b_run[wid] + my_run, base = b_edge[wid] + incl - len;
    if (base + len > cap_edges || r >= cap_edges) { atomicOr(&v.ctl->err, (uint32_t)ERR_CAPACITY); v.tmp_base[ds] = NONE; return; }
    v.run_off[r] = base; v.run_len[r] = len; v.run_next[r] = v.head[ds]; v.head[ds] = r;
    v.tmp_base[ds] = base;
}

__global__ void __launch_bounds__(256) k_g_link_fill(View v, const uint32_t *__restrict__ dep_off, uint32_t n, uint32_t E, uint32_t free_top0, uint32_t n_slots0,
                                                     const uint32_t *__restrict__ edge_ds, const uint32_t *__restrict__ edge_rank) {
    uint32_t e = blockIdx.x * 256 + threadIdx.x;
    if (e >= E) return;
    const uint32_t ds = edge_ds[e];
    if (ds == NONE) return;
    const uint32_t rank = edge_rank[e];
    if (!v.ctl->err) {
        uint32_t lo = 0, hi = n;
        while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (dep_off[mid] <= e) lo = mid; else hi = mid; }
        const uint32_t cs = new_slot(v, lo, free_top0, n_slots0), base = v.tmp_base[ds];
        if (base != NONE) v.edge[base + rank] = make_uint2(cs, v.gen[cs]);
    }
    if (rank == 0) v.tmp_cnt[ds] = 0;  // scratch back to all-zero
}

__global__ void __launch_bounds__(256) k_g_collect_ready(View v, uint32_t n, uint32_t free_top0, uint32_t n_slots0, uint64_t *__restrict__ okey, uint32_t *__restrict__ oval) {
    __shared__ uint32_t wcnt[4], wbase[4];
    const uint32_t i = blockIdx.x * 256 + threadIdx.x, wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    bool want = false; uint32_t sl = 0;
    if (i < n && !v.ctl->err) { sl = new_slot(v, i, free_top0, n_slots0); want = v.unfinished[sl] == 0; }
    const uint64_t m = __ballot(want);
    if (lane == 0) wcnt[wid] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {  // one atomic per workgroup
        const uint32_t tot = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        uint32_t b = tot ? atomicAdd(&v.ctl->n_out, tot) : 0;
        for (int w = 0; w < 4; w++) { wbase[w] = b; b += wcnt[w]; }
    }
    __syncthreads();
    if (want) { const uint32_t pos = wbase[wid] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)); okey[pos] = v.id[sl]; oval[pos] = sl; }
}

// ---------------------------------------------------------------------------------------------------------------- finish
// decrement the consumers of one run segment; `idx`/`stride` = this lane's position / the number of lanes sharing the segment
__device__ __forceinline__ void release_range(const View &v, uint32_t off, uint32_t len, uint32_t idx, uint32_t stride, uint64_t *okey, uint32_t *oval) {
    for (uint32_t e0 = 0; e0 < len; e0 += stride) {  // uniform trip count per wavefront
        const uint32_t e = e0 + idx;
        bool rel = false; uint32_t c = 0;
        if (e < len) {
            const uint2 ed = v.edge[off + e];
            c = ed.x;
            if (v.gen[c] == ed.y) rel = atomicSub(&v.unfinished[c], 1u) == 1u;   // Task::decrease_unfinished_deps  task.rs:207-216
        }
        const uint32_t pos = wave_append(&v.ctl->n_out, rel);
        if (rel) { okey[pos] = v.id[c]; oval[pos] = c; }
    }
}

// task_finished, reactor.rs:510-590, in three steps.
// (1) one thread per finished id: find it, claim it (exactly once even if listed twice), drop it from the table, recycle its slot
__global__ void __launch_bounds__(256) k_g_finish_claim(View v, const uint64_t *__restrict__ ids, uint32_t n, uint32_t *__restrict__ slot_of) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    uint32_t slot = NONE; bool unknown = false;
    if (i < n) {
        uint32_t pos = ht_find_pos(v, ids[i]);
        if (pos != NONE) {
            const uint32_t sl = v.ht_val[pos];
            const uint32_t old = atomicExch(&v.unfinished[sl], ST_FREE);
            if (old == ST_FREE) pos = NONE;  // the same id twice in this batch: the second one is unknown (reactor.rs:565-567)
            else {
                if (old != 0) atomicOr(&v.ctl->err, (uint32_t)ERR_NOT_READY);
                v.ht_key[pos] = HT_TOMB; v.gen[sl] += 1;
                slot = sl;
            }
        }
        unknown = pos == NONE;
    }
    const uint32_t fp = wave_append(&v.ctl->free_top, slot != NONE);
    if (slot != NONE) v.free_slot[fp] = slot;
    const uint64_t um = __ballot(unknown);
    if (um && __lane_id() == (uint32_t)__ffsll((long long)um) - 1u) atomicAdd(&v.ctl->n_unknown, (uint32_t)__popcll(um));
    if (i < n) slot_of[i] = slot;
}

constexpr uint32_t SHORT_RUN = 4;      // runs up to this length are released by the thread that owns the finished task
constexpr uint32_t RUN_CHUNK = 4096;   // longer runs are cut into chunks of this many edges, one wavefront each

// (2) one thread per finished task: decrement the consumers of its short runs, defer the long ones
__global__ void __launch_bounds__(256) k_g_finish_release(View v, const uint32_t *__restrict__ slot_of, uint32_t n, uint64_t *__restrict__ okey, uint32_t *__restrict__ oval,
                                                          uint2 *__restrict__ big, uint32_t big_cap) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    uint32_t dead = 0;
    const uint32_t slot = i < n ? slot_of[i] : NONE;
    if (slot != NONE) {
        for (uint32_t r = v.head[slot]; r != NONE; r = v.run_next[r]) {
            const uint32_t off = v.run_off[r], len = v.run_len[r];
            dead += len;
            if (len > SHORT_RUN) {
                const uint32_t nch = (len + RUN_CHUNK - 1) / RUN_CHUNK, b = atomicAdd(&v.ctl->n_big, nch);
                if (b + nch <= big_cap) for (uint32_t c = 0; c < nch; c++) big[b + c] = make_uint2(off + c * RUN_CHUNK, min(RUN_CHUNK, len - c * RUN_CHUNK));
                else atomicOr(&v.ctl->err, (uint32_t)ERR_CAPACITY);
                continue;
            }
            for (uint32_t e = 0; e < len; e++) {  // divergent trip counts: the ballot inside wave_append covers the lanes that are here
                const uint2 ed = v.edge[off + e];
                bool rel = false;
                if (v.gen[ed.x] == ed.y) rel = atomicSub(&v.unfinished[ed.x], 1u) == 1u;   // Task::decrease_unfinished_deps  task.rs:207-216
                const uint32_t pos = wave_append(&v.ctl->n_out, rel);
                if (rel) { okey[pos] = v.id[ed.x]; oval[pos] = ed.x; }
            }
        }
        v.head[slot] = NONE;
    }
    dead = wave_sum(dead);
    if (dead && __lane_id() == 0) atomicAdd(&v.ctl->edges_dead, dead);
}

// (3) hub tasks: one wavefront per chunk of a long run
__global__ void __launch_bounds__(256) k_g_big_runs(View v, const uint2 *__restrict__ big, uint64_t *__restrict__ okey, uint32_t *__restrict__ oval) {
    const uint32_t nb = v.ctl->n_big, lane = threadIdx.x & 63, n_waves = gridDim.x * 4;
    for (uint32_t b = (blockIdx.x * 256 + threadIdx.x) >> 6; b < nb; b += n_waves) release_range(v, big[b].x, big[b].y, lane, 64, okey, oval);
}

// ---------------------------------------------------------------------------------------------------------------- remove
__global__ void __launch_bounds__(256) k_g_remove_seed(View v, const uint64_t *__restrict__ ids, uint32_t n, uint64_t *__restrict__ okey, uint32_t *__restrict__ oval) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    bool got = false; uint32_t sl = 0;
    if (i < n) {
        sl = ht_find(v, ids[i]);
        if (sl == NONE) atomicAdd(&v.ctl->n_unknown, 1u);
        else got = atomicExch(&v.unfinished[sl], ST_FREE) != ST_FREE;  // claimed once even if the id is listed twice
    }
    uint32_t pos = wave_append(&v.ctl->n_out, got);
    if (got) { okey[pos] = v.id[sl]; oval[pos] = sl; }
}

__global__ void k_g_close_level(View v) { v.ctl->lev_begin = v.ctl->lev_end; v.ctl->lev_end = v.ctl->n_out; }

// one BFS level of collect_recursive_consumers (task.rs:235-250): the output list is the queue; wavefront per frontier task
__global__ void __launch_bounds__(256) k_g_remove_expand(View v, uint64_t *__restrict__ okey, uint32_t *__restrict__ oval) {
    const uint32_t begin = v.ctl->lev_begin, end = v.ctl->lev_end, lane = threadIdx.x & 63;
    const uint32_t n_waves = gridDim.x * 4;
    for (uint32_t q = begin + ((blockIdx.x * 256 + threadIdx.x) >> 6); q < end; q += n_waves) {
        const uint32_t slot = oval[q];
        uint32_t r = v.head[slot];
        while (r != NONE) {
            const uint32_t off = v.run_off[r], len = v.run_len[r];
            for (uint32_t e0 = 0; e0 < len; e0 += 64) {
                const uint32_t e = e0 + lane;
                bool got = false; uint32_t c = 0;
                if (e < len) {
                    const uint2 ed = v.edge[off + e];
                    c = ed.x;
                    if (v.gen[c] == ed.y) got = atomicExch(&v.unfinished[c], ST_FREE) != ST_FREE;
                }
                const uint32_t pos = wave_append(&v.ctl->n_out, got);
                if (got) { okey[pos] = v.id[c]; oval[pos] = c; }
            }
            r = v.run_next[r];
        }
    }
}

// the removed tasks leave the table and their slots are recycled; the gen bump invalidates the edges that still point at them
__global__ void __launch_bounds__(256) k_g_remove_finalize(View v, const uint64_t *__restrict__ okey, const uint32_t *__restrict__ oval, uint32_t n) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t sl = oval[i];
    uint32_t p = ht_find_pos(v, okey[i]);
    if (p != NONE) v.ht_key[p] = HT_TOMB;
    v.gen[sl] += 1;
    uint32_t dead = 0;
    for (uint32_t r = v.head[sl]; r != NONE; r = v.run_next[r]) dead += v.run_len[r];
    v.head[sl] = NONE;
    if (dead) atomicAdd(&v.ctl->edges_dead, dead);
    v.free_slot[atomicAdd(&v.ctl->free_top, 1u)] = sl;
}

// ----------------------------------------------------------------------------------------------------------------- misc
__global__ void __launch_bounds__(256) k_g_unfinished(View v, const uint64_t *__restrict__ ids, uint32_t n, uint32_t *__restrict__ out) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t sl = ht_find(v, ids[i]);
    out[i] = sl == NONE ? 0xFFFFFFFFu : v.unfinished[sl];
}

// ------------------------------------------------------------------------------------------------------- b-level (extension)
// BASELINE.json's config 5 names a "dynamic b-level recompute"; the reference has none (Priority's low 32 bits — "scheduler priority",
// common/priority.rs:43-66 — are never written), so this is an EXTENSION with no reference counterpart: parity unpinned, off unless the host calls
// hqtick_graph_blevel, and no parity run does.  b-level(t) = 0 for a task without a live consumer, else 1 + the largest b-level among its consumers: the
// length of the longest path from t to a sink of what is still in the graph.
// One sweep: every live slot recomputes its value from its consumers' current values (a pull over its own edge list — the edges run producer -> consumer, so
// nobody writes anybody else's word and no atomics are needed).  Values only grow, a stale read only delays: the fixed point is the longest path, reached after
// at most depth + 1 sweeps; `changed` says when.
__global__ void __launch_bounds__(256) k_g_blevel_sweep(View v, uint32_t n_slots, uint32_t *__restrict__ bl, uint32_t *__restrict__ changed) {
    const uint32_t sl = blockIdx.x * 256 + threadIdx.x;
    bool ch = false;
    if (sl < n_slots && v.unfinished[sl] != ST_FREE) {
        uint32_t best = 0;
        for (uint32_t r = v.head[sl]; r != NONE; r = v.run_next[r]) {
            const uint32_t off = v.run_off[r], len = v.run_len[r];
            for (uint32_t e = 0; e < len; e++) {
                const uint2 ed = v.edge[off + e];
                if (v.gen[ed.x] != ed.y || v.unfinished[ed.x] == ST_FREE) continue;  // the consumer has left the graph
                const uint32_t c = bl[ed.x];
                const uint32_t cand = c == 0xFFFFFFFFu ? c : c + 1;
                best = cand > best ? cand : best;
            }
        }
        if (best != bl[sl]) { bl[sl] = best; ch = true; }
    }
    if (__ballot(ch) && (threadIdx.x & 63) == 0) atomicOr(changed, 1u);
}
// the result into the low 32 bits of the tasks' priorities (where Priority::add_priority_u32 would put a scheduler priority); the largest value comes back
__global__ void __launch_bounds__(256) k_g_blevel_store(View v, uint32_t n_slots, const uint32_t *__restrict__ bl, uint32_t *__restrict__ max_out) {
    const uint32_t sl = blockIdx.x * 256 + threadIdx.x;
    uint32_t mine = 0;
    if (sl < n_slots && v.unfinished[sl] != ST_FREE) { mine = bl[sl]; v.prio[sl] = (v.prio[sl] & 0xFFFFFFFF00000000ull) | (uint64_t)mine; }
    for (int off = 32; off > 0; off >>= 1) { const uint32_t o = __shfl_down(mine, off, 64); mine = o > mine ? o : mine; }
    if ((threadIdx.x & 63) == 0 && mine) atomicMax(max_out, mine);
}
// ... and of the tasks that already sit in the resident ready set (ids -> slots through the hash table); tombstones and ids the graph does not hold are left alone
__global__ void __launch_bounds__(256) k_g_blevel_ready(View v, const uint64_t *__restrict__ ids, const uint32_t *__restrict__ rq, uint64_t *__restrict__ prio, uint64_t n, uint32_t *__restrict__ n_done) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    bool hit = false;
    if (i < n && rq[i] != 0xFFFFFFFFu) {
        const uint32_t sl = ht_find(v, ids[i]);
        if (sl != NONE) { prio[i] = v.prio[sl]; hit = true; }
    }
    const uint32_t c = (uint32_t)__popcll(__ballot(hit));
    if (c && (threadIdx.x & 63) == 0) atomicAdd(n_done, c);
}
__global__ void __launch_bounds__(256) k_g_priorities(View v, const uint64_t *__restrict__ ids, uint32_t n, uint64_t *__restrict__ out) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t sl = ht_find(v, ids[i]);
    out[i] = sl == NONE ? 0ull : v.prio[sl];
}

__global__ void k_g_publish(View v, Ctl *host_copy) { if (threadIdx.x < sizeof(Ctl) / 4) reinterpret_cast<uint32_t *>(host_copy)[threadIdx.x] = reinterpret_cast<uint32_t *>(v.ctl)[threadIdx.x]; }

__global__ void __launch_bounds__(256) k_g_gather(View v, const uint64_t *__restrict__ okey, const uint32_t *__restrict__ oval, uint32_t n, uint64_t *__restrict__ out_id,
                                                  uint64_t *__restrict__ out_prio, uint32_t *__restrict__ out_rq, uint64_t *__restrict__ host_id, int with_payload) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint64_t id = okey[i];
    out_id[i] = id; host_id[i] = id;
    if (with_payload) { const uint32_t sl = oval[i]; out_prio[i] = v.prio[sl]; out_rq[i] = v.rq[sl]; }
}

__global__ void __launch_bounds__(256) k_g_rehash(View v, uint32_t n_slots) {
    uint32_t sl = blockIdx.x * 256 + threadIdx.x;
    if (sl >= n_slots || v.unfinished[sl] == ST_FREE) return;
    if (ht_insert(v, v.id[sl], sl) != 0) atomicOr(&v.ctl->err, (uint32_t)ERR_CAPACITY);
}

// edge pool compaction: every live producer's runs become one run in the new pool; wavefront per slot
__global__ void __launch_bounds__(256) k_g_compact(View v, uint32_t n_slots, uint32_t *__restrict__ nrn, uint32_t *__restrict__ nro, uint32_t *__restrict__ nrl,
                                                   uint2 *__restrict__ nedge, uint32_t *__restrict__ tops /* [0] runs, [1] edges */) {
    const uint32_t sl = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (sl >= n_slots || v.unfinished[sl] == ST_FREE) return;
    uint32_t r = v.head[sl];
    if (r == NONE) return;
    uint32_t total = 0;
    for (uint32_t q = r; q != NONE; q = v.run_next[q]) total += v.run_len[q];
    uint32_t nr = 0, base = 0;
    if (lane == 0) { nr = atomicAdd(&tops[0], 1u); base = atomicAdd(&tops[1], total); }
    nr = __shfl(nr, 0); base = __shfl(base, 0);
    uint32_t at = base;
    for (uint32_t q = r; q != NONE; q = v.run_next[q]) {
        const uint32_t off = v.run_off[q], len = v.run_len[q];
        for (uint32_t e = lane; e < len; e += 64) nedge[at + e] = v.edge[off + e];
        at += len;
    }
    if (lane == 0) { nrn[nr] = NONE; nro[nr] = base; nrl[nr] = total; v.head[sl] = nr; }
}

// -------------------------------------------------------------------------------------------------------------------- sort
__device__ __forceinline__ void cmp_swap(uint64_t &ka, uint32_t &va, uint64_t &kb, uint32_t &vb, bool asc) {
    if ((ka > kb) == asc) { uint64_t tk = ka; ka = kb; kb = tk; uint32_t tv = va; va = vb; vb = tv; }
}

// LDS stages: with full = 1 the whole network up to the chunk size, else only the tail j = chunk/2 .. 1 of stage k
__global__ void __launch_bounds__(256) k_bsort_lds(uint64_t *__restrict__ key, uint32_t *__restrict__ val, uint32_t chunk, uint64_t k_stage, int full) {
    __shared__ uint64_t sk[SORT_CH];
    __shared__ uint32_t sv[SORT_CH];
    const uint64_t g0 = (uint64_t)blockIdx.x * chunk;
    for (uint32_t t = threadIdx.x; t < chunk; t += 256) { sk[t] = key[g0 + t]; sv[t] = val[g0 + t]; }
    __syncthreads();
    for (uint64_t k = full ? 2 : k_stage; k <= (full ? (uint64_t)chunk : k_stage); k <<= 1) {
        for (uint32_t j = (k >> 1) < (uint64_t)(chunk >> 1) ? (uint32_t)(k >> 1) : (chunk >> 1); j > 0; j >>= 1) {
            for (uint32_t t = threadIdx.x; t < (chunk >> 1); t += 256) {
                const uint32_t i = ((t / j) * (j << 1)) + (t % j);
                const bool asc = ((g0 + i) & k) == 0;
                cmp_swap(sk[i], sv[i], sk[i + j], sv[i + j], asc);
            }
            __syncthreads();
        }
    }
    for (uint32_t t = threadIdx.x; t < chunk; t += 256) { key[g0 + t] = sk[t]; val[g0 + t] = sv[t]; }
}

__global__ void __launch_bounds__(256) k_bsort_global(uint64_t *__restrict__ key, uint32_t *__restrict__ val, uint64_t half, uint64_t j, uint64_t k) {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= half) return;
    const uint64_t i = ((t / j) * (j << 1)) + (t % j);
    uint64_t ka = key[i], kb = key[i + j];
    const bool asc = (i & k) == 0;
    if ((ka > kb) == asc) { key[i] = kb; key[i + j] = ka; uint32_t va = val[i]; val[i] = val[i + j]; val[i + j] = va; }
}

__global__ void __launch_bounds__(256) k_fill_pad(uint64_t *__restrict__ key, uint64_t n, uint64_t n_pow2) {
    const uint64_t i = n + (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n_pow2) key[i] = ~0ull;
}

uint64_t pow2_at_least(uint64_t n) { uint64_t p = 1; while (p < n) p <<= 1; return p; }

}  // namespace

uint64_t sort_capacity(uint64_t n) { return pow2_at_least(std::max<uint64_t>(n, 2)); }

hipError_t sort_pairs(uint64_t *key, uint32_t *val, uint64_t n, hipStream_t s) {
    if (n <= 1) return hipSuccess;
    const uint64_t np = pow2_at_least(n);
    if (np > n) hipLaunchKernelGGL(k_fill_pad, dim3((unsigned)((np - n + 255) / 256)), dim3(256), 0, s, key, n, np);
    const uint32_t chunk = (uint32_t)std::min<uint64_t>(np, SORT_CH);
    const unsigned blocks = (unsigned)(np / chunk);
    hipLaunchKernelGGL(k_bsort_lds, dim3(blocks), dim3(256), 0, s, key, val, chunk, (uint64_t)0, 1);
    for (uint64_t k = (uint64_t)chunk << 1; k <= np; k <<= 1) {
        for (uint64_t j = k >> 1; j >= chunk; j >>= 1) hipLaunchKernelGGL(k_bsort_global, dim3((unsigned)((np / 2 + 255) / 256)), dim3(256), 0, s, key, val, np / 2, j, k);
        hipLaunchKernelGGL(k_bsort_lds, dim3(blocks), dim3(256), 0, s, key, val, chunk, k, 0);
    }
    return hipGetLastError();
}

// ============================================================================================================ host driver
#define G_HIP(call)                                                                                                    \
    do {                                                                                                               \
        hipError_t e_ = (call);                                                                                        \
        if (e_ != hipSuccess) return fail(HQTICK_E_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_));          \
    } while (0)

namespace {
// grows a device buffer, keeping its contents (DevBuf::ensure alone would drop them)
bool grow_keep(hqbuf::DevBuf &b, size_t old_bytes, size_t new_bytes, hipStream_t s) {
    if (new_bytes <= b.cap) return true;
    hqbuf::DevBuf nb;
    if (!nb.ensure(new_bytes)) return false;
    if (b.p && old_bytes) {
        if (hipMemcpyAsync(nb.p, b.p, old_bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) { nb.release(); return false; }
        if (hipStreamSynchronize(s) != hipSuccess) { nb.release(); return false; }
    }
    b.release();
    b = nb;
    return true;
}
unsigned nblk(uint64_t n) { return (unsigned)((n + 255) / 256); }
}  // namespace

View Graph::view() const {
    View v;
    v.id = d_id.as<uint64_t>(); v.prio = d_prio.as<uint64_t>(); v.order = d_order.as<uint64_t>();
    v.rq = d_rq.as<uint32_t>(); v.unfinished = d_unf.as<uint32_t>(); v.gen = d_gen.as<uint32_t>(); v.head = d_head.as<uint32_t>();
    v.free_slot = d_free.as<uint32_t>();
    v.ht_key = d_htk.as<uint64_t>(); v.ht_val = d_htv.as<uint32_t>(); v.ht_mask = (uint32_t)(ht_cap - 1);
    v.run_next = d_rn.as<uint32_t>(); v.run_off = d_ro.as<uint32_t>(); v.run_len = d_rl.as<uint32_t>(); v.edge = d_edge.as<uint2>();
    v.tmp_cnt = d_tmpc.as<uint32_t>(); v.tmp_base = d_tmpb.as<uint32_t>();
    v.ctl = d_ctl.as<Ctl>();
    return v;
}

bool Graph::init(hipStream_t s) {
    if (ready_) return true;
    if (!d_ctl.ensure(sizeof(Ctl)) || !h_ctl.ensure(sizeof(Ctl)) || !d_big.ensure(65536 * sizeof(uint2))) return false;
    if (hipMemsetAsync(d_ctl.p, 0, sizeof(Ctl), s) != hipSuccess) return false;
    if (hipEventCreate(&ev0) != hipSuccess || hipEventCreate(&ev1) != hipSuccess) return false;
    ready_ = true;
    return true;
}

void Graph::clear() {
    n_slots = n_live_ = 0; free_top = run_top = edge_top = edges_dead = 0; ht_used = 0; n_out_ = n_unknown_ = 0;
    if (d_ctl.p) hipMemset(d_ctl.p, 0, sizeof(Ctl));
    if (d_htk.p) hipMemset(d_htk.p, 0xFF, ht_cap * 8);
    if (d_tmpc.p) hipMemset(d_tmpc.p, 0, cap_slots * 4);
}

void Graph::release() {
    for (hqbuf::DevBuf *b : {&d_id, &d_prio, &d_order, &d_rq, &d_unf, &d_gen, &d_head, &d_free, &d_tmpc, &d_tmpb, &d_htk, &d_htv, &d_rn, &d_ro, &d_rl, &d_edge, &d_rn2, &d_ro2,
                             &d_rl2, &d_edge2, &d_ctl, &d_stage, &d_eds, &d_erk, &d_okey, &d_oval, &d_out_id, &d_out_prio, &d_out_rq, &d_big, &d_bl})
        b->release();
    h_ctl.release(); h_stage.release(); h_out.release();
    if (ev0) hipEventDestroy(ev0);
    if (ev1) hipEventDestroy(ev1);
    ev0 = ev1 = nullptr; ready_ = false; cap_slots = cap_edges = ht_cap = 0;
    n_slots = n_live_ = 0; free_top = run_top = edge_top = edges_dead = 0; ht_used = 0;
}

Stats Graph::stats() const {
    Stats st{};
    st.n_tasks = n_live_; st.n_slots = n_slots; st.n_edges_live = edge_top - edges_dead; st.n_edges_pool = edge_top; st.n_runs = run_top;
    st.hash_capacity = ht_cap; st.hash_tombstones = ht_used - n_live_;
    st.bytes_hbm = cap_slots * (8 + 8 + 8 + 4 * 7) + ht_cap * 12 + cap_edges * (8 + 12);
    return st;
}

bool Graph::grow_slots(uint64_t want, hipStream_t s) {
    if (want <= cap_slots) return true;
    uint64_t nc = std::max<uint64_t>(want + want / 2, 4096);
    const uint64_t oc = cap_slots;
    if (!grow_keep(d_id, oc * 8, nc * 8, s) || !grow_keep(d_prio, oc * 8, nc * 8, s) || !grow_keep(d_order, oc * 8, nc * 8, s) || !grow_keep(d_rq, oc * 4, nc * 4, s) ||
        !grow_keep(d_unf, oc * 4, nc * 4, s) || !grow_keep(d_gen, oc * 4, nc * 4, s) || !grow_keep(d_head, oc * 4, nc * 4, s) || !grow_keep(d_free, oc * 4, nc * 4, s) ||
        !grow_keep(d_tmpc, oc * 4, nc * 4, s) || !grow_keep(d_tmpb, oc * 4, nc * 4, s))
        return false;
    // new slots: free, generation 0, scratch zero
    if (hipMemsetAsync(d_unf.as<uint32_t>() + oc, 0xFF, (nc - oc) * 4, s) != hipSuccess) return false;
    if (hipMemsetAsync(d_gen.as<uint32_t>() + oc, 0, (nc - oc) * 4, s) != hipSuccess) return false;
    if (hipMemsetAsync(d_tmpc.as<uint32_t>() + oc, 0, (nc - oc) * 4, s) != hipSuccess) return false;
    const uint64_t sc = sort_capacity(nc);
    if (!d_okey.ensure(sc * 8) || !d_oval.ensure(sc * 4) || !d_out_id.ensure(nc * 8) || !d_out_prio.ensure(nc * 8) || !d_out_rq.ensure(nc * 4) || !h_out.ensure(nc * 8)) return false;
    cap_slots = nc;
    return true;
}

bool Graph::rebuild_hash(uint64_t want_entries, hipStream_t s) {
    uint64_t nc = std::max<uint64_t>(pow2_at_least(want_entries * 2), 8192);
    if (nc > (1ull << 31)) return false;
    if (nc > ht_cap) { if (!d_htk.ensure(nc * 8) || !d_htv.ensure(nc * 4)) return false; }
    else nc = ht_cap;
    ht_cap = nc;
    if (hipMemsetAsync(d_htk.p, 0xFF, nc * 8, s) != hipSuccess) return false;
    if (n_slots) hipLaunchKernelGGL(k_g_rehash, dim3(nblk(n_slots)), dim3(256), 0, s, view(), (uint32_t)n_slots);
    ht_used = n_live_;
    return hipGetLastError() == hipSuccess;
}

int Graph::ensure_edges(uint64_t extra, hipStream_t s) {
    if ((uint64_t)edge_top + extra <= cap_edges) return 0;
    const uint64_t live = edge_top - edges_dead;
    uint64_t nc = std::max<uint64_t>({cap_edges, (live + extra) * 2, 16384});
    if (nc > 0xFFFFFFF0ull) return fail(HQTICK_E_CAPACITY, "dependency graph: more than 2^32 edges");
    if (!d_rn2.ensure(nc * 4) || !d_ro2.ensure(nc * 4) || !d_rl2.ensure(nc * 4) || !d_edge2.ensure(nc * 8)) return fail(HQTICK_E_DEVICE, "hipMalloc edge pool");
    if (edge_top) {
        // compact into the alternate pool; its two counters live in the (otherwise unused here) scratch head of d_big
        uint32_t *tops = d_big.as<uint32_t>();
        G_HIP(hipMemsetAsync(tops, 0, 8, s));
        hipLaunchKernelGGL(k_g_compact, dim3(nblk(n_slots * 64)), dim3(256), 0, s, view(), (uint32_t)n_slots, d_rn2.as<uint32_t>(), d_ro2.as<uint32_t>(), d_rl2.as<uint32_t>(),
                           d_edge2.as<uint2>(), tops);
        uint32_t *h = h_ctl.as<uint32_t>();
        G_HIP(hipMemcpyAsync(h, tops, 8, hipMemcpyDeviceToHost, s));
        G_HIP(hipStreamSynchronize(s));
        run_top = h[0]; edge_top = h[1]; edges_dead = 0;
        Ctl *c = d_ctl.as<Ctl>();
        G_HIP(hipMemcpyAsync(&c->run_top, &h[0], 4, hipMemcpyHostToDevice, s));
        G_HIP(hipMemcpyAsync(&c->edge_top, &h[1], 4, hipMemcpyHostToDevice, s));
        G_HIP(hipMemsetAsync(&c->edges_dead, 0, 4, s));
        G_HIP(hipStreamSynchronize(s));
    }
    std::swap(d_rn, d_rn2); std::swap(d_ro, d_ro2); std::swap(d_rl, d_rl2); std::swap(d_edge, d_edge2);
    cap_edges = nc;
    return 0;
}

// ids of an operation: pinned staging -> device
int Graph::stage(uint64_t n_ids, const uint64_t *ids, hipStream_t s) {
    if (!h_stage.ensure(n_ids * 8 + 64) || !d_stage.ensure(n_ids * 8 + 64)) return fail(HQTICK_E_DEVICE, "hipMalloc graph staging");
    memcpy(h_stage.p, ids, n_ids * 8);
    G_HIP(hipMemcpyAsync(d_stage.p, h_stage.p, n_ids * 8, hipMemcpyHostToDevice, s));
    return 0;
}

// common tail: publish the counters, sort the output list by id, gather the payload, mirror the counters on the host
int Graph::finish_output(hipStream_t s, bool with_payload) {
    View v = view();
    hipLaunchKernelGGL(k_g_publish, dim3(1), dim3(64), 0, s, v, h_ctl.dev<Ctl>());
    G_HIP(hipStreamSynchronize(s));
    const Ctl c = *h_ctl.as<Ctl>();
    free_top = c.free_top; run_top = c.run_top; edge_top = c.edge_top; edges_dead = c.edges_dead; n_out_ = c.n_out; n_unknown_ = c.n_unknown;
    if (n_out_) {
        G_HIP(sort_pairs(d_okey.as<uint64_t>(), d_oval.as<uint32_t>(), n_out_, s));
        hipLaunchKernelGGL(k_g_gather, dim3(nblk(n_out_)), dim3(256), 0, s, v, d_okey.as<uint64_t>(), d_oval.as<uint32_t>(), n_out_, d_out_id.as<uint64_t>(), d_out_prio.as<uint64_t>(),
                           d_out_rq.as<uint32_t>(), h_out.dev<uint64_t>(), with_payload ? 1 : 0);
        G_HIP(hipGetLastError());
        G_HIP(hipStreamSynchronize(s));
    }
    if (ev0) { float ms = 0.f; if (hipEventElapsedTime(&ms, ev0, ev1) == hipSuccess) last_us_ = ms * 1000.0; }
    return 0;
}

int Graph::add(uint64_t n, const uint64_t *id, const uint64_t *prio, const uint32_t *rq, const uint32_t *dep_off, const uint64_t *dep_id, hipStream_t s) {
    n_out_ = 0; n_unknown_ = 0;
    if (n == 0) return 0;
    if (n > 0x7FFFFFFFull) return fail(HQTICK_E_CAPACITY, "more than 2^31 tasks in one batch");
    if (!init(s)) return fail(HQTICK_E_DEVICE, "dependency graph: device setup failed");
    const uint64_t E = dep_off ? dep_off[n] : 0;
    if (E > 0xFFFFFFF0ull) return fail(HQTICK_E_CAPACITY, "more than 2^32 dependency entries in one batch");
    if (dep_off) { if (dep_off[0] != 0) return fail(HQTICK_E_INVALID, "dep_off[0] != 0"); for (uint64_t i = 0; i < n; i++) if (dep_off[i + 1] < dep_off[i]) return fail(HQTICK_E_INVALID, "dep_off not monotone"); }
    for (uint64_t i = 0; i < n; i++) if (id[i] >= HT_TOMB) return fail(HQTICK_E_INVALID, "task id out of range");
    for (uint64_t i = 0; i < n; i++) if (rq[i] == 0xFFFFFFFFu) return fail(HQTICK_E_INVALID, "request id 0xFFFFFFFF is reserved");
    // capacity
    const uint64_t fresh = n > free_top ? n - free_top : 0;
    if (n_slots + fresh > 0xFFFFFFF0ull) return fail(HQTICK_E_CAPACITY, "more than 2^32 tasks in the graph");
    if (!grow_slots(n_slots + fresh, s)) return fail(HQTICK_E_DEVICE, "hipMalloc task table");
    if ((ht_used + n) * 2 > ht_cap) { if (!rebuild_hash(n_live_ + n, s)) return fail(HQTICK_E_DEVICE, "hipMalloc task hash table"); }
    if (int rc = ensure_edges(E, s)) return rc;
    // staging: ids | prio | dep ids | rq | dep_off   (8-byte columns first)
    const size_t o_prio = n * 8, o_dep = o_prio + n * 8, o_rq = o_dep + E * 8, o_off = o_rq + n * 4, bytes = o_off + (n + 1) * 4;
    if (!h_stage.ensure(bytes + 64) || !d_stage.ensure(bytes + 64) || !d_eds.ensure(E * 4 + 64) || !d_erk.ensure(E * 4 + 64)) return fail(HQTICK_E_DEVICE, "hipMalloc graph staging");
    unsigned char *h = h_stage.as<unsigned char>();
    memcpy(h, id, n * 8); memcpy(h + o_prio, prio, n * 8); if (E) memcpy(h + o_dep, dep_id, E * 8); memcpy(h + o_rq, rq, n * 4);
    if (dep_off) memcpy(h + o_off, dep_off, (n + 1) * 4); else memset(h + o_off, 0, (n + 1) * 4);
    G_HIP(hipMemcpyAsync(d_stage.p, h, bytes, hipMemcpyHostToDevice, s));
    const unsigned char *d = d_stage.as<unsigned char>();
    const uint64_t *d_ids = reinterpret_cast<const uint64_t *>(d), *d_pr = reinterpret_cast<const uint64_t *>(d + o_prio), *d_dep = reinterpret_cast<const uint64_t *>(d + o_dep);
    const uint32_t *d_rqs = reinterpret_cast<const uint32_t *>(d + o_rq), *d_off = reinterpret_cast<const uint32_t *>(d + o_off);
    View v = view();
    Ctl *c = d_ctl.as<Ctl>();
    G_HIP(hipMemsetAsync(&c->n_out, 0, 12, s));  // n_out, n_unknown, err
    batch_no++;
    const uint32_t ft0 = free_top, ns0 = (uint32_t)n_slots, nn = (uint32_t)n, EE = (uint32_t)E;
    hipLaunchKernelGGL(k_g_precheck, dim3(nblk(n)), dim3(256), 0, s, v, d_ids, nn);
    hipLaunchKernelGGL(k_g_insert, dim3(nblk(n)), dim3(256), 0, s, v, d_ids, d_pr, d_rqs, nn, ft0, ns0, batch_no);
    G_HIP(hipEventRecord(ev0, s));
    if (E) {
        hipLaunchKernelGGL(k_g_link_count, dim3(nblk(E)), dim3(256), 0, s, v, d_off, d_dep, nn, EE, ft0, ns0, batch_no, d_eds.as<uint32_t>(), d_erk.as<uint32_t>());
        hipLaunchKernelGGL(k_g_link_alloc, dim3(nblk(E)), dim3(256), 0, s, v, EE, d_eds.as<uint32_t>(), d_erk.as<uint32_t>(), (uint32_t)cap_edges);
        hipLaunchKernelGGL(k_g_link_fill, dim3(nblk(E)), dim3(256), 0, s, v, d_off, nn, EE, ft0, ns0, d_eds.as<uint32_t>(), d_erk.as<uint32_t>());
    }
    G_HIP(hipEventRecord(ev1, s));
    hipLaunchKernelGGL(k_g_collect_ready, dim3(nblk(n)), dim3(256), 0, s, v, nn, ft0, ns0, d_okey.as<uint64_t>(), d_oval.as<uint32_t>());
    G_HIP(hipGetLastError());
    // the counters the kernels did not maintain: free_top goes down by what the batch took
    hipLaunchKernelGGL(k_g_publish, dim3(1), dim3(64), 0, s, v, h_ctl.dev<Ctl>());
    G_HIP(hipStreamSynchronize(s));
    const uint32_t e = h_ctl.as<Ctl>()->err;
    if (e) {
        if (e & ERR_DUP_IN_BATCH) { hipLaunchKernelGGL(k_g_rollback, dim3(nblk(n)), dim3(256), 0, s, v, d_ids, nn, ft0, ns0); ht_used += n; }
        G_HIP(hipMemsetAsync(&c->n_out, 0, 12, s));
        G_HIP(hipStreamSynchronize(s));
        if (e & ERR_CAPACITY) return fail(HQTICK_E_CAPACITY, "dependency graph: internal pool overflow");
        return fail(HQTICK_E_INVALID, e & ERR_EXISTS ? "hqtick_graph_add_tasks: a task id is already in the graph" : "hqtick_graph_add_tasks: the same task id twice in one batch");
    }
    const uint32_t took = (uint32_t)std::min<uint64_t>(n, ft0);
    free_top = ft0 - took; n_slots += fresh; n_live_ += n; ht_used += n;
    G_HIP(hipMemcpyAsync(&c->free_top, &free_top, 4, hipMemcpyHostToDevice, s));
    if (int rc = finish_output(s, true)) return rc;
    return (int)n_out_;
}

int Graph::finish(uint64_t n, const uint64_t *id, hipStream_t s) {
    n_out_ = 0; n_unknown_ = 0;
    if (n == 0) return 0;
    if (n > 0x03FFFFFFull) return fail(HQTICK_E_CAPACITY, "more than 2^26 finished tasks in one batch");
    if (!init(s)) return fail(HQTICK_E_DEVICE, "dependency graph: device setup failed");
    if (n_slots == 0) { n_unknown_ = (uint32_t)n; return 0; }
    if (int rc = stage(n, id, s)) return rc;
    View v = view();
    Ctl *c = d_ctl.as<Ctl>();
    G_HIP(hipMemsetAsync(&c->n_out, 0, 16, s));  // n_out, n_unknown, err, n_big
    // deferred-run list: one entry per started RUN_CHUNK of a run longer than SHORT_RUN
    const uint64_t big_cap = (uint64_t)edge_top / RUN_CHUNK + (uint64_t)edge_top / SHORT_RUN + 64;
    if (!d_big.ensure(big_cap * sizeof(uint2)) || !d_erk.ensure(n * 4 + 64)) return fail(HQTICK_E_DEVICE, "hipMalloc graph staging");
    G_HIP(hipEventRecord(ev0, s));
    hipLaunchKernelGGL(k_g_finish_claim, dim3(nblk(n)), dim3(256), 0, s, v, d_stage.as<uint64_t>(), (uint32_t)n, d_erk.as<uint32_t>());
    hipLaunchKernelGGL(k_g_finish_release, dim3(nblk(n)), dim3(256), 0, s, v, d_erk.as<uint32_t>(), (uint32_t)n, d_okey.as<uint64_t>(), d_oval.as<uint32_t>(), d_big.as<uint2>(), (uint32_t)big_cap);
    hipLaunchKernelGGL(k_g_big_runs, dim3(BIG_GRID), dim3(256), 0, s, v, d_big.as<uint2>(), d_okey.as<uint64_t>(), d_oval.as<uint32_t>());
    G_HIP(hipEventRecord(ev1, s));
    G_HIP(hipGetLastError());
    if (int rc = finish_output(s, true)) return rc;
    const uint32_t e = h_ctl.as<Ctl>()->err;
    const uint64_t gone = n - n_unknown_;
    n_live_ -= gone;
    if (e & ERR_CAPACITY) return fail(HQTICK_E_CAPACITY, "dependency graph: deferred-run list overflow");
    if (e & ERR_NOT_READY) return fail(HQTICK_E_INVALID, "hqtick_graph_finish: a finished task still had unfinished dependencies");
    return (int)n_out_;
}

int Graph::remove(uint64_t n, const uint64_t *id, bool recursive, hipStream_t s) {
    n_out_ = 0; n_unknown_ = 0;
    if (n == 0) return 0;
    if (n > 0x7FFFFFFFull) return fail(HQTICK_E_CAPACITY, "more than 2^31 ids in one batch");
    if (!init(s)) return fail(HQTICK_E_DEVICE, "dependency graph: device setup failed");
    if (n_slots == 0) { n_unknown_ = (uint32_t)n; return 0; }
    if (int rc = stage(n, id, s)) return rc;
    View v = view();
    Ctl *c = d_ctl.as<Ctl>();
    G_HIP(hipMemsetAsync(&c->n_out, 0, 24, s));  // n_out, n_unknown, err, n_big, lev_begin, lev_end
    hipLaunchKernelGGL(k_g_remove_seed, dim3(nblk(n)), dim3(256), 0, s, v, d_stage.as<uint64_t>(), (uint32_t)n, d_okey.as<uint64_t>(), d_oval.as<uint32_t>());
    G_HIP(hipEventRecord(ev0, s));
    if (recursive) {
        for (;;) {
            for (uint32_t l = 0; l < BFS_LEVELS_PER_SYNC; l++) {
                hipLaunchKernelGGL(k_g_close_level, dim3(1), dim3(1), 0, s, v);
                hipLaunchKernelGGL(k_g_remove_expand, dim3(512), dim3(256), 0, s, v, d_okey.as<uint64_t>(), d_oval.as<uint32_t>());
            }
            hipLaunchKernelGGL(k_g_publish, dim3(1), dim3(64), 0, s, v, h_ctl.dev<Ctl>());
            G_HIP(hipStreamSynchronize(s));
            const Ctl hc = *h_ctl.as<Ctl>();
            if (hc.n_out == hc.lev_end) break;  // the last level added nothing
        }
    }
    G_HIP(hipEventRecord(ev1, s));
    hipLaunchKernelGGL(k_g_publish, dim3(1), dim3(64), 0, s, v, h_ctl.dev<Ctl>());
    G_HIP(hipStreamSynchronize(s));
    const uint32_t total = h_ctl.as<Ctl>()->n_out;
    if (total) hipLaunchKernelGGL(k_g_remove_finalize, dim3(nblk(total)), dim3(256), 0, s, v, d_okey.as<uint64_t>(), d_oval.as<uint32_t>(), total);
    G_HIP(hipGetLastError());
    if (int rc = finish_output(s, false)) return rc;
    n_live_ -= n_out_;
    return (int)n_out_;
}

int Graph::unfinished(uint64_t n, const uint64_t *id, uint32_t *out, hipStream_t s) {
    if (n == 0) return 0;
    if (!init(s)) return fail(HQTICK_E_DEVICE, "dependency graph: device setup failed");
    if (n_slots == 0) { for (uint64_t i = 0; i < n; i++) out[i] = 0xFFFFFFFFu; return 0; }
    if (int rc = stage(n, id, s)) return rc;
    if (!d_erk.ensure(n * 4 + 64)) return fail(HQTICK_E_DEVICE, "hipMalloc graph staging");
    hipLaunchKernelGGL(k_g_unfinished, dim3(nblk(n)), dim3(256), 0, s, view(), d_stage.as<uint64_t>(), (uint32_t)n, d_erk.as<uint32_t>());
    G_HIP(hipMemcpyAsync(out, d_erk.p, n * 4, hipMemcpyDeviceToHost, s));
    G_HIP(hipStreamSynchronize(s));
    return 0;
}

// b-level of every task in the graph into the low 32 bits of its priority (extension: see k_g_blevel_sweep).  Returns the number of sweeps (>= 1) or a negative
// HQTICK_E_* code; *max_level = the largest b-level.  ready_*: the resident ready columns to refresh from the graph's priorities (may be null).
int Graph::blevel(uint32_t *max_level, const uint64_t *ready_id, const uint32_t *ready_rq, uint64_t *ready_prio, uint64_t n_ready, uint32_t *n_ready_updated, hipStream_t s) {
    if (max_level) *max_level = 0;
    if (n_ready_updated) *n_ready_updated = 0;
    if (!init(s)) return fail(HQTICK_E_DEVICE, "dependency graph: device setup failed");
    if (n_slots == 0) return 0;
    if (!d_bl.ensure(n_slots * 4 + 64) || !h_stage.ensure(64)) return fail(HQTICK_E_DEVICE, "hipMalloc b-level");
    View v = view();
    // flag words behind the values, in HBM (a wavefront that changes something ORs into one: 15 k atomics per sweep — over PCIe into pinned memory they took 2.4 ms
    // per sweep of 1 M tasks, in HBM they are free): [0] a change in the group's first sweeps, [1] in its last one, [2] largest b-level, [3] ready tasks refreshed
    uint32_t *dflag = d_bl.as<uint32_t>() + n_slots, *flag = h_stage.as<uint32_t>();
    G_HIP(hipMemsetAsync(d_bl.p, 0, n_slots * 4 + 16, s));
    G_HIP(hipEventRecord(ev0, s));
    int sweeps = 0;
    for (;;) {
        for (uint32_t l = 0; l < BFS_LEVELS_PER_SYNC; l++) { hipLaunchKernelGGL(k_g_blevel_sweep, dim3(nblk(n_slots)), dim3(256), 0, s, v, (uint32_t)n_slots, d_bl.as<uint32_t>(), dflag); sweeps++; }
        hipLaunchKernelGGL(k_g_blevel_sweep, dim3(nblk(n_slots)), dim3(256), 0, s, v, (uint32_t)n_slots, d_bl.as<uint32_t>(), dflag + 1);  // (the group's last sweep on a word of its own: nothing changed in it = done)
        sweeps++;
        G_HIP(hipGetLastError());
        G_HIP(hipMemcpyAsync(flag, dflag, 16, hipMemcpyDeviceToHost, s));
        G_HIP(hipStreamSynchronize(s));
        if (!flag[1]) break;
        G_HIP(hipMemsetAsync(dflag, 0, 8, s));
        if (sweeps > 1 << 20) return fail(HQTICK_E_DEVICE, "b-level: the sweeps do not settle (a cycle in the dependency graph?)");
    }
    hipLaunchKernelGGL(k_g_blevel_store, dim3(nblk(n_slots)), dim3(256), 0, s, v, (uint32_t)n_slots, d_bl.as<uint32_t>(), dflag + 2);
    if (ready_id && ready_prio && n_ready) hipLaunchKernelGGL(k_g_blevel_ready, dim3((unsigned)((n_ready + 255) / 256)), dim3(256), 0, s, v, ready_id, ready_rq, ready_prio, n_ready, dflag + 3);
    G_HIP(hipEventRecord(ev1, s));
    G_HIP(hipGetLastError());
    G_HIP(hipMemcpyAsync(flag, dflag, 16, hipMemcpyDeviceToHost, s));
    G_HIP(hipStreamSynchronize(s));
    if (max_level) *max_level = flag[2];
    if (n_ready_updated) *n_ready_updated = flag[3];
    if (ev0) { float ms = 0.f; if (hipEventElapsedTime(&ms, ev0, ev1) == hipSuccess) last_us_ = ms * 1000.0; }
    return sweeps;
}

int Graph::priorities(uint64_t n, const uint64_t *id, uint64_t *out, hipStream_t s) {
    if (n == 0) return 0;
    if (!init(s)) return fail(HQTICK_E_DEVICE, "dependency graph: device setup failed");
    if (n_slots == 0) { for (uint64_t i = 0; i < n; i++) out[i] = 0; return 0; }
    if (int rc = stage(n, id, s)) return rc;
    if (!d_okey.ensure(n * 8 + 64)) return fail(HQTICK_E_DEVICE, "hipMalloc graph staging");
    hipLaunchKernelGGL(k_g_priorities, dim3(nblk(n)), dim3(256), 0, s, view(), d_stage.as<uint64_t>(), (uint32_t)n, d_okey.as<uint64_t>());
    G_HIP(hipMemcpyAsync(out, d_okey.p, n * 8, hipMemcpyDeviceToHost, s));
    G_HIP(hipStreamSynchronize(s));
    return 0;
}

}  // namespace hqgraph
