// k_price_sweep: one PRICE SWEEP of the coupled placement model on the MI355X — every worker's block solved exactly under the current prices of the
// model's wide rows (run_scheduling_solver's model, /root/reference/crates/tako/src/internal/scheduler/solver.rs:95-430, which the reference hands to
// HiGHS, solver/highs.rs:65-88; the method is in price.h, the per-block algorithm in price_core.h / block_core.h).
//
// Launch shape: grid = number of blocks (one per worker: 1024-4096), block = 256 threads = FOUR wave64 — one runs the block's chain, the others share its dual pool and run
// its greedy fills, then leave (block_core.h: pool_sections) —, 17.5 / 23.9 / 36.9 KB of LDS per block at 8 / 16 / 32 columns (dual vertices, level stack, greedy vectors,
// the block's wide-row entries) -> 6 blocks per CU (wavefront slots: 81 VGPRs), 1536 resident on the 256 CUs, dealt round-robin over the 8 XCDs by the dispatcher; the blocks
// share nothing but the prices (<= 1 KB, in the kernel arguments) — no XCD-aware mapping is needed.  Integer / f64 scalar work on LDS-resident data:
// not an HBM kernel (a block reads ~0.5-2 KB of tables) and not MFMA work; its figure of merit is block solves per second.
//
// A sweep is one link of a serial chain (master LP on the host -> prices -> sweep -> cut -> master ...), so its LATENCY is what counts:
//   * prices travel in the kernel arguments: no copy, no PCIe read by 1024 wavefronts;
//   * the wide rows' activities are integer, accumulated with 64-bit atomics in HBM: exact, order-free, the same on every replica;
//   * the end of a sweep is two levels of tickets in HBM (per part, then per sweep): the parts' last blocks and then the sweep's last block add up the per-block values in
//     a fixed order and write the sweep's totals + a sequence number straight into pinned host memory; the host waits on that word instead of a stream synchronisation
//     (the marker packet behind hipStreamSynchronize costs ~6 us per call);
//   * the patterns of the first 64 sweeps are written straight into pinned host memory (the primal side reads them in place), later ones stay in an HBM ring.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "dev_wave.h"
#include "price_core.h"
#include "price_dev.h"

namespace hqprice {

static_assert(PARTS == ASLOTS, "the master's parts are the kernel's activity slots");
// Activity vectors per part on the device (SweepOut::asub): block b adds into vector b % ASUB of its part, the part's last block adds them up.  With one, the 64 blocks
// of a part (1024-block model; 256 at 4096 blocks) queue at the same K addresses — one cache line on c3p — at the moment they all finish: the last ones waited 3-5 us
// for their acknowledgements (profiles/r06/price_sweep_waves.txt).  (Round 5 measured four vectors with no gain: the device fence behind them hid it then.)
constexpr int ASUB = 4;

namespace {

struct SweepResult {   // pinned host memory, written by the last workgroup of a sweep
    double cx, rc, bnd;
    uint32_t n_budget, max_steps;
    long long act[KMAX];
    double part_cx[ASLOTS];
    long long part_act[ASLOTS * KMAX];   // [part * K + k]: only the first ASLOTS * K entries are written
    uint32_t seq;      // written last (release, system scope)
};

struct PartVal { uint32_t nbud, mx; };   // one part's sums, handed from the part's last block to the sweep's last block (HBM)

struct SweepArgs {
    Tables t;
    SweepOut out;
    uint32_t budget, seq;
    uint32_t *tickets;     // [0] the sweep's, [1 + p] part p's
    long long *tact;       // [K] the sweep's activities: the parts' last blocks add their vectors up here, the sweep's last block reads and zeroes it
    PartVal *pval;         // [ASLOTS]
    SweepResult *res;
    // worker-range shards (price.h: ShardedSweeper): the grid covers the blocks [first, first + gridDim.x) only, and instead of the sweep's totals the last
    // workgroup leaves the range's per-block values in pinned memory (lv_*: arrays indexed by absolute block) — the totals are added up after the ranks' all-gather
    uint32_t first, local;
    double *lv_cx, *lv_rc, *lv_bnd; uint32_t *lv_steps;
    double pi[KMAX];
};

// Values that travel between workgroups of ONE launch (per-block results, the parts' sums, tickets) are written and read with device-scope relaxed atomics: the store
// goes through to where every XCD sees it, the load does not look into a cache that may hold last sweep's line.  Ordering comes from waiting for the stores'
// acknowledgements (a workgroup-scope release fence = s_waitcnt vmcnt(0)) before the ticket is taken — not from a device-scope release fence, whose L2 write-back
// cost every block 5-7 us (16 on the last ones) in front of a ticket 1024 blocks queued for (profiles/r06: price_sweep_waves.txt).
__device__ __forceinline__ double ld_dev(const double *p) { return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }
__device__ __forceinline__ long long ld_dev(const long long *p) { return (long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t ld_dev(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_dev(long long *p, long long v) { __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_dev(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// ... and what the host reads (pinned memory): system scope
__device__ __forceinline__ void st_host(double *p, double v) { __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void st_host(long long *p, long long v) { __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void st_host(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void acked() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); }   // every store / atomic of this wavefront so far is acknowledged

// SH = hqblock::SharedN<N>, N = 8 / 16 / 32: the smallest working set the model's widest block fits (block_core.h) — i.e. how many blocks a CU holds at once.  The
// answers do not depend on N.
// NW: wavefronts per block.  Wavefront 0 runs the block's chain; the others take part in the dual pool / greedy section (block_core.h: pool_sections) and leave.  A
// sweep is as long as its slowest block's chain, and on a model of <= 4 blocks per CU every helper sits on a SIMD that would otherwise idle (NW = 4: one wavefront per
// SIMD); VGPRs (one allocation for every wavefront of the kernel) bound the residency to 16-20 wavefronts per CU, so wider models take fewer helpers (launch()).
//
// The end of a sweep, in two levels (the master's 16 PARTS = contiguous ranges of blocks): every block takes a ticket of its PART; the part's last block adds the
// part's c.x, hands the part's activity vector over (and zeroes the accumulator for the next sweep), writes the part's rows of the result straight into the host's
// pinned memory, and takes the sweep's ticket; the last of those adds the per-block values of the whole sweep and writes totals + sequence word.  Two short queues
// (64-256 and 16 deep) instead of one of 1024-4096, and the parts' work runs on 16 CUs side by side.  The ORDER of every floating-point sum is the one of price.h:
// totals_from_blocks — as it was when one workgroup did all of this: a GPU tick, a sharded tick and the emulated tick walk the same sequence of prices.
template <class SH, int NW>
__global__ __launch_bounds__(WAVE * NW) void k_price_sweep(const SweepArgs a) {
    __shared__ SH S;
    using WV = typename std::conditional<NW == 1, hqblock::DevWave, hqblock::DevGroup<NW>>::type;
    WV wv;
    if (NW > 1 && threadIdx.x >= WAVE) { hqblock::pool_helper(wv, S); return; }
    const uint32_t lane = threadIdx.x;   // (from here on the main wavefront is alone)
    const uint32_t nb = a.t.n_blocks, K = a.t.K, per = part_size(nb), b = a.first + blockIdx.x, p = b / per;
    solve_priced_block(wv, S, a.t, a.pi, b, a.out, a.budget);
    uint64_t *prof = a.out.prof ? a.out.prof + (size_t)b * PSLOTS : nullptr;
    const uint32_t r0 = a.first, r1 = a.first + gridDim.x;   // the launch's blocks; the part's blocks among them:
    const uint32_t pb0 = p * per > r0 ? p * per : r0, pe = p * per + per < nb ? p * per + per : nb, pb1 = pe < r1 ? pe : r1;
    acked();  // the block's values and its activity atomics, before its ticket
    if (prof && lane == 0) prof[11] = wv.now();
    uint32_t last = 0;
    if (lane == 0) last = __hip_atomic_fetch_add(&a.tickets[1 + p], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == pb1 - pb0 - 1 ? 1u : 0u;
    last = (uint32_t)__builtin_amdgcn_readfirstlane((int)last);
    if (prof && lane == 0) prof[12] = wv.now();
    if (!last) return;
    {   // the part's last block.  Everything it reads is asked for before anything is used: one round trip (per 256 blocks of the part)
        long long vs[2][ASUB];   // the part's activity vectors (K <= KMAX = 128: two entries per lane)
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const uint32_t k = lane + (uint32_t)u * WAVE;
#pragma unroll
            for (int sub = 0; sub < ASUB; sub++) vs[u][sub] = k < K ? ld_dev(&a.out.act[((size_t)p * ASUB + sub) * K + k]) : 0;
        }
        // c.x of the part in the order of price.h: totals_from_blocks — four running sums, sum q over the part's blocks q, q + 4, q + 8, ... one after the other, then
        // ((s0 + s1) + s2) + s3: the loads side by side (256 blocks per round, staged in the pool's storage), lanes 0-3 then add from LDS in that order.  The search
        // steps of the part's blocks (budget flags, maximum) ride in the same rounds.
        double *scr = &S.py[0][0];
        static_assert(sizeof(S.py) >= 256 * sizeof(double), "a round of the part's c.x values is staged in the dual pool's storage");
        uint32_t nbud = 0, mx = 0;
        double s4 = 0.0;
        for (uint32_t c0 = p * per; c0 < pe; c0 += 256) {
            double v[4]; uint32_t vst[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { const uint32_t q = c0 + lane + (uint32_t)u * WAVE; v[u] = (q < pe && !a.local) ? ld_dev(&a.out.blk_cx[q]) : 0.0; vst[u] = (q >= pb0 && q < pb1) ? ld_dev(&a.out.blk_steps[q]) : 0u; }
#pragma unroll
            for (int u = 0; u < 4; u++) { scr[lane + (uint32_t)u * WAVE] = v[u]; nbud += vst[u] >> 31; const uint32_t st = vst[u] & 0x7FFFFFFFu; mx = st > mx ? st : mx; }
            wv.sync();
            const uint32_t cnt = pe - c0 < 256u ? pe - c0 : 256u;
            if (lane < 4 && !a.local) for (uint32_t i = lane; i < cnt; i += 4) s4 += scr[i];
            wv.sync();
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { nbud += __shfl_xor(nbud, off, 64); const uint32_t om = __shfl_xor(mx, off, 64); mx = om > mx ? om : mx; }
        const double s1 = __shfl(s4, 1, 64), s2 = __shfl(s4, 2, 64), s3 = __shfl(s4, 3, 64);
        const double pcx = ((s4 + s1) + s2) + s3;   // (lane 0's value is the part's)
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const uint32_t k = lane + (uint32_t)u * WAVE;
            if (k >= K) continue;
            long long v = 0;
#pragma unroll
            for (int sub = 0; sub < ASUB; sub++) { v += vs[u][sub]; st_dev(&a.out.act[((size_t)p * ASUB + sub) * K + k], 0); }   // (the accumulators, ready for the next sweep)
            st_host(&a.res->part_act[(size_t)p * K + k], v);
            if (v != 0 && !a.local) atomicAdd(reinterpret_cast<unsigned long long *>(&a.tact[k]), (unsigned long long)v);   // the sweep's activities: the parts' vectors added up (exact, order-free)
        }
        if (lane == 0) {
            st_dev(&a.pval[p].nbud, nbud); st_dev(&a.pval[p].mx, mx);
            if (!a.local) st_host(&a.res->part_cx[p], pcx);
            st_dev(&a.tickets[1 + p], 0u);
        }
        __threadfence_system();   // this block's rows of the host's result are in place before its ticket says so (16 blocks side by side: ~0.5 us)
    }
    const uint32_t p_first = r0 / per, p_last = (r1 - 1) / per;
    if (lane == 0) last = __hip_atomic_fetch_add(&a.tickets[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == p_last - p_first ? 1u : 0u;
    last = (uint32_t)__builtin_amdgcn_readfirstlane((int)last);
    if (!last) return;
    // the sweep's last block
    if (a.local) {  // this rank's share of a sharded sweep: the per-block values as they are (the totals are added up after the ranks' all-gather); its parts' activity rows are in place
        for (uint32_t b0 = r0 + lane; b0 < r1; b0 += WAVE * 8) {   // (eight blocks' loads in flight, then their stores)
            double vcx[8], vrc[8], vbd[8]; uint32_t vst[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { const uint32_t q = b0 + (uint32_t)u * WAVE; const bool in = q < r1; vcx[u] = in ? ld_dev(&a.out.blk_cx[q]) : 0.0; vrc[u] = in ? ld_dev(&a.out.blk_rc[q]) : 0.0; vbd[u] = in ? ld_dev(&a.out.blk_bnd[q]) : 0.0; vst[u] = in ? ld_dev(&a.out.blk_steps[q]) : 0u; }
#pragma unroll
            for (int u = 0; u < 8; u++) { const uint32_t q = b0 + (uint32_t)u * WAVE; if (q < r1) { st_host(&a.lv_cx[q], vcx[u]); st_host(&a.lv_rc[q], vrc[u]); st_host(&a.lv_bnd[q], vbd[u]); st_host(&a.lv_steps[q], vst[u]); } }
        }
    } else {
        const uint32_t n_parts = (nb + per - 1) / per;   // parts that hold blocks (the others' rows of the result are zero)
        // (asked for first, used last: the sweep's activities and the parts' step statistics)
        long long ta[2] = {0, 0};
#pragma unroll
        for (int u = 0; u < 2; u++) { const uint32_t k = lane + (uint32_t)u * WAVE; if (k < K) ta[u] = ld_dev(&a.tact[k]); }
        uint32_t pn = 0, pm = 0;
        if (lane < n_parts) { pn = ld_dev(&a.pval[lane].nbud); pm = ld_dev(&a.pval[lane].mx); }
        // the sweep's totals in the order of price.h: totals_from_blocks — lane l adds the blocks l, l + 64, ... one after the other, lane 0 then the 64 partial sums
        // in lane order (eight of a lane's blocks in flight per round)
        double cx = 0.0, rc = 0.0, bnd = 0.0;
        for (uint32_t b0 = lane; b0 < nb; b0 += WAVE * 8) {
            double vcx[8], vrc[8], vbd[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { const uint32_t q = b0 + (uint32_t)u * WAVE; const bool in = q < nb; vcx[u] = in ? ld_dev(&a.out.blk_cx[q]) : 0.0; vrc[u] = in ? ld_dev(&a.out.blk_rc[q]) : 0.0; vbd[u] = in ? ld_dev(&a.out.blk_bnd[q]) : 0.0; }
#pragma unroll
            for (int u = 0; u < 8; u++) { if (b0 + (uint32_t)u * WAVE >= nb) break; cx += vcx[u]; rc += vrc[u]; bnd += vbd[u]; }
        }
        double *red = &S.py[0][0];
        red[lane] = cx; red[WAVE + lane] = rc; red[2 * WAVE + lane] = bnd;
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) { pn += __shfl_xor(pn, off, 64); const uint32_t om = __shfl_xor(pm, off, 64); pm = om > pm ? om : pm; }   // (lanes 0-15 hold the parts)
        wv.sync();
        double tcx = 0.0, trc = 0.0, tb = 0.0;
        if (lane == 0) for (int l = 0; l < WAVE; l++) { tcx += red[l]; trc += red[WAVE + l]; tb += red[2 * WAVE + l]; }
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const uint32_t k = lane + (uint32_t)u * WAVE;
            if (k >= K) continue;
            st_dev(&a.tact[k], 0);
            st_host(&a.res->act[k], ta[u]);
            for (uint32_t i = n_parts; i < (uint32_t)ASLOTS; i++) st_host(&a.res->part_act[(size_t)i * K + k], 0);
        }
        if (lane >= n_parts && lane < (uint32_t)ASLOTS) st_host(&a.res->part_cx[lane], 0.0);
        if (lane == 0) { st_host(&a.res->cx, tcx); st_host(&a.res->rc, trc); st_host(&a.res->bnd, tb); st_host(&a.res->n_budget, pn); st_host(&a.res->max_steps, pm); }
    }
    if (lane == 0) st_dev(&a.tickets[0], 0u);
    if (prof && lane == 0) prof[13] = wv.now();
    __threadfence_system();
    wv.sync();
    if (prof && lane == 0) prof[14] = wv.now();
    if (lane == 0) __hip_atomic_store(&a.res->seq, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// d_sync: [tickets: 32 words][accumulators ASLOTS x ASUB x KMAX i64][the sweep's activities KMAX i64][PartVal x ASLOTS]
constexpr size_t SYNC_ACT = 128, SYNC_PACT = SYNC_ACT + (size_t)ASLOTS * ASUB * KMAX * 8, SYNC_PVAL = SYNC_PACT + (size_t)KMAX * 8, SYNC_BYTES = SYNC_PVAL + (size_t)ASLOTS * sizeof(PartVal);

// stage profile (HQTICK_PRICE_PROFILE=1): intervals between the stamps of price_core.h / the kernel's tail
constexpr int NPROF = 13;
const int PROF_FROM[NPROF] = {0, 1, 3, 4, 5, 5, 8, 9, 10, 6, 11, 12, 13};
const int PROF_TO[NPROF]   = {1, 2, 4, 5, 6, 8, 9, 10, 6, 11, 12, 13, 14};
const char *const PROF_NAME[NPROF] = {"reduced costs + compaction", "dual pool + order + greedy fills", "level lists + root bound", "walk", "results + activities", "[results: zero the sums", "pattern + sums in LDS", "atomics + cost loads", "values]",
                                      "acknowledgements before the ticket", "part ticket", "part sums + sweep ticket + totals (last block)", "system fence (last block)"};

size_t al16(size_t v) { return (v + 15) & ~(size_t)15; }
double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

DeviceSweeper::~DeviceSweeper() { h_stage.release(); h_res.release(); h_pats.release(); h_pin.release(); h_blkv.release(); d_tab.release(); d_pats.release(); d_blk.release(); d_sync.release(); d_prof.release(); }

bool DeviceSweeper::begin(const HostTables &t, uint32_t max_sweeps) {
    if (t.K > (uint32_t)KMAX || t.n_blocks == 0) return false;
    if (profile && (!h_prof.ensure((size_t)t.n_blocks * PSLOTS * 8 + 64) || !d_prof.ensure((size_t)t.n_blocks * PSLOTS * 8 + 64))) return false;
    T = &t; n_sweeps = 0; cap_sweeps = max_sweeps;
    max_block_cols = 0;
    for (uint32_t b = 0; b < t.n_blocks; b++) max_block_cols = std::max(max_block_cols, t.blk_off[b + 1] - t.blk_off[b]);
    if (force_nmax) max_block_cols = (uint32_t)hqblock::NMAX;   // (HQTICK_PRICE_NMAX=1: the full-size working set whatever the model — A/B switch)
    const size_t nw = t.w_row.size();
    o_off = 0; o_m = al16(o_off + (size_t)(t.n_blocks + 1) * 4); o_cap = al16(o_m + t.n_blocks); o_cost = al16(o_cap + (size_t)t.n_blocks * MMAX * 8);
    o_a = al16(o_cost + (size_t)t.n_cols * 8); o_ccap = al16(o_a + (size_t)t.n_cols * MMAX * 8); o_woff = al16(o_ccap + (size_t)t.n_cols * 4);
    o_wrow = al16(o_woff + (size_t)(t.n_cols + 1) * 4); o_wcoef = al16(o_wrow + nw * 2); tab_bytes = al16(o_wcoef + nw * 4);
    if (!h_stage.ensure(tab_bytes) || !d_tab.ensure(tab_bytes) || !h_res.ensure(sizeof(SweepResult) + 64)) return false;
    if (!h_pin.ensure((size_t)std::min<uint32_t>(max_sweeps, PIN_SWEEPS) * t.n_cols * 2 + 64)) return false;
    in_flight = false;
    if (!d_pats.ensure((size_t)max_sweeps * t.n_cols * 2) || !d_blk.ensure((size_t)t.n_blocks * 28 + 64) || !d_sync.ensure(SYNC_BYTES)) return false;
    unsigned char *h = h_stage.as<unsigned char>();
    memcpy(h + o_off, t.blk_off.data(), (size_t)(t.n_blocks + 1) * 4); memcpy(h + o_m, t.blk_m.data(), t.n_blocks); memcpy(h + o_cap, t.blk_cap.data(), (size_t)t.n_blocks * MMAX * 8);
    memcpy(h + o_cost, t.col_cost.data(), (size_t)t.n_cols * 8); memcpy(h + o_a, t.col_a.data(), (size_t)t.n_cols * MMAX * 8); memcpy(h + o_ccap, t.col_cap.data(), (size_t)t.n_cols * 4);
    memcpy(h + o_woff, t.col_woff.data(), (size_t)(t.n_cols + 1) * 4);
    if (nw) { memcpy(h + o_wrow, t.w_row.data(), nw * 2); memcpy(h + o_wcoef, t.w_coef.data(), nw * 4); }
    if (hipMemcpyAsync(d_tab.p, h, tab_bytes, hipMemcpyHostToDevice, stream) != hipSuccess) return false;
    // tickets + the wide rows' accumulators: every completed sweep leaves them zero (the parts' and the sweep's last blocks put back what they used), so only a fresh
    // buffer, or one a sweep did not come back from, is cleared (a fill kernel and ~5 us of runtime calls per coupled solve otherwise)
    if (!sync_clean) { if (hipMemsetAsync(d_sync.p, 0, SYNC_BYTES, stream) != hipSuccess) return false; }
    SweepResult *r = h_res.as<SweepResult>();
    seq = r->seq;  // (whatever the last solve left: the next sweep writes seq + 1)
    return true;
}

bool DeviceSweeper::set_caps(const int32_t *col_cap) {
    if (!T) return false;
    memcpy(h_stage.as<unsigned char>() + o_ccap, col_cap, (size_t)T->n_cols * 4);
    return hipMemcpyAsync(d_tab.as<unsigned char>() + o_ccap, h_stage.as<unsigned char>() + o_ccap, (size_t)T->n_cols * 4, hipMemcpyHostToDevice, stream) == hipSuccess;
}

bool DeviceSweeper::set_block_caps(const double *blk_cap) {
    if (!T) return false;
    const size_t bytes = (size_t)T->n_blocks * MMAX * 8;
    memcpy(h_stage.as<unsigned char>() + o_cap, blk_cap, bytes);
    return hipMemcpyAsync(d_tab.as<unsigned char>() + o_cap, h_stage.as<unsigned char>() + o_cap, bytes, hipMemcpyHostToDevice, stream) == hipSuccess;
}

bool DeviceSweeper::sweep(const double *pi, SweepTotals &out) { return launch(pi, 0, T ? T->n_blocks : 0, false, &out); }

bool DeviceSweeper::sweep_range(const double *pi, uint32_t b0, uint32_t b1, RangeValues &rv) {
    if (!T || b1 > T->n_blocks || b0 > b1) return false;
    const uint32_t nb = T->n_blocks;
    if (!h_blkv.ensure((size_t)nb * 28 + 64)) return false;
    if (b1 > b0) { if (!launch(pi, b0, b1, true, nullptr)) return false; }
    else {  // a rank without blocks (more ranks than parts): nothing to launch, the sweep still counts (the ring of patterns stays aligned across the ranks)
        if (n_sweeps >= cap_sweeps) return false;
        SweepResult *r = h_res.as<SweepResult>();
        memset(r->part_act, 0, sizeof(r->part_act));
        n_sweeps++;
    }
    unsigned char *h = h_blkv.as<unsigned char>();
    rv = RangeValues{(const double *)h, (const double *)(h + (size_t)nb * 8), (const double *)(h + (size_t)nb * 16), (const uint32_t *)(h + (size_t)nb * 24), h_res.as<SweepResult>()->part_act};
    return true;
}

bool DeviceSweeper::launch(const double *pi, uint32_t b0, uint32_t b1, bool local, SweepTotals *outp) { return launch_only(pi, b0, b1, local) && wait_done(outp); }
bool DeviceSweeper::sweep_launch(const double *pi) { return launch_only(pi, 0, T ? T->n_blocks : 0, false); }
bool DeviceSweeper::sweep_finish(SweepTotals &out) { return wait_done(&out); }

bool DeviceSweeper::launch_only(const double *pi, uint32_t b0, uint32_t b1, bool local) {
    if (!T || n_sweeps >= cap_sweeps || b1 <= b0 || in_flight) return false;
    const HostTables &t = *T;
    unsigned char *d = d_tab.as<unsigned char>();
    SweepArgs a;
    a.t = Tables{t.n_blocks, t.n_cols, t.K, (const uint32_t *)(d + o_off), (const uint8_t *)(d + o_m), (const double *)(d + o_cap), (const double *)(d + o_cost), (const double *)(d + o_a),
                 (const int32_t *)(d + o_ccap), (const uint32_t *)(d + o_woff), (const uint16_t *)(d + o_wrow), (const int32_t *)(d + o_wcoef)};
    unsigned char *blk = d_blk.as<unsigned char>();
    // the patterns of the first PIN_SWEEPS sweeps go straight into pinned host memory (16 B per block over PCIe, complete when the completion word is: every block waits
    // for its stores' acknowledgements before its ticket) — the primal side then reads them in place instead of behind a copy and a stream synchronisation; later sweeps
    // (long branch-and-price runs) keep theirs in the HBM ring
    uint16_t *xdst = n_sweeps < PIN_SWEEPS ? h_pin.dev<uint16_t>() + (size_t)n_sweeps * t.n_cols : d_pats.as<uint16_t>() + (size_t)n_sweeps * t.n_cols;
    a.out = SweepOut{xdst, (double *)blk, (double *)(blk + (size_t)t.n_blocks * 8), (double *)(blk + (size_t)t.n_blocks * 16),
                     (long long *)(d_sync.as<unsigned char>() + SYNC_ACT), (uint32_t *)(blk + (size_t)t.n_blocks * 24), profile ? d_prof.as<uint64_t>() : nullptr, (uint32_t)ASUB, dbg};
    a.budget = budget; a.seq = ++seq; a.tickets = d_sync.as<uint32_t>(); a.tact = (long long *)(d_sync.as<unsigned char>() + SYNC_PACT); a.pval = (PartVal *)(d_sync.as<unsigned char>() + SYNC_PVAL); a.res = h_res.dev<SweepResult>();
    a.first = b0; a.local = local ? 1u : 0u;
    if (local) { unsigned char *lv = h_blkv.dev<unsigned char>(); a.lv_cx = (double *)lv; a.lv_rc = (double *)(lv + (size_t)t.n_blocks * 8); a.lv_bnd = (double *)(lv + (size_t)t.n_blocks * 16); a.lv_steps = (uint32_t *)(lv + (size_t)t.n_blocks * 24); }
    else { a.lv_cx = a.lv_rc = a.lv_bnd = nullptr; a.lv_steps = nullptr; }
    memset(a.pi, 0, sizeof(a.pi));
    memcpy(a.pi, pi, (size_t)t.K * 8);
    const double t0 = now_us();
    // wavefronts per block: four.  (Measured on configs[3]'s 4096 sixteen-column blocks as well, where the helpers cost residency — six blocks per CU instead of
    // nine: 193 us per sweep against 219 with two and 226 with one; profiles/r06/price_sweep_waves.txt.)
    const uint32_t nblk = b1 - b0;
    const int nw = force_waves == 1 || force_waves == 2 ? force_waves : 4;
#define HQ_SWEEP(SH_, NW_) hipLaunchKernelGGL((k_price_sweep<SH_, NW_>), dim3(nblk), dim3(WAVE * NW_), 0, stream, a)
#define HQ_SWEEP_N(SH_) do { if (nw == 4) HQ_SWEEP(SH_, 4); else if (nw == 2) HQ_SWEEP(SH_, 2); else HQ_SWEEP(SH_, 1); } while (0)
    if (max_block_cols <= 8) HQ_SWEEP_N(hqblock::SharedN<8>);
    else if (max_block_cols <= 16) HQ_SWEEP_N(hqblock::SharedN<16>);
    else HQ_SWEEP_N(hqblock::SharedN<hqblock::NMAX>);
#undef HQ_SWEEP_N
#undef HQ_SWEEP
    if (hipGetLastError() != hipSuccess) return false;
    in_flight = true; sync_clean = false; flight_t0 = t0; flight_blocks = b1 - b0;
    return true;
}

bool DeviceSweeper::wait_done(SweepTotals *outp) {
    if (!T || !in_flight) return false;
    in_flight = false;
    const HostTables &t = *T;
    const double t0 = flight_t0;
    // wait for the sweep's own completion word (pinned memory); the stream synchronisation is the fallback after 2 s
    volatile SweepResult *r = h_res.as<SweepResult>();
    for (uint64_t spins = 0;; spins++) {
        if (__atomic_load_n(&r->seq, __ATOMIC_ACQUIRE) == seq) break;
        if ((spins & 0xFFFF) == 0xFFFF && now_us() - t0 > 2.0e6) {
            if (hipStreamSynchronize(stream) != hipSuccess) return false;
            if (__atomic_load_n(&r->seq, __ATOMIC_ACQUIRE) != seq) return false;
            break;
        }
    }
    last_kernel_us = now_us() - t0;
    sync_clean = true;   // (the completion word is the sweep's last store)
    if (profile) {  // HQTICK_PRICE_PROFILE=1: per-stage medians over the blocks of this sweep (100 MHz wavefront clock), accumulated for end()
        // (the stamps are written to HBM — stores into pinned memory would put a PCIe acknowledgement into every wait of the block — and copied once the kernel has ended)
        if (hipMemcpyAsync(h_prof.p, d_prof.p, (size_t)t.n_blocks * PSLOTS * 8, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) return false;
        const uint64_t *pr = h_prof.as<uint64_t>();
        for (int st = 0; st < NPROF; st++) {
            const int from = PROF_FROM[st], to = PROF_TO[st];
            std::vector<double> v; v.reserve(t.n_blocks);
            for (uint32_t b = 0; b < t.n_blocks; b++) { const uint64_t t0s = pr[(size_t)b * PSLOTS + from], t1s = pr[(size_t)b * PSLOTS + to]; if (t0s && t1s >= t0s) v.push_back((double)(t1s - t0s) / 100.0); }
            if (v.empty()) continue;
            std::sort(v.begin(), v.end()); prof_med[st] += v[v.size() / 2]; prof_max[st] += v.back();
        }
        uint64_t first = ~0ull, last_done = 0, seq_at = 0;
        for (uint32_t b = 0; b < t.n_blocks; b++) { const uint64_t *q = pr + (size_t)b * PSLOTS; if (q[0] && q[0] < first) first = q[0]; if (q[6] > last_done) last_done = q[6]; if (q[14] > seq_at) seq_at = q[14]; }
        if (first != ~0ull) { prof_span += (double)(last_done - first) / 100.0; if (seq_at) prof_tail += (double)(seq_at - last_done) / 100.0; }
        double smax = 0; for (uint32_t b = 0; b < t.n_blocks; b++) smax = std::max(smax, (double)pr[(size_t)b * PSLOTS + 7]);
        prof_steps += smax; prof_n++;
    }
    total_sweeps++; total_block_solves += flight_blocks; total_us += last_kernel_us;
    n_sweeps++;
    if (!outp) return true;
    SweepTotals &out = *outp;
    out.cx = r->cx; out.rc = r->rc; out.bnd = r->bnd; out.n_budget = r->n_budget; out.max_steps = r->max_steps;
    out.act.resize(t.K);
    for (uint32_t k = 0; k < t.K; k++) out.act[k] = r->act[k];
    out.part_cx.assign(r->part_cx, r->part_cx + ASLOTS);
    out.part_act.resize((size_t)ASLOTS * t.K);
    for (size_t i = 0; i < (size_t)ASLOTS * t.K; i++) out.part_act[i] = r->part_act[i];
    return true;
}

const uint16_t *DeviceSweeper::patterns(uint32_t first, uint32_t count) {
    if (!T || first + count > n_sweeps || in_flight) return nullptr;
    const size_t nc = T->n_cols, bytes = (size_t)count * nc * 2;
    if (first + count <= PIN_SWEEPS) return h_pin.as<uint16_t>() + (size_t)first * nc;   // in place (launch_only)
    if (!h_pats.ensure(bytes + 64)) return nullptr;
    if (count == 0) return h_pats.as<uint16_t>();
    const uint32_t n_pin = first < PIN_SWEEPS ? PIN_SWEEPS - first : 0;   // sweeps of the range that sit in pinned memory
    if (n_pin) memcpy(h_pats.p, h_pin.as<uint16_t>() + (size_t)first * nc, (size_t)n_pin * nc * 2);
    if (hipMemcpyAsync(h_pats.as<uint16_t>() + (size_t)n_pin * nc, d_pats.as<uint16_t>() + (size_t)(first + n_pin) * nc, bytes - (size_t)n_pin * nc * 2, hipMemcpyDeviceToHost, stream) != hipSuccess) return nullptr;
    if (hipStreamSynchronize(stream) != hipSuccess) return nullptr;
    return h_pats.as<uint16_t>();
}

void DeviceSweeper::end() {
    if (in_flight) { hipStreamSynchronize(stream); in_flight = false; }   // (a solve that gave up between sweep_launch and sweep_finish: the kernel must not outlive the tables' owner)
    if (profile && prof_n) {
        fprintf(stderr, "[price profile] %d sweeps, per block and sweep (us; median over blocks / slowest block, averaged over the sweeps):", prof_n);
        for (int st = 0; st < NPROF; st++) fprintf(stderr, "  %s %.1f / %.1f;", PROF_NAME[st], prof_med[st] / prof_n, prof_max[st] / prof_n);
        fprintf(stderr, "  first block's start -> last block's results %.1f, -> completion word stored %.1f more;", prof_span / prof_n, prof_tail / prof_n);
        fprintf(stderr, "  most search steps of a block %.0f; sweep as the host saw it %.1f us\n", prof_steps / prof_n, total_us / (double)std::max<uint64_t>(1, total_sweeps));
        for (int st = 0; st < NPROF; st++) prof_med[st] = prof_max[st] = 0; prof_steps = 0; prof_span = prof_tail = 0; prof_n = 0;
    }
    T = nullptr;
}

}  // namespace hqprice
