// One PRICE SWEEP of the coupled placement model: every worker's block solved exactly under the current prices of the model's wide rows.
//
// Where it sits.  run_scheduling_solver (/root/reference/crates/tako/src/internal/scheduler/solver.rs:95-192) creates one block of `nat` columns
// per worker — one column per (batch, variant) the worker can run, one row per resource — and couples the blocks through a handful of WIDE rows:
// the batch-size rows of unsaturated batches (:264-271), the "blocker short" rows (:233-253) and the priority-cut rows over the workers where a
// blocker leaves no gap (:395-429).  The reference hands the whole thing to HiGHS (solver/highs.rs:65-88).  Here the wide rows are priced out
// (csrc/price.cpp: Dantzig-Wolfe / Lagrangian over the wide rows, a cutting-plane master of K + 1 variables on the host), and what is left per set
// of prices pi is W independent bounded integer knapsacks
//        V_w(pi) = max { (c_w - pi A_w) . x :  R_w x <= free_w,  0 <= x <= cap,  x integer }        (<= 32 columns, <= 4 rows)
// — the W-way data parallelism of the coupled tick, one workgroup per worker (one wavefront of it runs the block's chain, the others share its dual pool and run its greedy fills), solved EXACTLY (integer blocks, not their LP relaxation: the bound
// pi . h + sum_w V_w(pi) is then at least as tight as the LP bound HiGHS starts from, and every sweep's maximisers are integer patterns the primal
// side can use as they are).  The block solver is block_core.h's: dual-vertex pool, 64 greedy fills, depth-first walk with 64 children per step.
//
// Everything lane-varying goes through the Wave policy, so the CPU tests execute the same code with the wavefront emulated (hqblock::HostWave).
#pragma once
#include "block_core.h"

namespace hqprice {

using hqblock::MMAX;
using hqblock::NMAX;
using hqblock::Shared;
using hqblock::WAVE;

constexpr int KMAX = 128;  // distinct wide left-hand sides of one model (prices staged in LDS)
constexpr int ASLOTS = 16; // the wide rows' activities are accumulated in ASLOTS partial vectors, one per PART of the model = contiguous range of blocks (block b adds into
                           // slot b / ceil(n_blocks / ASLOTS)): 1024 blocks on one address serialise in L2 — and the master takes the parts' cuts one by one (price.h)
HQB_HD uint32_t part_size(uint32_t n_blocks) { return (n_blocks + (uint32_t)ASLOTS - 1) / (uint32_t)ASLOTS; }

// The model's blocks, flattened (device-visible memory).  Column q of block b is entry blk_off[b] + q of the col_* arrays.
struct Tables {
    uint32_t n_blocks, n_cols, K;
    const uint32_t *blk_off;   // [n_blocks + 1]
    const uint8_t *blk_m;      // [n_blocks] resource rows of the block (<= MMAX)
    const double *blk_cap;     // [n_blocks * MMAX] capacities on the row's own grid (integers carried in f64)
    const double *col_cost;    // [n_cols] objective coefficient (the component's scaled costs, >= 0)
    const double *col_a;       // [n_cols * MMAX] amounts on the row's grid (integers carried in f64; 0 = the column does not use the row)
    const int32_t *col_cap;    // [n_cols] upper bound of the column
    const uint32_t *col_woff;  // [n_cols + 1] the column's entries in the wide rows
    const uint16_t *w_row;     // wide row index (< K)
    const int32_t *w_coef;     // integer coefficient (in the row's <= form: a `>=` row enters negated)
};

// What one sweep leaves behind.
struct SweepOut {
    uint16_t *x;          // [n_cols] the maximisers: one integer pattern per block
    double *blk_cx;       // [n_blocks] c . x of the block's pattern (original costs)
    double *blk_rc;       // [n_blocks] (c - pi A) . x
    double *blk_bnd;      // [n_blocks] >= V_w(pi): equal to blk_rc (plus rounding slack) unless the block's search ran out of budget, then its LP bound
    long long *act;       // [ASLOTS * K] partial sums over blocks of A_w x — integer coefficients, so the atomic sums are exact and order-free
    uint32_t *blk_steps;  // [n_blocks] search steps (0 = closed at the root); bit 31: budget exhausted
    uint64_t *prof;       // optional [n_blocks * PSLOTS]: wavefront clock at the stage boundaries (HQTICK_PRICE_PROFILE=1); nullptr = off
    // `act` may hold asub (a power of two) vectors per part instead of one: block b adds into vector b % asub of its part, whoever adds the totals up sums them
    // (price.hip: ASUB; the host emulation keeps one vector per part — 0 or 1 here).
    uint32_t asub = 1;
    uint32_t dbg = 0;     // experiments only (HQTICK_PRICE_DBG): bit 0 = no global atomics (the sweep's activities are then wrong: timing runs), bit 1 = pools above 8 entries are rebuilt by the main wavefront (same answers: tests)
};

// columns of a priced block whose reduced cost is at most this fraction of the block's largest original cost stay at zero (their possible
// contribution is added to the block's bound)
constexpr double RC_DROP = 1e-12;
constexpr int PSLOTS = 16;  // profile stamps per block: 0-7 the stages of solve_priced_block (7: search steps), 8-10 inside its last stage, 11-14 the kernel's tail (price.hip)

template <class W, class SH>
HQB_HD void solve_priced_block(W &wv, SH &S, const Tables &t, const double *pi, uint32_t b, const SweepOut &out, uint32_t budget) {
    const uint32_t c0 = t.blk_off[b], nb = t.blk_off[b + 1] - c0;
    const int m = (int)t.blk_m[b];
    uint64_t *prof = out.prof ? out.prof + (size_t)b * PSLOTS : nullptr;
    if (prof && wv.first()) { prof[0] = wv.now(); for (int i = 1; i < PSLOTS; i++) prof[i] = 0; }
    // stage the prices: the dual pool is empty at this point, its storage is the staging area
    double *spi = &S.py[0][0];
    static_assert(hqblock::PCAP * MMAX >= KMAX, "the prices are staged in the dual pool's storage");
    // Round trip 1: lane q reads everything the block needs of its column q — where its wide-row entries begin, cost, bound, amounts (they wait in the work problem's
    // storage, idle until setup_work, for the test below and the compaction) — while the prices are copied into LDS.
    wv.each([&](int lane) {
        for (uint32_t k = (uint32_t)lane; k < t.K; k += WAVE) spi[k] = pi[k];
        if (lane < SH::NN) S.colcap[lane] = 2147483647;
        if (lane < MMAX) S.cap[lane] = lane < m ? t.blk_cap[(size_t)b * MMAX + lane] : 0.0;
        if ((uint32_t)lane < nb) {
            const uint32_t j = c0 + (uint32_t)lane;
            S.woff[lane] = t.col_woff[j]; if ((uint32_t)lane + 1 == nb) S.woff[nb] = t.col_woff[j + 1];
            S.wc[lane] = t.col_cost[j];
            S.wcap[lane] = t.col_cap[j];
            HQB_UNROLL
            for (int r = 0; r < MMAX; r++) S.wa[r][lane] = t.col_a[(size_t)j * MMAX + r];
        }
    });
    if (wv.first()) { S.status = hqblock::ST_OK; S.steps = 0; S.steps_p1 = 0; S.n = 0; S.m = m; S.npool = 0; S.usedres = 0; }
    wv.sync();
    // Round trip 2: the block's wide-row entries — one contiguous range — lane = ENTRY, side by side into LDS (when they fit: ECAP; a chain of loads per column
    // otherwise, four at a time).  The reduced costs then add up from LDS in the entries' order, and the results stage finds them there again.
    const uint32_t E0 = nb ? S.woff[0] : 0u, E = nb ? S.woff[nb] - E0 : 0u;
    const bool cached = E <= (uint32_t)SH::ECAP;
    if (cached) {
        wv.each([&](int lane) { for (uint32_t r = (uint32_t)lane; r < E; r += WAVE) { S.ewr[r] = t.w_row[E0 + r]; S.ewc[r] = t.w_coef[E0 + r]; } });
        wv.sync();
    }
    wv.each([&](int lane) {
        double rc = -1.0;
        if ((uint32_t)lane < nb) {
            const uint32_t e0 = S.woff[lane], e1 = S.woff[lane + 1];
            rc = S.wc[lane];
            if (cached) { for (uint32_t e = e0 - E0; e < e1 - E0; e++) rc -= spi[S.ewr[e]] * (double)S.ewc[e]; }
            else for (uint32_t e = e0; e < e1; e += 4) {
                uint16_t wr[4]; int32_t wc[4];
                HQB_UNROLL
                for (int u = 0; u < 4; u++) { const bool in = e + (uint32_t)u < e1; wr[u] = in ? t.w_row[e + (uint32_t)u] : (uint16_t)0; wc[u] = in ? t.w_coef[e + (uint32_t)u] : 0; }
                HQB_UNROLL
                for (int u = 0; u < 4; u++) { if (e + (uint32_t)u >= e1) break; rc -= spi[wr[u]] * (double)wc[u]; }
            }
        }
        S.lane_val[lane] = rc;
    });
    wv.sync();
    int lmax = -1;
    const double cmax = wv.argmax([&](int lane) { return (uint32_t)lane < nb ? S.wc[lane] : -1.0; }, &lmax);
    const double thr = (cmax > 0.0 ? cmax : 1.0) * RC_DROP;
    const uint64_t elig = wv.ballot([&](int lane) { return (uint32_t)lane < nb && S.lane_val[lane] > thr && S.wcap[lane] >= 1; });
    const int n = __builtin_popcountll(elig);
    wv.sync();  // every lane has read the prices: the pool's storage may be written again
    // the columns that stay: compacted, reduced cost as the block's cost
    wv.each([&](int lane) {
        if (!((elig >> lane) & 1)) return;
        const int q = __builtin_popcountll(elig & ((1ull << lane) - 1ull));
        S.c[q] = S.lane_val[lane];
        S.gcol[q] = lane;
        S.colcap[q] = S.wcap[lane];
        for (int r = 0; r < MMAX; r++) { const double v = r < m ? S.wa[r][lane] : 0.0; S.a[r][q] = v; S.ainv[r][q] = v > 0.0 ? 1.0 / v : 0.0; }
    });
    // what the dropped columns could add at most (0 < rc <= thr): part of the block's bound
    double dropped = 0.0;
    for (uint32_t q = 0; q < nb; q++) {
        const double rc = S.lane_val[q];
        if (((elig >> q) & 1) || !(rc > 0.0)) continue;
        dropped += rc * (double)(S.wcap[q] < 65536 ? S.wcap[q] : 65536);
    }
    wv.sync();
    for (int q = n; q < SH::NN; q++) if (wv.first()) { S.c[q] = 0.0; for (int r = 0; r < MMAX; r++) { S.a[r][q] = 0.0; S.ainv[r][q] = 0.0; } }
    if (wv.first()) S.n = n;
    wv.sync();
    uint16_t *x = out.x + c0;
    if (n == 0) {
        hqblock::pool_skip(wv);
        wv.each([&](int lane) { if ((uint32_t)lane < nb) x[lane] = 0; });
        if (wv.first()) { wv.put(&out.blk_cx[b], 0.0); wv.put(&out.blk_rc[b], 0.0); wv.put(&out.blk_bnd[b], dropped); wv.put(&out.blk_steps[b], 0u); }
        return;
    }
    if (prof && wv.first()) prof[1] = wv.now();  // reduced costs, columns compacted
    // search order (ascending size) and greedy order (descending value density), by rank counting — as build_block does for a class block (both keys in one pass,
    // both ranks in the next: the density keys wait in the greedy's value array, which nobody reads before the fills have written it)
    wv.each([&](int lane) {
        if (lane >= n) return;
        double sz = 0.0, w = 0.0;
        for (int r = 0; r < m; r++) if (S.a[r][lane] > 0.0) { sz += S.a[r][lane] / (S.cap[r] + 1.0); w += S.cap[r] > 0.0 ? S.a[r][lane] / S.cap[r] : 1e30; }
        S.lane_val[lane] = sz;
        S.gval[lane] = w > 0.0 ? S.c[lane] / w : 0.0;
    });
    wv.sync();
    wv.each([&](int lane) {
        if (lane >= n) return;
        const double mine = S.lane_val[lane], dens = S.gval[lane];
        int rank = 0, drank = 0;
        for (int i = 0; i < n; i++) {
            const double o = S.lane_val[i], od = S.gval[i];
            if (o < mine || (o == mine && i < lane)) rank++;
            if (od > dens || (od == dens && i < lane)) drank++;
        }
        S.pi[rank] = (uint8_t)lane;
        S.pd[drank] = (uint8_t)lane;
    });
    wv.sync();
    // dual pool (ordered) and greedy fills: the section the workgroup's other wavefronts take part in (block_core.h: pool_sections)
    hqblock::pool_main(wv, S, (out.dbg & 2u) ? 8u : (uint32_t)hqblock::PCAP);   // (dbg bit 1: every pool of more than 8 entries takes the rebuild path — tests)
    if (prof && wv.first()) prof[2] = wv.now();  // dual pool built and ordered
    if (prof && wv.first()) prof[3] = wv.now();
    const uint32_t all = n >= 32 ? 0xFFFFFFFFu : ((1u << n) - 1u);
    hqblock::setup_work(wv, S, all, -1, 0);
    hqblock::pool_await(wv, S);  // the greedy fills (they ran beside the pool and the level lists): the first incumbent
    double capv[MMAX];
    for (int r = 0; r < MMAX; r++) capv[r] = S.cap[r];
    const double root = hqblock::lp_bound(S, n, capv);
    if (prof && wv.first()) prof[4] = wv.now();  // level lists, root bound
    bool ok = true;
    uint32_t left = budget;
    if (!(root <= S.best + 1e-12 * S.best)) {
        if (wv.first()) { for (int r = 0; r < MMAX; r++) S.rem[n][r] = S.cap[r]; S.zfix[n] = 0.0; }
        wv.sync();
        ok = hqblock::walk(wv, S, hqblock::MODE_MAX, 0.0, &left, nullptr);
    }
    wv.sync();
    if (prof && wv.first()) prof[5] = wv.now();  // walk
    // results: the pattern, its value at the original and at the reduced costs, the block's contribution to the wide rows — summed per block in LDS first
    // (the dual pool's storage: the walk is over — and the pool is the one array whose size does not shrink with the working set's column capacity), then ONE global
    // atomic per row the block touches
    long long *lact = reinterpret_cast<long long *>(&S.py[0][0]);
    static_assert(sizeof(S.py) >= sizeof(long long) * KMAX, "the block's activities are summed in the dual pool's storage");
    wv.each([&](int lane) { for (uint32_t k = (uint32_t)lane; k < t.K; k += WAVE) lact[k] = 0; });
    wv.sync();
    if (prof && wv.first()) prof[8] = wv.now();
    // (lane = column: the pattern; then lane = ENTRY of the block's wide-row lists — they are one contiguous range — two per lane in flight, the entry's column found by
    // counting the list starts at or before it: one or two round trips for the block instead of a chain of them per column, 0.5 us median / 20 us on the last blocks)
    wv.each([&](int lane) {
        if ((uint32_t)lane >= nb) return;
        uint32_t xv = 0;
        if ((elig >> lane) & 1) xv = S.xbest[__builtin_popcountll(elig & ((1ull << lane) - 1ull))];
        x[lane] = (uint16_t)xv;
        S.wcap[lane] = (int32_t)xv;   // (the work problem is done with its caps)
    });
    wv.sync();
    wv.each([&](int lane) {
        for (uint32_t r0 = (uint32_t)lane; r0 < E; r0 += 2u * WAVE) {
            const uint32_t r1 = r0 + (uint32_t)WAVE;
            const bool two = r1 < E;
            const uint16_t wr0 = cached ? S.ewr[r0] : t.w_row[E0 + r0], wr1 = !two ? (uint16_t)0 : cached ? S.ewr[r1] : t.w_row[E0 + r1];
            const int32_t wc0 = cached ? S.ewc[r0] : t.w_coef[E0 + r0], wc1 = !two ? 0 : cached ? S.ewc[r1] : t.w_coef[E0 + r1];
            uint32_t q0 = 0, q1 = 0;
            for (uint32_t i = 1; i < nb; i++) { const uint32_t o = S.woff[i] - E0; q0 += o <= r0 ? 1u : 0u; q1 += o <= r1 ? 1u : 0u; }
            const long long x0 = (long long)S.wcap[q0], x1 = two ? (long long)S.wcap[q1] : 0;
            if (x0) wv.lds_add_i64(&lact[wr0], (long long)wc0 * x0);
            if (x1) wv.lds_add_i64(&lact[wr1], (long long)wc1 * x1);
        }
    });
    wv.sync();
    if (prof && wv.first()) prof[9] = wv.now();
    wv.each([&](int lane) {
        const uint32_t nsub = out.asub ? out.asub : 1u;
        long long *slot = out.act + ((size_t)(b / part_size(t.n_blocks)) * nsub + (b & (nsub - 1u))) * t.K;
        for (uint32_t k = (uint32_t)lane; k < t.K; k += WAVE) if (lact[k] != 0 && !(out.dbg & 1u)) wv.atomic_add_i64(&slot[k], lact[k]);
    });
    // (the original costs of the kept columns: one load per lane, side by side — lane 0 reading them one after the other was a chain of n global loads, 16 us on an
    // eight-column block against 3 us on the median one: the tail of every sweep)
    wv.each([&](int lane) { if (lane < n) S.lane_val[lane] = t.col_cost[c0 + (uint32_t)S.gcol[lane]]; });  // (S.wc holds the work problem's costs by now)
    wv.sync();
    if (prof && wv.first()) prof[10] = wv.now();
    if (wv.first()) {
        double cx = 0.0, rc = 0.0;  // fixed order: the same sums on every replica
        for (int q = 0; q < n; q++) { const double xv = (double)S.xbest[q]; cx += S.lane_val[q] * xv; rc += S.c[q] * xv; }
        wv.put(&out.blk_cx[b], cx);
        wv.put(&out.blk_rc[b], rc);
        // the walk closes a node whose bound is within 1e-12 (relative) of the incumbent: the optimum is not above best * (1 + 1e-12)
        wv.put(&out.blk_bnd[b], (ok ? rc * (1.0 + 2e-12) : (root > rc ? root : rc)) + dropped);
        wv.put(&out.blk_steps[b], S.steps | (ok ? 0u : 0x80000000u));
        if (prof) { prof[6] = wv.now(); prof[7] = S.steps; }
    }
}

}  // namespace hqprice
