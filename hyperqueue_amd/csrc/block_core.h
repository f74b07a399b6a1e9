// Exact solver of ONE worker-class block of the separable placement model: one wavefront runs the block's chain (up to three more share its dual pool and run its
// greedy fills: pool_sections).
//
// Where it sits: run_scheduling_solver (/root/reference/crates/tako/src/internal/scheduler/solver.rs:95-192) creates, per worker, one `nat`
// column per (batch, variant) the worker can run now and one row per resource: sum_j a[r][j] x_j <= free[r].  On a tick without priority
// cuts, blockers and multi-node batches these per-worker blocks are the whole model (host_model.cpp, "separable instances"), workers with
// the same (free, total, eligibility) share one block, and a steady-state cluster has about as many such classes as workers.  Each block is a
// bounded integer knapsack with <= 4 resource rows and <= 32 columns; the blocks are independent -- the W-way data parallelism of the tick.
//
// Result convention = csrc/milp.h's canonical optimum: among the integer points within 1e-9 (relative) of the optimum the one that
// minimises the LAST column, then the one before it, ...  Feasibility is decided in exact integer `ResourceAmount` arithmetic (the host solver
// works in f64 units with a 1e-9 tolerance; on the 1/10000 grid the two agree).  The integers are CARRIED in f64: every amount, capacity and
// product stays below 2^52, where f64 add / mul / fma are exact — the MI355X runs f64 at full rate while 64-bit integer multiply and divide
// are multi-instruction sequences, and a single wavefront per block exposes every instruction's latency.
//
// Algorithm (all of a wavefront's 64 lanes work on the same block):
//   build     columns / rows from the class descriptor, rows divided by their gcd, costs in the reference's operation order (solver.rs:550-568)
//   duals     every basis of the dual polyhedron {y >= 0, A^T y >= c} is tried (lane-strided over the C(n + m, m) choices of m tight
//             constraints; 4 x 4 elimination in registers).  A basic point y that covers a column set C (a_j . y >= c_j for j in C) bounds every
//             sub-problem over columns F within C:  LP_F(rem) <= y . rem.  The pool keeps (y, cover mask, tight-column mask).
//   greedy    64 column orders, one per lane, each raised to its maximum: the best one is the first incumbent
//   walk      depth-first over the columns of a WORK PROBLEM in its search order — large requests are decided first, the two smallest are left
//             for the leaves (that order cuts the trees ~10x against the model's own column order).  The 64 lanes evaluate 64 values of the
//             current column at once (child bound = fixed part + min over the level's duals of y . rem), __ballot gives the survivors;
//             with two columns left the lanes enumerate one and the other follows exactly -> 64 complete solutions per step.
//   phase 1   walk over all columns, maximising.
//   phase 2   for j = n-1 .. 0: the smallest value of column j for which the columns before it can still reach optimum - 1e-9, by probes
//             "any point with x_j <= mid?" (a walk that stops at the first leaf; the cap enters the bound as a Lagrangian penalty), first just
//             below the current value, then bisecting.  The result is the lexicographic minimum read from the last column = the canonical optimum.
// Everything lane-varying is a function of (lane) handed to the Wave policy (ballot / arg-max / each); the uniform control flow is shared by
// the device kernel (block_solve.hip, Wave = one wavefront) and the host emulation the CPU tests run (lanes in a loop).
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define HQB_HD __host__ __device__ inline
#define HQB_UNROLL _Pragma("unroll")
#else
#define HQB_HD inline
#define HQB_UNROLL
#endif

namespace hqblock {

constexpr int NMAX = 32;     // columns of one block
constexpr int MMAX = 4;      // resource rows of one block
constexpr int WAVE = 64;
constexpr int PCAP = 192;    // dual points kept per block (a C3-shaped block has at most 80, a C4-shaped one — 16 columns — 175: measured over 68 698 blocks of the steady-state
                             // and unsaturated ticks; what does not fit is dropped: weaker bounds, same answer).  400 until round 4: with the
                             // level lists and the greedy's vectors sharing their storage the block is 25.9 KB of LDS instead of 40.8: six resident per CU instead of four
constexpr int DPRE = 48;     // dual points a level of the walk looks at (C3-shaped: <= 43)
constexpr int GCOLS = 64;    // (batch, variant) columns of a tick the eligibility mask can address
constexpr int32_t UB_LIMIT = 65535;  // a column that could be taken more often than this goes to the host solver

enum { ST_OK = 0, ST_BUDGET = 1, ST_UNSUPPORTED = 2 };

// The tick's (batch, variant) columns, shared by every class: request entries, weight, resource pool sums (solver.rs:68-82).
struct ColTable {
    uint32_t n_cols, R;
    const uint32_t *ent_off;     // [n_cols + 1]
    const uint32_t *ent_res;     // resource id
    const uint8_t *ent_kind;     // HQ_ENTRY_AMOUNT = 0 / HQ_ENTRY_ALL = 1
    const uint64_t *ent_amount;
    const uint32_t *weight;      // [n_cols] ResourceWeight, 10 000 = 1.0
    const double *pool;          // [R] resource_sums
    // When the arrays above lie in ONE allocation [blob, blob + blob_bytes) (16-byte aligned, blob_bytes a multiple of 16, <= BLOB_MAX) the kernel
    // stages it into LDS with one 16-byte load per lane and reads the tables from there; nullptr: the arrays are read where they are.
    const void *blob;
    uint32_t blob_bytes;
};
// One worker class: what the block depends on.
struct ClassTable {
    uint32_t n_classes;
    const uint64_t *free_;   // [n_classes * R]   never HQ_AMOUNT_MAX on a resource an eligible column uses (the host keeps those classes)
    const uint64_t *total;   // [n_classes * R]
    const uint64_t *elig;    // [n_classes]       bit g: column g exists on this class's workers (solver.rs:123-126)
};
struct Output {
    uint32_t *x;        // [n_classes * n_cols] count per column (0 where not eligible)
    uint32_t *status;   // [n_classes] ST_*
    uint32_t *steps;    // [n_classes] search steps of both phases (statistics)
    uint64_t *prof;     // optional [n_classes * 8] timestamps (wavefront clock) at the stage boundaries of solve_block; nullptr = off
};

// Templated on the column capacity N: the working set of a block of <= N columns.  The algorithm never looks at N beyond "does the block fit" (n <= N), so a block's
// answer is the same whatever N it is solved under; what changes is the footprint — 25.9 KB at N = 32, 17.6 KB at N = 16, 13.5 KB at N = 8 — i.e. how many blocks a
// CU holds at once (six / nine / eleven): k_price_sweep picks the smallest N its model's widest block fits (price.hip).
template <int N_>
struct SharedN {  // one block's working set: LDS on the device
    static constexpr int NN = N_;
    int n, m, status;
    uint32_t steps, steps_p1;
    uint64_t usedres;               // resources some eligible column touches
    int gcol[N_];                 // block column -> tick column
    double c[N_];
    double a[MMAX][N_], ainv[MMAX][N_];  // amounts on the row's own grid (integers carried in f64) and their reciprocals (0 where the column does not use the row)
    double cap[MMAX];
    uint8_t pi[N_];               // block columns by ascending size (the search decides the large ones first)
    uint8_t pd[N_];               // block columns by descending value density (first greedy order)
    // dual pool
    uint32_t npool;
    alignas(16) double py[PCAP][MMAX];  // (16-byte aligned: the staging area of build_block overlays it)
    uint32_t pcover[PCAP], ptight[PCAP];
    uint16_t porder[PCAP];          // pool entries by ascending y . cap: the tightest bounds at the root come first in every level's list
    float pkey[PCAP];               // y . cap of an entry (the sort key), written with the entry
    uint16_t ptix[PCAP];            // the entry's basis number: equal keys are ordered by it, so that the order does not depend on who found which entry first
    uint32_t cnext;                 // next chunk of basis numbers to hand out (blocks worked on by several wavefronts: pool_sections)
    uint32_t bar, gdone;            // ... the pool wavefronts' barrier counter, and the greedy wavefront's flag
    uint32_t redo_at;               // a pool with more entries than this is rebuilt by the main wavefront alone (PCAP; tests force the path with a small value)
    // work problem: columns in search order (position wn - 1 is decided first)
    int wn;
    uint8_t wcol[N_];
    double wc[N_];
    double wa[MMAX][N_], winv[MMAX][N_];
    uint32_t wmask[N_ + 1];       // block columns at positions < k
    // level lists of the walk
    uint16_t dl[N_ + 1][DPRE];    // per level: pool entries that are vertices of that level's dual polyhedron
    float dpen[N_ + 1][DPRE];     // ... and the penalty of a capped column (phase 2), rounded up
    // the greedy's vectors.  (Until round 6 they shared the level lists' storage; with the greedy fills on a wavefront of their own the main wavefront builds the
    // first level lists WHILE the fills run — pool_main / pool_await — and the two need their own 1.5 / 3 / 6 KB at 8 / 16 / 32 columns.)
    uint8_t perm[N_][WAVE];       // [column slot][lane] — lanes side by side, so that a wavefront's accesses spread over the LDS banks
    alignas(8) uint16_t gx[N_][WAVE];   // (8-byte aligned: build_block overlays its int64 amounts here — UBSan found them on a 4-byte boundary at N = 32)
    double gval[WAVE];            // the fills' values
    int32_t wcap[N_];             // upper cap of a position (INT32_MAX = none)
    int32_t colcap[N_];           // upper cap of a block column (INT32_MAX = none): the priced blocks of the coupled solve (price_core.h) carry their model bounds here
    uint32_t woff[N_ + 1];        // priced blocks: where the columns' wide-row entries begin (price_core.h)
    static constexpr int ECAP = 16 * N_;   // ... and the entries themselves, when the block has at most this many (c3p: 70 on 8 columns)
    uint16_t ewr[ECAP]; int32_t ewc[ECAP];
    uint32_t dcnt[N_ + 1];
    // level stack of the walk (level k = number of positions still free)
    double rem[N_ + 1][MMAX];
    double zfix[N_ + 1];
    int32_t ptr[N_ + 1], ub[N_ + 1];
    uint32_t xsel[N_];            // by position
    // incumbent / completion, by block column
    uint32_t xbest[N_];
    double best;
    // greedy
    double lane_val[WAVE];
    uint32_t lane_rng[WAVE];        // setup_work: level ranges of the pool entry a lane is looking at
    uint64_t lmask[N_ + 1];       // setup_work: per level, which of the 64 entries of the current pass enter its list
};
using Shared = SharedN<NMAX>;

constexpr int BLOB_MAX = 8192;               // largest column table the kernel stages in LDS
struct alignas(16) V16 { uint64_t lo, hi; };
constexpr int64_t VAL_LIMIT = 1ll << 52;    // amounts and capacities stay exact in f64

// How often an amount `a` (> 0, reciprocal `inv`) fits into `rem` (>= 0): floor(rem / a) without a division.  rem * inv is within 2^-51 relative
// of the quotient, so its floor is off by at most one; the remainder fma(-q, a, rem) is exact (integers below 2^53) and repairs it.
HQB_HD int32_t fits(double rem, double a, double inv) {
    double q = floor(rem * inv);
    const double r = fma(-q, a, rem);
    if (r < 0.0) q -= 1.0;
    else if (r >= a) q += 1.0;
    return q > 2147483647.0 ? 2147483647 : (int32_t)q;
}
// C(i, p) for p <= 4, i <= NMAX + MMAX: closed form (the divisions are by constants)
HQB_HD uint32_t binom(uint32_t i, int p) {
    switch (p) {
        case 0: return 1u;
        case 1: return i;
        case 2: return i < 2 ? 0u : i * (i - 1) / 2;
        case 3: return i < 3 ? 0u : i * (i - 1) * (i - 2) / 6;
        default: return i < 4 ? 0u : i * (i - 1) * (i - 2) * (i - 3) / 24;
    }
}
HQB_HD int64_t gcd64(int64_t a, int64_t b) {  // binary gcd: shifts and subtractions only
    if (a == 0) return b;
    if (b == 0) return a;
    const int sh = __builtin_ctzll((unsigned long long)(a | b));
    a >>= __builtin_ctzll((unsigned long long)a);
    while (b) {
        b >>= __builtin_ctzll((unsigned long long)b);
        if (a > b) { const int64_t t = a; a = b; b = t; }
        b -= a;
    }
    return a << sh;
}

// min over the level's dual points of y . rem: an upper bound of the LP over the positions [0, k), hence of its integer optimum
template <class SH>
HQB_HD double lp_bound(const SH &S, int k, const double *rem) {
    const int cnt = (int)(S.dcnt[k] < (uint32_t)DPRE ? S.dcnt[k] : (uint32_t)DPRE);
    if (cnt == 0) return 1e300;
    const double r0 = rem[0], r1 = rem[1], r2 = rem[2], r3 = rem[3];
    double best = 1e300;
    for (int i = 0; i < cnt; i++) {
        const double *d = S.py[S.dl[k][i]];
        const double v = d[0] * r0 + d[1] * r1 + d[2] * r2 + d[3] * r3 + (double)S.dpen[k][i];
        best = v < best ? v : best;
    }
    return best;
}

// ---- step 1: the block of one class -----------------------------------------------------------------------------------------------------------
// The tables arrive in device-visible HOST memory (pinned): a dependent chain of loads from there costs a PCIe round trip each, so the wavefront
// first copies what it needs into LDS with one wide load per lane (the staging area overlays the dual pool, which is empty at this point), then
// lane g builds column g.
// The tables arrive in device-visible HOST memory (pinned): every dependent load from there is a PCIe round trip (~2 us), so the wavefront
// first copies the column table (one allocation, a few hundred bytes) and its class's free / total rows into LDS with one wide load per lane —
// the staging area overlays the dual pool, which is empty at this point — and lane g then builds column g from LDS.
template <class W, class SH>
HQB_HD void build_block(W &wv, SH &S, const ColTable &ct_in, const ClassTable &cl, uint32_t cls) {
    const uint32_t R = ct_in.R, NC = ct_in.n_cols;
    uint8_t *area = reinterpret_cast<uint8_t *>(&S.py[0][0]);
    // (the pool, and behind it the work problem's arrays up to `wcap`: nothing in there is written before dual_candidate / setup_work, which run after the block is built)
    static_assert(offsetof(SH, dl) - offsetof(SH, py) >= BLOB_MAX + 64 * 8 * 2, "the staging area overlays the dual pool and the work problem behind it, up to the level lists (whose storage holds a64 meanwhile)");
    uint64_t *sfree = reinterpret_cast<uint64_t *>(area + BLOB_MAX), *stotal = sfree + 64;
    // integer amounts while the rows are brought onto their own grid (gcd); overlays the greedy vectors, which are not in use yet
    static_assert(sizeof(uint16_t) * SH::NN * WAVE >= sizeof(int64_t) * (MMAX * SH::NN + MMAX), "a64 overlays gx");
    int64_t (*a64)[SH::NN] = reinterpret_cast<int64_t (*)[SH::NN]>(&S.gx[0][0]);
    int64_t *cap64 = &a64[MMAX - 1][SH::NN - 1] + 1;
    ColTable ct = ct_in;
    if (wv.first()) { S.status = ST_OK; S.steps = 0; S.steps_p1 = 0; S.n = 0; S.m = 0; S.npool = 0; S.usedres = 0; }
    wv.each([&](int lane) { if (lane < SH::NN) S.colcap[lane] = 2147483647; });
    if (NC > (uint32_t)GCOLS || R > 64 || NC == 0) { if (wv.first()) S.status = ST_UNSUPPORTED; wv.sync(); return; }
    const uint64_t elig = cl.elig[cls] & (NC >= 64 ? ~0ull : ((1ull << NC) - 1ull));
    const bool staged = ct_in.blob != nullptr && ct_in.blob_bytes <= (uint32_t)BLOB_MAX && (ct_in.blob_bytes & 15u) == 0;
    wv.each([&](int lane) {
        if (staged) {
            const V16 *src = reinterpret_cast<const V16 *>(ct_in.blob);
            V16 *dst = reinterpret_cast<V16 *>(area);
            for (uint32_t i = (uint32_t)lane; i < ct_in.blob_bytes / 16; i += WAVE) dst[i] = src[i];
        }
        for (uint32_t i = (uint32_t)lane; i < R; i += WAVE) { sfree[i] = cl.free_[(size_t)cls * R + i]; stotal[i] = cl.total[(size_t)cls * R + i]; }
        for (int i = lane; i < MMAX * SH::NN; i += WAVE) a64[i / SH::NN][i % SH::NN] = 0;
        if (lane < MMAX) cap64[lane] = 0;
    });
    wv.sync();
    if (staged) {  // the table pointers now point into LDS
        const uint8_t *base = reinterpret_cast<const uint8_t *>(ct_in.blob);
        auto re = [&](const void *p) { return area + (reinterpret_cast<const uint8_t *>(p) - base); };
        ct.ent_off = reinterpret_cast<const uint32_t *>(re(ct_in.ent_off)); ct.ent_res = reinterpret_cast<const uint32_t *>(re(ct_in.ent_res));
        ct.ent_kind = re(ct_in.ent_kind); ct.ent_amount = reinterpret_cast<const uint64_t *>(re(ct_in.ent_amount));
        ct.weight = reinterpret_cast<const uint32_t *>(re(ct_in.weight)); ct.pool = reinterpret_cast<const double *>(re(ct_in.pool));
    }
    const int n = __builtin_popcountll(elig);
    if (n > SH::NN) { if (wv.first()) S.status = ST_UNSUPPORTED; wv.sync(); return; }
    // column g by lane g: cost in the reference's operation order (create_sn_var, solver.rs:550-568), the resources it touches
    wv.each([&](int lane) {
        const uint32_t g = (uint32_t)lane;
        if (g >= NC || !((elig >> g) & 1)) return;
        double sc = 0.0; bool any = false, bad = false;
        for (uint32_t e = ct.ent_off[g]; e < ct.ent_off[g + 1]; e++) {
            const uint32_t r = ct.ent_res[e];
            const uint64_t amt = ct.ent_kind[e] ? stotal[r] : ct.ent_amount[e];
            const double pool = ct.pool[r];
            sc += pool < 0.000001 ? 0.0 : ((double)amt / 10000.0) / pool;
            if (sfree[r] == UINT64_MAX) bad = true;  // unbounded row: the reference's carry-over (solver.rs:183-185) is a host matter
            if (amt == 0) continue;
            if (amt >= (uint64_t)VAL_LIMIT || (sfree[r] >= (uint64_t)VAL_LIMIT)) bad = true;
            any = true;
            wv.atomic_or64(&S.usedres, 1ull << r);
        }
        if (!any) bad = true;  // a column no row bounds
        const int j = __builtin_popcountll(elig & ((1ull << g) - 1ull));
        S.c[j] = sc * ((double)ct.weight[g] / 10000.0);
        S.gcol[j] = (int)g;
        if (bad) S.status = ST_UNSUPPORTED;
    });
    wv.sync();
    const uint64_t used = S.usedres;
    const int m = __builtin_popcountll(used);
    if (S.status != ST_OK || m > MMAX) { if (wv.first()) S.status = ST_UNSUPPORTED; wv.sync(); return; }
    wv.each([&](int lane) {
        const uint32_t g = (uint32_t)lane;
        if (g < R && ((used >> g) & 1)) cap64[__builtin_popcountll(used & ((1ull << g) - 1ull))] = (int64_t)sfree[g];
        if (g >= NC || !((elig >> g) & 1)) return;
        const int j = __builtin_popcountll(elig & ((1ull << g) - 1ull));
        for (uint32_t e = ct.ent_off[g]; e < ct.ent_off[g + 1]; e++) {
            const uint32_t r = ct.ent_res[e];
            const uint64_t amt = ct.ent_kind[e] ? stotal[r] : ct.ent_amount[e];
            if (amt) a64[__builtin_popcountll(used & ((1ull << r) - 1ull))][j] += (int64_t)amt;
        }
    });
    wv.sync();
    // rows on their own grid: amounts and capacity divided by the row's gcd (the capacity rounds down: what is cut off no column can use); from
    // here on the integers live in f64
    wv.each([&](int lane) {
        if (lane >= MMAX) return;
        int64_t g = 0;
        if (lane < m) for (int j = 0; j < n; j++) g = gcd64(g, a64[lane][j]);
        if (g < 1) g = 1;
        for (int j = 0; j < SH::NN; j++) {
            const double v = (lane < m && j < n) ? (double)a64[lane][j] / (double)g : 0.0;  // exact: g divides the amount, both below 2^52
            S.a[lane][j] = v; S.ainv[lane][j] = v > 0.0 ? 1.0 / v : 0.0;
        }
        S.cap[lane] = lane < m ? (double)(cap64[lane] / g) : 0.0;
    });
    wv.sync();
    // search order: columns by ascending size (stable), by rank counting; a column that fits too often goes to the host
    wv.each([&](int lane) {
        if (lane >= n) return;
        int32_t ub = 2147483647; double sz = 0.0;
        for (int r = 0; r < m; r++) if (S.a[r][lane] > 0.0) { const int32_t q = fits(S.cap[r], S.a[r][lane], S.ainv[r][lane]); ub = q < ub ? q : ub; sz += S.a[r][lane] / (S.cap[r] + 1.0); }
        if (ub > UB_LIMIT) S.status = ST_UNSUPPORTED;
        S.lane_val[lane] = sz;
    });
    wv.sync();
    wv.each([&](int lane) {
        if (lane >= n) return;
        const double mine = S.lane_val[lane];
        int rank = 0;
        for (int i = 0; i < n; i++) { const double o = S.lane_val[i]; if (o < mine || (o == mine && i < lane)) rank++; }
        S.pi[rank] = (uint8_t)lane;
    });
    wv.sync();
    wv.each([&](int lane) {  // value density c_j / sum_r a_rj / cap_r
        if (lane >= n) return;
        double w = 0.0;
        for (int r = 0; r < m; r++) if (S.a[r][lane] > 0.0) w += S.cap[r] > 0.0 ? S.a[r][lane] / S.cap[r] : 1e30;
        S.lane_val[lane] = w > 0.0 ? S.c[lane] / w : 0.0;
    });
    wv.sync();
    wv.each([&](int lane) {
        if (lane >= n) return;
        const double mine = S.lane_val[lane];
        int rank = 0;
        for (int i = 0; i < n; i++) { const double o = S.lane_val[i]; if (o > mine || (o == mine && i < lane)) rank++; }
        S.pd[rank] = (uint8_t)lane;
    });
    if (wv.first()) { S.n = n; S.m = m; }
    wv.sync();
}

// ---- step 2: dual points ------------------------------------------------------------------------------------------------------------------------
// Basis number t of the C(n + m, m) choices of m tight constraints among {column j: a_j . y = c_j} and {row r: y_r = 0}.
template <class W, class SH>
HQB_HD void dual_candidate(W &wv, SH &S, uint32_t t) {
    const int n = S.n, m = S.m;
    double M[MMAX][MMAX + 1];
    HQB_UNROLL
    for (int i = 0; i < MMAX; i++) {
        HQB_UNROLL
        for (int q = 0; q <= MMAX; q++) M[i][q] = (i >= m && q == i) ? 1.0 : 0.0;  // padding rows: y_r = 0
    }
    uint32_t tight = 0;
    int maxcol = -1;
    {   // unrank: items come out in descending order
        uint32_t rest = t;
        int hi = n + m;
        HQB_UNROLL
        for (int eq = 0; eq < MMAX; eq++) {
            if (eq < m) {
                const int p = m - eq;
                int i = hi - 1;
                while (binom((uint32_t)i, p) > rest) i--;
                rest -= binom((uint32_t)i, p);
                hi = i;
                if (i < n) {
                    HQB_UNROLL
                    for (int r = 0; r < MMAX; r++) M[eq][r] = S.a[r][i];
                    M[eq][MMAX] = S.c[i];
                    tight |= 1u << i;
                    if (i > maxcol) maxcol = i;
                } else {
                    HQB_UNROLL
                    for (int r = 0; r < MMAX; r++) M[eq][r] = (r == i - n) ? 1.0 : 0.0;
                }
            }
        }
    }
    // Gauss-Jordan with partial pivoting, fixed 4 x 4 so that everything stays in registers
    HQB_UNROLL
    for (int col = 0; col < MMAX; col++) {
        int piv = col; double pv = M[col][col] < 0 ? -M[col][col] : M[col][col];
        HQB_UNROLL
        for (int i = col + 1; i < MMAX; i++) { double v = M[i][col] < 0 ? -M[i][col] : M[i][col]; if (v > pv) { pv = v; piv = i; } }
        if (!(pv > 1e-300)) return;  // singular
        HQB_UNROLL
        for (int i = col + 1; i < MMAX; i++) if (i == piv) {
            HQB_UNROLL
            for (int q = 0; q <= MMAX; q++) { double tmp = M[col][q]; M[col][q] = M[i][q]; M[i][q] = tmp; }
        }
        const double inv = 1.0 / M[col][col];
        HQB_UNROLL
        for (int q = 0; q <= MMAX; q++) M[col][q] *= inv;
        HQB_UNROLL
        for (int i = 0; i < MMAX; i++) if (i != col) {
            const double f = M[i][col];
            HQB_UNROLL
            for (int q = 0; q <= MMAX; q++) M[i][q] -= f * M[col][q];
        }
    }
    double y[MMAX], ymax = 0.0;
    HQB_UNROLL
    for (int r = 0; r < MMAX; r++) { y[r] = M[r][MMAX]; if (!(y[r] == y[r]) || y[r] > 1e280 || y[r] < -1e280) return; double v = y[r] < 0 ? -y[r] : y[r]; ymax = v > ymax ? v : ymax; }
    HQB_UNROLL
    for (int r = 0; r < MMAX; r++) { if (y[r] < -1e-9 * ymax) return; if (y[r] < 0.0) y[r] = 0.0; }
    // the columns the point covers, and the factor that makes the cover exact in floating point (tight columns come out equal up to rounding)
    uint32_t cover = 0; double f = 1.0;
    for (int j = 0; j < n; j++) {
        double s = 0.0;
        HQB_UNROLL
        for (int r = 0; r < MMAX; r++) s += S.a[r][j] * y[r];
        if (s < S.c[j] * (1.0 - 1e-11)) continue;
        cover |= 1u << j;
        if (s < S.c[j]) { const double q = S.c[j] / s; f = q > f ? q : f; }
    }
    if ((tight & ~cover) != 0) return;
    // Useful for some walk?  The walks visit the column sets {first k columns of pi that are < j} (j = n: phase 1).  The point is a vertex for such a
    // set F iff tight within F within cover; the most permissive j is maxcol + 1, the smallest k the one that reaches the last tight column of pi.
    {
        const uint32_t low = maxcol + 1 >= 32 ? 0xFFFFFFFFu : ((1u << (maxcol + 1)) - 1u);
        uint32_t pref = 0, left = tight;
        for (int i = 0; i < n && left; i++) { const uint32_t b = 1u << S.pi[i]; pref |= b; left &= ~b; }
        if (maxcol < 0) pref = 0;
        if ((pref & low & ~cover) != 0) return;
    }
    f *= 1.0 + 1e-15;
    const uint32_t slot = wv.atomic_inc(&S.npool);
    if (slot < (uint32_t)PCAP) {
        double yf[MMAX];
        HQB_UNROLL
        for (int r = 0; r < MMAX; r++) { yf[r] = y[r] * f; S.py[slot][r] = yf[r]; }
        S.pcover[slot] = cover; S.ptight[slot] = tight;
        S.pkey[slot] = (float)(yf[0] * S.cap[0] + yf[1] * S.cap[1] + yf[2] * S.cap[2] + yf[3] * S.cap[3]);
        S.ptix[slot] = (uint16_t)t;   // (C(NMAX + MMAX, MMAX) = 58 905 basis numbers at most)
    }
}

// ---- step 3: greedy incumbents -----------------------------------------------------------------------------------------------------------------
HQB_HD uint32_t xorshift32(uint32_t &s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }

template <class SH>
HQB_HD void greedy_lane(SH &S, int lane) {
    const int n = S.n, m = S.m;
    // lane 0: by value density (descending); 1: large requests first; 2: small first; 3 / 4: the model's order and its reverse; others: shuffles
    for (int j = 0; j < n; j++) S.perm[j][lane] = lane == 0 ? S.pd[j] : lane == 1 ? S.pi[n - 1 - j] : lane == 2 ? S.pi[j] : lane == 4 ? (uint8_t)(n - 1 - j) : (uint8_t)j;
    if (lane >= 5) { uint32_t s = 0x9E3779B9u * (uint32_t)(lane + 1); for (int i = n - 1; i > 0; i--) { int q = (int)(xorshift32(s) % (uint32_t)(i + 1)); uint8_t tmp = S.perm[i][lane]; S.perm[i][lane] = S.perm[q][lane]; S.perm[q][lane] = tmp; } }
    double rem[MMAX];
    for (int r = 0; r < MMAX; r++) rem[r] = S.cap[r];
    for (int j = 0; j < n; j++) S.gx[j][lane] = 0;
    for (int i = 0; i < n; i++) {
        const int j = S.perm[i][lane];
        if (!(S.c[j] > 0.0)) continue;
        int32_t ub = S.colcap[j];
        for (int r = 0; r < m; r++) if (S.a[r][j] > 0.0) { const int32_t q = fits(rem[r], S.a[r][j], S.ainv[r][j]); ub = q < ub ? q : ub; }
        if (ub <= 0) continue;
        S.gx[j][lane] = (uint16_t)ub;
        for (int r = 0; r < m; r++) rem[r] -= (double)ub * S.a[r][j];
    }
    double z = 0.0;
    for (int j = n - 1; j >= 0; j--) z = z + S.c[j] * (double)S.gx[j][lane];
    S.gval[lane] = z;
}

// ---- steps 2 + 3 as a section of the block that EVERY wavefront of its workgroup takes part in ---------------------------------------------------
// A block is one dependent chain (build -> duals -> greedy -> level lists -> walk), and with one wavefront per block the chain's longest links are the pool — C(n + m, m)
// 4 x 4 eliminations through 64 lanes: 8 rounds on an 8-column block, 76 on a 16-column one (16 of a c3p block's 44 us, 41-82 of a configs[3] block's) — and the 64
// greedy fills behind it (8-11 us).  Neither needs the other, and the candidates are independent of one another, so a block may bring W::WAVES wavefronts (its
// workgroup: one per SIMD of the CU for four): the MAIN wavefront (index 0) runs the chain, the others wait at group barrier A, take part in this section and leave
// (pool_helper).  The LAST wavefront runs the greedy fills and raises a flag; the others ("pool wavefronts", the main one among them) take the basis numbers in
// chunks of 64 from a counter in LDS, then order the pool together.  The pool wavefronts meet at barriers of their own (a counter in LDS: the workgroup's s_barrier
// would make them wait for the greedy fills); the main wavefront waits for the flag when it leaves the section.
// Who finds which entry first is then a matter of timing, and nothing may depend on it: the pool's ORDER (porder) is by (key, basis number), a total order, and a pool
// that overflows — where the slot an entry gets decides whether it is kept at all — is thrown away and built again by the main wavefront alone in the one-wavefront
// order (never seen on 68 698 measured blocks; it keeps the answers of a block a function of the block).
template <class W, class SH>
HQB_HD void pool_sections(W &wv, SH &S) {
    const int n = S.n, m = S.m;
    const uint32_t total = binom((uint32_t)(n + m), m);
    auto all_by_lane = [&]() { wv.each([&](int lane) { for (uint32_t t = (uint32_t)lane; t < total; t += WAVE) dual_candidate(wv, S, t); }); };
    uint32_t phase = 0;  // barriers of the pool wavefronts passed so far
    if (W::WAVES == 1) {
        all_by_lane();
        wv.each([&](int lane) { greedy_lane(S, lane); });
        wv.sync();
    } else {
        if (wv.wave_index() == W::WAVES - 1) {  // the greedy wavefront
            wv.each([&](int lane) { greedy_lane(S, lane); });
            wv.raise(&S.gdone);
            return;
        }
        for (;;) {
            const uint32_t base = wv.grab(&S.cnext, (uint32_t)WAVE);
            if (base >= total) break;
            wv.each([&](int lane) { const uint32_t t = base + (uint32_t)lane; if (t < total) dual_candidate(wv, S, t); });
        }
        wv.pool_barrier(&S.bar, ++phase);  // the pool is complete
        if (S.npool > S.redo_at) {
            wv.pool_barrier(&S.bar, ++phase);  // (everyone has looked at npool)
            if (wv.wave_index() == 0) { if (wv.first()) S.npool = 0; wv.sync(); all_by_lane(); }
            wv.pool_barrier(&S.bar, ++phase);
        }
    }
    // pool order: ascending y . cap, equal keys by basis number (rank counting)
    const uint32_t np = S.npool < (uint32_t)PCAP ? S.npool : (uint32_t)PCAP;
    wv.each_of_pool([&](int tid, int nthreads) {
        for (uint32_t i = (uint32_t)tid; i < np; i += (uint32_t)nthreads) {
            const float mine = S.pkey[i]; const uint16_t mt = S.ptix[i];
            uint32_t rank = 0;
            for (uint32_t j = 0; j < np; j++) { const float o = S.pkey[j]; rank += (o < mine || (o == mine && S.ptix[j] < mt)) ? 1u : 0u; }
            S.porder[rank] = (uint16_t)i;
        }
    });
    wv.pool_barrier(&S.bar, ++phase);  // the pool is ordered: the helpers among the pool wavefronts are done
}
// the main wavefront: open the section (group barrier A) and take part; the pool is complete and ordered when it returns, the greedy fills may still be running ...
template <class W, class SH>
HQB_HD void pool_main(W &wv, SH &S, uint32_t redo_at = (uint32_t)PCAP) {
    if (wv.first()) { S.cnext = 0; S.bar = 0; S.gdone = 0; S.redo_at = redo_at; }
    wv.group_sync();
    pool_sections(wv, S);
}
// ... until here: the first incumbent = the best of the 64 fills (the main wavefront builds the first level lists in between: setup_work needs the pool, not the fills)
template <class W, class SH>
HQB_HD void pool_await(W &wv, SH &S) {
    wv.await(&S.gdone);
    int l = 0;
    const double top = wv.argmax([&](int lane) { return S.gval[lane]; }, &l);
    if (wv.first()) { S.best = top; for (int j = 0; j < S.n; j++) S.xbest[j] = S.gx[j][l]; }
    wv.sync();
}
// ... which the main wavefront also passes when there is no block to work on (S.n == 0 or a status other than ST_OK), so that the helpers are released
template <class W>
HQB_HD void pool_skip(W &wv) { wv.group_sync(); }
// every other wavefront of the workgroup: this is all it does
template <class W, class SH>
HQB_HD void pool_helper(W &wv, SH &S) {
    wv.group_sync();  // A: the block is built (n, m, a, c, cap, pi, pd, colcap)
    if (S.status != ST_OK || S.n == 0) return;
    pool_sections(wv, S);
}

// ---- the work problem of one walk ---------------------------------------------------------------------------------------------------------------
// Columns: the ones of `cols` (mask of block columns) in the order of pi.  `capcol` (or -1) is a column whose value is capped at `capval`
// (phase 2's probes).  Level lists: a pool point bounds the level with free set F when it is a vertex of F's dual polyhedron (tight within F
// within cover).  With a capped column j in F the LP has one more dual variable, the multiplier of x_j <= L; its vertices are the ones above
// plus (y, mu = c_j - a_j . y > 0) with y a vertex for F \ {j}: those enter with the penalty L * mu   (bound = y . rem + penalty).
template <class W, class SH>
HQB_HD void setup_work(W &wv, SH &S, uint32_t cols, int capcol, int32_t capval) {
    const uint64_t selmask = wv.ballot([&](int lane) { return lane < S.n && ((cols >> S.pi[lane]) & 1); });
    wv.each([&](int lane) {
        if (lane >= S.n || !((selmask >> lane) & 1)) return;
        const int p = __builtin_popcountll(selmask & ((1ull << lane) - 1ull)), j = S.pi[lane];
        S.wcol[p] = (uint8_t)j;
        S.wc[p] = S.c[j];
        S.wcap[p] = (j == capcol && capval < S.colcap[j]) ? capval : S.colcap[j];
        for (int r = 0; r < MMAX; r++) { S.wa[r][p] = S.a[r][j]; S.winv[r][p] = S.ainv[r][j]; }
    });
    if (wv.first()) S.wn = __builtin_popcountll(selmask);
    wv.sync();
    wv.each([&](int lane) {
        if (lane > S.wn) return;
        uint32_t mask = 0;
        for (int q = 0; q < lane; q++) mask |= 1u << S.wcol[q];
        S.wmask[lane] = mask;
    });
    wv.sync();
    const int wn = S.wn;
    const uint32_t np = S.npool < (uint32_t)PCAP ? S.npool : (uint32_t)PCAP;
    const uint32_t capbit = capcol >= 0 ? 1u << capcol : 0u;
    // Level lists in pool order (porder: tightest at the root first).  The level sets are nested (F_1 < F_2 < ... ), so an entry is a vertex for
    // a contiguous range of levels — [lo, hi] as it is, [lo2, hi2] through the capped column's multiplier — which its lane works out once; one
    // ballot per level then gives every entry its slot (prefix popcount), with nothing but registers between the ballots.
    if (wv.first()) for (int k = 0; k <= SH::NN; k++) S.dcnt[k] = 0;
    wv.sync();
    for (uint32_t base = 0; base < np; base += WAVE) {
        wv.each([&](int lane) {
            const uint32_t rk = base + (uint32_t)lane;
            uint32_t rng = 0x00FF00FFu;  // lo = 255 > hi = 0: empty ranges
            double mu = 0.0;
            if (rk < np) {
                const uint32_t i = S.porder[rk], cover = S.pcover[i], tight = S.ptight[i];
                int lo = 255, hi = 0, lo2 = 255, hi2 = 0;
                for (int k = 1; k <= wn; k++) {
                    const uint32_t F = S.wmask[k];
                    if ((tight & ~F) == 0 && (F & ~cover) == 0) { if (k < lo) lo = k; hi = k; }
                    else if (F & capbit) { const uint32_t G = F & ~capbit; if ((tight & ~G) == 0 && (G & ~cover) == 0) { if (k < lo2) lo2 = k; hi2 = k; } }
                }
                if (hi2) {
                    double sdot = 0.0;
                    for (int r = 0; r < MMAX; r++) sdot += S.a[r][capcol] * S.py[i][r];
                    mu = (S.c[capcol] - sdot) * (1.0 + 1e-12);
                    if (mu < 0.0) mu = 0.0;
                }
                rng = (uint32_t)lo | ((uint32_t)hi << 8) | ((uint32_t)lo2 << 16) | ((uint32_t)hi2 << 24);
            }
            S.lane_rng[lane] = rng; S.lane_val[lane] = mu;
        });
        wv.sync();
        for (int k = 1; k <= wn; k++) {
            const uint64_t mask = wv.ballot([&](int lane) {
                const uint32_t rng = S.lane_rng[lane];
                const uint32_t kk = (uint32_t)k;
                return ((rng & 0xFFu) <= kk && kk <= ((rng >> 8) & 0xFFu)) || (((rng >> 16) & 0xFFu) <= kk && kk <= (rng >> 24));
            });
            if (wv.first()) S.lmask[k] = mask;
        }
        wv.sync();
        wv.each([&](int lane) {
            const uint32_t rng = S.lane_rng[lane];
            if (rng == 0x00FF00FFu) return;
            const uint16_t i = S.porder[base + (uint32_t)lane];
            const uint32_t lo = rng & 0xFFu, hi = (rng >> 8) & 0xFFu, lo2 = (rng >> 16) & 0xFFu, hi2 = rng >> 24;
            const double pen = (double)capval * S.lane_val[lane];
            const float penf = pen > 0.0 ? (float)(pen * (1.0 + 2e-7)) : 0.0f;
            for (uint32_t k = 1; k <= (uint32_t)wn; k++) {
                const bool plain = lo <= k && k <= hi, capped = lo2 <= k && k <= hi2;
                if (!plain && !capped) continue;
                const uint32_t slot = S.dcnt[k] + (uint32_t)__builtin_popcountll(S.lmask[k] & ((1ull << lane) - 1ull));
                if (slot >= (uint32_t)DPRE) continue;
                S.dl[k][slot] = i;
                S.dpen[k][slot] = plain ? 0.0f : penf;
            }
        });
        wv.sync();
        if (wv.first()) for (int k = 1; k <= wn; k++) S.dcnt[k] += (uint32_t)__builtin_popcountll(S.lmask[k]);
        wv.sync();
    }
}

template <class SH>
HQB_HD int32_t level_ub(const SH &S, int k) {  // how often the column at position k - 1 fits into rem[k]
    const int p = k - 1;
    int32_t ub = S.wcap[p];
    for (int r = 0; r < S.m; r++) if (S.wa[r][p] > 0.0) { const int32_t q = fits(S.rem[k][r], S.wa[r][p], S.winv[r][p]); ub = q < ub ? q : ub; }
    return ub;
}

// Two positions left (1 and 0): position 1 takes v1, position 0 follows exactly with its maximum.
template <class SH>
HQB_HD bool terminal_lane(const SH &S, int32_t v1, int32_t *x0_out, double *val_out) {
    if (v1 < 0 || v1 > S.ub[2]) return false;
    int32_t x0max = S.wcap[0];
    for (int r = 0; r < S.m; r++) {
        const double left = fma(-(double)v1, S.wa[r][1], S.rem[2][r]);
        if (S.wa[r][0] > 0.0) { const int32_t q = left < 0.0 ? -1 : fits(left, S.wa[r][0], S.winv[r][0]); x0max = q < x0max ? q : x0max; }
    }
    *x0_out = x0max;
    *val_out = (S.zfix[2] + S.wc[1] * (double)v1) + S.wc[0] * (double)x0max;
    return true;
}

enum { MODE_MAX = 0, MODE_FIND = 1 };
struct Probe { double rem[MMAX]; double base, b; };  // one child of a level while its bound is being evaluated
#ifdef HQB_TRACE
static uint32_t g_trace_maxlist = 0, g_trace_maxpool = 0; static unsigned long g_trace_probes = 0;
#endif

// One walk over the work problem from the state the caller put into rem[wn] / zfix[wn]; every level takes its largest values first.
//   MODE_MAX   maximise into S.best / S.xbest.
//   MODE_FIND  stop at the first complete point whose objective is >= thr and write it into S.xbest (work columns only); *found_out tells
//              whether there was one.
// Returns false when the step budget ran out.
template <class W, class SH>
HQB_HD bool walk(W &wv, SH &S, int mode, double thr, uint32_t *budget, bool *found_out) {
    const int wn = S.wn, m = S.m;
    bool found = false;
    if (wn == 1) {  // a single position: no search
        if (wv.first()) {
            int32_t ub = S.wcap[0];
            for (int r = 0; r < m; r++) if (S.wa[r][0] > 0.0) { const int32_t q = fits(S.rem[1][r], S.wa[r][0], S.winv[r][0]); ub = q < ub ? q : ub; }
            const double val = S.zfix[1] + S.wc[0] * (double)ub;
            if (mode == MODE_MAX) { if (val > S.best) { S.best = val; S.xbest[S.wcol[0]] = (uint32_t)ub; } }
            else { S.ptr[1] = val >= thr ? 1 : 0; if (val >= thr) S.xbest[S.wcol[0]] = (uint32_t)ub; }
        }
        wv.sync();
        if (mode == MODE_FIND) found = S.ptr[1] != 0;
        if (found_out) *found_out = found;
        return true;
    }
    if (wv.first()) { S.ub[wn] = level_ub(S, wn); S.ptr[wn] = S.ub[wn]; }
    wv.sync();
    int k = wn;
    uint32_t steps = 0;
    bool in_budget = true;
    // The state of the CURRENT level lives in registers (it is the same in every lane): next value to try, upper bound, what is left of the rows, the fixed part of the
    // objective — and the incumbent's value.  A descent works the next level's state out in every lane from those registers; lane 0 also writes it to the level stack
    // in LDS, which is read again only when the walk comes BACK to a level (until round 6 every step began by reading its level from LDS, behind the barrier that
    // followed lane 0's writes: two or three LDS round trips of a step's seven).
    int32_t p = 0, ubk = 0; double remk[MMAX], zk = 0.0;
    auto load_level = [&](int kk) { p = S.ptr[kk]; ubk = S.ub[kk]; for (int r = 0; r < MMAX; r++) remk[r] = S.rem[kk][r]; zk = S.zfix[kk]; };
    load_level(k);
    double best = S.best;
    while (k <= wn) {
        if (steps >= *budget) { in_budget = false; break; }
        steps++;
        if (p < 0) { k++; if (k <= wn) load_level(k); continue; }  // level exhausted
        const int pos = k - 1;
        if (k == 2) {  // leaves: every lane a complete point
            auto eval = [&](int lane, int32_t *x0, double *val) { return terminal_lane(S, p - lane, x0, val); };
            if (mode == MODE_FIND) {
                const uint64_t mask = wv.ballot([&](int lane) { int32_t x0; double val; return eval(lane, &x0, &val) && val >= thr; });
                if (mask) {
                    const int l = wv.ctz(mask);
                    if (wv.first()) {
                        int32_t x0 = 0; double val = 0; eval(l, &x0, &val);
                        S.xsel[1] = (uint32_t)(p - l); S.xsel[0] = (uint32_t)x0;
                        for (int q = 0; q < wn; q++) S.xbest[S.wcol[q]] = S.xsel[q];
                    }
                    wv.sync();
                    found = true;
                    break;
                }
            } else {
                int l = -1;
                const double top = wv.argmax([&](int lane) { int32_t x0; double val; return eval(lane, &x0, &val) ? val : -1.0; }, &l);
                if (l >= 0 && top > best) {
                    if (wv.first()) {
                        int32_t x0 = 0; double val = 0; eval(l, &x0, &val);
                        S.xsel[1] = (uint32_t)(p - l); S.xsel[0] = (uint32_t)x0;
                        for (int q = 0; q < wn; q++) S.xbest[S.wcol[q]] = S.xsel[q];
                        S.best = val;
                    }
                    best = top;   // (= val: the same lane's evaluation)
                    wv.sync();
                }
            }
            p -= WAVE;
            continue;
        }
        // inner level: 64 values of the column at `pos` at once
        const double cut = mode == MODE_FIND ? thr : best + 1e-12 * (best < 0 ? -best : best);
        const double cj = S.wc[pos];
        double waj[MMAX];
        for (int r = 0; r < MMAX; r++) waj[r] = S.wa[r][pos];
        // child bounds: child v survives when fixed part + min over the level's duals of (y . rem(v) + penalty) clears the cut.  The minimum is taken over ALL of the
        // level's duals whatever the order they are looked at in (wv.ballot_bound: 64 children x chunks of 8 duals, tightest first, stopping once every child is
        // pruned): the surviving set does not depend on it.
        const int cnt = (int)(S.dcnt[k - 1] < (uint32_t)DPRE ? S.dcnt[k - 1] : (uint32_t)DPRE);
        const uint64_t mask = wv.ballot_bound(cnt, p + 1 < WAVE ? p + 1 : WAVE,
            [&](int child, Probe &st) {
                const int32_t v = p - child;
                if (v < 0 || v > ubk) return false;
                for (int r = 0; r < MMAX; r++) st.rem[r] = fma(-(double)v, waj[r], remk[r]);
                st.base = zk + cj * (double)v; st.b = 1e300;
                return true;
            },
            [&](const Probe &st, int i0, int i1) {   // min over the duals [i0, i1)
                double b = 1e300;
                for (int i = i0; i < i1; i++) {
                    const double *d = S.py[S.dl[k - 1][i]];
                    const double val = d[0] * st.rem[0] + d[1] * st.rem[1] + d[2] * st.rem[2] + d[3] * st.rem[3] + (double)S.dpen[k - 1][i];
                    b = val < b ? val : b;
                }
                return b;
            },
            [&](const Probe &st, double b) {
                const double bound = st.base + b;
                return mode == MODE_FIND ? bound >= cut : bound > cut;
            });
        if (!mask) { p -= WAVE; continue; }
        const int l = wv.ctz(mask);
        const int32_t v = p - l;
        // descend: the next level's state, in every lane
        double nrem[MMAX];
        for (int r = 0; r < MMAX; r++) nrem[r] = fma(-(double)v, waj[r], remk[r]);
        const double nz = zk + cj * (double)v;
        int32_t nub = S.wcap[pos - 1];
        for (int r = 0; r < m; r++) if (S.wa[r][pos - 1] > 0.0) { const int32_t q = fits(nrem[r], S.wa[r][pos - 1], S.winv[r][pos - 1]); nub = q < nub ? q : nub; }
        if (wv.first()) {
            S.xsel[pos] = (uint32_t)v;
            S.ptr[k] = v - 1;   // (where this level goes on when the walk comes back)
            for (int r = 0; r < MMAX; r++) S.rem[k - 1][r] = nrem[r];
            S.zfix[k - 1] = nz;
            S.ub[k - 1] = nub;
            S.ptr[k - 1] = nub;
        }
        wv.sync();
        k--;
        p = nub; ubk = nub; zk = nz;
        for (int r = 0; r < MMAX; r++) remk[r] = nrem[r];
    }
    *budget -= steps < *budget ? steps : *budget;
    if (wv.first()) S.steps += steps;
    wv.sync();
    if (found_out) *found_out = found;
    return in_budget;
}

// ---- the whole block ----------------------------------------------------------------------------------------------------------------------------
template <class W, class SH>
HQB_HD void solve_block(W &wv, SH &S, const ColTable &ct, const ClassTable &cl, uint32_t cls, const Output &out, uint32_t budget) {
    uint64_t *prof = out.prof ? out.prof + (size_t)cls * 8 : nullptr;
    if (prof && wv.first()) prof[0] = wv.now();
    build_block(wv, S, ct, cl, cls);
    if (prof && wv.first()) prof[1] = wv.now();
    uint32_t *x = out.x + (size_t)cls * ct.n_cols;
    wv.each([&](int lane) { for (uint32_t g = lane; g < ct.n_cols; g += WAVE) x[g] = 0; });
    if (S.status != ST_OK || S.n == 0) {
        pool_skip(wv);
        if (wv.first()) { out.status[cls] = (uint32_t)S.status; out.steps[cls] = 0; }
        return;
    }
    const int n = S.n;
    pool_main(wv, S);  // dual pool (ordered) and greedy fills
    if (prof && wv.first()) prof[2] = wv.now();
    if (prof && wv.first()) prof[3] = wv.now();
    const uint32_t all = n >= 32 ? 0xFFFFFFFFu : ((1u << n) - 1u);
    uint32_t left = budget;
    bool ok = true;
    // phase 1 unless the incumbent already meets the root bound
    setup_work(wv, S, all, -1, 0);
    pool_await(wv, S);
    {
        double cap[MMAX];
        for (int r = 0; r < MMAX; r++) cap[r] = S.cap[r];
        const double root = lp_bound(S, n, cap);
        if (!(root <= S.best + 1e-12 * S.best)) {
            if (wv.first()) { for (int r = 0; r < MMAX; r++) S.rem[n][r] = S.cap[r]; S.zfix[n] = 0.0; }
            wv.sync();
            ok = walk(wv, S, MODE_MAX, 0.0, &left, nullptr);
        }
        if (wv.first()) S.steps_p1 = S.steps;
        wv.sync();
        if (prof && wv.first()) prof[4] = wv.now();
    }
    // phase 2: S.xbest is a point with objective >= thr throughout; column by column from the last one its value is pushed down by probes
    // "is there a point with x_j <= mid (the later columns fixed) that still reaches thr" — first just below the current value (most columns
    // fail that at once), then by bisection, as csrc/milp.cpp does
    if (ok) {
        const double thr = S.best - 1e-9 * (S.best < 0 ? -S.best : S.best);
        double capf[MMAX];
        for (int r = 0; r < MMAX; r++) capf[r] = S.cap[r];
        double zf = 0.0;
        for (int j = n - 1; j >= 0 && ok; j--) {
            int32_t lo = 0, hi = (int32_t)S.xbest[j];
            bool first = true;
            while (lo < hi && ok) {
                const int32_t mid = first ? hi - 1 : (lo + hi) / 2;
                first = false;
                {   // Root bound of the probe before its level lists are built (setup_work is ~5 us of a ~7 us probe, and most probes fail): a pool point that
                    // is a dual vertex for the columns 0..j — as it is, or through the multiplier mu = c_j - a_j . y of the cap x_j <= mid — bounds the probe's
                    // LP by y . cap + mu * mid.  If even the best of them stays below the threshold no point with x_j <= mid reaches it.
                    const uint32_t F = j >= 31 ? all : (all & ((2u << j) - 1u)), jb = 1u << j, G = F & ~jb;
                    const uint32_t np = S.npool < (uint32_t)PCAP ? S.npool : (uint32_t)PCAP;
                    int lbest = -1;
                    wv.sync();
                    wv.argmax([&](int lane) {
                        double best = 1e300;
                        for (uint32_t i = (uint32_t)lane; i < np; i += WAVE) {
                            const uint32_t cover = S.pcover[i], tight = S.ptight[i];
                            double pen;
                            if ((tight & ~F) == 0 && (F & ~cover) == 0) pen = 0.0;
                            else if ((tight & ~G) == 0 && (G & ~cover) == 0) {
                                double sdot = 0.0;
                                for (int r = 0; r < MMAX; r++) sdot += S.a[r][j] * S.py[i][r];
                                const double mu = (S.c[j] - sdot) * (1.0 + 1e-12);
                                pen = mu > 0.0 ? mu * (double)mid * (1.0 + 2e-7) : 0.0;
                            } else continue;
                            const double v = S.py[i][0] * capf[0] + S.py[i][1] * capf[1] + S.py[i][2] * capf[2] + S.py[i][3] * capf[3] + pen;
                            best = v < best ? v : best;
                        }
                        S.lane_val[lane] = best;
                        return best < 1e299 ? 1.0 / (1.0 + (best > 0.0 ? best : 0.0)) : -1.0;  // (decreasing in the bound: the arg-max is the lane with the smallest one; -1: no point applies)
                    }, &lbest);
                    wv.sync();
                    const bool fails = lbest >= 0 && zf + S.lane_val[lbest] < thr;
                    wv.sync();
                    if (fails) { lo = mid + 1; continue; }
                }
                setup_work(wv, S, j >= 31 ? all : (all & ((2u << j) - 1u)), j, mid);
                if (wv.first()) { for (int r = 0; r < MMAX; r++) S.rem[S.wn][r] = capf[r]; S.zfix[S.wn] = zf; }
                wv.sync();
                bool found = false;
                ok = walk(wv, S, MODE_FIND, thr, &left, &found);
                wv.sync();
#ifdef HQB_TRACE
                { uint32_t mx = 0; for (int k = 1; k <= S.wn; k++) mx = S.dcnt[k] > mx ? S.dcnt[k] : mx; if (mx > g_trace_maxlist) g_trace_maxlist = mx; if (S.npool > g_trace_maxpool) g_trace_maxpool = S.npool; g_trace_probes++; }
#endif
                if (found) hi = (int32_t)S.xbest[j]; else lo = mid + 1;
            }
            const uint32_t xj = S.xbest[j];
            for (int r = 0; r < MMAX; r++) capf[r] -= (double)xj * S.a[r][j];
            zf = zf + S.c[j] * (double)xj;
        }
    }
    wv.sync();
    if (prof && wv.first()) { prof[5] = wv.now(); prof[6] = S.steps_p1; prof[7] = S.npool; }
    if (wv.first()) {
        out.status[cls] = ok ? (uint32_t)ST_OK : (uint32_t)ST_BUDGET;
        out.steps[cls] = S.steps;
        if (ok) for (int j = 0; j < n; j++) x[S.gcol[j]] = S.xbest[j];
    }
}

// ---- host emulation of a wavefront (CPU tests; also what documents the contract of the Wave policy) ---------------------------------------------
struct HostWave {
    static constexpr int WAVES = 1;   // wavefronts of a block's workgroup (pool_sections); the emulation is one
    int wave_index() const { return 0; }
    void group_sync() {}
    void pool_barrier(uint32_t *, uint32_t) {}
    void raise(uint32_t *p) { *p = 1; }
    void await(uint32_t *) {}
    template <class T> void put(T *p, T v) { *p = v; }   // a result other workgroups of the launch read (device: a store that is visible device-wide once acknowledged)
    uint32_t grab(uint32_t *p, uint32_t n) { const uint32_t v = *p; *p += n; return v; }
    template <class F> void each_of_pool(F f) { for (int l = 0; l < WAVE; l++) f(l, WAVE); }
    uint64_t now() const { return 0; }
    bool first() const { return true; }
    void sync() {}
    uint32_t atomic_inc(uint32_t *p) { return (*p)++; }
    void atomic_or64(uint64_t *p, uint64_t v) { *p |= v; }
    void atomic_add_i64(long long *p, long long v) { *p += v; }
    void lds_add_i64(long long *p, long long v) { *p += v; }
    static int ctz(uint64_t m) { int i = 0; while (!((m >> i) & 1)) i++; return i; }
    template <class F> void each(F f) { for (int l = 0; l < WAVE; l++) f(l); }
    template <class F> uint64_t ballot(F f) { uint64_t m = 0; for (int l = 0; l < WAVE; l++) if (f(l)) m |= 1ull << l; return m; }
    // children (lanes < nchild) whose init() holds and whose decide() holds for the minimum of part() over all cnt duals
    template <class I, class P, class D> uint64_t ballot_bound(int cnt, int nchild, I init, P part, D decide) {
        uint64_t m = 0;
        for (int l = 0; l < nchild; l++) {
            Probe st;
            if (!init(l, st)) continue;
            if (decide(st, part(st, 0, cnt))) m |= 1ull << l;
        }
        return m;
    }
    template <class F> double argmax(F f, int *lane) {  // largest value, lowest lane among equals; *lane = -1 when every value is negative
        double best = -1.0; int bl = -1;
        for (int l = 0; l < WAVE; l++) { double v = f(l); if (v > best) { best = v; bl = l; } }
        *lane = bl;
        return best;
    }
};

}  // namespace hqblock
