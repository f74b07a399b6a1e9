// The coupled placement solve by price sweeps: host side (see price.h for the method, price_core.h for what one sweep does per block).
#include "price.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <numeric>

#include "lp_tab.h"
#include "milp.h"

namespace hqprice {
namespace {

using hqmilp::lp::BASIC;
using hqmilp::lp::INF;
using hqmilp::lp::LP_OPT;
using hqmilp::lp::Rows;
using hqmilp::lp::Tab;

constexpr int KMAX_HOST = 128;   // == price_core.h's KMAX (not included here: this file is host-only and also part of libhqalloc.so)
constexpr int NMAX_BLOCK = 32, MMAX_BLOCK = 4;
constexpr double FIRST_PROPOSAL_SCALE = 1.0 / 16.0;   // kelley(): what is evaluated of the first master's proposal (see there)
constexpr double GRID = 10000.0;  // ResourceAmount fractions per unit  common/resources/amount.rs:7
constexpr int MAX_ROUNDS = 6;
constexpr int BP_MAX_NODES = 600;   // nodes of the branch-and-price phase (a deterministic count, like every limit in here)
// (per second of the caller's configured time limit: generous — what keeps the phase inside the limit is its wall-clock guard at 70 % of it)
constexpr double BP_MAX_STEPS = 3.0e6;   // ... and search steps of the slowest blocks summed over the sweeps (0.3-1 us each: see Solver::sweep_steps)
constexpr double BP_MAX_WORK = 3.0e9; // ... and tableau elements its masters may touch (lp_tab.h's `ops`: ~1e9 per second): with thousands of cuts in the master a node
                                      // costs milliseconds, and a 10 k-column model spent 8 s here against a 5 s time limit before this cap (deterministic like the counts)

double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct CapRow { int flat; double rhs; std::vector<std::pair<int, double>> g; };  // x_flat + sum_g coef B_g <= rhs (the column's own coefficient divided out)

struct Prob {
    HostTables T;
    std::vector<int> flat_of, model_of;       // model column <-> flat column (-1: global)
    std::vector<int> gmodel;                   // global columns (model index)
    std::vector<double> gcost;
    int K = 0, G = 0;
    std::vector<double> h;                     // wide rows in <= form
    std::vector<uint8_t> ge;                   // the row was a `>=` row (entered negated)
    std::vector<std::vector<std::pair<int, double>>> g_rows;   // per global column: (wide row, coefficient in <= form)
    // Wide rows with the same left-hand side over the block columns (the cut rows of one batch against its several blockers differ in their flag only) share
    // ONE activity: the device prices and accumulates per GROUP (T.K groups, price of a group = sum of its rows' prices), and the host keeps the left-hand
    // sides per group as well — row-wise here, column-wise in T.col_woff / T.w_row / T.w_coef — and expands to rows where a row's own right-hand side matters
    // (c3p at BASELINE size: 70 wide rows, 16 distinct left-hand sides, 71 680 against 16 384 terms).
    std::vector<int> grp_of;   // [K] row -> group
    int KG = 0;
    std::vector<int> g_off, g_col; std::vector<int32_t> g_coef;   // [KG] the distinct left-hand sides over the flat columns, terms in the model's order
    std::vector<int> gr_off, gr_row;                              // [KG] the rows of a group, ascending
    size_t row_terms = 0;                                         // terms of the K rows as the model wrote them (statistics)
    std::vector<CapRow> caps;
    std::vector<int32_t> base_cap;
    std::vector<int> block_of_flat;
};

// round half away from zero without the call into libm (-msse4.1 gives floor / nearbyint as one instruction, not round); |v| < 2^51
inline double round_fast(double v) { return (double)(long long)(v + (v >= 0.0 ? 0.5 : -0.5)); }
long long gcd_ll(long long a, long long b) { while (b) { const long long t = a % b; a = b; b = t; } return a < 0 ? -a : a; }

// The groups' left-hand sides -> P's row-wise and column-wise tables (P.grp_of, P.KG set; lhs[g] = (flat column, coefficient) in the model's order)
void finish_groups(Prob &P, const std::vector<std::vector<std::pair<int, int32_t>>> &lhs) {
    HostTables &T = P.T;
    const int KG = P.KG;
    T.K = (uint32_t)KG;
    size_t tot = 0; for (auto &l : lhs) tot += l.size();
    // (sized once and written through pointers: ~16 k terms per tick on c3p, three passes — a push_back per term was a third of the flattening)
    P.g_off.assign((size_t)KG + 1, 0); P.g_col.resize(tot); P.g_coef.resize(tot);
    std::vector<uint32_t> gcnt(T.n_cols + 1, 0);
    {
        auto *gc = P.g_col.data(); auto *gk = P.g_coef.data(); uint32_t *cnt = gcnt.data() + 1; size_t pos = 0;
        for (int g = 0; g < KG; g++) {
            const std::pair<int, int32_t> *l = lhs[(size_t)g].data();
            for (size_t i = 0, e = lhs[(size_t)g].size(); i < e; i++) { gc[pos] = l[i].first; gk[pos] = l[i].second; cnt[l[i].first]++; pos++; }
            P.g_off[(size_t)g + 1] = (int)pos;
        }
    }
    T.col_woff.assign(T.n_cols + 1, 0);
    for (uint32_t f = 0; f < T.n_cols; f++) T.col_woff[f + 1] = T.col_woff[f] + gcnt[f + 1];
    T.w_row.resize(T.col_woff[T.n_cols]); T.w_coef.resize(T.col_woff[T.n_cols]);
    {
        std::vector<uint32_t> &cur = gcnt;   // (the counts are done with: the cursors take their place)
        for (uint32_t f = 0; f < T.n_cols; f++) cur[f] = T.col_woff[f];
        uint32_t *cp = cur.data(); uint16_t *wr = T.w_row.data(); auto *wc = T.w_coef.data();
        const auto *gc = P.g_col.data(); const auto *gk = P.g_coef.data();
        for (int g = 0; g < KG; g++) for (int e = P.g_off[g], e1 = P.g_off[g + 1]; e < e1; e++) { const uint32_t p = cp[gc[e]]++; wr[p] = (uint16_t)g; wc[p] = gk[e]; }
    }
    // rows of every group, ascending
    P.gr_off.assign((size_t)KG + 1, 0);
    for (int k = 0; k < P.K; k++) P.gr_off[P.grp_of[k] + 1]++;
    for (int g = 0; g < KG; g++) P.gr_off[g + 1] += P.gr_off[g];
    P.gr_row.resize(P.K);
    { std::vector<int> c2(P.gr_off.begin(), P.gr_off.end() - 1); for (int k = 0; k < P.K; k++) P.gr_row[c2[P.grp_of[k]]++] = k; }
}

// The component -> blocks + wide rows.  Returns nullptr on success, else what kept the model on the host.
const char *build(const Request &rq, Prob &P) {
    const int n = rq.n;
    if (!rq.col_group) return "no block hints";
    // blocks in order of their first column (the tick: worker order)
    std::vector<int> gid; gid.reserve(n);
    int max_group = -1;
    for (int j = 0; j < n; j++) max_group = std::max(max_group, rq.col_group[j]);
    std::vector<int> block_of_group((size_t)max_group + 1, -1);
    std::vector<int> bsize;
    std::vector<int> blk(n, -1);
    for (int j = 0; j < n; j++) {
        const int g = rq.col_group[j];
        if (g < 0) { P.gmodel.push_back(j); continue; }
        if (block_of_group[g] < 0) { block_of_group[g] = (int)bsize.size(); bsize.push_back(0); }
        blk[j] = block_of_group[g];
        if (++bsize[blk[j]] > NMAX_BLOCK) return "block with more than 32 columns";
    }
    const int nb = (int)bsize.size();
    if (nb < 8) return "fewer than 8 blocks";
    P.G = (int)P.gmodel.size();
    for (int j : P.gmodel) {
        if (rq.ub[j] != 1.0) return "global column that is not 0/1";
        if (rq.c[j] < 0.0) return "global column with negative cost";
        P.gcost.push_back(rq.c[j]);
    }
    std::vector<int> gidx(n, -1);
    for (int g = 0; g < P.G; g++) gidx[P.gmodel[g]] = g;
    HostTables &T = P.T;
    T.n_blocks = (uint32_t)nb;
    T.blk_off.assign((size_t)nb + 1, 0);
    for (int b = 0; b < nb; b++) T.blk_off[b + 1] = T.blk_off[b] + (uint32_t)bsize[b];
    T.n_cols = T.blk_off[nb];
    P.flat_of.assign(n, -1); P.model_of.assign(T.n_cols, -1); P.block_of_flat.assign(T.n_cols, -1);
    {
        std::vector<uint32_t> cur(T.blk_off.begin(), T.blk_off.end() - 1);
        for (int j = 0; j < n; j++) if (blk[j] >= 0) { const int f = (int)cur[blk[j]]++; P.flat_of[j] = f; P.model_of[f] = j; P.block_of_flat[f] = blk[j]; }
    }
    T.col_cost.assign(T.n_cols, 0.0); T.col_a.assign((size_t)T.n_cols * MMAX_BLOCK, 0.0); T.col_cap.assign(T.n_cols, 0);
    T.blk_m.assign(nb, 0); T.blk_cap.assign((size_t)nb * MMAX_BLOCK, 0.0);
    for (uint32_t f = 0; f < T.n_cols; f++) {
        const int j = P.model_of[f];
        if (rq.c[j] < 0.0) return "negative cost";
        if (!(rq.ub[j] <= 65535.0)) return "column bound above 65535";
        T.col_cost[f] = rq.c[j];
        T.col_cap[f] = (int32_t)std::floor(rq.ub[j] + 1e-9);
    }
    struct WideTmp { double h; uint8_t ge; std::vector<std::pair<int, int32_t>> cols; std::vector<std::pair<int, double>> g; };
    std::vector<WideTmp> wide;
    std::vector<std::pair<int, long long>> ai;  // (one allocation for all block rows)
    for (int i = 0; i < rq.m; i++) {
        const int a = rq.roff[i], e = rq.roff[i + 1];
        if (a == e) continue;
        const double sc = rq.row_scale[i];
        int b0 = -2; bool multi = false, has_g = false, nonneg = true;
        int n_bcols = 0;
        for (int k = a; k < e; k++) {
            const int j = rq.rcol[k];
            if (rq.rcoef[k] < 0.0) nonneg = false;
            if (blk[j] < 0) { has_g = true; continue; }
            n_bcols++;
            if (b0 == -2) b0 = blk[j]; else if (blk[j] != b0) multi = true;
        }
        const bool is_le = rq.rlo[i] <= -INF, is_ge = rq.rhi[i] >= INF;
        if (is_le && is_ge) continue;  // no constraint at all
        if (!multi && !has_g && b0 >= 0) {  // a row of one block
            if (rq.row_implied && rq.row_implied[i]) continue;  // implied for integer points by the block's other rows: the sweeps solve the blocks in integers
            if (is_le && nonneg) {  // no point within the column bounds (which the block carries as caps) can violate it: e.g. the per-worker cut rows of a large cut
                double amax = 0.0;
                for (int k = a; k < e; k++) amax += rq.rcoef[k] * rq.ub[rq.rcol[k]];
                if (amax <= rq.rhi[i] * (1.0 + 1e-12) + 1e-9) continue;
            }
            if (!is_le || !nonneg || rq.rhi[i] < 0.0) return "block row that is not a packing row";
            const int r = T.blk_m[b0];
            if (r >= MMAX_BLOCK) {
                if (rq.trace) { fprintf(stderr, "[price] 5th row of block %d: hi %.6f scale %.6f:", b0, rq.rhi[i], sc); for (int k = a; k < e; k++) fprintf(stderr, " %.6f*x%d(ub %.0f)", rq.rcoef[k] * sc, rq.rcol[k], rq.ub[rq.rcol[k]]); fprintf(stderr, "\n"); }
                return "block with more than 4 rows";
            }
            long long g = 0; bool ok = true;
            ai.clear();
            for (int k = a; k < e && ok; k++) {
                const double v = rq.rcoef[k] * sc * GRID, rv = std::round(v);
                if (rv < 1.0 || std::fabs(v - rv) > 1e-6 * std::max(1.0, rv) || rv >= 4.0e15) ok = false;
                else { ai.push_back({P.flat_of[rq.rcol[k]], (long long)rv}); g = gcd_ll(g, (long long)rv); }
            }
            if (!ok) return "block row off the ResourceAmount grid";
            const double capv = rq.rhi[i] * sc * GRID;
            if (capv >= 4.0e15) return "block row capacity too large";
            const long long capi = (long long)std::floor(capv + 1e-6);
            if (g < 1) g = 1;
            for (auto &t : ai) T.col_a[(size_t)t.first * MMAX_BLOCK + r] += (double)(t.second / g);  // (duplicate terms of one row are summed)
            T.blk_cap[(size_t)b0 * MMAX_BLOCK + r] = (double)(capi / g);
            T.blk_m[b0] = (uint8_t)(r + 1);
            continue;
        }
        if (!is_le && !is_ge) return "equality or range row across blocks";
        // vacuous rows: a `<=` row no point within the column bounds can violate, a `>=` row that holds at zero
        if (is_le && nonneg) {
            double amax = 0.0;
            for (int k = a; k < e; k++) amax += rq.rcoef[k] * rq.ub[rq.rcol[k]];
            if (amax <= rq.rhi[i] * (1.0 + 1e-12) + 1e-9) continue;
        }
        if (is_ge && nonneg && rq.rlo[i] <= 1e-12) continue;
        if (!multi && has_g && n_bcols >= 1 && is_le && nonneg) {  // one block + flags: a conditional bound of the block's column(s)
            if (n_bcols != 1) return "conditional bound over several columns of a block";
            CapRow cr; cr.flat = -1; cr.rhs = 0.0;
            double cx = 0.0;
            for (int k = a; k < e; k++) if (blk[rq.rcol[k]] >= 0) { cr.flat = P.flat_of[rq.rcol[k]]; cx = rq.rcoef[k]; }
            if (!(cx > 0.0)) return "conditional bound with a zero coefficient";
            cr.rhs = rq.rhi[i] / cx;
            for (int k = a; k < e; k++) if (blk[rq.rcol[k]] < 0) cr.g.push_back({gidx[rq.rcol[k]], rq.rcoef[k] / cx});
            P.caps.push_back(std::move(cr));
            continue;
        }
        if (n_bcols == 0) return "row over global columns only";
        WideTmp w; w.ge = is_ge ? 1 : 0;
        w.cols.reserve((size_t)(e - a));
        const double sign = is_le ? 1.0 : -1.0;
        w.h = sign * (is_le ? rq.rhi[i] : rq.rlo[i]) * sc;
        for (int k = a; k < e; k++) {
            const int j = rq.rcol[k];
            const double v = sign * rq.rcoef[k] * sc;
            if (blk[j] < 0) { w.g.push_back({gidx[j], v}); continue; }
            const double rv = std::round(v);
            if (std::fabs(v - rv) > 1e-7 * std::max(1.0, std::fabs(rv)) || std::fabs(rv) > 1.0e9) return "wide row with a non-integer coefficient";
            if (rv != 0.0) w.cols.push_back({P.flat_of[j], (int32_t)rv});
        }
        wide.push_back(std::move(w));
        if ((int)wide.size() > 1024) return "more than 1024 wide rows";
    }
    for (int b = 0; b < nb; b++) if (T.blk_m[b] == 0) {
        // every resource row of this worker is slack at its column bounds (a nearly idle worker and a short ready set: the caps bind, not the resources).  The kernel
        // wants a row: the sum of the block's columns against the sum of their caps — never binding either.
        double total = 0.0;
        for (uint32_t f = T.blk_off[b]; f < T.blk_off[b + 1]; f++) { T.col_a[(size_t)f * MMAX_BLOCK] = 1.0; total += (double)T.col_cap[f]; }
        T.blk_cap[(size_t)b * MMAX_BLOCK] = total;
        T.blk_m[b] = 1;
    }
    // every column of a block must be bounded by its block: a cap below 65536 is there (checked above); amounts that do not fit even once leave ub 0
    P.K = (int)wide.size();
    if (P.K == 0) return "no wide row";
    P.h.resize(P.K); P.ge.resize(P.K); P.g_rows.assign(P.G, {});
    std::vector<uint64_t> sig(P.K);   // a hash of every wide row's left-hand side (the groups of identical left-hand sides below compare hashes first)
    for (int k = 0; k < P.K; k++) {
        P.h[k] = wide[k].h; P.ge[k] = wide[k].ge;
        uint64_t hsh = 1469598103934665603ull;
        for (auto &t : wide[k].cols) { hsh = (hsh ^ (uint64_t)(uint32_t)t.first) * 1099511628211ull; hsh = (hsh ^ (uint64_t)(uint32_t)t.second) * 1099511628211ull; }
        sig[k] = hsh; P.row_terms += wide[k].cols.size();
        for (auto &t : wide[k].g) P.g_rows[t.first].push_back({k, t.second});
    }
    // groups of rows with identical left-hand sides
    P.grp_of.assign(P.K, -1);
    std::vector<int> rep;
    // (the hash of the left-hand side first: the rows are a thousand terms long, an ordered map of them compares them term by term)
    for (int k = 0; k < P.K; k++) {
        int g = -1;
        for (size_t i = 0; i < rep.size() && g < 0; i++) if (sig[rep[i]] == sig[k] && wide[rep[i]].cols == wide[k].cols) g = (int)i;
        if (g < 0) { g = (int)rep.size(); rep.push_back(k); }
        P.grp_of[k] = g;
    }
    P.KG = (int)rep.size();
    if (P.KG > KMAX_HOST) return "more than 128 distinct wide left-hand sides";
    std::vector<std::vector<std::pair<int, int32_t>>> lhs((size_t)P.KG);
    for (int g = 0; g < P.KG; g++) lhs[(size_t)g] = std::move(wide[rep[g]].cols);
    finish_groups(P, lhs);
    P.base_cap = T.col_cap;
    return nullptr;
}

struct Cut {
    double cx = 0, bnd = 0; std::vector<long long> act; std::vector<double> pi;
    // the maximisers, one integer pattern per block, in the model's own variables: fetched from the device on demand (the sweep solved in variables shifted by `lo`)
    std::vector<uint16_t> x; bool fetched = false; std::vector<std::pair<uint32_t, int32_t>> lo;
    // the same per part of the model (PARTS contiguous ranges of blocks): the root master keeps one value function per part
    std::vector<double> pcx; std::vector<long long> pact;  // [PARTS], [PARTS * K]
};

// a node of the branch-and-price phase: flags fixed or free, bounds on single block columns
struct BPNode { std::vector<int8_t> flag; std::vector<std::pair<uint32_t, int32_t>> lo, hi; double bound = INF; int depth = 0; };

struct Solver {
    const Request &rq; Sweeper &sw; Prob P;
    std::vector<Cut> cuts;
    double relaxed_bound = INF;   // min over every evaluated pi of the Lagrangian with the flags relaxed to [0, 1]
    std::vector<double> relaxed_pi;
    double theta_scale = 1.0;
    std::vector<double> pmax;
    bool failed = false;
    size_t cut_lo = 0;            // the master works on cuts[cut_lo ..): points evaluated under the column bounds now in force
    double work = 0.0;            // tableau elements touched by the masters so far (Tab::ops)
    // What the sweeps themselves cost, in the kernel's own unit: a sweep lasts as long as its slowest block, ~1 us per search step on top of ~60 us.  Blocks that
    // run out of their step budget return an LP bound instead of their optimum: the total stays a valid bound, but the master's cuts (patterns) and its evaluations
    // (bounds) no longer describe the same function and the cutting-plane loop stops converging — a model whose blocks do that sweep after sweep (16-column blocks of a
    // busy C4 cluster) is not one for this path.
    double sweep_steps = 0.0; int budget_sweeps = 0;
    int max_sweeps = 256;
    // branch-and-price: the node whose bounds the device tables carry right now (empty lo list = the model's own bounds), and what its lower bounds add per sweep
    std::vector<std::pair<uint32_t, int32_t>> node_lo, node_caps; double node_cl = 0.0; std::vector<double> node_Al; bool at_root = true;
    bool base_caps = true;        // the sweeps run under the model's own column bounds (only those bound the relaxed model)
    Solver(const Request &r, Sweeper &s) : rq(r), sw(s) {}

    // one sweep at pi; returns the cut's index or -1.  In two halves for the caller with work to do meanwhile: evaluate_launch(pi), ..., evaluate(pi, true).
    bool evaluate_launch(const std::vector<double> &pi) {
        if ((int)cuts.size() >= max_sweeps) return false;
        std::vector<double> pig(P.KG, 0.0);
        for (int k = 0; k < P.K; k++) pig[P.grp_of[k]] += pi[k];
        const double t0 = now_us();
        sw.pending_k = (uint32_t)P.KG;
        if (!sw.sweep_launch(pig.data())) { failed = true; return false; }
        sw.stat_sweep_us += now_us() - t0;
        return true;
    }
    int evaluate(const std::vector<double> &pi, bool launched = false) {
        if ((int)cuts.size() >= max_sweeps) return -1;
        SweepTotals tot;
        const double t0 = now_us();
        if (launched) { if (!sw.sweep_finish(tot)) { failed = true; return -1; } }
        else {
            std::vector<double> pig(P.KG, 0.0);
            for (int k = 0; k < P.K; k++) pig[P.grp_of[k]] += pi[k];
            if (!sw.sweep(pig.data(), tot)) { failed = true; return -1; }
        }
        sw.stat_sweep_us += now_us() - t0; sw.stat_sweeps++;
        if (!sw.merges_clock() && now_us() * 1e-6 > sw.guard_s) sw.time_up = true;  // (a sharded sweeper has merged the ranks' readings inside sweep(): a local reading here could split the replicas)
        sweep_steps += 64.0 + (double)tot.max_steps;
        if (tot.n_budget > std::max<uint32_t>(4, P.T.n_blocks / 64)) budget_sweeps++;
        Cut c; c.cx = tot.cx; c.bnd = tot.bnd; c.pi = pi;
        c.act.resize(P.K);
        for (int k = 0; k < P.K; k++) c.act[k] = tot.act[P.grp_of[k]];
        if (node_lo.empty() && tot.part_cx.size() == (size_t)PARTS && tot.part_act.size() == (size_t)PARTS * P.KG) {
            c.pcx = tot.part_cx; c.pact.resize((size_t)PARTS * P.K);
            for (int p = 0; p < PARTS; p++) for (int k = 0; k < P.K; k++) c.pact[(size_t)p * P.K + k] = tot.part_act[(size_t)p * P.KG + P.grp_of[k]];
        }
        if (!node_lo.empty()) {  // the sweep ran in variables shifted by the node's lower bounds: back to the model's own (c.l, A.l and the bound's (c - pi A).l)
            c.lo = node_lo; c.cx += node_cl; c.bnd += node_cl;
            for (int k = 0; k < P.K; k++) { c.act[k] += (long long)std::llround(node_Al[k]); c.bnd -= pi[k] * node_Al[k]; }
        }
        cuts.push_back(std::move(c));
        // the model's bound at these prices, flags relaxed: pi.h + sum_w V_w(pi) + sum_g max(0, c_g - pi.A_g)
        double L = cuts.back().bnd;
        for (int k = 0; k < P.K; k++) L += pi[k] * P.h[k];
        for (int g = 0; g < P.G; g++) { double r = P.gcost[g]; for (auto &t : P.g_rows[g]) r -= pi[t.first] * t.second; if (r > 0.0) L += r; }
        if (base_caps && at_root && L < relaxed_bound) { relaxed_bound = L; relaxed_pi = pi; }
        return (int)cuts.size() - 1;
    }
    double fixed_value(const Cut &c, const std::vector<double> &hB, double cB) const {
        double L = c.bnd + cB;
        for (int k = 0; k < P.K; k++) L += c.pi[k] * hB[k];
        return L;
    }

    // Cutting-plane master (Kelley) for the model with its flags fixed: minimise pi.hB + theta over theta + pi.act_k >= cx_k.  Returns false when the
    // configuration cannot beat `cutoff` (or is infeasible); on success `lambda` holds the master's multipliers of the cuts (summing to 1) and pi_out
    // the final prices.
    //
    // The master is DISAGGREGATED over the model's parts: sum_w V_w(pi) is a sum over PARTS contiguous ranges of workers, every sweep returns each part's
    // own c.x and activities, and the master keeps one theta per part — its model of the dual function is the sum of PARTS piecewise-linear models
    // instead of one, which takes a third of the sweeps to the same accuracy (the first wave of the layered config-5 DAG: 61 -> 19-22).  The workers
    // are in objective order (factor (W - idx) / W, solver.rs:542-571), so contiguous ranges are the parts that differ most from one another.
    //
    // `probe` (optional; called after a master round that is within 10 %, at most three times): shown the master's fractional point — the columns' activities in every wide row — and its
    // lower bound; true = the caller goes to another configuration, this walk ends (false is returned).  `no_sweeps`: only what the cuts at hand settle — false as
    // soon as the master would need another sweep.
    // Starting prices from the model AGGREGATED over chunks of its blocks (round 6).  Where the blocks of a part are the same block up to their costs — a cluster of identical
    // workers: every cold tick, every DAG tick on an idle cluster — the part is one block with `count` times the capacities and the mean cost, and the model is an LP of
    // PARTS x (columns of a block) variables under the priced wide rows: a hundred microseconds of dual simplex.  Its duals of the wide rows are where the decomposition's
    // prices end up, up to what integrality and the cost spread inside a part change — a far better first proposal than a fraction of the price caps.  A proposal like any
    // other (the cut it yields is exact); false: the blocks of some part differ, the caller keeps its scaled proposal.
    bool aggregate_prices(const std::vector<double> &hB, const std::vector<int> &lk, std::vector<double> &pi) {
        const HostTables &T = P.T;
        // chunks of blocks = the aggregated model's granularity: as fine as ~256 LP variables allow (the dense dual simplex of lp_tab.h: ~0.1 ms there, milliseconds at 1024) —
        // 32 chunks of eight-column blocks, 16 of sixteen-column ones.  Measured in sweeps (emulated, deterministic): config-5 first wave 10 -> 7, the 20 % probe 15 -> 11,
        // configs[3]'s unsaturated cluster 19 -> 16; with 64 chunks 7 / 10 / 14, at ten times the LP.
        static const int chunks_env = getenv("HQPRICE_AGG_CHUNKS") ? atoi(getenv("HQPRICE_AGG_CHUNKS")) : 0;
        uint32_t widest = 1;
        for (uint32_t b = 0; b < T.n_blocks; b++) widest = std::max(widest, T.blk_off[b + 1] - T.blk_off[b]);
        const int agg_chunks = chunks_env > 0 ? chunks_env : std::min(64, std::max(PARTS, (int)(256u / widest)));
        const uint32_t nb = T.n_blocks, per = (nb + (uint32_t)agg_chunks - 1) / (uint32_t)agg_chunks;
        const int NP = (int)((nb + per - 1) / per), KL = (int)lk.size();
        if (NP < 2 || KL == 0) return false;
        // Only where a sweep is expensive: the aggregated LP (~256 variables through the dense dual simplex) costs ~1 ms on the MI355X box's host — three sweeps of a
        // 4096-block model, a dozen of a 1024-block one.  (Measured with it on everywhere: layered DAG loop 17 -> 14 sweeps but 2.00 -> 2.13 ms per tick; configs[3]'s
        // unsaturated cluster 19 -> 16 sweeps and 8.8 -> 7.6 ms.)
        static const uint64_t min_cols_env = getenv("HQPRICE_AGG_MINCOLS") ? (uint64_t)atoll(getenv("HQPRICE_AGG_MINCOLS")) : 32768;   // (experiments)
        if ((uint64_t)T.n_cols < min_cols_env) return false;
        if (P.G != 0) return false;   // (models with flags: the aggregated duals of one configuration mislead the walk over configurations — c3p 3 -> 4 sweeps, c4p 9 -> 16)
        std::vector<uint32_t> pb0(NP), pcnt(NP), pnc(NP), voff(NP + 1, 0);
        for (int p = 0; p < NP; p++) {
            const uint32_t b0 = (uint32_t)p * per, b1 = std::min(nb, b0 + per);
            pb0[p] = b0; pcnt[p] = b1 - b0; pnc[p] = T.blk_off[b0 + 1] - T.blk_off[b0];
            const uint32_t c0 = T.blk_off[b0], nc = pnc[p];
            for (uint32_t b = b0 + 1; b < b1; b++) {   // the same block? (amounts, bounds, capacities, membership in the wide rows — everything but the costs)
                const uint32_t c = T.blk_off[b];
                if (T.blk_off[b + 1] - c != nc || T.blk_m[b] != T.blk_m[b0]) return false;
                if (memcmp(&T.blk_cap[(size_t)b * MMAX_BLOCK], &T.blk_cap[(size_t)b0 * MMAX_BLOCK], sizeof(double) * MMAX_BLOCK) != 0) return false;
                if (memcmp(&T.col_a[(size_t)c * MMAX_BLOCK], &T.col_a[(size_t)c0 * MMAX_BLOCK], sizeof(double) * MMAX_BLOCK * nc) != 0) return false;
                if (memcmp(&T.col_cap[c], &T.col_cap[c0], sizeof(int32_t) * nc) != 0) return false;
                for (uint32_t q = 0; q < nc; q++) {
                    const uint32_t e0 = T.col_woff[c0 + q], e1 = T.col_woff[c0 + q + 1], f0 = T.col_woff[c + q];
                    if (T.col_woff[c + q + 1] - f0 != e1 - e0) return false;
                    if (e1 > e0 && (memcmp(&T.w_row[f0], &T.w_row[e0], sizeof(uint16_t) * (e1 - e0)) != 0 || memcmp(&T.w_coef[f0], &T.w_coef[e0], sizeof(int32_t) * (e1 - e0)) != 0)) return false;
                }
            }
            voff[p + 1] = voff[p] + nc;
        }
        const int nv = (int)voff[NP];
        if (nv == 0 || nv > 8192) return false;
        Rows A; A.n = nv;
        std::vector<double> ac(nv), alb(nv, 0.0), aub(nv);
        std::vector<std::pair<int, double>> terms;
        for (int p = 0; p < NP; p++) {
            const uint32_t c0 = T.blk_off[pb0[p]], nc = pnc[p];
            for (uint32_t q = 0; q < nc; q++) {
                double sum = 0.0;
                for (uint32_t b = 0; b < pcnt[p]; b++) sum += T.col_cost[T.blk_off[pb0[p] + b] + q];
                ac[voff[p] + q] = sum / (double)pcnt[p] / theta_scale;
                aub[voff[p] + q] = (double)pcnt[p] * (double)T.col_cap[c0 + q];
            }
            for (int r = 0; r < (int)T.blk_m[pb0[p]]; r++) {
                terms.clear(); double sc = 0.0;
                for (uint32_t q = 0; q < nc; q++) { const double a = T.col_a[(size_t)(c0 + q) * MMAX_BLOCK + r]; if (a != 0.0) { terms.push_back({(int)(voff[p] + q), a}); sc = std::max(sc, std::fabs(a)); } }
                if (terms.empty()) continue;
                for (auto &t : terms) t.second /= sc;
                A.add(terms, -INF, (double)pcnt[p] * T.blk_cap[(size_t)pb0[p] * MMAX_BLOCK + r] / sc);
            }
        }
        const int first_wide = A.m;
        std::vector<double> wscale(KL, 1.0);
        for (int i = 0; i < KL; i++) {
            const int g = P.grp_of[lk[i]];
            terms.clear(); double sc = 0.0;
            for (int p = 0; p < NP; p++) {
                const uint32_t c0 = T.blk_off[pb0[p]];
                for (uint32_t q = 0; q < pnc[p]; q++) for (uint32_t e = T.col_woff[c0 + q]; e < T.col_woff[c0 + q + 1]; e++) if ((int)T.w_row[e] == g) { terms.push_back({(int)(voff[p] + q), (double)T.w_coef[e]}); sc = std::max(sc, std::fabs((double)T.w_coef[e])); }
            }
            if (terms.empty()) { A.add({{0, 0.0}}, -INF, INF); continue; }
            for (auto &t : terms) t.second /= sc;
            wscale[i] = sc;
            A.add(terms, -INF, hB[lk[i]] / sc);
        }
        Tab at; at.init(&A, ac, alb, aub);
        if (at.solve(20000) != LP_OPT) return false;
        work += at.ops;
        pi.assign(P.K, 0.0);
        bool any = false;
        for (int i = 0; i < KL; i++) {
            const int a = at.where[first_wide + i];
            if (a < 0 || at.st[A.n + a] == BASIC) continue;
            double y = std::fabs(at.d[A.n + a]) / wscale[i] * theta_scale;   // the row's dual, back in the units of the prices (cost per unit of the row's activity)
            if (!(y > 0.0)) continue;
            pi[lk[i]] = std::min(y, pmax[lk[i]]);
            any = true;
        }
        return any;
    }

    std::function<bool(const std::vector<double> &, double)> probe;
    // what a no_sweeps walk that converged found: the caller goes to that configuration next, whose own walk would solve the same master again
    struct Settled { bool valid = false; std::vector<double> hB, lambda, pi; double cB = 0.0, bound = 0.0; size_t n_cuts = 0, cut_lo = 0; } settled;
    bool kelley(const std::vector<double> &hB, double cB, double cutoff, double tol, std::vector<double> &lambda, std::vector<double> &pi_out, double *bound_out, bool no_sweeps = false) {
        if (!no_sweeps && settled.valid) {
            settled.valid = false;
            if (settled.n_cuts == cuts.size() && settled.cut_lo == cut_lo && settled.cB == cB && settled.hB == hB) {
                if (settled.bound < cutoff) return false;
                lambda = std::move(settled.lambda); pi_out = std::move(settled.pi); *bound_out = settled.bound;
                return true;
            }
        }
        const int K = P.K;
        const uint32_t per = (P.T.n_blocks + PARTS - 1) / PARTS;
        const int NP = (int)((P.T.n_blocks + per - 1) / per);  // parts that hold blocks
        // With the flags fixed, the rows of one group are ONE left-hand side against several right-hand sides: only the tightest can bind, the others get price 0.  The
        // master prices that one row per group (a three-level tick: 16 of 70 rows — a tableau a third as wide, and none of its degenerate ties among identical columns).
        std::vector<int> lk; lk.reserve(P.KG);
        for (int g = 0; g < P.KG; g++) {
            int best = -1;
            for (int q = P.gr_off[g]; q < P.gr_off[g + 1]; q++) { const int k = P.gr_row[q]; if (best < 0 || hB[k] < hB[best]) best = k; }
            if (best >= 0) lk.push_back(best);
        }
        std::sort(lk.begin(), lk.end());
        const int KL = (int)lk.size();
        Rows M; M.n = KL + NP;
        std::vector<double> mc(KL + NP), mlb(KL + NP, 0.0), mub(KL + NP);
        for (int i = 0; i < KL; i++) { mc[i] = -hB[lk[i]] / theta_scale; mub[i] = pmax[lk[i]]; }
        for (int p = 0; p < NP; p++) { mc[KL + p] = -1.0; mub[KL + p] = 4.0; }
        std::vector<double> cut_scale;
        std::vector<std::pair<int, double>> terms;
        size_t in_master = cut_lo;
        auto push_cuts = [&](Tab *mt) {
            for (; in_master < cuts.size(); in_master++) {
                const Cut &c = cuts[in_master];
                for (int p = 0; p < NP; p++) {
                    terms.clear(); double sc = 1.0;
                    for (int i = 0; i < KL; i++) { const long long a = c.pact[(size_t)p * K + lk[i]]; if (a != 0) { const double v = (double)a / theta_scale; terms.push_back({i, v}); sc = std::max(sc, std::fabs(v)); } }
                    terms.push_back({KL + p, 1.0});
                    for (auto &t : terms) t.second /= sc;
                    M.add(terms, c.pcx[p] / theta_scale / sc, INF);
                    cut_scale.push_back(sc);
                    if (mt) mt->where.push_back(-1);
                }
            }
        };
        for (size_t k = cut_lo; k < cuts.size(); k++) if (cuts[k].pcx.empty()) { if (rq.trace) fprintf(stderr, "[price] master: cut %zu without part sums\n", k); return false; }  // (a sweeper without part sums: cannot happen with the two in this tree)
        push_cuts(nullptr);
        Tab mt; mt.init(&M, mc, mlb, mub);
        double ub_best = INF; std::vector<double> pi_best(K, 0.0);
        for (size_t k = cut_lo; k < cuts.size(); k++) { const double L = fixed_value(cuts[k], hB, cB); if (L < ub_best) { ub_best = L; pi_best = cuts[k].pi; } }
        std::vector<double> pi(K, 0.0), pi_prev_master;
        double lb_master = -INF, lp_us = 0.0;
        bool converged = false; int probes = 0;
        // multipliers per (cut, part) of the master as it stands (after a solve): every part's sum to 1
        auto multipliers = [&](std::vector<double> &lambda) {
            lambda.assign(cuts.size() * PARTS, 0.0);
            std::vector<double> lsum(NP, 0.0);
            for (size_t r = 0; r / NP + cut_lo < cuts.size() && r < (size_t)M.m; r++) {
                const int a = mt.where[r];
                if (a < 0 || mt.st[M.n + a] == BASIC) continue;
                const size_t k = r / NP + cut_lo; const int p = (int)(r % NP);
                const double v = std::fabs(mt.d[M.n + a]) / cut_scale[r];
                lambda[k * PARTS + p] = v; lsum[p] += v;
            }
            for (int p = 0; p < NP; p++) if (!(lsum[p] > 0.0)) {
                // no cut of the part binds: its theta sits on its bound 0 — at the final prices the part's workers take nothing (or nothing worth anything).
                // Its point is the pattern of the cut that is worth most at those prices.
                size_t bk = cut_lo; double bv = -INF;
                for (size_t k = cut_lo; k < cuts.size(); k++) {
                    double v = cuts[k].pcx[p];
                    for (int i = 0; i < KL; i++) v -= mt.x[i] * (double)cuts[k].pact[(size_t)p * K + lk[i]];
                    if (v > bv) { bv = v; bk = k; }
                }
                lambda[bk * PARTS + p] = 1.0; lsum[p] = 1.0;
            }
            for (size_t k = cut_lo; k < cuts.size(); k++) for (int p = 0; p < NP; p++) lambda[k * PARTS + p] /= lsum[p];
        };
        for (int it = 0; it < 200; it++) {
            { const double tl0 = now_us(); const int st = mt.solve(200000); lp_us += now_us() - tl0; if (st != LP_OPT) { if (rq.trace) fprintf(stderr, "[price] master LP status %d at iteration %d (%d cuts)\n", st, it, M.m); return false; } }
            lb_master = -mt.objective() * theta_scale + cB;
            if (ub_best < cutoff) { if (rq.trace) fprintf(stderr, "[price] configuration bounded by %.9f, below the incumbent %.9f\n", ub_best, cutoff); return false; }  // even the relaxation of this configuration is below the incumbent
            if (ub_best - lb_master <= tol * std::fabs(ub_best)) { converged = true; break; }
            if (no_sweeps) return false;
            if (probe && probes < 3 && it >= 1 && ub_best - lb_master <= 0.1 * std::fabs(ub_best)) {
                probes++;
                std::vector<double> lam, colact(K, 0.0);
                multipliers(lam);
                for (size_t k = cut_lo; k < cuts.size(); k++) for (int p = 0; p < NP; p++) {
                    const double l = lam[k * PARTS + p];
                    if (l == 0.0) continue;
                    const long long *pa = &cuts[k].pact[(size_t)p * K];
                    for (int r = 0; r < K; r++) colact[r] += l * (double)pa[r];
                }
                auto keep = std::move(probe); probe = nullptr;   // (the probe runs a master of its own: not this one's again)
                const bool go = keep(colact, lb_master);
                probe = std::move(keep);
                if (go) return false;
            }
            bool same = !pi_prev_master.empty();
            for (int i = 0; i < KL && same; i++) same = std::fabs(mt.x[i] - pi_prev_master[i]) <= 1e-15 + 1e-12 * std::fabs(mt.x[i]);
            pi_prev_master.assign(mt.x.begin(), mt.x.begin() + KL);
            const double alpha = (it < 3 || same) ? 0.0 : 0.3;       // in-out: between the best point so far and the master's proposal
            for (int k = 0; k < K; k++) pi[k] = alpha * pi_best[k];
            for (int i = 0; i < KL; i++) pi[lk[i]] += (1.0 - alpha) * mt.x[i];
            // The very first master has one cut (the patterns at pi = 0) and nothing that holds its prices back: its proposal sits on the price caps, where every column
            // of a priced row has stopped paying — far beyond the prices that only have to break ties or shave the marginal tasks.  A sixteenth of it is evaluated
            // instead (a valid point like any other: the cut it yields is exact): the 65 536-column unsaturated tick 22 -> 19 sweeps, 78 fuzz seeds 36 924 -> 34 596.
            if (it == 0 && cut_lo == 0 && cuts.size() == 1) {
                std::vector<double> pa;
                static const bool agg_on = !(getenv("HQPRICE_AGG_START") && atoi(getenv("HQPRICE_AGG_START")) == 0);   // (A/B switch)
                if (agg_on && aggregate_prices(hB, lk, pa)) { pi = pa; if (rq.trace) fprintf(stderr, "[price]   first proposal: the duals of the model aggregated over chunks of blocks\n"); }
                else for (int k = 0; k < K; k++) pi[k] *= FIRST_PROPOSAL_SCALE;
            }
            const int ci = evaluate(pi);
            if (ci < 0) break;
            if (sw.time_up) { if (rq.trace) fprintf(stderr, "[price] 70 %% of the time limit gone inside the master loop\n"); return false; }
            if (budget_sweeps >= 8) { if (rq.trace) fprintf(stderr, "[price] blocks keep running out of their search budget (%d sweeps): not a model for the sweeps\n", budget_sweeps); return false; }
            const double L = fixed_value(cuts[ci], hB, cB);
            if (L < ub_best) { ub_best = L; pi_best = pi; }
            push_cuts(&mt);
        }
        if (!converged) {
            if (mt.solve(200000) != LP_OPT) return false;
            lb_master = -mt.objective() * theta_scale + cB;
            if (ub_best - lb_master > 1e-3 * std::fabs(ub_best)) { if (rq.trace) fprintf(stderr, "[price] master not converged: %.9f vs %.9f\n", ub_best, lb_master); return false; }  // nowhere near: no usable multipliers
        }
        multipliers(lambda);
        if (rq.trace) fprintf(stderr, "[price]   master: %d priced rows + %d parts, %d cut rows, %ld pivots, %.1f us inside the LP (%d active rows)\n", KL, NP, M.m, mt.iters, lp_us, mt.ma);
        pi_out.assign(K, 0.0);
        for (int i = 0; i < KL; i++) pi_out[lk[i]] = mt.x[i];
        *bound_out = ub_best;
        if (no_sweeps) { settled.valid = true; settled.hB = hB; settled.cB = cB; settled.lambda = lambda; settled.pi = pi_out; settled.bound = ub_best; settled.n_cuts = cuts.size(); settled.cut_lo = cut_lo; }
        return true;
    }

    // the patterns of the cuts that have none yet, from the device's ring (one copy for the lot)
    bool fetch_patterns() {
        size_t first = cuts.size();
        for (size_t k = 0; k < cuts.size(); k++) if (!cuts[k].fetched) { first = k; break; }
        if (first == cuts.size()) return true;
        const uint32_t count = (uint32_t)(cuts.size() - first);
        const uint16_t *pat = sw.patterns((uint32_t)first, count);
        if (!pat) { failed = true; return false; }
        const uint32_t nc = P.T.n_cols;
        for (uint32_t i = 0; i < count; i++) {
            Cut &c = cuts[first + i];
            if (c.fetched) continue;
            c.x.assign(pat + (size_t)i * nc, pat + (size_t)(i + 1) * nc);
            for (auto &l : c.lo) c.x[l.first] = (uint16_t)(c.x[l.first] + l.second);
            c.fetched = true;
        }
        return true;
    }

    // One integer pattern per block out of the active cuts' maximisers, by error diffusion over the wide rows' running totals; then the repair.
    // `hB`: right-hand sides with the flags fixed.  Returns the point over the flat columns (empty: no usable point).
    // `usable` (optional): cuts whose patterns may be used at all (branch-and-price: those that respect the node's bounds); `lo_of` (optional): lower
    // bounds of single columns the repair must not go below.
    // `variant`: 0 = the blocks in their own order (worker order: the objective's), 1 = from the last block to the first — another point of the same quality
    // class, for the ticks whose first point misses the gap by a hair.
    std::vector<uint16_t> round_patterns(const std::vector<double> &lambda, const std::vector<double> &pi, const std::vector<double> &hB,
                                         const std::vector<char> *usable = nullptr, const std::vector<int32_t> *lo_of = nullptr, int variant = 0) {
        const int K = P.K; const HostTables &T = P.T;
        const uint32_t S = (uint32_t)cuts.size();
        const double t_r0 = now_us();
        if (!fetch_patterns()) return {};
        auto pat_of = [&](int k) { return cuts[(size_t)k].x.data(); };
        // lambda: [cut * PARTS + part], every part's multipliers summing to 1 (a caller with one multiplier per cut repeats it for every part)
        const uint32_t per = (T.n_blocks + PARTS - 1) / PARTS;
        auto lam = [&](int k, uint32_t p) { return lambda[(size_t)k * PARTS + p]; };
        std::vector<int> active;
        for (uint32_t k = 0; k < S; k++) {
            if (usable && !(*usable)[k]) continue;
            bool any = false; for (uint32_t p = 0; p < (uint32_t)PARTS && !any; p++) any = lam((int)k, p) > 1e-9;
            if (any) active.push_back((int)k);
        }
        if (active.empty()) return {};
        const int Q = (int)active.size();
        // row weights: relative deviation; rows the LP leaves slack count little
        std::vector<double> lpact(K, 0.0), wgt(K);
        for (int k : active) {
            const Cut &c = cuts[k];
            if (c.pact.empty()) { for (int r = 0; r < K; r++) lpact[r] += lam(k, 0) * (double)c.act[r]; continue; }
            for (uint32_t p = 0; p < (uint32_t)PARTS; p++) { const double l = lam(k, p); if (l > 0.0) for (int r = 0; r < K; r++) lpact[r] += l * (double)c.pact[(size_t)p * P.K + r]; }
        }
        for (int r = 0; r < K; r++) {
            const bool tight = pi[r] > 1e-12 || hB[r] - lpact[r] <= 1e-6 * std::max(1.0, std::fabs(hB[r]));
            wgt[r] = (tight ? 1.0 : 0.02) / std::max(1.0, std::fabs(hB[r]));
        }
        auto block_act = [&](uint32_t b, const uint16_t *x, std::vector<double> &out) {  // A_w x of block b for the pattern row x (flat layout)
            // per group (the rows of a group share its left-hand side), then every row its group's value: the sums run over the block's columns in the same order either way
            double ga[KMAX_HOST];
            for (int g = 0; g < P.KG; g++) ga[g] = 0.0;
            for (uint32_t f = T.blk_off[b]; f < T.blk_off[b + 1]; f++) { const uint16_t xv = x[f]; if (!xv) continue; for (uint32_t e = T.col_woff[f]; e < T.col_woff[f + 1]; e++) ga[T.w_row[e]] += (double)T.w_coef[e] * (double)xv; }
            for (int r = 0; r < K; r++) out[r] = ga[P.grp_of[r]];
        };
        std::vector<uint16_t> x(T.n_cols, 0);
        std::vector<double> cum(K, 0.0);
        std::vector<int> chosen(T.n_blocks, 0);
        {   // The diffusion runs per GROUP of rows with one left-hand side: their running totals, targets and candidates are the same numbers (a row's own part is its
            // right-hand side, which enters through the weight), so the error sum_r |cum_r + cand_r - tgt_r| w_r is sum_g |cum_g + cand_g - tgt_g| W_g with W_g the
            // group's summed weights — a quarter of the terms on a three-level tick (70 rows, 16 left-hand sides).
            const int KG = P.KG;
            std::vector<double> wg(KG, 0.0), cumg(KG, 0.0), tgtg(KG, 0.0), candg((size_t)Q * KG);
            for (int r = 0; r < K; r++) wg[P.grp_of[r]] += wgt[r];
            std::vector<char> on(Q); std::vector<const double *> cgp(Q, nullptr);
            for (uint32_t bi = 0; bi < T.n_blocks; bi++) {
                const uint32_t b = variant == 1 ? T.n_blocks - 1 - bi : bi;
                const uint32_t part = b / per;
                // (three blocks in four carry the SAME pattern in every cut of their part: its activities are worked out once, the targets take the same additions in
                // the same order, and with nothing to choose between the first cut's pattern is the choice — as the comparison below would make it)
                int first = -1; bool all_same = true;
                const size_t blk_bytes = (size_t)(T.blk_off[b + 1] - T.blk_off[b]) * 2;
                for (int q = 0; q < Q; q++) {
                    const double l = lam(active[q], part);
                    on[q] = l > 1e-9;   // (only the cuts the part's LP point is made of: their patterns are optimal at the final prices)
                    if (!on[q]) continue;
                    const uint16_t *px = pat_of(active[q]);
                    if (first >= 0 && memcmp(px + T.blk_off[b], pat_of(active[first]) + T.blk_off[b], blk_bytes) == 0) cgp[q] = cgp[first];
                    else {
                        if (first < 0) first = q; else all_same = false;
                        double *cg = &candg[(size_t)q * KG];
                        for (int g = 0; g < KG; g++) cg[g] = 0.0;
                        for (uint32_t f = T.blk_off[b]; f < T.blk_off[b + 1]; f++) { const uint16_t xv = px[f]; if (!xv) continue; for (uint32_t e = T.col_woff[f]; e < T.col_woff[f + 1]; e++) cg[T.w_row[e]] += (double)T.w_coef[e] * (double)xv; }
                        cgp[q] = cg;
                    }
                    const double *cg = cgp[q];
                    for (int g = 0; g < KG; g++) tgtg[g] += l * cg[g];
                }
                int bq = -1; double be = INF;
                if (all_same) bq = first;
                else for (int q = 0; q < Q; q++) {
                    if (!on[q]) continue;
                    const double *cg = cgp[q];
                    double e = 0.0;
                    for (int g = 0; g < KG; g++) e += std::fabs(cumg[g] + cg[g] - tgtg[g]) * wg[g];
                    if (e < be - 1e-15) { be = e; bq = q; }
                }
                if (bq < 0) return {};
                const double *cb = cgp[bq];
                for (int g = 0; g < KG; g++) cumg[g] += cb[g];
                chosen[b] = active[bq];
                memcpy(&x[T.blk_off[b]], pat_of(active[bq]) + T.blk_off[b], (size_t)(T.blk_off[b + 1] - T.blk_off[b]) * 2);
            }
            for (int r = 0; r < K; r++) cum[r] = cumg[P.grp_of[r]];
        }
        if (rq.trace) fprintf(stderr, "[price]   rounding: diffusion done at %.3f us, %d active cuts\n", now_us() - t_r0, Q);
        // `>=` rows that came out short: single-block pattern switches (any sweep's pattern of that block) that close the shortfall at the least loss
        auto ge_short = [&](const std::vector<double> &c) { double s = 0.0; for (int r = 0; r < K; r++) if (P.ge[r] && c[r] > hB[r] + 1e-9) s += c[r] - hB[r]; return s; };
        if (ge_short(cum) > 0.0) {
            std::vector<int> pool = active;
            for (int k = (int)S - 1; k >= 0 && (int)pool.size() < Q + 12; k--) if ((!usable || (*usable)[k]) && std::find(pool.begin(), pool.end(), k) == pool.end()) pool.push_back(k);
            std::vector<double> a0(K), a1(K), c2(K);
            double cmax = 0.0; for (double c : T.col_cost) cmax = std::max(cmax, c);
            // Passes: every switch that would help under the current totals is scored once, the list is walked best first and a switch applied if it still
            // helps under the totals as they are by then (one block once per pass) — a pass costs what ONE move of a best-move-at-a-time loop costs.
            struct Cand { double score; uint32_t b; int k; };
            std::vector<Cand> cl; std::vector<double> colshort, coltight;
            auto evaluate_switch = [&](uint32_t b, int k, double v0, double &score) {  // a0 = the block's current activities
                const uint16_t *px = pat_of(k);
                block_act(b, px, a1);
                double gain = 0.0, newviol = 0.0;
                for (int r = 0; r < K; r++) {
                    if (a0[r] == a1[r]) continue;
                    const double c = cum[r] - a0[r] + a1[r];
                    if (P.ge[r]) gain += std::max(0.0, cum[r] - hB[r]) - std::max(0.0, c - hB[r]);
                    else newviol += std::max(0.0, c - hB[r]) - std::max(0.0, cum[r] - hB[r]);
                }
                if (!(gain > 1e-9)) return false;
                double v1 = 0.0; for (uint32_t f = T.blk_off[b]; f < T.blk_off[b + 1]; f++) v1 += T.col_cost[f] * (double)px[f];
                score = gain * 10.0 * cmax - (v0 - v1) - std::max(0.0, newviol) * cmax;
                return true;
            };
            auto value_of = [&](uint32_t b) { double v = 0.0; for (uint32_t f = T.blk_off[b]; f < T.blk_off[b + 1]; f++) v += T.col_cost[f] * (double)x[f]; return v; };
            for (int pass = 0; pass < 64 && ge_short(cum) > 0.0; pass++) {
                const double base = ge_short(cum);
                cl.clear();
                std::vector<int> short_in(P.KG, 0);   // short `>=` rows per group
                for (int r = 0; r < K; r++) if (P.ge[r] && cum[r] > hB[r] + 1e-9) short_in[P.grp_of[r]]++;
                // what one more task of column f takes off the short rows (its entries in their groups, once per pass instead of once per candidate)
                // ... and what one task LESS of it costs on the `>=` rows that hold with (almost) nothing to spare.  A row with slack s that the switch moves by D is short
                // by max(0, D - s) afterwards; summed over those rows that is at least sum(D) - sum(s), the linear form below: a bound from BELOW on what the full look
                // subtracts, as the gain here is one from above on what it adds — the filter drops no switch the full look would take.  (The three-level C3 tick:
                // 323 full looks at switches that close the short row by emptying its neighbour, before the first one that does not.)
                std::vector<int> tight_in(P.KG, 0);
                double spare = 0.0;
                for (int r = 0; r < K; r++) if (P.ge[r] && cum[r] <= hB[r] + 1e-9 && hB[r] - cum[r] <= 2.0 + 1e-9) { tight_in[P.grp_of[r]]++; spare += std::max(0.0, hB[r] - cum[r]); }
                colshort.assign(T.n_cols, 0.0); coltight.assign(T.n_cols, 0.0);
                for (uint32_t f = 0; f < T.n_cols; f++) for (uint32_t e = T.col_woff[f]; e < T.col_woff[f + 1]; e++) {
                    const int ns = short_in[T.w_row[e]], nt = tight_in[T.w_row[e]];
                    if (ns) colshort[f] += (double)ns * (double)T.w_coef[e];
                    if (nt) coltight[f] += (double)nt * (double)T.w_coef[e];
                }
                for (uint32_t b = 0; b < T.n_blocks; b++) {
                    for (int k : pool) {
                        if (k == chosen[b]) continue;
                        const uint16_t *px = pat_of(k);
                        // a cheap score first — what the switch takes off the short rows (its columns' entries in those rows only) against the value it gives
                        // up; the full look (every row, new violations) happens when the switch is about to be applied
                        double gain = 0.0, dv = 0.0, dmg = 0.0; bool differs = false;
                        for (uint32_t f = T.blk_off[b]; f < T.blk_off[b + 1]; f++) {
                            const int d = (int)px[f] - (int)x[f];
                            if (!d) continue;
                            differs = true; dv += T.col_cost[f] * (double)d;
                            gain -= colshort[f] * (double)d;  // (integers in f64: exact, whatever the grouping)
                            dmg += coltight[f] * (double)d;
                        }
                        if (!differs || !(gain - std::max(0.0, dmg - spare) > 1e-9)) continue;
                        cl.push_back({std::min(gain, base) * 10.0 * cmax + dv, b, k});
                    }
                }
                if (rq.trace) fprintf(stderr, "[price]     pass %d: short by %.1f, %zu candidate switches, at %.3f us\n", pass, base, cl.size(), now_us() - t_r0);
                if (cl.empty()) break;
                // (the best few hundred are all a pass ever applies: the rest wait for the next pass, which scores them against the totals as they are then)
                const size_t top = std::min<size_t>(cl.size(), 512);
                std::partial_sort(cl.begin(), cl.begin() + (long)top, cl.end(), [](const Cand &p, const Cand &q) { return p.score > q.score || (p.score == q.score && (p.b < q.b || (p.b == q.b && p.k < q.k))); });
                cl.resize(top);
                std::vector<char> moved(T.n_blocks, 0);
                for (const Cand &cd : cl) {
                    if (!(ge_short(cum) > 0.0)) break;
                    if (moved[cd.b]) continue;
                    {   // still worth the full look?  (the rows it was scored on may have been closed by the switches before it)
                        const uint16_t *px = pat_of(cd.k); double gain = 0.0;
                        for (uint32_t f = T.blk_off[cd.b]; f < T.blk_off[cd.b + 1]; f++) {
                            const int d = (int)px[f] - (int)x[f];
                            if (d) for (uint32_t e = T.col_woff[f]; e < T.col_woff[f + 1]; e++) { const int g = T.w_row[e]; for (int q = P.gr_off[g]; q < P.gr_off[g + 1]; q++) { const int r = P.gr_row[q]; if (P.ge[r] && cum[r] > hB[r] + 1e-9) gain -= (double)T.w_coef[e] * (double)d; } }
                        }
                        if (!(gain > 1e-9)) continue;
                    }
                    block_act(cd.b, pat_of(chosen[cd.b]), a0);
                    double score;
                    if (!evaluate_switch(cd.b, cd.k, value_of(cd.b), score)) continue;  // (a1 = the new pattern's activities)
                    for (int r = 0; r < K; r++) cum[r] += a1[r] - a0[r];
                    chosen[cd.b] = cd.k; moved[cd.b] = 1;
                    memcpy(&x[T.blk_off[cd.b]], pat_of(cd.k) + T.blk_off[cd.b], (size_t)(T.blk_off[cd.b + 1] - T.blk_off[cd.b]) * 2);
                }
                if (ge_short(cum) >= base - 1e-9) break;
            }
            if (ge_short(cum) > 0.0) return {};
        }
        if (rq.trace) fprintf(stderr, "[price]   rounding: >= repair done at %.3f us\n", now_us() - t_r0);
        // `<=` rows that came out over: take single tasks away, the cheapest first, never pushing a `>=` row short
        for (int r = 0; r < K; r++) {
            if (cum[r] <= hB[r] + 1e-9) continue;
            std::vector<int> cols;
            const int gr = P.grp_of[r];
            for (int k = P.g_off[gr]; k < P.g_off[gr + 1]; k++) if (P.g_coef[k] > 0 && x[P.g_col[k]] > 0) cols.push_back(k);
            std::sort(cols.begin(), cols.end(), [&](int a, int b) { const double ca = T.col_cost[P.g_col[a]] / P.g_coef[a], cb = T.col_cost[P.g_col[b]] / P.g_coef[b]; return ca < cb || (ca == cb && a > b); });
            for (int k : cols) {
                const int f = P.g_col[k];
                while (x[f] > (lo_of ? (uint16_t)(*lo_of)[f] : 0) && cum[r] > hB[r] + 1e-9) {
                    bool ok = true;
                    for (uint32_t e = T.col_woff[f]; e < T.col_woff[f + 1] && ok; e++) if (T.w_coef[e] < 0) { const int g = T.w_row[e]; for (int q = P.gr_off[g]; q < P.gr_off[g + 1] && ok; q++) if (cum[P.gr_row[q]] - (double)T.w_coef[e] > hB[P.gr_row[q]] + 1e-9) ok = false; }
                    if (!ok) break;
                    x[f]--;
                    for (uint32_t e = T.col_woff[f]; e < T.col_woff[f + 1]; e++) { const int g = T.w_row[e]; for (int q = P.gr_off[g]; q < P.gr_off[g + 1]; q++) cum[P.gr_row[q]] -= (double)T.w_coef[e]; }
                }
                if (cum[r] <= hB[r] + 1e-9) break;
            }
            if (cum[r] > hB[r] + 1e-9) return {};
        }
        return x;
    }

    // ------------------------------------------------------------------------------------------------ branch and price
    // What the root leaves open is the integrality of the FLAGS and of the few blocks the wide rows force to mix patterns.  Nodes fix flags (first) or
    // bound one block column (x <= floor / x >= ceil of its value in the node's LP point); a node's bound is the same Lagrangian, its sweeps run with the
    // column caps hi - lo and the block capacities minus what the lower bounds use, in variables shifted by lo; cuts of other nodes are reused wherever
    // their patterns respect the node's bounds.  Depth first, the child nearer to the LP point first; a node is closed when its bound is within rel_gap
    // of the incumbent.  All limits are counts (nodes, sweeps): replicas decide alike.
    struct NodeLP { double bound = INF; bool pruned = false; std::vector<double> lambda, pi, bfrac; };

    bool apply_node(const BPNode &nd, std::vector<int32_t> &lo_arr, std::vector<int32_t> &hi_arr) {
        const HostTables &T = P.T;
        lo_arr.assign(T.n_cols, 0); hi_arr = P.base_cap;
        for (auto &l : nd.lo) lo_arr[l.first] = std::max(lo_arr[l.first], l.second);
        for (auto &h : nd.hi) hi_arr[h.first] = std::min(hi_arr[h.first], h.second);
        node_caps.clear();
        for (const CapRow &cr : P.caps) {  // a conditional bound applies once every flag of its row is fixed (before that the node is relaxed by leaving it out)
            double r = cr.rhs; bool all_fixed = true;
            for (auto &t : cr.g) { if (nd.flag[t.first] < 0) { all_fixed = false; break; } r -= t.second * (double)nd.flag[t.first]; }
            if (!all_fixed) continue;
            const double c = std::floor(r + 1e-9);
            const int32_t cap = c < 0.0 ? 0 : (c > 65535.0 ? 65535 : (int32_t)c);
            if (cap < hi_arr[cr.flat]) { hi_arr[cr.flat] = cap; node_caps.push_back({(uint32_t)cr.flat, cap}); }
        }
        std::vector<int32_t> caps(T.n_cols); std::vector<double> bcap(T.blk_cap);
        node_lo.clear(); node_cl = 0.0; node_Al.assign(P.K, 0.0);
        for (uint32_t f = 0; f < T.n_cols; f++) {
            if (hi_arr[f] < lo_arr[f]) return false;
            caps[f] = hi_arr[f] - lo_arr[f];
            const int32_t l = lo_arr[f];
            if (l <= 0) continue;
            node_lo.push_back({f, l}); node_cl += T.col_cost[f] * (double)l;
            for (uint32_t e = T.col_woff[f]; e < T.col_woff[f + 1]; e++) { const int g = T.w_row[e]; for (int q = P.gr_off[g]; q < P.gr_off[g + 1]; q++) node_Al[P.gr_row[q]] += (double)T.w_coef[e] * (double)l; }
            const int b = P.block_of_flat[f];
            for (int r = 0; r < MMAX_BLOCK; r++) bcap[(size_t)b * MMAX_BLOCK + r] -= T.col_a[(size_t)f * MMAX_BLOCK + r] * (double)l;
        }
        for (double v : bcap) if (v < 0.0) return false;  // the lower bounds alone overfill a worker
        if (!sw.set_caps(caps.data()) || !sw.set_block_caps(bcap.data())) { failed = true; return false; }
        at_root = nd.lo.empty() && nd.hi.empty() && node_caps.empty();
        return true;
    }

    void node_master(const BPNode &nd, double cutoff, double tol, NodeLP &out, std::vector<char> &usable) {
        const int K = P.K, G = P.G;
        out = NodeLP(); out.bound = nd.bound;
        if (!fetch_patterns()) return;
        usable.assign(cuts.size(), 1);
        for (size_t k = 0; k < cuts.size(); k++) {
            const std::vector<uint16_t> &x = cuts[k].x;
            for (auto &l : nd.lo) if ((int32_t)x[l.first] < l.second) { usable[k] = 0; break; }
            if (usable[k]) for (auto &h : nd.hi) if ((int32_t)x[h.first] > h.second) { usable[k] = 0; break; }
            if (usable[k]) for (auto &h : node_caps) if ((int32_t)x[h.first] > h.second) { usable[k] = 0; break; }
        }
        std::vector<double> hN(P.h); double cN = 0.0;
        std::vector<int> freeg;
        for (int g = 0; g < G; g++) {
            if (nd.flag[g] < 0) { freeg.push_back(g); continue; }
            cN += P.gcost[g] * (double)nd.flag[g];
            for (auto &t : P.g_rows[g]) hN[t.first] -= t.second * (double)nd.flag[g];
        }
        const int F = (int)freeg.size();
        Rows M; M.n = K + 1 + F;
        std::vector<double> mc(M.n), mlb(M.n, 0.0), mub(M.n);
        for (int k = 0; k < K; k++) { mc[k] = -hN[k] / theta_scale; mub[k] = pmax[k]; }
        mc[K] = -1.0; mub[K] = 4.0;
        for (int i = 0; i < F; i++) { mc[K + 1 + i] = -1.0; mub[K + 1 + i] = 4.0; }
        std::vector<std::pair<int, double>> terms;
        std::vector<double> row_scale; std::vector<int> row_cut;
        for (int i = 0; i < F; i++) {  // mu_g + pi.A_g >= c_g
            const int g = freeg[i];
            terms.clear(); double sc = 1.0;
            for (auto &t : P.g_rows[g]) { const double v = t.second / theta_scale; terms.push_back({t.first, v}); sc = std::max(sc, std::fabs(v)); }
            terms.push_back({K + 1 + i, 1.0});
            for (auto &t : terms) t.second /= sc;
            M.add(terms, P.gcost[g] / theta_scale / sc, INF);
            row_scale.push_back(sc); row_cut.push_back(-1);
        }
        Tab mt;
        auto add_cut = [&](size_t k, bool live) {
            const Cut &c = cuts[k];
            terms.clear(); double sc = 1.0;
            for (int r = 0; r < K; r++) if (c.act[r] != 0) { const double v = (double)c.act[r] / theta_scale; terms.push_back({r, v}); sc = std::max(sc, std::fabs(v)); }
            terms.push_back({K, 1.0});
            for (auto &t : terms) t.second /= sc;
            M.add(terms, c.cx / theta_scale / sc, INF);
            row_scale.push_back(sc); row_cut.push_back((int)k);
            if (live) mt.where.push_back(-1);
        };
        for (size_t k = 0; k < cuts.size(); k++) if (usable[k]) add_cut(k, false);
        mt.init(&M, mc, mlb, mub);
        auto node_value = [&](const Cut &c) {  // the node's Lagrangian at the cut's prices
            double L = c.bnd + cN;
            for (int k = 0; k < K; k++) L += c.pi[k] * hN[k];
            for (int g : freeg) { double r = P.gcost[g]; for (auto &t : P.g_rows[g]) r -= c.pi[t.first] * t.second; if (r > 0.0) L += r; }
            return L;
        };
        std::vector<double> pi(K), pi_best, prev;
        bool ok = false;
        for (int it = 0; it < 80; it++) {
            ok = false;
            if (mt.solve(200000) != LP_OPT) break;
            ok = true;
            const double lbm = -mt.objective() * theta_scale + cN;
            if (out.bound <= cutoff) { out.pruned = true; break; }
            if (out.bound - lbm <= tol * std::fabs(out.bound)) break;
            bool same = !prev.empty();
            for (int k = 0; k < K && same; k++) same = std::fabs(mt.x[k] - prev[k]) <= 1e-15 + 1e-12 * std::fabs(mt.x[k]);
            prev.assign(mt.x.begin(), mt.x.begin() + K);
            const double alpha = (it < 2 || same || pi_best.empty()) ? 0.0 : 0.3;
            for (int k = 0; k < K; k++) pi[k] = (alpha > 0.0 ? alpha * pi_best[k] : 0.0) + (1.0 - alpha) * mt.x[k];
            const int ci = evaluate(pi);
            if (ci < 0) { ok = false; break; }
            usable.push_back(1);
            const double L = node_value(cuts[ci]);
            if (L < out.bound) { out.bound = L; pi_best = pi; }
            add_cut((size_t)ci, true);
        }
        work += mt.ops;
        if (!ok || out.pruned) return;
        out.lambda.assign(cuts.size(), 0.0); out.bfrac.assign(G, 0.0);
        double lsum = 0.0;
        for (int r = 0; r < M.m; r++) {
            const int a = mt.where[r];
            if (a < 0 || mt.st[M.n + a] == BASIC) continue;
            const double v = std::fabs(mt.d[M.n + a]) / row_scale[r];
            if (row_cut[r] >= 0) { out.lambda[row_cut[r]] = v; lsum += v; } else out.bfrac[freeg[r]] = std::min(1.0, v);
        }
        for (int g = 0; g < G; g++) if (nd.flag[g] >= 0) out.bfrac[g] = (double)nd.flag[g];
        if (lsum > 0.0) for (double &l : out.lambda) l /= lsum; else out.lambda.clear();
        out.pi.assign(mt.x.begin(), mt.x.begin() + K);
    }
};

}  // namespace

namespace {
// everything after the model has been flattened into S.P
Answer run_solver(Solver &S, const double tp0) {
    Answer ans;
    const Request &rq = S.rq; Sweeper &sw = S.sw;
    auto tmark = [&](const char *what) { if (rq.trace) fprintf(stderr, "[price] %s at %.3f ms (sweeps so far %.3f ms)\n", what, (now_us() - tp0) / 1e3, sw.stat_sweep_us / 1e3); };
    tmark("model flattened");
    Prob &P = S.P;
    if (rq.trace) fprintf(stderr, "[price] %u blocks, %u block columns, %d wide rows with %d distinct left-hand sides (%zu / %zu terms), %d flags, %zu conditional bounds\n", P.T.n_blocks, P.T.n_cols, P.K, P.KG, P.row_terms, P.T.w_row.size(), P.G, P.caps.size());
    const int K = P.K, G = P.G;
    sw.stat_sweeps = 0; sw.stat_sweep_us = 0;
    sw.guard_s = rq.deadline_s - 0.3 * rq.time_limit_s; sw.time_up = false;  // (first read at the first sweep)
    S.max_sweeps = (int)std::min<size_t>(4096, std::max<size_t>(256, ((size_t)64 << 20) / ((size_t)P.T.n_cols * 2 + 1)));  // the device keeps every sweep's patterns: at most 64 MB of them
    if (!sw.begin(P.T, (uint32_t)S.max_sweeps)) { ans.why = "sweeper refused the model"; return ans; }
    struct Ender { Sweeper &s; ~Ender() { s.end(); } } ender{sw};
    tmark("tables handed to the sweeper");
    // the first sweep runs at zero prices: it needs the tables and nothing else, so it is on its way while the host works out the price caps
    std::vector<double> pi0(K, 0.0);
    if (!S.evaluate_launch(pi0)) { ans.why = "sweep failed"; return ans; }
    // price caps: beyond pmax every column of the row has a negative reduced cost (`<=` rows); for `>=` rows a multiple of the largest cost per unit
    S.pmax.assign(K, 0.0);
    {
        double cmax = 0.0; for (double c : P.T.col_cost) cmax = std::max(cmax, c);
        // (per GROUP: the rows of a group share their left-hand side — one division per term of a list, not one per term of every row that carries it)
        std::vector<double> gp(P.KG, 0.0);
        for (int g = 0; g < P.KG; g++) {
            double p = 0.0; bool neg = false; int32_t amin = INT32_MAX;
            for (int t = P.g_off[g]; t < P.g_off[g + 1]; t++) {
                const int32_t a = P.g_coef[t];
                if (a > 0) p = std::max(p, P.T.col_cost[P.g_col[t]] / (double)a); else { neg = true; amin = std::min(amin, -a); }
            }
            if (neg) p = std::max(p, 64.0 * cmax / (double)std::max<int32_t>(1, amin));
            gp[g] = p;
        }
        for (int k = 0; k < K; k++) S.pmax[k] = gp[P.grp_of[k]];
    }
    tmark("price caps done");
    if (S.evaluate(pi0, true) < 0) { ans.why = "sweep failed"; return ans; }
    tmark("first sweep (prices 0) done");
    S.theta_scale = std::max(S.cuts[0].bnd, 1e-9);
    ans.ran = true;
    double best_value = rq.incumbent ? rq.incumbent_value : -INF;
    std::vector<double> hB(K);
    std::vector<int32_t> caps(P.base_cap);
    std::vector<double> final_pi;
    std::vector<std::vector<double>> tried;   // flag configurations already solved
    auto was_tried = [&](const std::vector<double> &B) { for (auto &t : tried) if (t == B) return true; return false; };
    auto certified = [&]() { return best_value > -INF && S.relaxed_bound <= best_value + rq.rel_gap * std::fabs(best_value); };
    // One flag configuration: master to convergence, one pattern per block, repair, the caller's polish.  Returns the point's value (-INF: nothing
    // usable came out) and leaves the point in x.
    std::vector<double> jump_to;   // set by try_config's probe: the configuration to go to instead (try_config returns -INF then)
    auto try_config = [&](const std::vector<double> &B, std::vector<double> &x) -> double {
        tried.push_back(B);
        // (the one clock of this path: read at the sweeps, and — where no other rank depends on the reading — here, so that the host-only stretches between sweeps
        // (master LPs, rounding, polish) cannot carry a tick past its guard either; a sharded sweeper merges the ranks' readings inside its exchange, nowhere else)
        if (!sw.merges_clock() && now_us() * 1e-6 > sw.guard_s) sw.time_up = true;
        if (sw.time_up) return -INF;
        ans.rounds++;
        double cB = 0.0;
        for (int k = 0; k < K; k++) hB[k] = P.h[k];
        for (int g = 0; g < G; g++) { cB += P.gcost[g] * B[g]; for (auto &t : P.g_rows[g]) hB[t.first] -= t.second * B[g]; }
        if (!P.caps.empty()) {  // conditional bounds with the flags fixed
            caps = P.base_cap;
            for (const CapRow &cr : P.caps) { double r = cr.rhs; for (auto &t : cr.g) r -= t.second * B[t.first]; const double c = std::floor(r + 1e-9); if (c < (double)caps[cr.flat]) caps[cr.flat] = c < 0.0 ? 0 : (int32_t)c; }
            if (!sw.set_caps(caps.data())) { S.failed = true; return -INF; }
            // (the cuts of earlier rounds are points of THEIR bounds: a round with other bounds starts a new master)
            S.base_caps = false; S.cut_lo = S.cuts.size();
            if (S.evaluate(pi0) < 0) return -INF;
        }
        std::vector<double> lambda, pi;
        double bound_B = INF;
        const double cutoff = best_value > -INF ? best_value * (1.0 - 1e-12) : -INF;
        const double tol = std::max(1e-6, rq.rel_gap / 50.0);
        // Once this configuration's master is within 10 %: would the flags its fractional point does not need, dropped, give a configuration whose master is ALREADY
        // converged on the cuts at hand, at a bound no worse?  Then that is where the walk goes — this configuration would be left for it anyway after its own
        // convergence, rounding and polish (the three-level C3 tick: 9 sweeps on the incumbent's flags, then a successor that converged without one more; now 3).
        jump_to.clear();
        S.probe = nullptr;
        if (G > 0 && P.caps.empty()) S.probe = [&](const std::vector<double> &colact, double lb_here) -> bool {
            std::vector<double> Bp = B, act = colact;
            for (int g = 0; g < G; g++) if (Bp[g] == 1.0) for (auto &t : P.g_rows[g]) act[t.first] += t.second;
            bool dropped = false;
            for (int g = 0; g < G; g++) {
                if (Bp[g] != 1.0 || P.gcost[g] != 0.0) continue;
                bool ok = true;
                for (auto &t : P.g_rows[g]) if (act[t.first] - t.second > P.h[t.first] + 1e-7 * (1.0 + std::fabs(P.h[t.first]))) ok = false;
                if (!ok) continue;
                Bp[g] = 0.0; dropped = true;
                for (auto &t : P.g_rows[g]) act[t.first] -= t.second;
            }
            if (!dropped || was_tried(Bp)) return false;
            std::vector<double> hBp(K); double cBp = 0.0;
            for (int k = 0; k < K; k++) hBp[k] = P.h[k];
            for (int g = 0; g < G; g++) { cBp += P.gcost[g] * Bp[g]; for (auto &t : P.g_rows[g]) hBp[t.first] -= t.second * Bp[g]; }
            std::vector<double> lam2, pi2; double b2 = INF;
            if (!S.kelley(hBp, cBp, -INF, tol, lam2, pi2, &b2, true) || b2 < lb_here) { S.settled.valid = false; return false; }
            if (rq.trace) fprintf(stderr, "[price] configuration %u left for the one its fractional point names: converged on the %zu cuts at hand, bound %.9f\n", ans.rounds, S.cuts.size(), b2);
            jump_to = Bp;
            return true;
        };
        const bool conv = S.kelley(hB, cB, cutoff, tol, lambda, pi, &bound_B);
        S.probe = nullptr;
        if (!conv) return -INF;
        tmark("master converged");
        final_pi = pi;
        std::vector<uint16_t> xf = S.round_patterns(lambda, pi, hB);
        tmark("patterns rounded");
        if (xf.empty()) return -INF;
        x.assign(rq.n, 0.0);
        for (uint32_t f = 0; f < P.T.n_cols; f++) x[P.model_of[f]] = (double)xf[f];
        for (int g = 0; g < G; g++) x[P.gmodel[g]] = B[g];
        double value = 0.0;
        if (!rq.polish || !rq.polish(x, value)) return -INF;  // the caller's rows say no
        tmark("polished");
        if (rq.trace) fprintf(stderr, "[price] configuration %u: %zu sweeps so far, bound with these flags %.9f, point %.9f, model bound %.9f\n", ans.rounds, S.cuts.size(), bound_B, value, S.relaxed_bound);
        if (value > best_value) { best_value = value; ans.x = x; ans.x_value = value; }
        // not certified by this point although the configuration's bound allows more: the same multipliers rounded in the other block order — a third of a
        // millisecond against the dozens of sweeps branch-and-price would spend on the same question (the layered DAG loop's 57-sweep ticks: 17)
        if (bound_B > best_value * (1.0 + rq.rel_gap)) {
            std::vector<uint16_t> xr = S.round_patterns(lambda, pi, hB, nullptr, nullptr, 1);
            if (!xr.empty()) {
                std::vector<double> x2(rq.n, 0.0);
                for (uint32_t f = 0; f < P.T.n_cols; f++) x2[P.model_of[f]] = (double)xr[f];
                for (int g = 0; g < G; g++) x2[P.gmodel[g]] = B[g];
                double v2 = 0.0;
                if (rq.polish(x2, v2)) {
                    if (rq.trace) fprintf(stderr, "[price] configuration %u, blocks in reverse order: point %.9f\n", ans.rounds, v2);
                    if (v2 > value) { value = v2; x = x2; }
                    if (v2 > best_value) { best_value = v2; ans.x = x2; ans.x_value = v2; }
                }
            }
        }
        return value;
    };
    // flags of x whose rows hold without them are dropped (they only ever restrict): the rule of milp.cpp's sparse_greedy.  True if B changed.
    auto drop_flags = [&](std::vector<double> &B, const std::vector<double> &x) {
        bool dropped = false;
        std::vector<double> act(K, 0.0);
        {
            std::vector<double> ga(P.KG, 0.0);
            for (int g = 0; g < P.KG; g++) for (int t = P.g_off[g]; t < P.g_off[g + 1]; t++) ga[g] += (double)P.g_coef[t] * x[P.model_of[P.g_col[t]]];
            for (int k = 0; k < K; k++) act[k] += ga[P.grp_of[k]];
        }
        for (int g = 0; g < G; g++) for (auto &t : P.g_rows[g]) act[t.first] += t.second * x[P.gmodel[g]];
        for (int g = 0; g < G; g++) {
            if (B[g] != 1.0 || P.gcost[g] != 0.0 || x[P.gmodel[g]] != 1.0) continue;
            bool ok = true;
            for (auto &t : P.g_rows[g]) if (act[t.first] - t.second > P.h[t.first] + 1e-7 * (1.0 + std::fabs(P.h[t.first]))) ok = false;
            if (!ok) continue;
            B[g] = 0.0; dropped = true;
            for (auto &t : P.g_rows[g]) act[t.first] -= t.second;
        }
        return dropped;
    };
    // a configuration and what dropping leads to from there
    auto descend = [&](std::vector<double> B) {
        for (int round = 0; round < MAX_ROUNDS && !S.failed && !was_tried(B); round++) {
            std::vector<double> x;
            const double v = try_config(B, x);
            if (!jump_to.empty()) { B = jump_to; jump_to.clear(); continue; }
            if (v == -INF) break;
            if (certified()) break;
            if (!drop_flags(B, x)) break;
        }
    };
    if (rq.incumbent && G > 0) {  // the caller's incumbent (milp.cpp's sparse_greedy: flags on, raise, drop the flags no longer needed, raise again) names a configuration
        std::vector<double> B(G);
        for (int g = 0; g < G; g++) B[g] = rq.incumbent[P.gmodel[g]] > 0.5 ? 1.0 : 0.0;
        descend(B);
    }
    if (ans.x.empty() && !S.failed) descend(std::vector<double>(G, 1.0));   // every flag on: always feasible for the tick's models (every lower-priority batch capped at its cut)
    if (S.failed) { ans.ran = false; ans.why = "sweep failed"; ans.x.clear(); return ans; }
    // Not certified against the bound seen so far and the model has flags: the master of the RELAXED model (flags in [0, 1]) may still bring the bound
    // down — its cuts are the points already evaluated, the flags enter through one extra variable each (mu_g >= c_g - pi.A_g, mu_g >= 0).  Its
    // multipliers of the flag rows are the flags' LP values: the configuration they round to, and single flags forced off in the order of those
    // values, are tried next ("place every higher-priority task of that class": often worth a little utilisation for what it unlocks).
    std::vector<double> Blp;
    if (G > 0 && best_value > -INF && !certified()) {
        if (!S.base_caps) {  // back to the model's own column bounds: only sweeps under those bound the relaxed model
            if (!sw.set_caps(P.base_cap.data())) { ans.ran = false; ans.why = "sweep failed"; ans.x.clear(); return ans; }
            S.base_caps = true; S.cut_lo = S.cuts.size();
            if (S.evaluate(pi0) < 0) { ans.ran = false; ans.why = "sweep failed"; ans.x.clear(); return ans; }
        }
        Rows M; M.n = K + 1 + G;
        std::vector<double> mc(M.n), mlb(M.n, 0.0), mub(M.n);
        for (int k = 0; k < K; k++) { mc[k] = -P.h[k] / S.theta_scale; mub[k] = S.pmax[k]; }
        mc[K] = -1.0; mub[K] = 4.0;
        for (int g = 0; g < G; g++) { mc[K + 1 + g] = -1.0; mub[K + 1 + g] = 4.0; }
        std::vector<std::pair<int, double>> terms;
        for (int g = 0; g < G; g++) {  // mu_g + pi.A_g >= c_g
            terms.clear(); double sc = 1.0;
            for (auto &t : P.g_rows[g]) { const double v = t.second / S.theta_scale; terms.push_back({t.first, v}); sc = std::max(sc, std::fabs(v)); }
            terms.push_back({K + 1 + g, 1.0});
            for (auto &t : terms) t.second /= sc;
            M.add(terms, P.gcost[g] / S.theta_scale / sc, INF);
        }
        size_t in_master = S.cut_lo;
        Tab mt;
        auto push_cuts = [&](bool live) {
            for (; in_master < S.cuts.size(); in_master++) {
                const Cut &c = S.cuts[in_master];
                terms.clear(); double sc = 1.0;
                for (int k = 0; k < K; k++) if (c.act[k] != 0) { const double v = (double)c.act[k] / S.theta_scale; terms.push_back({k, v}); sc = std::max(sc, std::fabs(v)); }
                terms.push_back({K, 1.0});
                for (auto &t : terms) t.second /= sc;
                M.add(terms, c.cx / S.theta_scale / sc, INF);
                if (live) mt.where.push_back(-1);
            }
        };
        push_cuts(false);
        mt.init(&M, mc, mlb, mub);
        std::vector<double> pi(K), pi_best = S.relaxed_pi.empty() ? std::vector<double>(K, 0.0) : S.relaxed_pi, prev;
        bool master_ok = false;
        for (int it = 0; it < 120; it++) {
            master_ok = false;
            if (mt.solve(200000) != LP_OPT) break;
            master_ok = true;
            const double lb_master = -mt.objective() * S.theta_scale;
            if (S.relaxed_bound <= best_value + rq.rel_gap * std::fabs(best_value)) break;   // certified
            if (S.relaxed_bound - lb_master <= 1e-6 * std::fabs(S.relaxed_bound)) break;      // the bound is what it is
            bool same = !prev.empty();
            for (int k = 0; k < K && same; k++) same = std::fabs(mt.x[k] - prev[k]) <= 1e-15 + 1e-12 * std::fabs(mt.x[k]);
            prev.assign(mt.x.begin(), mt.x.begin() + K);
            const double alpha = (it < 2 || same) ? 0.0 : 0.3;
            for (int k = 0; k < K; k++) pi[k] = alpha * pi_best[k] + (1.0 - alpha) * mt.x[k];
            const double before = S.relaxed_bound;
            if (S.evaluate(pi) < 0) break;
            if (S.relaxed_bound < before) pi_best = pi;
            push_cuts(true);
        }
        if (S.failed) { ans.ran = false; ans.why = "sweep failed"; ans.x.clear(); return ans; }
        if (master_ok || mt.solve(200000) == LP_OPT) {
            Blp.assign(G, 0.0);
            for (int g = 0; g < G; g++) { const int ar = mt.where[g]; if (ar >= 0 && mt.st[M.n + ar] != BASIC) { double sc = 1.0; for (auto &t : P.g_rows[g]) sc = std::max(sc, std::fabs(t.second / S.theta_scale)); Blp[g] = std::min(1.0, std::fabs(mt.d[M.n + ar]) / sc); } }
        }
    }
    if (!Blp.empty() && !certified() && P.caps.empty()) {
        std::vector<double> B(G);
        for (int g = 0; g < G; g++) B[g] = (Blp[g] > 0.5 || P.gcost[g] > 0.0) ? 1.0 : 0.0;
        descend(B);
        // single flags of the best configuration forced off, least LP value first
        for (int trial = 0; trial < 12 && !certified() && !S.failed && !ans.x.empty(); trial++) {
            std::vector<double> cur(G);
            for (int g = 0; g < G; g++) cur[g] = ans.x[P.gmodel[g]];
            int pick = -1; double pv = INF;
            for (int g = 0; g < G; g++) {
                if (cur[g] != 1.0 || P.gcost[g] != 0.0) continue;
                std::vector<double> t = cur; t[g] = 0.0;
                if (was_tried(t)) continue;
                if (Blp[g] < pv) { pv = Blp[g]; pick = g; }
            }
            if (pick < 0) break;
            cur[pick] = 0.0;
            descend(cur);
        }
        if (S.failed) { ans.ran = false; ans.why = "sweep failed"; ans.x.clear(); return ans; }
    }
    // ---- branch and price: what the configurations above leave open, on models small enough for a few hundred more sweeps to be cheap ----
    double bp_bound = INF;
    if (best_value > -INF && !certified() && !S.failed && P.T.n_cols <= 131072 && S.budget_sweeps < 8) {
        S.base_caps = true;  // every node sets its own column bounds from here on
        int col_nodes = 0;  // branching on single columns moves the bound of these models very little (a mixing worker passes its role to the next one): a short leash
        auto closes = [&](double bound) { return bound <= best_value + rq.rel_gap * std::fabs(best_value); };
        std::vector<BPNode> stack;
        { BPNode root; root.flag.assign(G, -1); root.bound = S.relaxed_bound; stack.push_back(std::move(root)); }
        double closed_max = -INF;  // the largest bound among the closed nodes: with the open ones it bounds the model
        int nodes = 0;
        std::vector<int32_t> lo_arr, hi_arr; std::vector<char> usable;
        while (!stack.empty() && nodes < BP_MAX_NODES && (int)S.cuts.size() + 8 < S.max_sweeps && S.work < BP_MAX_WORK * rq.time_limit_s && S.sweep_steps < BP_MAX_STEPS * rq.time_limit_s &&
               !sw.time_up && !S.failed) {
            BPNode nd = std::move(stack.back()); stack.pop_back();
            if (!sw.merges_clock() && now_us() * 1e-6 > sw.guard_s) { sw.time_up = true; stack.push_back(std::move(nd)); break; }
            if (closes(nd.bound)) { closed_max = std::max(closed_max, nd.bound); continue; }
            nodes++;
            if (!S.apply_node(nd, lo_arr, hi_arr)) { if (S.failed) break; continue; }  // empty node
            Solver::NodeLP lp;
            S.node_master(nd, best_value, std::max(2e-6, rq.rel_gap / 20.0), lp, usable);
            if (S.failed) break;
            if (lp.pruned || closes(lp.bound)) { closed_max = std::max(closed_max, std::min(lp.bound, nd.bound)); continue; }
            if (lp.lambda.empty()) { stack.push_back(std::move(nd)); break; }  // no usable multipliers (master trouble): leave the node open
            nd.bound = std::min(nd.bound, lp.bound);
            // flags first: the free flag whose LP value is least decided
            int bg = -1; double bd = 1e-6;
            for (int g = 0; g < G; g++) if (nd.flag[g] < 0) { const double d = std::min(lp.bfrac[g], 1.0 - lp.bfrac[g]); if (d > bd) { bd = d; bg = g; } }
            if (bg < 0) {  // every flag is 0 or 1 in the LP point: the node's patterns make a point
                std::vector<double> Bn(G), hBn(P.h);
                for (int g = 0; g < G; g++) { Bn[g] = lp.bfrac[g] > 0.5 ? 1.0 : 0.0; for (auto &t : P.g_rows[g]) hBn[t.first] -= t.second * Bn[g]; }
                std::vector<double> lam_parts(lp.lambda.size() * PARTS);
                for (size_t k = 0; k < lp.lambda.size(); k++) for (int p = 0; p < PARTS; p++) lam_parts[k * PARTS + p] = lp.lambda[k];
                std::vector<uint16_t> xf = S.round_patterns(lam_parts, lp.pi, hBn, &usable, &lo_arr);
                if (!xf.empty()) {
                    std::vector<double> x(rq.n, 0.0);
                    for (uint32_t f = 0; f < P.T.n_cols; f++) x[P.model_of[f]] = (double)xf[f];
                    for (int g = 0; g < G; g++) x[P.gmodel[g]] = Bn[g];
                    double value = 0.0;
                    if (rq.polish && rq.polish(x, value) && value > best_value) { best_value = value; ans.x = x; ans.x_value = value; }
                }
                if (closes(nd.bound)) { closed_max = std::max(closed_max, nd.bound); continue; }
            }
            BPNode a = nd, b = nd; a.depth = b.depth = nd.depth + 1;
            if (bg >= 0) {
                const int8_t first = lp.bfrac[bg] > 0.5 ? 1 : 0;
                a.flag[bg] = first; b.flag[bg] = (int8_t)(1 - first);
            } else {  // the block column whose LP value is furthest from an integer (cost-weighted)
                if (++col_nodes > 48) { stack.push_back(std::move(nd)); break; }
                int bf = -1; double bs = 1e-7, bv = 0.0;
                std::vector<double> xl(P.T.n_cols, 0.0);
                for (size_t k = 0; k < S.cuts.size(); k++) { const double l = k < lp.lambda.size() ? lp.lambda[k] : 0.0; if (l <= 0.0 || !usable[k]) continue; const uint16_t *px = S.cuts[k].x.data(); for (uint32_t f = 0; f < P.T.n_cols; f++) xl[f] += l * (double)px[f]; }
                for (uint32_t f = 0; f < P.T.n_cols; f++) { const double fr = xl[f] - std::floor(xl[f]); const double sc = std::min(fr, 1.0 - fr) * (P.T.col_cost[f] + 1e-9); if (std::min(fr, 1.0 - fr) > 1e-6 && sc > bs) { bs = sc; bf = (int)f; bv = xl[f]; } }
                if (bf < 0) { closed_max = std::max(closed_max, nd.bound); continue; }  // an integral LP point that still did not close: nothing to branch on (its value IS the bound)
                const int32_t fl = (int32_t)std::floor(bv);
                BPNode dn = nd, up = nd; dn.depth = up.depth = nd.depth + 1;
                dn.hi.push_back({(uint32_t)bf, fl}); up.lo.push_back({(uint32_t)bf, fl + 1});
                if (bv - fl < 0.5) { a = std::move(dn); b = std::move(up); } else { a = std::move(up); b = std::move(dn); }
            }
            stack.push_back(std::move(b)); stack.push_back(std::move(a));  // (a = the child nearer to the LP point: popped first)
        }
        bp_bound = closed_max;
        for (const BPNode &nd : stack) bp_bound = std::max(bp_bound, nd.bound);
        if (rq.trace) fprintf(stderr, "[price] branch and price: %d nodes, %zu left open, %zu sweeps in all, bound %.9f, incumbent %.9f\n", nodes, stack.size(), S.cuts.size(), bp_bound, best_value);
        // back to the model's own bounds (a later caller of this sweeper starts from begin() anyway)
        S.node_lo.clear(); S.at_root = true;
        if (S.failed) { ans.ran = false; ans.why = "sweep failed"; ans.x.clear(); return ans; }
    }
    ans.bound = std::min(S.relaxed_bound, bp_bound > -INF ? bp_bound : INF);
    ans.sweeps = (uint32_t)sw.stat_sweeps;
    // for the host's guided windows: blocks, their values and the reduced costs at the last prices
    // (only what the sweeps leave open goes on to the host's windows: a certified answer needs neither)
    const bool closed = best_value > -INF && ans.bound <= best_value + rq.rel_gap * std::fabs(best_value);
    if (!final_pi.empty() && !closed) {
        ans.block_of.assign(rq.n, -1); ans.rcost.assign(rq.n, 0.0);
        std::vector<std::pair<int, double>> rows;
        for (uint32_t f = 0; f < P.T.n_cols; f++) {
            const int j = P.model_of[f];
            ans.block_of[j] = P.block_of_flat[f];
            double r = P.T.col_cost[f];
            rows.clear();   // the column's rows in ascending order (the order the prices are subtracted in is part of the value's last bits)
            for (uint32_t e = P.T.col_woff[f]; e < P.T.col_woff[f + 1]; e++) { const int g = P.T.w_row[e]; for (int q = P.gr_off[g]; q < P.gr_off[g + 1]; q++) rows.push_back({P.gr_row[q], (double)P.T.w_coef[e]}); }
            std::sort(rows.begin(), rows.end());
            for (auto &t : rows) r -= final_pi[t.first] * t.second;
            ans.rcost[j] = r;
        }
    }
    return ans;
}
}  // namespace

namespace {

thread_local bool g_check_hints = false; thread_local int g_hint_mismatches = 0;
// ---- the model as its builder wrote it -> blocks + wide rows (what build() does for a scaled component copy; same tables, same numbering) -------------------------
// Returns nullptr on success, else what keeps the model on the classic path.  ub: the derived column bounds (model columns), c: obj / cmax.
const char *build_from_model(const ModelView &mv, Prob &P, std::vector<double> &ub, std::vector<double> &c, double &cmax) {
    const int n = mv.n, m = mv.m;
    if (!mv.col_group || !mv.row_lhs || !mv.row_lhs_len) return "no structure hints";
    // blocks in order of their first column (the tick: worker order)
    int max_group = -1;
    for (int j = 0; j < n; j++) max_group = std::max(max_group, mv.col_group[j]);
    std::vector<int> block_of_group((size_t)max_group + 1, -1), bsize, blk(n, -1);
    for (int j = 0; j < n; j++) {
        const int g = mv.col_group[j];
        if (g < 0) { P.gmodel.push_back(j); continue; }
        if (block_of_group[g] < 0) { block_of_group[g] = (int)bsize.size(); bsize.push_back(0); }
        blk[j] = block_of_group[g];
        if (++bsize[blk[j]] > NMAX_BLOCK) return "block with more than 32 columns";
    }
    const int nb = (int)bsize.size();
    if (nb < 8) return "fewer than 8 blocks";
    cmax = 0.0;
    for (int j = 0; j < n; j++) cmax = std::max(cmax, std::fabs(mv.obj[j]));
    if (cmax == 0.0) cmax = 1.0;
    c.resize(n);
    for (int j = 0; j < n; j++) c[j] = mv.obj[j] / cmax;
    P.G = (int)P.gmodel.size();
    for (int j : P.gmodel) {
        if (mv.kind[j] != 1) return "global column that is not 0/1";
        if (c[j] < 0.0) return "global column with negative cost";
        P.gcost.push_back(c[j]);
    }
    std::vector<int> gidx(n, -1);
    for (int g = 0; g < P.G; g++) gidx[P.gmodel[g]] = g;
    // ---- column bounds, as hqmilp::solve derives them: a row without a negative coefficient bounds every column it holds by floor(rhs / coef + 1e-9).  A shared list of
    // leading terms is walked once, against the smallest right-hand side of its rows (the bound is monotone in the right-hand side).
    const int n_lhs = mv.n_lists;
    for (int i = 0; i < m; i++) if (mv.row_lhs[i] >= n_lhs) return "bad structure hint";
    struct Fam { int first = -1, len = 0; bool has_neg = false, scanned = false; double rhs_min = INF;
                 // of the list's block columns: their block (-2 none yet), more than one block, a global column inside, count, integer coefficients
                 int b0 = -2; bool multi = false, has_g = false, integral = true; int n_bcols = 0; double amax = 0.0; };
    std::vector<Fam> fam((size_t)n_lhs);
    ub.assign(n, INF);
    for (int j = 0; j < n; j++) if (mv.kind[j] == 1) ub[j] = 1.0;
    const bool hinted = mv.row_block && mv.col_ub;   // the builder's own bounds of the block columns: the rows of single blocks need not be walked for theirs
    if (hinted) for (int j = 0; j < n; j++) if (mv.col_ub[j] != UINT32_MAX) ub[j] = std::min(ub[j], (double)mv.col_ub[j]);
    auto bound_terms = [&](int a, int e, double rhs) { for (int k = a; k < e; k++) if (mv.rcoef[k] > 0.0) { double &u = ub[mv.rcol[k]]; u = std::min(u, std::floor(rhs / mv.rcoef[k] + 1e-9)); } };
    for (int l = 0; l < n_lhs; l++) for (int k = mv.list_off[l]; k < mv.list_off[l + 1]; k++) if (mv.list_col[k] < 0 || mv.list_col[k] >= n) return "bad structure hint";
    for (int i = 0; i < m; i++) {
        const int a = mv.roff[i], e = mv.roff[i + 1];
        if (a == e && mv.row_lhs[i] < 0) { const double b = mv.rhs[i]; const bool ok = mv.rtype[i] == 1 ? b >= -1e-9 : (mv.rtype[i] == 0 ? b <= 1e-9 : std::fabs(b) <= 1e-9); if (!ok) return "infeasible empty row"; continue; }
        if (hinted && mv.row_block[i] >= 0) continue;
        const int L = mv.row_lhs[i];
        const int tail = a;   // (a shared list's terms are not among the row's stored ones: those are all tail)
        bool neg = false;
        if (L >= 0) { Fam &f = fam[(size_t)L]; if (f.first < 0) { f.first = i; f.len = mv.list_off[L + 1] - mv.list_off[L]; } }
        for (int k = tail; k < e; k++) if (mv.rcoef[k] < 0.0) neg = true;
        if (mv.rtype[i] == 0 || neg) continue;   // a `>=` row, or a negative coefficient: no bound from here (the multi-node rows that do bound through one are not for this path)
        if (mv.rhs[i] < -1e-9) return "infeasible row";
        if (L >= 0) fam[(size_t)L].rhs_min = std::min(fam[(size_t)L].rhs_min, mv.rhs[i]);
        bound_terms(tail, e, mv.rhs[i]);
    }
    for (int l = 0; l < n_lhs; l++) if (fam[(size_t)l].first >= 0 && fam[(size_t)l].rhs_min < INF) {   // coefficient 1: the bound is the right-hand side itself
        const double bnd = std::floor(fam[(size_t)l].rhs_min + 1e-9);
        for (int k = mv.list_off[l]; k < mv.list_off[l + 1]; k++) { double &u = ub[mv.list_col[k]]; u = std::min(u, bnd); }
    }
    if (hinted && g_check_hints) {   // tests: the builder's column bounds must be what the skipped single-block rows give — not tighter (a point lost), not looser
        std::vector<double> chk(n, INF);
        for (int j = 0; j < n; j++) if (mv.kind[j] == 1) chk[j] = 1.0;
        for (int i = 0; i < m; i++) {
            if (mv.row_block[i] < 0 || mv.rtype[i] == 0) continue;
            bool neg = false;
            for (int k = mv.roff[i]; k < mv.roff[i + 1]; k++) if (mv.rcoef[k] < 0.0) neg = true;
            if (neg) continue;
            for (int k = mv.roff[i]; k < mv.roff[i + 1]; k++) if (mv.rcoef[k] > 0.0) chk[mv.rcol[k]] = std::min(chk[mv.rcol[k]], std::floor(mv.rhs[i] / mv.rcoef[k] + 1e-9));
        }
        for (int j = 0; j < n; j++) if (mv.col_ub[j] != UINT32_MAX && chk[j] < INF && (double)mv.col_ub[j] != chk[j]) { g_hint_mismatches++; return "structure hint: col_ub differs from what the block's rows give"; }
    }
    
    HostTables &T = P.T;
    T.n_blocks = (uint32_t)nb;
    T.blk_off.assign((size_t)nb + 1, 0);
    for (int b = 0; b < nb; b++) T.blk_off[b + 1] = T.blk_off[b] + (uint32_t)bsize[b];
    T.n_cols = T.blk_off[nb];
    P.flat_of.assign(n, -1); P.model_of.assign(T.n_cols, -1); P.block_of_flat.assign(T.n_cols, -1);
    {
        std::vector<uint32_t> cur(T.blk_off.begin(), T.blk_off.end() - 1);
        for (int j = 0; j < n; j++) if (blk[j] >= 0) { const int f = (int)cur[blk[j]]++; P.flat_of[j] = f; P.model_of[f] = j; P.block_of_flat[f] = blk[j]; }
    }
    T.col_cost.assign(T.n_cols, 0.0); T.col_a.assign((size_t)T.n_cols * MMAX_BLOCK, 0.0); T.col_cap.assign(T.n_cols, 0);
    T.blk_m.assign(nb, 0); T.blk_cap.assign((size_t)nb * MMAX_BLOCK, 0.0);
    for (uint32_t f = 0; f < T.n_cols; f++) {
        const int j = P.model_of[f];
        if (c[j] < 0.0) return "negative cost";
        if (!(ub[j] <= 65535.0)) return "column bound above 65535";
        T.col_cost[f] = c[j];
        T.col_cap[f] = (int32_t)std::floor(ub[j] + 1e-9);
    }
    // ---- the rows
    struct WideRow { double h; uint8_t ge; int lhs_key; int own = -1; std::vector<std::pair<int, double>> g; };   // lhs_key: 2 * L + (`>=` row), or -1 with its own list `own`
    std::vector<WideRow> wide;
    std::vector<std::vector<std::pair<int, int32_t>>> own_lists;   // left-hand sides of wide rows outside every family (sign applied)
    struct RowMemo { int n = -1; long long g = 1; double raw[32]; double scaled[64]; };
    RowMemo row_memo[8], memo_big; unsigned memo_next = 0;
    // what a range of terms looks like to the classification below
    struct Scan { int b0 = -2; bool multi = false, has_g = false, nonneg = true; int n_bcols = 0; double amax = 0.0; };
    auto scan = [&](int a, int e, Scan &sc) {
        for (int k = a; k < e; k++) {
            const int j = mv.rcol[k];
            if (mv.rcoef[k] < 0.0) sc.nonneg = false;
            sc.amax += mv.rcoef[k] * ub[j];
            if (blk[j] < 0) { sc.has_g = true; continue; }
            sc.n_bcols++;
            if (sc.b0 == -2) sc.b0 = blk[j]; else if (blk[j] != sc.b0) sc.multi = true;
        }
    };
    for (int i = 0; i < m; i++) {
        const int a = mv.roff[i], e = mv.roff[i + 1];
        if (a == e && mv.row_lhs[i] < 0) continue;
        const bool is_le = mv.rtype[i] == 1, is_ge = mv.rtype[i] == 0;
        const double rhs = mv.rhs[i];
        const int L = mv.row_lhs[i];
        Scan sc; const int tail = a;
        bool fam_row = false;
        if (L >= 0) {
            Fam &f = fam[(size_t)L];
            if (!f.scanned) {   // the list itself, once: which blocks its columns are of, what it can reach at most (coefficient 1 each)
                f.scanned = true;
                for (int k = mv.list_off[L]; k < mv.list_off[L + 1]; k++) {
                    const int j = mv.list_col[k];
                    f.amax += ub[j];
                    if (blk[j] < 0) { f.has_g = true; continue; }
                    f.n_bcols++;
                    if (f.b0 == -2) f.b0 = blk[j]; else if (blk[j] != f.b0) f.multi = true;
                }
            }
            if (!(f.multi && !f.has_g)) return "shared list that is not a wide left-hand side";   // (one block's list, or a flag inside it: the plain form's business)
            sc.b0 = f.b0; sc.multi = true; sc.has_g = false; sc.nonneg = true; sc.n_bcols = f.n_bcols; sc.amax = f.amax;
            fam_row = true;
        }
        const int hint_b = (hinted && !fam_row && mv.row_block[i] >= 0 && mv.row_block[i] <= max_group) ? block_of_group[mv.row_block[i]] : -1;
        if (hint_b >= 0) {   // the builder says: one block's row — no need to look its columns' blocks up
            if (mv.row_implied && mv.row_implied[i]) continue;
            sc.b0 = hint_b; sc.n_bcols = e - a;
            const int32_t hb = mv.row_block[i];
            for (int k = a; k < e; k++) {
                if (mv.col_group[mv.rcol[k]] != hb) return "structure hint: a row_block row holds a column of another block";   // (hints are checked, not trusted)
                if (mv.rcoef[k] < 0.0) sc.nonneg = false;
                sc.amax += mv.rcoef[k] * ub[mv.rcol[k]];
            }
        } else scan(tail, e, sc);   // (a row with a shared list: its own terms behind the list; any other row: all of it)
        if (!sc.multi && !sc.has_g && sc.b0 >= 0) {  // a row of one block
            if (mv.row_implied && mv.row_implied[i]) continue;  // implied for integer points by the block's other rows: the sweeps solve the blocks in integers
            if (is_le && sc.nonneg && sc.amax <= rhs * (1.0 + 1e-12) + 1e-9) continue;  // no point within the column bounds can violate it
            if (!is_le || !sc.nonneg || rhs < 0.0) return "block row that is not a packing row";
            const int b0 = sc.b0, r = T.blk_m[b0];
            if (r >= MMAX_BLOCK) return "block with more than 4 rows";
            // identical workers repeat the same few coefficient lists: a list seen before (compared as the doubles it is) has its grid integers and their gcd ready —
            // no product, rounding and 64-bit remainder per term
            const int nt = e - a;
            const RowMemo *hit = nullptr;
            for (const RowMemo &rm : row_memo) if (rm.n == nt && memcmp(rm.raw, mv.rcoef + a, (size_t)nt * sizeof(double)) == 0) { hit = &rm; break; }
            if (!hit) {
                if (nt > 64) return "block row with more than 64 terms";
                long long vi[64], g = 0;
                for (int k = 0; k < nt; k++) {
                    const double v = mv.rcoef[a + k] * GRID, rv = round_fast(v);
                    if (rv < 1.0 || std::fabs(v - rv) > 1e-6 * std::max(1.0, rv) || rv >= 4.0e15) return "block row off the ResourceAmount grid";
                    vi[k] = (long long)rv;
                    g = gcd_ll(g, vi[k]);
                }
                if (g < 1) g = 1;
                RowMemo &rm = nt <= 32 ? row_memo[memo_next++ % 8] : memo_big;
                rm.n = nt <= 32 ? nt : -1; rm.g = g;
                if (nt <= 32) memcpy(rm.raw, mv.rcoef + a, (size_t)nt * sizeof(double));
                for (int k = 0; k < nt; k++) rm.scaled[k] = (double)(vi[k] / g);
                hit = &rm;
            }
            const long long g = hit->g;
            const double capv = rhs * GRID;
            if (capv >= 4.0e15) return "block row capacity too large";
            const long long capi = (long long)std::floor(capv + 1e-6);
            for (int k = 0; k < nt; k++) T.col_a[(size_t)P.flat_of[mv.rcol[a + k]] * MMAX_BLOCK + r] += hit->scaled[k];  // (duplicate terms of one row are summed)
            T.blk_cap[(size_t)b0 * MMAX_BLOCK + r] = (double)(capi / g);
            T.blk_m[b0] = (uint8_t)(r + 1);
            continue;
        }
        if (!is_le && !is_ge) return "equality or range row across blocks";
        if (is_le && sc.nonneg && sc.amax <= rhs * (1.0 + 1e-12) + 1e-9) continue;   // vacuous within the column bounds
        if (is_ge && sc.nonneg && rhs <= 1e-12) continue;                            // holds at zero
        if (!sc.multi && sc.has_g && sc.n_bcols >= 1 && is_le && sc.nonneg) {  // one block + flags: a conditional bound of the block's column(s)
            if (sc.n_bcols != 1) return "conditional bound over several columns of a block";
            CapRow cr; cr.flat = -1; cr.rhs = 0.0;
            double cx = 0.0;
            for (int k = a; k < e; k++) if (blk[mv.rcol[k]] >= 0) { cr.flat = P.flat_of[mv.rcol[k]]; cx = mv.rcoef[k]; }
            if (!(cx > 0.0)) return "conditional bound with a zero coefficient";
            cr.rhs = rhs / cx;
            for (int k = a; k < e; k++) if (blk[mv.rcol[k]] < 0) cr.g.push_back({gidx[mv.rcol[k]], mv.rcoef[k] / cx});
            P.caps.push_back(std::move(cr));
            continue;
        }
        if (sc.n_bcols == 0) return "row over global columns only";
        WideRow w; w.ge = is_ge ? 1 : 0;
        const double sign = is_le ? 1.0 : -1.0;
        w.h = sign * rhs;
        if (fam_row) {
            w.lhs_key = 2 * L + (is_ge ? 1 : 0);
        } else { w.lhs_key = -1; w.own = (int)own_lists.size(); own_lists.emplace_back(); own_lists.back().reserve((size_t)(e - a)); }
        for (int k = tail; k < e; k++) {
            const int j = mv.rcol[k];
            const double v = sign * mv.rcoef[k];
            if (blk[j] < 0) { w.g.push_back({gidx[j], v}); continue; }
            if (fam_row) return "block column behind a shared list";
            const double rv = round_fast(v);
            if (std::fabs(v - rv) > 1e-7 * std::max(1.0, std::fabs(rv)) || std::fabs(rv) > 1.0e9) return "wide row with a non-integer coefficient";
            if (rv != 0.0) own_lists.back().push_back({P.flat_of[j], (int32_t)rv});
        }
        P.row_terms += (size_t)(e - a) - w.g.size() + (fam_row ? (size_t)fam[(size_t)L].len : 0);
        wide.push_back(std::move(w));
        if ((int)wide.size() > 1024) return "more than 1024 wide rows";
    }
    
    for (int b = 0; b < nb; b++) if (T.blk_m[b] == 0) {   // every resource row of this worker is slack at its column bounds: a never-binding row (the kernel wants one)
        double total = 0.0;
        for (uint32_t f = T.blk_off[b]; f < T.blk_off[b + 1]; f++) { T.col_a[(size_t)f * MMAX_BLOCK] = 1.0; total += (double)T.col_cap[f]; }
        T.blk_cap[(size_t)b * MMAX_BLOCK] = total;
        T.blk_m[b] = 1;
    }
    P.K = (int)wide.size();
    if (P.K == 0) return "no wide row";
    P.h.resize(P.K); P.ge.resize(P.K); P.g_rows.assign(P.G, {});
    // groups: rows with the same (list, sign) share one; lists of different families with the same content are merged too (build() does, by content), and the groups are
    // numbered in the order of their first row — the numbering build() arrives at
    P.grp_of.assign(P.K, -1);
    std::vector<std::vector<std::pair<int, int32_t>>> lhs;
    std::vector<uint64_t> lhs_hash;
    std::vector<int> group_of_key((size_t)2 * n_lhs, -1);
    auto materialise = [&](int key, std::vector<std::pair<int, int32_t>> &out) {   // the family's list with the row's sign, zero coefficients dropped
        const int l = key >> 1;
        const int32_t sign = (key & 1) ? -1 : 1;
        const int k0 = mv.list_off[l], k1 = mv.list_off[l + 1];
        out.resize((size_t)(k1 - k0));
        std::pair<int, int32_t> *o = out.data(); const int *fo = P.flat_of.data();
        for (int k = k0; k < k1; k++) o[k - k0] = {fo[mv.list_col[k]], sign};
    };
    auto hash_of = [](const std::vector<std::pair<int, int32_t>> &l) { uint64_t h = 1469598103934665603ull; for (auto &t : l) h = (h ^ ((uint64_t)(uint32_t)t.first | ((uint64_t)(uint32_t)t.second << 32))) * 1099511628211ull; return h; };
    std::vector<std::pair<int, int32_t>> tmp;
    for (int k = 0; k < P.K; k++) {
        WideRow &w = wide[(size_t)k];
        P.h[k] = w.h; P.ge[k] = w.ge;
        for (auto &t : w.g) P.g_rows[t.first].push_back({k, t.second});
        if (w.lhs_key >= 0 && group_of_key[(size_t)w.lhs_key] >= 0) { P.grp_of[k] = group_of_key[(size_t)w.lhs_key]; continue; }
        std::vector<std::pair<int, int32_t>> *l;
        if (w.lhs_key >= 0) { materialise(w.lhs_key, tmp); l = &tmp; } else l = &own_lists[(size_t)w.own];
        const uint64_t hsh = hash_of(*l);
        int g = -1;
        for (size_t i = 0; i < lhs.size() && g < 0; i++) if (lhs_hash[i] == hsh && lhs[i] == *l) g = (int)i;
        if (g < 0) { g = (int)lhs.size(); lhs.push_back(std::move(*l)); lhs_hash.push_back(hsh); }
        if (w.lhs_key >= 0) group_of_key[(size_t)w.lhs_key] = g;
        P.grp_of[k] = g;
    }
    P.KG = (int)lhs.size();
    if (P.KG > KMAX_HOST) return "more than 128 distinct wide left-hand sides";
    
    finish_groups(P, lhs);
    
    P.base_cap = T.col_cap;
    return nullptr;
}

// The candidate point of a flag configuration against the flattened model, raised greedily (most valuable columns first) as far as the blocks' rows, the wide rows
// and the conditional bounds allow — what CompSolver::polish_point does on the component's scaled rows.  The integers are exact here: block rows on their own grid,
// wide rows with integer coefficients.  x: the model's columns (in / out).
struct Polisher {
    const Prob &P; int n;
    explicit Polisher(const Prob &p, int n_) : P(p), n(n_) {}
    bool run(std::vector<double> &x, double &value) const {
        const HostTables &T = P.T; const int K = P.K, G = P.G;
        if ((int)x.size() != n) return false;
        std::vector<double> xf(T.n_cols), B(G);
        for (uint32_t f = 0; f < T.n_cols; f++) { const double v = x[P.model_of[f]], rv = round_fast(v); if (v < -1e-9 || v > (double)P.base_cap[f] + 1e-9 || std::fabs(v - rv) > 1e-9) return false; xf[f] = rv; }
        for (int g = 0; g < G; g++) { B[g] = x[P.gmodel[g]]; if (B[g] != 0.0 && B[g] != 1.0) return false; }
        // block rows
        std::vector<double> bact((size_t)T.n_blocks * MMAX_BLOCK, 0.0);
        for (uint32_t b = 0; b < T.n_blocks; b++) {
            double *ba = &bact[(size_t)b * MMAX_BLOCK];
            for (uint32_t f = T.blk_off[b]; f < T.blk_off[b + 1]; f++) if (xf[f] != 0.0) for (int r = 0; r < MMAX_BLOCK; r++) ba[r] += T.col_a[(size_t)f * MMAX_BLOCK + r] * xf[f];
            for (int r = 0; r < (int)T.blk_m[b]; r++) if (ba[r] > T.blk_cap[(size_t)b * MMAX_BLOCK + r]) return false;
        }
        // wide rows: activity per group, the flags' part per row
        std::vector<double> ga(P.KG, 0.0), fl(K, 0.0);
        for (int g = 0; g < P.KG; g++) for (int t = P.g_off[g]; t < P.g_off[g + 1]; t++) ga[g] += (double)P.g_coef[t] * xf[P.g_col[t]];
        for (int g = 0; g < G; g++) if (B[g] != 0.0) for (auto &t : P.g_rows[g]) fl[t.first] += t.second * B[g];
        for (int k = 0; k < K; k++) if (ga[P.grp_of[k]] + fl[k] > P.h[k] + 1e-9 * std::max(1.0, std::fabs(P.h[k]))) return false;
        // conditional bounds with these flags
        std::vector<double> cap(T.n_cols);
        for (uint32_t f = 0; f < T.n_cols; f++) cap[f] = (double)P.base_cap[f];
        for (const CapRow &cr : P.caps) { double r = cr.rhs; for (auto &t : cr.g) r -= t.second * B[t.first]; const double cc = std::floor(r + 1e-9); if (xf[cr.flat] > cc) return false; cap[cr.flat] = std::min(cap[cr.flat], cc); }
        // The raise goes through the columns by descending cost (ties: ascending model column — CompSolver::polish_point's order).  Raising only ever uses room up, so
        // a column whose own block has no room for one more of it NOW never gets any: the few that do are found first, and only those are ordered.
        std::vector<int> order;
        // (the same holds for the wide rows: a group whose tightest row has no room for the column's coefficient rules the column out for good — on an unsaturated
        // tick the batch-size rows are tight after the rounding and nearly every worker has room: without this test 60 000 columns were candidates, for nothing)
        std::vector<double> gslack(P.KG, INF);
        for (int k = 0; k < K; k++) { const int g = P.grp_of[k]; gslack[g] = std::min(gslack[g], P.h[k] - (ga[g] + fl[k])); }
        for (uint32_t b = 0; b < T.n_blocks; b++) {
            const double *ba = &bact[(size_t)b * MMAX_BLOCK], *bc = &T.blk_cap[(size_t)b * MMAX_BLOCK];
            const int mb = (int)T.blk_m[b];
            for (uint32_t f = T.blk_off[b]; f < T.blk_off[b + 1]; f++) {
                if (!(T.col_cost[f] > 0.0) || cap[f] - xf[f] < 1.0) continue;
                bool room = true;
                for (uint32_t e = T.col_woff[f]; e < T.col_woff[f + 1] && room; e++) { const double a = (double)T.w_coef[e]; if (a > 0.0 && gslack[T.w_row[e]] < 0.5 * a) room = false; }
                for (int r = 0; r < mb && room; r++) { const double a = T.col_a[(size_t)f * MMAX_BLOCK + r]; if (a > 0.0 && bc[r] - ba[r] < a) room = false; }
                if (room) order.push_back((int)f);
            }
        }
        std::sort(order.begin(), order.end(), [&](int p, int q) { return T.col_cost[p] > T.col_cost[q] || (T.col_cost[p] == T.col_cost[q] && P.model_of[p] < P.model_of[q]); });
        for (int f : order) {
            double step = cap[f] - xf[f];
            if (step < 1.0) continue;
            const uint32_t b = (uint32_t)P.block_of_flat[f];
            const double *ba = &bact[(size_t)b * MMAX_BLOCK];
            for (int r = 0; r < (int)T.blk_m[b] && step >= 1.0; r++) { const double a = T.col_a[(size_t)f * MMAX_BLOCK + r]; if (a > 0.0) { const double room = T.blk_cap[(size_t)b * MMAX_BLOCK + r] - ba[r]; if (room < a) { step = 0.0; break; } step = std::min(step, std::floor(room / a + 1e-9)); } }
            for (uint32_t e = T.col_woff[f]; e < T.col_woff[f + 1] && step >= 1.0; e++) {
                const double a = (double)T.w_coef[e];
                if (!(a > 0.0)) continue;   // (a `>=` row entered negated: more of the column only helps it)
                const int g = T.w_row[e];
                for (int q = P.gr_off[g]; q < P.gr_off[g + 1]; q++) { const int k = P.gr_row[q]; const double room = P.h[k] - (ga[g] + fl[k]); if (room < 0.5 * a) { step = 0.0; break; } step = std::min(step, std::floor(room / a + 1e-9)); }
            }
            if (step < 1.0) continue;
            xf[f] += step;
            double *bw = &bact[(size_t)b * MMAX_BLOCK];
            for (int r = 0; r < MMAX_BLOCK; r++) bw[r] += T.col_a[(size_t)f * MMAX_BLOCK + r] * step;
            for (uint32_t e = T.col_woff[f]; e < T.col_woff[f + 1]; e++) ga[T.w_row[e]] += (double)T.w_coef[e] * step;
        }
        double z = 0.0;
        for (uint32_t f = 0; f < T.n_cols; f++) { x[P.model_of[f]] = xf[f]; z += T.col_cost[f] * xf[f]; }
        for (int g = 0; g < G; g++) z += P.gcost[g] * B[g];
        value = z;
        return true;
    }
};

}  // namespace

Answer solve_model(const ModelView &mv, double rel_gap, double time_limit_s, double deadline_s, bool trace, Sweeper &sw, double *cost_scale) {
    const double tp0 = now_us();
    Request rq;
    rq.n = mv.n; rq.m = mv.m; rq.rel_gap = rel_gap; rq.time_limit_s = time_limit_s; rq.deadline_s = deadline_s; rq.trace = trace;
    Solver S(rq, sw);
    std::vector<double> ub, c; double cmax = 1.0;
        if (const char *why = build_from_model(mv, S.P, ub, c, cmax)) { Answer ans; ans.why = why; return ans; }
        if (cost_scale) *cost_scale = cmax;
    Polisher pol(S.P, mv.n);
    rq.polish = [&pol](std::vector<double> &x, double &value) { return pol.run(x, value); };
    return run_solver(S, tp0);
}

void set_check_hints(bool on) { g_check_hints = on; g_hint_mismatches = 0; }
int hint_mismatches() { return g_hint_mismatches; }

Answer solve(const Request &rq, Sweeper &sw) {
    Solver S(rq, sw);
    const double tp0 = now_us();
    if (const char *why = build(rq, S.P)) { Answer ans; ans.why = why; return ans; }
    return run_solver(S, tp0);
}

}  // namespace hqprice
