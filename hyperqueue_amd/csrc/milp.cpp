// Exact small-MILP solver (see milp.h).  Branch-and-bound over a bounded-variable DUAL simplex on a dense tableau that holds only
// the ACTIVE rows: constraints enter the tableau when the current LP point violates them (the placement model carries one cut row per
// worker, blocker and cut — scheduler/solver.rs:274-429 — almost all of them slack at the optimum), children are warm-started from the
// parent's tableau, and a lexicographic canonicalisation pass follows.
//
// Why dual simplex: every objective coefficient of the tick's model is >= 0 (scheduler/solver.rs:542-597) and every placement column
// has a finite upper bound implied by its worker's resource rows, so "costly columns at their upper bound, no row active" is dual
// feasible from the start; a violated row that enters, or a B&B child that differs from its parent by one bound, is exactly the case the
// dual method re-optimises in a handful of pivots.
#include "milp.h"
#include "lp_tab.h"
#include "price.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <climits>
#include <cstring>
#include <functional>
#include <numeric>
#include <string>
#include <unordered_map>

namespace hqmilp {
namespace {

using namespace lp;

struct DSUlite {
    std::vector<int> p;
    explicit DSUlite(int n) : p(n) { std::iota(p.begin(), p.end(), 0); }
    int find(int a) { while (p[a] != a) { p[a] = p[p[a]]; a = p[a]; } return a; }
    void unite(int a, int b) { a = find(a); b = find(b); if (a != b) p[std::max(a, b)] = std::min(a, b); }
};

void order_by_cost_desc(const double *c, int n, std::vector<int> &out);

const bool cuts_on = !(getenv("HQMILP_CUTS") && atoi(getenv("HQMILP_CUTS")) == 0);
const bool hull_on = !(getenv("HQMILP_HULL_CUTS") && atoi(getenv("HQMILP_HULL_CUTS")) == 0);  // (A/B switch: block-hull cuts in the root rounds)
const bool tree_cuts_on = !(getenv("HQMILP_TREE_CUTS") && atoi(getenv("HQMILP_TREE_CUTS")) == 0);  // (A/B switch: the certification tree on the rows + root cuts)

struct CompSolver {
    int n = 0;
    Rows R;
    std::vector<double> c, lb, ub;
    // the coupled solve by price sweeps (csrc/price.h): where the sweeps run, and what the model's builder said about its structure
    hqprice::Sweeper *sweeper = nullptr;
    std::function<void()> lazy_incumbent;  // set instead of an incumbent for a component that goes to the sweeps first: called when they leave it open (solve())
    std::vector<double> row_scale;       // R's row i times row_scale[i] = the model's row
    std::vector<uint8_t> row_implied;
    std::vector<int32_t> col_group;
    int price_sweeps = 0, price_rounds = 0; double price_us = 0.0;
    double deadline = 0; bool timed_out = false;
    double time_limit_s = 5.0;  // the configured limit itself (not what is left of it): what count-based budgets are scaled by, so that replicas scale them alike
    long nodes = 0, lp_iters = 0;
    // incumbent
    bool have = false; double best = -INF; std::vector<double> bx;
    bool canonical_done = true;
    // search control of phase 1: a plain dive first (good incumbents fast), then — if that does not finish within its node budget — a restart
    // from the root with strong branching, which proves what the dive found with far fewer nodes
    bool strong = false, aborted = false; long node_budget = -1;
    bool cert_stop = false;  // the incumbent came within rel_gap of the component's bound (root LP, or the tighter bound of the price sweeps) while the tree was running: stop it

    // Deterministic work budget of the tie-break phase, in tableau element updates (a pivot or a tableau copy touches ma x width of them);
    // the wall clock stays the backstop.  Work, not seconds: every replica of a sharded scheduler must take the same decision here.
    double work = 0.0, work_limit = -1.0;
    bool time_up() { if (timed_out) return true; if ((work_limit >= 0 && work > work_limit) || ((nodes & 31) == 0 && wall() > deadline)) timed_out = true; return timed_out; }
    int solve_counted(Tab &t) { double before = t.ops; int r = t.solve(200000); work += t.ops - before; return r; }

    // column-wise view of R for the primal heuristic
    std::vector<int> coff, crow; std::vector<double> ccoef; std::vector<int> by_cost;
    void build_columns() {
        coff.assign(n + 1, 0);
        for (int k = 0; k < (int)R.col.size(); k++) coff[R.col[k] + 1]++;
        for (int j = 0; j < n; j++) coff[j + 1] += coff[j];
        crow.resize(R.col.size()); ccoef.resize(R.col.size());
        std::vector<int> cur(coff.begin(), coff.end() - 1);
        for (int i = 0; i < R.m; i++) for (int k = R.off[i]; k < R.off[i + 1]; k++) { int j = R.col[k]; crow[cur[j]] = i; ccoef[cur[j]++] = R.coef[k]; }
        order_by_cost_desc(c.data(), n, by_cost);
    }
    // Primal heuristic: from an integer point inside the bounds (e.g. the floor of an LP solution), keep it only if every row holds,
    // then raise columns greedily — most valuable first — as far as the rows allow.  The placement models are packing
    // problems whose LP bound is usually attained, so a maximal point found here often closes the search at the root.
    void greedy_from(std::vector<double> x) {
        if (coff.empty()) build_columns();
        std::vector<double> act(R.m);
        for (int i = 0; i < R.m; i++) { act[i] = R.activity(i, x.data()); if (act[i] < R.lo[i] - FEAS_TOL || act[i] > R.hi[i] + FEAS_TOL) return; }
        for (int j : by_cost) {
            if (c[j] <= 0.0) break;
            double step = ub[j] - x[j];
            for (int k = coff[j]; k < coff[j + 1] && step >= 1.0; k++) {
                const int i = crow[k]; const double a = ccoef[k];
                // (room < a / 2 means floor(room / a + 1e-9) = 0 whatever the rounding: most columns of a packed point end here, without the division)
                if (a > 0.0 && R.hi[i] < INF) { const double room = R.hi[i] - act[i]; if (room < 0.5 * a) { step = 0.0; break; } step = std::min(step, std::floor(room / a + 1e-9)); }
                else if (a < 0.0 && R.lo[i] > -INF) { const double room = act[i] - R.lo[i]; if (room < -0.5 * a) { step = 0.0; break; } step = std::min(step, std::floor(room / -a + 1e-9)); }
            }
            if (step < 1.0) continue;
            x[j] += step;
            for (int k = coff[j]; k < coff[j + 1]; k++) act[crow[k]] += ccoef[k] * step;
        }
        double z = 0.0; for (int j = 0; j < n; j++) z += c[j] * x[j];
        if (!have || z > best + 1e-12 * std::fabs(best)) { have = true; best = z; bx = x; }
    }
    // A point from outside (the price sweeps): false when it violates a row or a bound; else raised greedily like greedy_from, returned in place
    bool polish_point(std::vector<double> &x, double &value) {
        if (coff.empty()) build_columns();
        if ((int)x.size() != n) return false;
        for (int j = 0; j < n; j++) if (x[j] < lb[j] - FEAS_TOL || x[j] > ub[j] + FEAS_TOL) return false;
        std::vector<double> act(R.m);
        for (int i = 0; i < R.m; i++) { act[i] = R.activity(i, x.data()); if (act[i] < R.lo[i] - FEAS_TOL || act[i] > R.hi[i] + FEAS_TOL) return false; }
        for (int j : by_cost) {
            if (c[j] <= 0.0) break;
            double step = ub[j] - x[j];
            for (int k = coff[j]; k < coff[j + 1] && step >= 1.0; k++) {
                const int i = crow[k]; const double a = ccoef[k];
                // (room < a / 2 means floor(room / a + 1e-9) = 0 whatever the rounding: most columns of a packed point end here, without the division)
                if (a > 0.0 && R.hi[i] < INF) { const double room = R.hi[i] - act[i]; if (room < 0.5 * a) { step = 0.0; break; } step = std::min(step, std::floor(room / a + 1e-9)); }
                else if (a < 0.0 && R.lo[i] > -INF) { const double room = act[i] - R.lo[i]; if (room < -0.5 * a) { step = 0.0; break; } step = std::min(step, std::floor(room / -a + 1e-9)); }
            }
            if (step < 1.0) continue;
            x[j] += step;
            for (int k = coff[j]; k < coff[j + 1]; k++) act[crow[k]] += ccoef[k] * step;
        }
        double z = 0.0; for (int j = 0; j < n; j++) z += c[j] * x[j];
        value = z;
        return true;
    }
    // see run(): true (and the point in xout) when the LP optimum is integral, equals the incumbent and no other integer point comes within the canonical window of it
    bool unique_optimum(std::vector<double> &xout) {
        if (!have || n == 0 || n > 4096) return false;
        Tab u; u.init(&R, c, lb, ub); u.deadline = deadline;
        if (solve_counted(u) != LP_OPT) return false;
        lp_iters += u.iters;
        const double z = u.objective();
        if (std::fabs(z - best) > 1e-9 * std::max(1.0, std::fabs(best))) { if (tracing) fprintf(stderr, "[uniq] gap z %.12f best %.12f\n", z, best); return false; }   // an integrality gap: the incumbent is not the LP's vertex
        for (int j = 0; j < n; j++) if (std::fabs(u.x[j] - std::round(u.x[j])) > INT_TOL || std::round(u.x[j]) != std::round(bx[(size_t)j])) { if (tracing) fprintf(stderr, "[uniq] x[%d] %.6f vs %.1f\n", j, u.x[j], bx[(size_t)j]); return false; }
        if (slack_unit.size() != (size_t)R.m) find_slack_units();
        const double window = 2e-9 * std::fabs(best) + 1e-12;
        for (int k = 0; k < u.width(); k++) {
            if (u.st[k] == BASIC || u.lb[k] == u.ub[k]) continue;
            double step = 1.0;
            if (k >= n) { step = slack_unit[(size_t)u.arow[k - n]]; if (!(step > 0.0)) { if (tracing) fprintf(stderr, "[uniq] row %d without unit, d %.3e\n", u.arow[k - n], u.d[k]); return false; } }
            if (!(std::fabs(u.d[k]) * step > window)) { if (tracing) fprintf(stderr, "[uniq] col %d d %.3e step %.3e window %.3e\n", k, u.d[k], step, window); return false; }
        }
        xout.assign((size_t)n, 0.0);
        for (int j = 0; j < n; j++) xout[(size_t)j] = std::round(u.x[j]);
        return true;
    }
    // ---- Gomory mixed-integer cuts at the root ------------------------------------------------------------------------------------------------------------------
    // The tick's coupled models are pure integer programs whose LP bound sits percent above the optimum (every worker a knapsack with fractional requests, the
    // batch-size rows across them) and whose Lagrangian / Dantzig-Wolfe bound still sits 2e-4..6e-4 above it on small clusters mid-run — twice the reference's
    // mip_rel_gap, with thousands of near-tied block patterns below it that no branching on single columns separates (tools/price_fuzz.py: the ticks HiGHS
    // certifies and round 3 did not).  HiGHS closes exactly those at its ROOT: a few rounds of cuts take the LP bound down to the optimum itself.  The same here:
    // every basic integer column with a fractional LP value gives one GMI cut from its tableau row (the tableau is dense and explicit, lp_tab.h) —
    //     x_k + sum_j abar_j t_j = beta,  t_j >= 0 the nonbasic columns' distances from their bounds  =>  sum_j gamma_j t_j >= 1,
    //     gamma_j = f_j / f_0 or (1 - f_j) / (1 - f_0) for an integer t_j (structural columns; slacks of rows whose coefficients share a unit), abar_j / f_0 or
    //     -abar_j / (1 - f_0) for a continuous one —
    // written back over the structural columns and appended to R like any other row (the LP activates it when violated; the tree's nodes inherit it).  Guarded the
    // usual way: fractionality of the source row in [0.02, 0.98], bounded dynamism, a small rhs relaxation, and a cut the incumbent violates is thrown away.
    std::vector<double> slack_unit;   // per row of R: its activity is an integer multiple of this (0: unknown / continuous)
    int cuts_added = 0;
    void find_slack_units() {
        slack_unit.assign((size_t)R.m, 0.0);
        for (int i = 0; i < R.m; i++) {
            const int a = R.off[i], b = R.off[i + 1];
            if (b <= a) continue;
            // coefficients are the model's (multiples of 1e-4) divided by row_scale: on the grid 1e-4 / scale
            const double sc = (size_t)i < row_scale.size() ? row_scale[(size_t)i] : 1.0;
            long long g = 0; bool ok = true;
            for (int k = a; k < b && ok; k++) {
                const double v = R.coef[k] * sc * 10000.0; const double rv = std::round(v);
                if (std::fabs(v - rv) > 1e-6 * std::max(1.0, std::fabs(v)) || std::fabs(rv) > 1e15) { ok = false; break; }
                long long x = (long long)std::fabs(rv), y = g; while (y) { const long long tt = x % y; x = y; y = tt; } g = x;
            }
            if (!ok || g <= 0) continue;
            const double unit = (double)g / (sc * 10000.0);
            // the bounds the slack is measured from must lie on the same grid (presolve snaps the right-hand sides of `<=` rows to reachable values)
            auto on_grid = [&](double v) { if (!(std::fabs(v) < INF)) return true; const double q = v / unit; return std::fabs(q - std::round(q)) <= 1e-7 * std::max(1.0, std::fabs(q)); };
            if (on_grid(R.lo[i]) && on_grid(R.hi[i])) slack_unit[(size_t)i] = unit;
        }
    }
    int gmi_round(Tab &t, Rows &RC, int max_cuts) {
        const int N = t.width();
        struct Cut { std::vector<std::pair<int, double>> terms; double rhs, eff; };
        std::vector<Cut> found;
        std::vector<double> alpha(n);
        for (int r = 0; r < t.ma; r++) {
            const int k = t.B[r];
            if (k >= n) continue;
            const double beta = t.x[k], f0 = beta - std::floor(beta);
            if (f0 < 0.02 || f0 > 0.98) continue;
            const double *row = &t.T[(size_t)r * t.stride];
            std::fill(alpha.begin(), alpha.end(), 0.0);
            double rho = 1.0; bool bad = false;
            for (int j = 0; j < N && !bad; j++) {
                if (t.st[j] == BASIC || row[j] == 0.0) continue;
                if (t.lb[j] == t.ub[j]) continue;  // fixed: t_j = 0 always
                const bool up = t.st[j] == AT_UP;
                const double abar = up ? -row[j] : row[j];
                double gamma, unit = 1.0; bool integer_t = j < n;
                if (j >= n) { const double u = slack_unit[(size_t)t.arow[j - n]]; if (u > 0.0) { integer_t = true; unit = u; } }
                if (integer_t) {
                    const double au = abar * unit; double fj = au - std::floor(au);
                    if (fj < 1e-9 || fj > 1.0 - 1e-9) fj = 0.0;
                    gamma = (fj <= f0 ? fj / f0 : (1.0 - fj) / (1.0 - f0)) / unit;
                } else gamma = abar >= 0.0 ? abar / f0 : -abar / (1.0 - f0);
                if (gamma == 0.0) continue;
                if (!(gamma < 1e9)) { bad = true; break; }
                // gamma * t_j with t_j = x_j - lb_j (at lower) or ub_j - x_j (at upper); a slack x_j is its row's activity
                const double bound = up ? t.ub[j] : t.lb[j];
                if (!(std::fabs(bound) < 1e15)) { bad = true; break; }
                const double sgn = up ? -1.0 : 1.0;
                rho += sgn * gamma * bound;
                if (j < n) alpha[j] += sgn * gamma;
                else { const int i = t.arow[j - n]; for (int q = RC.off[i]; q < RC.off[i + 1]; q++) alpha[RC.col[q]] += sgn * gamma * RC.coef[q]; }
            }
            if (bad) continue;
            // sum alpha_i x_i >= rho.  Tiny coefficients are dropped against the column's bound (a relaxation), then the dynamism is checked
            double amax = 0.0; for (int i = 0; i < n; i++) amax = std::max(amax, std::fabs(alpha[i]));
            if (!(amax > 1e-12)) continue;
            Cut cut; double amin = INF, lhs = 0.0, norm = 0.0;
            for (int i = 0; i < n; i++) {
                const double a = alpha[i];
                if (a == 0.0) continue;
                if (std::fabs(a) < 1e-7 * amax) { const double w = a > 0.0 ? ub[i] : lb[i]; if (!(std::fabs(w) < 1e9)) { bad = true; break; } rho -= a * w; continue; }
                cut.terms.push_back({i, a}); amin = std::min(amin, std::fabs(a)); lhs += a * t.x[i]; norm += a * a;
            }
            if (bad || cut.terms.empty() || amax / amin > 1e6) continue;
            rho -= 1e-9 * (std::fabs(rho) + amax);  // numerical safety: the cut is relaxed a hair
            const double viol = rho - lhs;
            if (viol <= 1e-6 * amax) continue;
            if (have) {  // never cut the incumbent off
                double li = 0.0; for (auto &tm : cut.terms) li += tm.second * bx[tm.first];
                if (li < rho - 1e-9 * (std::fabs(rho) + amax)) continue;
            }
            for (auto &tm : cut.terms) tm.second /= amax;
            cut.rhs = rho / amax; cut.eff = viol / std::sqrt(norm);
            found.push_back(std::move(cut));
        }
        std::sort(found.begin(), found.end(), [](const Cut &a, const Cut &b) { return a.eff > b.eff; });
        int added = 0;
        for (auto &cu : found) {
            if (added >= max_cuts) break;
            if (cu.eff < 1e-5) break;
            RC.add(cu.terms, cu.rhs, INF);
            slack_unit.push_back(0.0);
            t.where.push_back(-1);
            added++;
        }
        cuts_added += added;
        return added;
    }
    // ---- Block-hull cuts at the root (what the bound of small coupled models was missing) ------------------------------------------------------
    // The model is one small block per worker (its builder says which: col_group) plus rows across them.  For ANY cost vector r over a block's columns,
    //     r . x_b <= V_b(r) = max { r . x_b : the block's own rows and bounds, x_b integer }
    // holds at every integer point of the model — a facet-or-face of the block's integer hull, found by solving the block EXACTLY (8-16 columns, <= 4 rows: tens of
    // microseconds).  With r = the Lagrangian costs at the current LP optimum (c minus the duals of every row that is not the block's own) these are the cuts of
    // Dantzig-Wolfe / Kelley written in the original space: rounds of them take the LP bound down to the decomposition bound, which on clusters mid-run sits an order
    // of magnitude closer to the optimum than 40 rounds of GMI cuts alone reach (tools/price_fuzz.py seed 2047: 2.9e-3 against 7e-4), and the GMI rounds that follow
    // start from THERE — cuts across blocks on top of the blocks' hulls, which is the combination HiGHS closes these models with at its root.
    // Everything counted, nothing timed: block solves run on a node cap.
    struct HullBlocks {
        bool ready = false, usable = false;
        std::vector<std::vector<int>> cols;      // per block: its columns
        std::vector<Rows> rows;                  // per block: the rows that live inside it (local column ids)
        std::vector<uint8_t> row_internal;       // per row of R: 1 = inside one block
    } hb;
    void find_hull_blocks() {
        hb.ready = true; hb.usable = false;
        if ((int)col_group.size() != n) return;
        int gmax = -1; for (int j = 0; j < n; j++) gmax = std::max(gmax, (int)col_group[j]);
        if (gmax < 1) return;
        std::vector<int> id((size_t)gmax + 1, -1), local(n, -1);
        for (int j = 0; j < n; j++) { const int g = col_group[j]; if (g < 0) continue; if (id[g] < 0) { id[g] = (int)hb.cols.size(); hb.cols.emplace_back(); } local[j] = (int)hb.cols[id[g]].size(); hb.cols[id[g]].push_back(j); }
        const int nb = (int)hb.cols.size();
        if (nb < 2) return;
        for (auto &bc : hb.cols) if (bc.size() > 64) return;   // (a block of that size is not a worker's)
        hb.rows.assign(nb, Rows());
        for (int b = 0; b < nb; b++) hb.rows[b].n = (int)hb.cols[b].size();
        hb.row_internal.assign((size_t)R.m, 0);
        std::vector<std::pair<int, double>> terms;
        for (int i = 0; i < R.m; i++) {
            int b = -1; bool inside = R.off[i + 1] > R.off[i];
            for (int k = R.off[i]; k < R.off[i + 1] && inside; k++) { const int g = col_group[R.col[k]]; if (g < 0) { inside = false; break; } const int bb = id[g]; if (b < 0) b = bb; else if (bb != b) inside = false; }
            if (!inside) continue;
            hb.row_internal[(size_t)i] = 1;
            terms.clear();
            for (int k = R.off[i]; k < R.off[i + 1]; k++) terms.push_back({local[R.col[k]], R.coef[k]});
            hb.rows[b].add(terms, R.lo[i], R.hi[i]);
        }
        hb.usable = true;
    }
    long hull_solves = 0;
    // one round: cuts for the blocks whose LP point beats their integer optimum at the current Lagrangian costs; returns how many were added to RC
    int hull_round(Tab &t, Rows &RC) {
        if (!hb.ready) find_hull_blocks();
        if (!hb.usable) return 0;
        // Lagrangian costs: c_j minus the duals of the active rows that are NOT internal to a block (cut rows included).  The dual of active row a is the reduced
        // cost of its slack column (zero while the slack is basic).
        std::vector<double> r(c.begin(), c.end());
        for (int a = 0; a < t.ma; a++) {
            const int i = t.arow[a];
            if (i < R.m && hb.row_internal[(size_t)i]) continue;
            const double y = t.st[t.n + a] == BASIC ? 0.0 : t.d[t.n + a];
            if (y == 0.0) continue;
            for (int k = RC.off[i]; k < RC.off[i + 1]; k++) r[RC.col[k]] -= y * RC.coef[k];
        }
        int added = 0;
        std::vector<std::pair<int, double>> terms;
        for (size_t b = 0; b < hb.cols.size(); b++) {
            const std::vector<int> &bc = hb.cols[b];
            const int nbc = (int)bc.size();
            double lpv = 0.0, rmax = 0.0;
            for (int l = 0; l < nbc; l++) { const int j = bc[l]; lpv += r[j] * t.x[j]; rmax = std::max(rmax, std::fabs(r[j])); }
            if (!(rmax > 1e-12)) continue;
            if (lpv <= 1e-7 * rmax) continue;   // (V_b >= r . lb-point; with lb = 0 nothing below zero can be violated)
            CompSolver sub; sub.n = nbc; sub.in_lns = true; sub.deadline = deadline; sub.node_cap = 4000; sub.tracing = false;
            sub.c.resize(nbc); sub.lb.resize(nbc); sub.ub.resize(nbc); sub.R = hb.rows[b];
            for (int l = 0; l < nbc; l++) { const int j = bc[l]; sub.c[l] = r[j] / rmax; sub.lb[l] = lb[j]; sub.ub[l] = r[j] > 0.0 ? ub[j] : lb[j]; }   // a column that does not pay stays at its lower bound
            std::vector<double> xb;
            const int st = sub.run(false, xb);
            nodes += sub.nodes; work += sub.work; hull_solves++;
            if (tracing && getenv("HQMILP_HULL_TRACE")) fprintf(stderr, "[hull] block %zu: %d cols st %d nodes %ld work %.3g\n", b, nbc, st, sub.nodes, sub.work);
            if (st != 1 || (int)xb.size() != nbc) continue;   // not solved to the end within its cap: no cut from this block
            double V = 0.0; for (int l = 0; l < nbc; l++) V += r[bc[l]] * xb[l];
            if (lpv <= V + 1e-6 * rmax * std::max(1.0, std::fabs(V / rmax))) continue;
            // r . x_b <= V, scaled to max |coef| = 1 and relaxed a hair; never against the incumbent (cannot be: it is an integer point of the block)
            terms.clear();
            for (int l = 0; l < nbc; l++) { const int j = bc[l]; if (r[j] > 0.0 && std::fabs(r[j]) >= 1e-9 * rmax) terms.push_back({j, r[j] / rmax}); }
            // (columns with r_j <= 0 were held at their lower bound in the block solve: dropping them from the cut keeps it valid only for lb = 0 — check)
            bool lb0 = true; for (int l = 0; l < nbc; l++) if (lb[bc[l]] != 0.0) lb0 = false;
            if (!lb0) continue;
            double rhs = V / rmax; rhs += 1e-9 * (std::fabs(rhs) + 1.0);
            if (have) { double li = 0.0; for (auto &tm : terms) li += tm.second * bx[tm.first]; if (li > rhs) continue; }
            RC.add(terms, -INF, rhs);
            slack_unit.push_back(0.0);
            t.where.push_back(-1);
            added++;
        }
        cuts_added += added;
        return added;
    }
    // Dual feasibility of a solved tableau (max problem: a nonbasic column at its lower bound must not gain, one at its upper bound must not lose): what makes its
    // objective a BOUND.  The dense tableau is never refactored, and cuts bring coefficient ranges of 1e6 into it: a warm re-solve can come back "optimal" from a
    // basis whose reduced costs have drifted (tools/price_fuzz.py seed 29: 142.262 where a cold solve of the same rows gives 142.399) — such a value bounds nothing.
    static bool dual_feasible(const Tab &t, double tol) {
        const int N = t.width();
        for (int j = 0; j < N; j++) {
            if (t.st[j] == BASIC || t.lb[j] == t.ub[j]) continue;
            if (t.st[j] == AT_LO ? t.d[j] > tol : t.d[j] < -tol) return false;
        }
        return true;
    }
    // ... and the primal side of the same question: every active row's slack column must be the row's activity at the structural columns' values (both are updated
    // incrementally by the pivots).  A tableau that fails either test bounds nothing and closes nothing.
    static bool consistent(const Tab &q) {
        if (!dual_feasible(q, 1e-7)) return false;
        for (int a = 0; a < q.ma; a++) { const double act = q.R->activity(q.arow[a], q.x.data()); if (std::fabs(act - q.x[q.n + a]) > 1e-7 * std::max(1.0, std::fabs(act))) return false; }
        return true;
    }
    // A bound that does not need the tableau to be right (Neumaier / Shcherbina's safe dual bound).  For ANY multipliers lambda of the active rows
    //        max { c.x : rows, boxes }  <=  sum_j max_{lb_j <= x_j <= ub_j} (c_j - sum_a lambda_a A_aj) x_j  +  sum_a max_{lo_a <= s <= hi_a} lambda_a s
    // (add lambda_a (s_a - A_a.x) = 0 to the objective and drop the coupling) — evaluated from the ROWS' own coefficients in plain arithmetic, a few thousand
    // multiply-adds.  With the slack columns' reduced costs as multipliers it reproduces the LP value of an exact tableau; from a drifted tableau it is a little
    // weaker and still a bound.  Rows that are not active are simply relaxed; a multiplier that would need an infinite side of its row is set to zero.  Both sign
    // conventions are tried (each is valid) and the smaller value returned; INF when neither is finite.
    static double safe_dual_bound(const Tab &q) {
        const Rows &RR = *q.R;
        double out = INF;
        std::vector<double> r((size_t)q.n);
        for (int sign = 1; sign >= -1; sign -= 2) {
            for (int j = 0; j < q.n; j++) r[(size_t)j] = q.cost[j];
            double b = 0.0; bool finite = true;
            for (int a = 0; a < q.ma; a++) {
                double lam = (double)sign * q.d[q.n + a];
                const double lo = q.lb[q.n + a], hi = q.ub[q.n + a];
                if (!(lam == lam)) { lam = 0.0; }
                if (lam > 0.0 && hi >= INF * 0.5) lam = 0.0;
                if (lam < 0.0 && lo <= -INF * 0.5) lam = 0.0;
                if (lam == 0.0) continue;
                b += lam > 0.0 ? lam * hi : lam * lo;
                const int i = q.arow[a];
                for (int k = RR.off[i]; k < RR.off[i + 1]; k++) r[(size_t)RR.col[k]] -= lam * RR.coef[k];
            }
            for (int j = 0; j < q.n && finite; j++) {
                const double rj = r[(size_t)j];
                if (rj > 0.0) { if (q.ub[j] >= INF * 0.5) finite = false; else b += rj * q.ub[j]; }
                else if (rj < 0.0) { if (q.lb[j] <= -INF * 0.5) finite = false; else b += rj * q.lb[j]; }
            }
            if (finite && b == b && b < out) out = b;
        }
        return out;
    }
    // Rounds of cuts on a COPY of the rows, for the bound only: the tree below keeps working on the model's own rows (its canonical answers must not depend on the
    // numerics of cut rows).  A round's value counts when its tableau is dual feasible; the final value is taken from a COLD solve over the final rows, the larger of
    // the two when they differ.
    // The rows with the accepted cuts stay (`RCm`): the CERTIFICATION phases of the tree run on them (search(): a node's LP bound then starts from the cut-tightened
    // root instead of the plain LP's, percent above it — the difference between 18 nodes and millions on small clusters mid-run); the exact /
    // canonical pass after a certificate goes back to the model's own rows.
    Rows RCm; bool rc_valid = false; double rc_bound = INF; long cut_infeas_refuted = 0, drift_resolves = 0;
    // (debugging, HQMILP_WATCH_X=<file of n values>: a known feasible point; every node that is closed while it still contains the point says why)
    std::vector<double> watch_x; bool watch_loaded = false;
    bool watch_inside(const Tab &q) {
        if (!watch_loaded) { watch_loaded = true; if (const char *f = getenv("HQMILP_WATCH_X")) if (!in_lns && n > 100) { FILE *fp = fopen(f, "r"); if (fp) { double v; while ((int)watch_x.size() < n && fscanf(fp, "%lf", &v) == 1) watch_x.push_back(v); fclose(fp); if ((int)watch_x.size() != n) watch_x.clear(); } } }
        if (watch_x.empty()) return false;
        for (int k = 0; k < n; k++) if (watch_x[k] < q.lb[k] - 1e-7 || watch_x[k] > q.ub[k] + 1e-7) return false;
        for (int a = 0; a < q.ma; a++) { const double act = q.R->activity(q.arow[a], watch_x.data()); if (act < q.lb[q.n + a] - 1e-7 || act > q.ub[q.n + a] + 1e-7) return false; }
        return true;
    }
    void watch_closed(const Tab &q, const char *why, double z) { if (watch_inside(q)) { double zw = 0.0; for (int k = 0; k < n; k++) zw += c[k] * watch_x[k]; fprintf(stderr, "[watch] node %ld CLOSED (%s) at z %.9f while it holds the watched point of value %.9f; best %.9f, rows %s\n", nodes, why, z, zw, best, q.R == &RCm ? "cuts" : "plain"); } }
    std::vector<double> cut_lp_x;   // the LP point over RCm
    // RINS at the cut LP's point: the columns on which that point and the incumbent agree stay where they are, the rest is a small model solved exactly.  With the
    // bound a few 1e-4 above the incumbent the two agree almost everywhere (price_fuzz seed 2056: the windows' incumbent sits 8e-5 below the optimum, 1.1e-4 below
    // the bound — one exchange between three workers away from the certificate).
    bool rins_at_cut_point() {
        if (!rc_valid || !have || in_lns || (int)cut_lp_x.size() != n) return false;
        std::vector<int> wcols;
        for (int j = 0; j < n; j++) if (std::fabs(cut_lp_x[j] - bx[j]) > 0.5 - 1e-9 || std::fabs(cut_lp_x[j] - std::round(cut_lp_x[j])) > INT_TOL) wcols.push_back(j);
        if (wcols.empty() || (int)wcols.size() > 160) return false;
        const double before = best;
        const bool imp = lns_solve(wcols, deadline, 50000);
        if (tracing) fprintf(stderr, "[milp] n=%d RINS at the cut LP's point (%d free columns): %.9f -> %.9f\n", n, (int)wcols.size(), before, best);
        return imp;
    }
    // Two passes, the second only when the first leaves the model open: GMI rounds alone, then block-hull cuts with GMI rounds on top.  Neither dominates (price_fuzz
    // 2020: GMI alone names the optimum to 1e-9, with the hull cuts the rounds stall 1.6e-4 above it; 2017: GMI alone stalls 8.4e-4 above, with the hull cuts the
    // root closes) — the tree gets the rows of the pass with the lower bound.
    void root_cuts(const Tab &root0) {
        root_cuts_pass(root0, false);
        if (hull_on && have && !certified() && !in_lns && n <= 600 && rel_gap > 0.0) { if (!hb.ready) find_hull_blocks(); if (hb.usable) root_cuts_pass(root0, true); }
    }
    void root_cuts_pass(const Tab &root0, const bool with_hull) {
        if (in_lns || n > 600 || rel_gap <= 0.0) return;  // (the models the sweeps never see, 256 columns and below by default, and a little beyond; an LP of 1000+ columns is too slow to re-solve 40 times)
        const double work_cap = work + (getenv("HQMILP_CUT_WORK") ? atof(getenv("HQMILP_CUT_WORK")) : 3.0e8) * std::max(0.2, time_limit_s / 5.0);  // a deterministic budget (tableau element updates: ~0.3 s of a 5 s limit), like every other one in here
        Rows RC = R;
        find_slack_units();
        Tab root = root0; root.R = &RC;
        double prev = root.objective(), accepted = prev;
        double best_safe = INF;   // the smallest safe dual bound of an accepted round's tableau: a bound on the model whatever rows come later
        int stall = 0, hull_stall = 0;
        int round = 0;
        for (; round < 80 && !time_up() && work < work_cap; round++) {
            if (have && accepted <= best + rel_gap * std::fabs(best)) break;
            // block-hull cuts while they move the bound; GMI cuts (across blocks) when they no longer do — and then the hulls again, at the point the GMI rounds moved to
            int added = 0; const char *kind = "GMI";
            if (with_hull && hull_stall < 2) { added = hull_round(root, RC); if (added) kind = "block-hull"; }
            const bool was_hull = added > 0;
            if (!added) { added = gmi_round(root, RC, 40 + n / 8); hull_stall = 0; }
            if (!added) { if (tracing) fprintf(stderr, "[milp] n=%d cut round %d: nothing to add\n", n, round); break; }
            // (the warm re-solve on a pivot allowance: behind block-hull cuts the vertex is highly degenerate and the dual simplex can stall for thousands of pivots — price_fuzz
            // seed 2047: 4 300 against the ~700 a cold solve of the same rows takes; past the allowance the cold solve below takes over)
            bool ok;
            { const double before = root.ops; const int r = root.solve(with_hull ? std::max(600L, 3L * (long)root.ma) : 200000L); work += root.ops - before; ok = r == LP_OPT && consistent(root); }
            if (!ok) {  // once more from a cold start over the same rows
                root = Tab(); root.init(&RC, c, lb, ub); root.deadline = deadline;
                ok = solve_counted(root) == LP_OPT && consistent(root);
                if (!ok) { if (tracing) fprintf(stderr, "[milp] n=%d cut round %d: LP not re-solved\n", n, round); break; }
            }
            const double z = root.objective();
            if (tracing) fprintf(stderr, "[milp] n=%d cut round %d: %d %s cuts, LP bound %.9f -> %.9f (%d rows active of %d) work %.3g iters %ld\n", n, round, added, kind, prev, z, root.ma, RC.m, work, (long)root.iters);
            if (was_hull && prev - z < 1e-6 * std::fabs(prev)) hull_stall++;
            if (z > accepted * (1.0 + 1e-9) + 1e-12) break;  // a bound cannot rise when rows are added: the arithmetic has gone wrong, keep what was accepted
            accepted = z;
            best_safe = std::min(best_safe, safe_dual_bound(root));
            if (prev - z < 1e-7 * std::fabs(prev)) { if (++stall >= 3) break; } else stall = 0;
            prev = z;
            if (RC.m > 40 * n + 4000) break;
        }
        if (tracing) fprintf(stderr, "[milp] n=%d cut rounds over after %d: work %.3g of %.3g, time_up %d\n", n, round, work, work_cap, (int)timed_out);
        if (RC.m > R.m && accepted < root0.objective()) {  // the certificate's bound: confirmed by a cold solve of the final rows
            Tab cold; cold.init(&RC, c, lb, ub); cold.deadline = deadline;
            static const bool force_safe = getenv("HQMILP_FORCE_SAFE_BOUND") != nullptr;   // tests: take the fallback below whatever the cold solve says
            if (solve_counted(cold) == LP_OPT && consistent(cold) && !force_safe) {
                const double zc = cold.objective();
                if (tracing) fprintf(stderr, "[milp] n=%d cuts: %d rows added, bound %.9f (cold solve of the final rows: %.9f)\n", n, RC.m - R.m, accepted, zc);
                root_bound = std::min(root_bound, std::max(accepted, zc) * (1.0 + 1e-9) + 1e-12);
            } else {
                // The cold tableau over several hundred cut rows has drifted past the consistency test (price_fuzz 2317: the rounds had the bound 7e-5 above the
                // incumbent and the certificate was dropped here).  A tableau that cannot be trusted still NAMES multipliers, and the bound they give is computed from
                // the rows themselves: the smallest of what the accepted rounds', the last warm and the cold tableau's multipliers prove.
                const double sw = std::min(best_safe, safe_dual_bound(root)), sc = safe_dual_bound(cold), sb = std::min(sw, sc);
                if (tracing) fprintf(stderr, "[milp] n=%d cuts: %d rows added, bound %.9f; the cold solve of the final rows is not consistent: safe dual bound %.9f (warm multipliers %.9f, cold %.9f)\n", n, RC.m - R.m, accepted, sb, sw, sc);
                if (sb < INF * 0.5) root_bound = std::min(root_bound, sb * (1.0 + 1e-9) + 1e-12);
            }
        }
        cuts_added = RC.m - R.m;
        if (RC.m > R.m && have && !certified()) {
            // The cut LP's own point as a primal lead (RENS): where the rounds took the bound down to within a few 1e-4 of the optimum, the point that attains it is integral
            // in most columns — those are fixed, the fractional ones move between floor and ceiling, and the small model that is left is solved exactly.  (The 100-column
            // clusters of tools/price_fuzz.py on the host-only path: the windows' incumbent sits 2e-4 below an optimum the cut bound names to 1e-9.)
            std::vector<double> base(root.x.begin(), root.x.begin() + n);
            std::vector<int> wcols;
            for (int j = 0; j < n; j++) { const double r = std::round(base[j]); if (std::fabs(base[j] - r) <= INT_TOL) base[j] = std::min(ub[j], std::max(lb[j], r)); else wcols.push_back(j); }
            if (wcols.empty()) greedy_from(base);
            else if ((int)wcols.size() <= 96) { const double before = best; lns_solve(wcols, deadline, 20000, &base); if (tracing) fprintf(stderr, "[milp] n=%d RENS at the cut LP's point (%d fractional columns): %.9f -> %.9f\n", n, (int)wcols.size(), before, best); }
        }
        if (RC.m > R.m && tree_cuts_on && (!rc_valid || accepted < rc_bound)) { RCm = std::move(RC); rc_valid = true; rc_bound = accepted; row_unit.resize((size_t)RCm.m, 0.0); cut_lp_x.assign(root.x.begin(), root.x.begin() + n); }
    }

    void round_and_repair(const Tab &t) {
        std::vector<double> x(n);
        for (int j = 0; j < n; j++) x[j] = std::min(ub[j], std::max(lb[j], std::floor(t.x[j] + INT_TOL)));
        greedy_from(std::move(x));
    }

    // Objective lattice: if every cost is an integer multiple of q (the tick's costs are amount/pool x weight x (W - idx)/W, rationals with a
    // common denominator), an integral point better than the incumbent is better by at least q: nodes whose LP bound is below best + q hold none.
    double quantum = 0.0, lattice = -1.0;  // lattice >= 0: handed down by the parent model (window sub-problems), no search
    void find_quantum() {
        if (lattice >= 0.0) { quantum = lattice; return; }
        double cmaxv = 0.0; for (int j = 0; j < n; j++) cmaxv = std::max(cmaxv, std::fabs(c[j]));
        if (cmaxv == 0.0) return;
        const double tol = 1e-10 * cmaxv;
        double g = 0.0;
        for (int j = 0; j < n; j++) {
            double a = std::fabs(c[j]), b = g;
            if (a < tol) continue;
            if (b == 0.0) { g = a; continue; }
            if (a < b) std::swap(a, b);
            for (int it = 0; it < 200 && b > tol; it++) { double r = std::fmod(a, b); if (b - r < tol) r = 0.0; a = b; b = r; }
            g = a;
            if (g < 1e-6 * cmaxv) return;  // no useful lattice
        }
        if (g <= 0.0) return;
        for (int j = 0; j < n; j++) { double k = c[j] / g; if (std::fabs(k - std::round(k)) > 1e-7) return; }
        quantum = g;
    }
    // a node with LP bound z can be dropped when no integral point in it beats the incumbent
    // What "optimal" means (milp.h): the reference's HiGHS stops at mip_rel_gap = 1e-4, so a node whose LP bound is within rel_gap of the incumbent
    // is closed.  `gap_pruned` records that this rule (and not the exact one) closed a node: the incumbent is then certified, not proven exact.
    double rel_gap = 0.0;
    bool gap_pruned = false;
    double root_bound = INF;  // LP bound of the whole component, once the root LP is solved
    // Branch on a fractional FLAG before any placement column (round 6).  The flags are the model's big-M switches (blocker / cut rows, scheduler/solver.rs:233-253,395-429):
    // with one of them fractional the LP pays for a little of everything, and no branching on placement counts pins it down — price_fuzz seeds 2056 / 2057 / 2130 go from
    // `NeedMoreCompute` after 5 s to certificates in 0.3-4 s (emulated sweeps) with nothing else changed.  The static rule ranks columns by cost, and a flag costs nothing:
    // it used to come last.  HQMILP_FLAGS_FIRST=0: the old order (A/B).
    bool flags_first = !(getenv("HQMILP_FLAGS_FIRST") && atoi(getenv("HQMILP_FLAGS_FIRST")) == 0);
    bool certified() const { return have && rel_gap > 0.0 && root_bound <= best + rel_gap * std::fabs(best); }
    // (read once per process: a CompSolver is constructed per window of the window search and per class block — thousands per tick)
    static bool trace_enabled() { static const bool on = getenv("HQMILP_TRACE") != nullptr; return on; }
    bool tracing = trace_enabled();
    double t_begin = tracing ? wall() : 0.0;
    void trace(const char *what) { if (tracing && !in_lns) fprintf(stderr, "[milp] n=%d t=%.6fs %s: incumbent %.9f nodes %ld\n", n, wall() - t_begin, what, have ? best : -1.0, nodes); }
    bool cannot_improve(double z) {
        if (!have) return false;
        if (z <= best + 1e-12 * std::fabs(best)) return true;
        if (quantum > 0.0 && z < best + quantum * (1.0 - 1e-6)) return true;
        if (rel_gap > 0.0 && z <= best + rel_gap * std::fabs(best)) { gap_pruned = true; return true; }
        return false;
    }

    // Branching column: among the fractional ones the most valuable (largest cost), ties by fractionality.  The tick's objective is
    // (W - idx)/W x (normalised resources of the request): deciding the expensive placements first — big requests on the low-index workers —
    // fixes the part of the packing everything else has to fit around, and the up-branch-first dive lands on good incumbents early.  Against
    // "most fractional" this turns unsaturated multi-class models from time-outs into millisecond proofs at <= 8 workers.
    static int pick_fractional(const Tab &t) {
        int j = -1; double bc = -1.0, bf = 0.0;
        for (int k = 0; k < t.n; k++) {
            const double v = t.x[k], fr = std::fabs(v - std::round(v));
            if (fr <= INT_TOL) continue;
            const double c = t.cost[k];
            if (c > bc + 1e-12 || (c > bc - 1e-12 && fr > bf)) { bc = c; bf = fr; j = k; }
        }
        return j;
    }

    // phase 1: maximise
    void dfs_opt(Tab &t) {
        nodes++;
        if (aborted || time_up()) return;
        if (cert_stop || (rel_gap > 0.0 && !in_lns && work_limit < 0 && certified())) { cert_stop = true; aborted = true; return; }
        if (node_budget >= 0 && nodes > node_budget) { aborted = true; return; }
        {   // a node of a large tableau costs a copy of it (~20 ns per element with the page faults): check the clock every time, and do not start
            // a copy that cannot finish in the time that is left
            const double elems = (double)t.ma * (double)t.width();
            if (elems > 1.0e6 && deadline - wall() < 2.0e-8 * elems) { timed_out = true; return; }
        }
        int s = solve_counted(t);
        static const bool node_trace = getenv("HQMILP_NODE_TRACE") != nullptr;
        if (node_trace && !in_lns && n > 100) fprintf(stderr, "[node] %ld status %d z %.9f best %.9f rows %s ma %d\n", nodes, s, s == LP_OPT ? t.objective() : -1.0, best, t.R == &RCm ? "cuts" : "plain", t.ma);
        if (s == LP_INFEAS && rc_valid && t.R == &RCm) {
            // "Infeasible" on the rows + cuts closes a whole subtree, and the dense tableau is never refactored: with a few hundred cut rows of mixed scale in it the dual
            // ratio test can come up empty on an LP that has points (price_fuzz seed 2057: four such nodes closed a tree in eleven, 1.04e-4 below a point the host path
            // held).  The claim is checked on the model's OWN rows under the node's bounds, from a cold start: infeasible there too — closed; otherwise the subtree is
            // searched on those rows (slower bound, sound).
            Tab plain; plain.init(&R, c, std::vector<double>(t.lb.begin(), t.lb.begin() + n), std::vector<double>(t.ub.begin(), t.ub.begin() + n)); plain.deadline = deadline;
            const int ps = solve_counted(plain);
            if (node_trace && !in_lns && n > 100) fprintf(stderr, "[node] %ld infeasible on the cut rows; on the model's own rows: status %d z %.9f\n", nodes, ps, ps == LP_OPT ? plain.objective() : -1.0);
            if (ps == LP_LIMIT) { timed_out = true; return; }
            if (ps != LP_OPT) { watch_closed(t, "infeasible, confirmed on the model's rows", -1.0); return; }
            cut_infeas_refuted++;
            t = std::move(plain);
            s = LP_OPT;
        }
        if (s != LP_OPT) { if (s == LP_LIMIT) timed_out = true; else watch_closed(t, "infeasible", -1.0); return; }
        double z = t.objective();
        if (nodes == 1 && tracing && !in_lns) fprintf(stderr, "[milp] n=%d root LP %.9f incumbent %.9f rel gap %.3e\n", n, z, have ? best : -1.0, have ? (z - best) / best : 0.0);
        bool on_cut_rows = rc_valid && t.R == &RCm;
        bool drifted = false;
        if (on_cut_rows) {
            // A node's bound is its tableau's objective only while the tableau is DUAL FEASIBLE — and this one is never refactored: cut rows bring coefficient ranges of
            // 1e6 into it, and after a few hundred pivots the reduced costs can have drifted (price_fuzz seed 2057: a node closed at 21.586 whose cold solve gives
            // 21.614, above the threshold; a point 1.04e-4 better than the "certified" incumbent sat in it).  On the rows + cuts a node that is about to be closed by
            // its bound on a drifted tableau is solved again from a cold start over the same rows and bounds, and a drifted tableau tightens no bounds below.
            drifted = !consistent(t);
            const bool closes = have && (z <= best + 1e-12 * std::fabs(best) || (quantum > 0.0 && z < best + quantum * (1.0 - 1e-6)) || (rel_gap > 0.0 && z <= best + rel_gap * std::fabs(best)));
            if (drifted && closes) {
                Tab cold; cold.init(&RCm, c, std::vector<double>(t.lb.begin(), t.lb.begin() + n), std::vector<double>(t.ub.begin(), t.ub.begin() + n)); cold.deadline = deadline;
                const int cs = solve_counted(cold);
                if (node_trace && !in_lns && n > 100) fprintf(stderr, "[node] %ld would close at %.9f on a drifted tableau; cold solve of the same rows and bounds: status %d z %.9f\n", nodes, z, cs, cs == LP_OPT ? cold.objective() : -1.0);
                if (cs == LP_LIMIT) { timed_out = true; return; }
                if (cs == LP_INFEAS || (cs == LP_OPT && !consistent(cold))) {   // "infeasible" on the cut rows is confirmed on the model's own rows (as above) — and a cold solve that is ITSELF
                    // inconsistent (the cut rows' scale again) decides nothing: the node goes on on the model's own rows
                    Tab plain; plain.init(&R, c, std::vector<double>(t.lb.begin(), t.lb.begin() + n), std::vector<double>(t.ub.begin(), t.ub.begin() + n)); plain.deadline = deadline;
                    const int ps = solve_counted(plain);
                    if (ps == LP_LIMIT) { timed_out = true; return; }
                    if (ps != LP_OPT) { watch_closed(t, "cold solve infeasible, confirmed on the model's rows", -1.0); return; }
                    t = std::move(plain);
                } else t = std::move(cold);
                drift_resolves++;
                z = t.objective();
                on_cut_rows = rc_valid && t.R == &RCm;
                drifted = on_cut_rows && !consistent(t);
            }
        }
        if (cannot_improve(z)) { watch_closed(t, "bound", z); return; }
        int j = pick_fractional(t);
        if (flags_first && (int)col_group.size() == n) {   // a fractional FLAG (a global 0/1 column of the builder: col_group < 0) before any placement column
            int jf = -1; double bf = 0.0;
            for (int k = 0; k < n; k++) { if (col_group[k] >= 0) continue; const double fr = std::fabs(t.x[k] - std::round(t.x[k])); if (fr > INT_TOL && fr > bf) { bf = fr; jf = k; } }
            if (jf >= 0) j = jf;
        }
        const int j_flag = (flags_first && j >= 0 && (int)col_group.size() == n && col_group[j] < 0) ? j : -1;
        if (j >= 0 && (nodes == 1 || (nodes & 63) == 0)) {  // root and every 64th node: try to close the gap from this LP point
            round_and_repair(t);
            if (cannot_improve(z)) { watch_closed(t, "bound after round-and-repair", z); return; }
        }
        if (j < 0) {
            std::vector<double> xi(t.x.begin(), t.x.begin() + n);
            for (auto &v : xi) v = std::round(v);
            if (on_cut_rows) {   // (a drifted tableau's "integral optimum" need not satisfy the model: checked against its own rows and bounds first)
                bool ok = true;
                for (int k = 0; k < n && ok; k++) if (xi[k] < lb[k] - FEAS_TOL || xi[k] > ub[k] + FEAS_TOL) ok = false;
                for (int i = 0; i < R.m && ok; i++) { const double a = R.activity(i, xi.data()); if (a < R.lo[i] - FEAS_TOL || a > R.hi[i] + FEAS_TOL) ok = false; }
                if (!ok) {
                    if (drifted) {   // once more from a cold start: the node is searched on from an exact tableau
                        Tab cold; cold.init(&RCm, c, std::vector<double>(t.lb.begin(), t.lb.begin() + n), std::vector<double>(t.ub.begin(), t.ub.begin() + n)); cold.deadline = deadline;
                        nodes--;   // (the same node again)
                        drift_resolves++;
                        dfs_opt(cold);
                    }
                    return;
                }
            }
            double zz = 0.0; for (int k = 0; k < n; k++) zz += c[k] * xi[k];
            if (!have || zz > best || !on_cut_rows) { have = true; bx = std::move(xi); best = zz; }
            return;
        }
        if (have && !drifted) {
            // Reduced-cost bound tightening: moving a nonbasic column k by delta away from its bound costs at least |d_k| delta of the LP bound z (the
            // duals of the active rows bound the whole model), and only points worth `need` or more are of interest below this node.  With the
            // near-optimal incumbents of the window search the room z - need is a few objective quanta, which pins most columns.
            const double need = quantum > 0.0 ? best + quantum * (1.0 - 1e-6) : best + 1e-12 * std::fabs(best);
            const double room = z - need;
            if (room >= 0.0) {
                for (int k = 0; k < n; k++) {
                    if (t.st[k] == BASIC || t.lb[k] == t.ub[k]) continue;
                    const double dk = std::fabs(t.d[k]);
                    if (dk < 1e-9) continue;
                    const double steps = std::floor(room / dk + 1e-9);
                    if (steps >= t.ub[k] - t.lb[k]) continue;
                    if (t.st[k] == AT_UP) t.lb[k] = t.ub[k] - steps; else t.ub[k] = t.lb[k] + steps;
                }
            }
        }
        {   // Branch on a ROW first where one offers itself: an active row whose coefficients are all equal (a batch-size row: how many tasks of the batch are
            // placed in total) has an integral activity at every integer point, and its slack is a column of the tableau like any other.  With identical
            // workers the LP optimum keeps such a total fractional while it shuffles single (worker, class) columns around: branching on the columns
            // enumerates the symmetric packings one by one (8 workers x 5 classes, 86 of 88 tasks fit: 2.4 M nodes without proof in 10 s), branching on
            // the total decides what the symmetric packings have in common.
            int ba = -1; double bfr = 1e-6;
            for (int a = 0; a < t.ma && strong && j_flag < 0; a++) {
                const double u = row_unit[(size_t)t.arow[a]];
                if (u <= 0.0) continue;
                const double act = t.x[t.n + a] / u, fr = std::fabs(act - std::round(act));
                if (fr > bfr) { bfr = fr; ba = a; }
            }
            if (ba >= 0) {
                const int k = t.n + ba;
                const double u = row_unit[(size_t)t.arow[ba]], act = t.x[k] / u;
                {
                    Tab up = t;
                    up.set_lb(k, std::ceil(act - INT_TOL) * u);
                    dfs_opt(up);
                    lp_iters += up.iters - t.iters;
                }
                t.set_ub(k, std::floor(act + INT_TOL) * u);
                dfs_opt(t);
                return;
            }
        }
        const int SB = 32;
        if (strong && have && j_flag < 0 && (double)t.ma * (double)t.width() <= 4.0e6) {  // two tableau copies per candidate: not for the large models
            // strong branching over the SB most valuable fractional columns: both children are solved, the column whose children lose the most
            // bound is branched on, and a child that cannot hold anything better fixes the column the other way at once
            std::vector<std::pair<double, int>> cand;
            for (int k = 0; k < n; k++) { double fr = std::fabs(t.x[k] - std::round(t.x[k])); if (fr > INT_TOL) cand.push_back({-t.cost[k], k}); }
            std::sort(cand.begin(), cand.end());
            if ((int)cand.size() > SB) cand.resize(SB);
            double best_score = -1.0; int best_k = -1;
            for (auto &cd : cand) {
                const int k = cd.second; const double vk = t.x[k];
                if (std::fabs(vk - std::round(vk)) <= INT_TOL) continue;  // became integral through an earlier fixing
                double dz[2]; bool dead[2];
                if (wall() > deadline) { timed_out = true; return; }  // a child LP of a large model takes milliseconds: check the clock per candidate
                for (int side = 0; side < 2; side++) {
                    Tab c = t;
                    if (side == 0) c.set_lb(k, std::ceil(vk - INT_TOL)); else c.set_ub(k, std::floor(vk + INT_TOL));
                    int cs = solve_counted(c);
                    nodes++;
                    if (cs == LP_LIMIT) { timed_out = true; return; }
                    // (on the rows + cuts a child's warm tableau may have drifted — see the top of this function: there the children only SCORE the candidates; whether
                    // one of them is closed is decided when the search gets to it, with the cold re-solve behind it)
                    dead[side] = !on_cut_rows && (cs != LP_OPT || cannot_improve(c.objective()));
                    dz[side] = cs == LP_OPT ? std::max(0.0, z - c.objective()) : (on_cut_rows ? z : 1e9);
                    if (!on_cut_rows && !dead[side] && pick_fractional(c) < 0) {  // the child's LP optimum is integral: a better incumbent, and this child is finished
                        bx.assign(c.x.begin(), c.x.begin() + n);
                        for (auto &v : bx) v = std::round(v);
                        double zz = 0.0; for (int q = 0; q < n; q++) zz += this->c[q] * bx[q];
                        best = zz; have = true; dead[side] = true;
                    }
                }
                if (dead[0] && dead[1]) { watch_closed(t, "both strong-branching children dead", z); return; }  // neither child can improve: the node is done
                if (dead[0] || dead[1]) {        // one child is empty: tighten the column here and re-solve the node
                    if (dead[0]) t.set_ub(k, std::floor(vk + INT_TOL)); else t.set_lb(k, std::ceil(vk - INT_TOL));
                    int cs = solve_counted(t);
                    if (cs != LP_OPT) { if (cs == LP_LIMIT) timed_out = true; return; }
                    z = t.objective();
                    if (cannot_improve(z)) { watch_closed(t, "bound after a strong-branching fixing", z); return; }
                    continue;
                }
                const double score = std::max(dz[0], 1e-9) * std::max(dz[1], 1e-9);
                if (score > best_score) { best_score = score; best_k = k; }
            }
            const int jj = pick_fractional(t);
            if (jj < 0) { dfs_opt(t); return; }  // integral after the fixings: let the node handler record it (cheap re-solve)
            if (best_k >= 0 && best_score > 1e-15 && std::fabs(t.x[best_k] - std::round(t.x[best_k])) > INT_TOL) j = best_k; else j = jj;  // degenerate scores: the static rule
        }
        double v = t.x[j];
        {
            Tab up = t;
            up.set_lb(j, std::ceil(v - INT_TOL));
            dfs_opt(up);
            lp_iters += up.iters - t.iters;
        }
        t.set_ub(j, std::floor(v + INT_TOL));
        dfs_opt(t);
    }
    // feasibility search (costs kept, so the dual ratio test stays non-degenerate): first integral point or false
    bool dfs_feas(Tab &t, std::vector<double> &out) {
        nodes++;
        if (time_up()) return false;
        int s = solve_counted(t);
        if (s != LP_OPT) { if (s == LP_LIMIT) timed_out = true; return false; }
        int j = pick_fractional(t);
        if (j < 0) { out.assign(t.x.begin(), t.x.begin() + n); for (auto &v : out) v = std::round(v); return true; }
        if ((double)t.ma * (double)t.width() <= 4.0e6) {
            // strong branching for the feasibility search: a probe of the tie-break phase is mostly a PROOF that nothing fits (the objective row pins
            // the point to the optimal face); solving both children of the most valuable fractional columns finds the column whose children die
            // soonest, and a column with one dead child is fixed on the spot
            std::vector<std::pair<double, int>> cand;
            for (int k = 0; k < n; k++) { double fr = std::fabs(t.x[k] - std::round(t.x[k])); if (fr > INT_TOL) cand.push_back({-t.cost[k], k}); }
            std::sort(cand.begin(), cand.end());
            if ((int)cand.size() > 16) cand.resize(16);
            const double z = t.objective();
            double best_score = -1.0; int best_k = -1;
            for (auto &cd : cand) {
                const int k = cd.second; const double vk = t.x[k];
                if (std::fabs(vk - std::round(vk)) <= INT_TOL) continue;
                if (time_up()) return false;
                double dz[2]; bool dead[2];
                for (int side = 0; side < 2; side++) {
                    Tab c = t;
                    work += (double)(t.ma + 1) * (double)t.width();
                    if (side == 0) c.set_lb(k, std::ceil(vk - INT_TOL)); else c.set_ub(k, std::floor(vk + INT_TOL));
                    const int cs = solve_counted(c);
                    nodes++;
                    if (cs == LP_LIMIT) { timed_out = true; return false; }
                    dead[side] = cs != LP_OPT;
                    dz[side] = dead[side] ? 1e9 : std::max(0.0, z - c.objective());
                    if (!dead[side] && pick_fractional(c) < 0) { out.assign(c.x.begin(), c.x.begin() + n); for (auto &v : out) v = std::round(v); return true; }
                }
                if (dead[0] && dead[1]) return false;
                if (dead[0] || dead[1]) {
                    if (dead[0]) t.set_ub(k, std::floor(vk + INT_TOL)); else t.set_lb(k, std::ceil(vk - INT_TOL));
                    const int cs = solve_counted(t);
                    if (cs != LP_OPT) { if (cs == LP_LIMIT) timed_out = true; return false; }
                    continue;
                }
                const double score = std::max(dz[0], 1e-9) * std::max(dz[1], 1e-9);
                if (score > best_score) { best_score = score; best_k = k; }
            }
            j = pick_fractional(t);
            if (j < 0) { out.assign(t.x.begin(), t.x.begin() + n); for (auto &v : out) v = std::round(v); return true; }
            if (best_k >= 0 && best_score > 1e-15 && std::fabs(t.x[best_k] - std::round(t.x[best_k])) > INT_TOL) j = best_k;
        }
        double v = t.x[j];
        {
            Tab up = t;
            up.set_lb(j, std::ceil(v - INT_TOL));
            if (dfs_feas(up, out)) return true;
        }
        t.set_ub(j, std::floor(v + INT_TOL));
        return dfs_feas(t, out);
    }

    // Large-neighbourhood improvement of the incumbent: slide a window over the columns (the tick's models are worker-major, so a window is a
    // few neighbouring workers), keep every column outside it at its incumbent value, and solve the small model that is left exactly.  Any
    // improvement is an improvement of the whole incumbent (the rows are checked with the fixed part moved to their bounds).  This is where a
    // time-limited solve gets its quality from once the search cannot be finished: the dive's incumbent is 0.3-0.7 % below HiGHS's on
    // half-full clusters, the windows recover most of that.  Deterministic: window order, sizes and per-window node budgets are fixed.
    bool in_lns = false;
    // window = columns [start, start + win) and, with stride > 0, also [start + stride, start + stride + win): neighbouring workers, or two groups
    // of workers far apart (tasks move between the early, well-paid workers and the late ones)
    std::vector<int> lns_local; std::vector<char> lns_row_mark; double lns_stat[5] = {0, 0, 0, 0, 0};
    // one neighbourhood: the columns `wcols` are free, every other column keeps its incumbent value; true when the incumbent improved
    bool lns_solve(const std::vector<int> &wcols, double until, long cap, const std::vector<double> *base = nullptr) {
        const std::vector<double> &fx = base ? *base : bx;  // values of the columns outside the neighbourhood
        const int wn = (int)wcols.size();
        if (coff.empty()) build_columns();
        if (lns_local.empty()) { lns_local.assign(n, -1); lns_row_mark.assign(R.m, 0); }
        std::vector<int> &local = lns_local; std::vector<char> &row_mark = lns_row_mark;
        bool any_room = false;
        for (int j : wcols) if (ub[j] > lb[j]) any_room = true;
        if (!any_room) return false;
        std::vector<int> rows_used; std::vector<std::pair<int, double>> terms;
        CompSolver sub; sub.n = wn; sub.in_lns = true; sub.deadline = until;  // per-window limit = the node cap below, not a clock: same answer on every replica
        sub.c.resize(wn); sub.lb.resize(wn); sub.ub.resize(wn);
        for (int q = 0; q < wn; q++) { sub.c[q] = c[wcols[q]]; sub.lb[q] = lb[wcols[q]]; sub.ub[q] = ub[wcols[q]]; }
        sub.R.n = wn;
        for (int q = 0; q < wn; q++) { const int j = wcols[q]; local[j] = q; for (int k = coff[j]; k < coff[j + 1]; k++) if (!row_mark[crow[k]]) { row_mark[crow[k]] = 1; rows_used.push_back(crow[k]); } }
        std::sort(rows_used.begin(), rows_used.end());
        for (int i : rows_used) {
            row_mark[i] = 0;
            terms.clear(); double fixed = 0.0;
            for (int k = R.off[i]; k < R.off[i + 1]; k++) { const int j = R.col[k]; if (local[j] >= 0) terms.push_back({local[j], R.coef[k]}); else fixed += R.coef[k] * fx[j]; }
            sub.R.add(terms, R.lo[i] <= -INF ? -INF : R.lo[i] - fixed, R.hi[i] >= INF ? INF : R.hi[i] - fixed);
        }
        for (int j : wcols) local[j] = -1;
        double z0 = 0.0;
        if (!base) {
            sub.bx.resize(wn); for (int q = 0; q < wn; q++) sub.bx[q] = bx[wcols[q]];
            for (int j = 0; j < wn; j++) z0 += sub.c[j] * sub.bx[j];
            sub.have = true; sub.best = z0;
        } else {  // RENS: the neighbourhood of the LP point — its columns between floor and ceiling, no incumbent of its own; worth it only above the incumbent
            double zfix = 0.0; for (int j = 0; j < n; j++) zfix += c[j] * fx[j];
            for (int q = 0; q < wn; q++) { const int j = wcols[q]; zfix -= c[j] * fx[j]; sub.lb[q] = std::max(lb[j], std::floor(fx[j] + INT_TOL)); sub.ub[q] = std::min(ub[j], sub.lb[q] + 1.0); }
            z0 = best - zfix;  // what the neighbourhood has to beat
            sub.have = true; sub.best = z0; sub.bx.assign(wn, 0.0);  // a cutoff, not a point: adopted below only if beaten
        } sub.lattice = quantum;  // the window's costs are a subset of this model's: an improvement is at least one quantum
        sub.node_cap = cap;
        std::vector<double> xw;
        const int st = sub.run(false, xw);
        nodes += sub.nodes; lp_iters += sub.lp_iters;
        if (base && tracing) fprintf(stderr, "[milp] rens sub: st %d best %.9f cutoff %.9f nodes %ld\n", st, sub.best, z0, sub.nodes);
        if ((st == 1 || st == 2) && sub.best > z0 + 1e-12 * std::fabs(best)) {
            if (base) bx = fx;
            for (int q = 0; q < wn; q++) bx[wcols[q]] = xw[q];
            double zz = 0.0; for (int j = 0; j < n; j++) zz += c[j] * bx[j];
            best = zz;
            return true;
        }
        return false;
    }
    bool lns_windows(double until, int win, int stride = 0, int groups = 2) {
        if (!have || in_lns || n <= win) return false;
        bool improved = false;
        std::vector<int> wcols;
        for (int start = 0; start < n && wall() < until && !timed_out && !certified(); start += (stride > 0 ? win : win / 2)) {
            wcols.clear();
            for (int j = start; j < std::min(n, start + win); j++) wcols.push_back(j);
            if (stride > 0) {
                if (start + stride >= n) break;
                for (int g = 1; g < groups; g++) for (int j = start + g * stride; j < std::min(n, start + g * stride + win); j++) wcols.push_back(j);
            }
            const long n0 = nodes; const double t0 = wall();
            const bool imp = lns_solve(wcols, until, lns_cap);
            improved |= imp;
            if (tracing) { lns_stat[0] += 1; lns_stat[1] += imp; lns_stat[2] += (double)(nodes - n0); lns_stat[3] += wall() - t0; if (nodes - n0 >= lns_cap) lns_stat[4] += 1; }
        }
        if (tracing) { fprintf(stderr, "[milp]   windows win=%d stride=%d: %g solved, %g improved, %g nodes, %.3fs, %g capped; quantum %g incumbent %.9f\n", win, stride, lns_stat[0], lns_stat[1], lns_stat[2], lns_stat[3], lns_stat[4], quantum, best); for (auto &v : lns_stat) v = 0; }
        return improved;
    }
    // the whole schedule: neighbours (32, 64 columns), then pairs of 32-column groups at halving distances; repeated while something improves
    // LP-guided windows.  Blocks = the connected components of the model without its wide rows (a worker's columns; the batch-size rows are what
    // is left out).  Where the root LP pays a block more than the incumbent does there is something to gain, where it pays less there is something to
    // give: windows pair the blocks with the largest deficit with those of the largest surplus, which the index-based windows only meet by chance.
    std::vector<int> block_of; int n_blocks = 0;
    std::vector<double> row_unit;  // per row: u > 0 when every coefficient of the row equals u (three terms or more) — its activity / u is an integer at every integer point
    std::vector<double> lag_value, lag_rcost;  // after lagrangian_bound(): V_b(pi*) per block and the reduced costs c - pi* A per column
    void find_blocks() {
        DSUlite d(n);
        const int wide = std::max(12, n / 32);
        for (int i = 0; i < R.m; i++) { if (R.off[i + 1] - R.off[i] > wide) continue; for (int k = R.off[i] + 1; k < R.off[i + 1]; k++) d.unite(R.col[R.off[i]], R.col[k]); }
        block_of.assign(n, -1); n_blocks = 0;
        std::vector<int> id(n, -1);
        for (int j = 0; j < n; j++) { const int r = d.find(j); if (id[r] < 0) id[r] = n_blocks++; block_of[j] = id[r]; }
    }
    bool lns_guided(double until) {
        if (lp_x.empty() || in_lns || !have) return false;
        if (block_of.empty()) find_blocks();
        if (n_blocks < 4 || n_blocks > n / 2) return false;
        bool improved = false;
        for (int pass = 0; pass < 6 && wall() < until && !timed_out && !certified(); pass++) {
            std::vector<double> gain(n_blocks, 0.0);
            if (!lag_value.empty()) {  // regret at the Lagrangian's prices: V_b(pi*) - (c - pi* A).x_b >= 0; the regrets and the priced slack of the wide rows add up to the whole gap
                for (int b = 0; b < n_blocks; b++) gain[b] = lag_value[b];
                for (int j = 0; j < n; j++) gain[block_of[j]] -= lag_rcost[j] * bx[j];
            } else
            for (int j = 0; j < n; j++) gain[block_of[j]] += c[j] * (lp_x[j] - bx[j]);
            std::vector<int> order(n_blocks); std::iota(order.begin(), order.end(), 0);
            std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return gain[a] > gain[b]; });
            const int half = 4, shift = pass * 2;
            if (2 * (half + shift) > n_blocks) break;
            std::vector<char> take(n_blocks, 0);
            for (int k = 0; k < half; k++) { take[order[shift + k]] = 1; take[order[n_blocks - 1 - shift - k]] = 1; }
            std::vector<int> wcols;
            for (int j = 0; j < n; j++) if (take[block_of[j]]) wcols.push_back(j);
            if (wcols.size() > 160) continue;
            const double before = best;
            if (lns_solve(wcols, until, std::max<long>(lns_cap, 8000))) { improved = true; pass = -1; }  // the ranking has changed: start over
            if (tracing) fprintf(stderr, "[milp]   guided window (%d columns, shift %d): %.9f -> %.9f\n", (int)wcols.size(), shift, before, best);
        }
        return improved;
    }
    // Lagrangian bound over the wide rows.  The model without its wide rows (batch sizes; a handful) is one small LP per block (a worker: 8 columns x
    // 3 rows), and for ANY multipliers pi >= 0 on the wide rows   pi.h + sum_blocks max { (c - pi A).x : x in block }   bounds the model from above —
    // at the best pi it IS the LP bound.  That makes the bound of an 8 192 x 3 080 model a matter of 9-variable cutting-plane LPs (Kelley) and a few
    // hundred sweeps over 1 024 tiny block LPs, where the dense tableau of the whole model does not finish in seconds.  Any iterate gives a valid
    // bound; the loop stops when the master's lower estimate is within 1e-5 of the best bound seen.
    bool lagrangian_bound(double until, double *bound_out) {
        if (block_of.empty()) find_blocks();
        if (n_blocks < 8) return false;
        for (int j = 0; j < n; j++) if (lb[j] != 0.0) return false;
        // rows: inside one block, or wide
        std::vector<int> wide; std::vector<int> row_block(R.m, -1);
        for (int i = 0; i < R.m; i++) {
            int b = -1; bool multi = false;
            for (int k = R.off[i]; k < R.off[i + 1]; k++) { const int bb = block_of[R.col[k]]; if (b < 0) b = bb; else if (bb != b) { multi = true; break; } }
            if (multi) { if (R.lo[i] > -INF || R.hi[i] >= INF) return false; wide.push_back(i); }
            else { if (R.lo[i] > FEAS_TOL || R.hi[i] < -FEAS_TOL) return false; row_block[i] = b; }
        }
        const int nw = (int)wide.size();
        if (nw == 0 || nw > 64) return false;
        struct Block { std::vector<int> cols; Rows rows; std::vector<double> lb, ub, cost; Tab tab; };
        std::vector<Block> blocks(n_blocks);
        std::vector<int> local(n, -1);
        for (int j = 0; j < n; j++) { Block &B = blocks[block_of[j]]; local[j] = (int)B.cols.size(); B.cols.push_back(j); B.lb.push_back(0.0); B.ub.push_back(ub[j]); }
        for (auto &B : blocks) { B.rows.n = (int)B.cols.size(); B.cost.resize(B.cols.size()); }
        std::vector<std::pair<int, double>> terms;
        for (int i = 0; i < R.m; i++) {
            if (row_block[i] < 0) continue;
            terms.clear();
            for (int k = R.off[i]; k < R.off[i + 1]; k++) terms.push_back({local[R.col[k]], R.coef[k]});
            if (!terms.empty()) blocks[row_block[i]].rows.add(terms, R.lo[i], R.hi[i]);
        }
        // wide rows column-wise: for column j the (wide row, coefficient) pairs
        std::vector<int> woff(n + 1, 0), wrow; std::vector<double> wcoef;
        for (int q = 0; q < nw; q++) for (int k = R.off[wide[q]]; k < R.off[wide[q] + 1]; k++) woff[R.col[k] + 1]++;
        for (int j = 0; j < n; j++) woff[j + 1] += woff[j];
        wrow.resize(woff[n]); wcoef.resize(woff[n]);
        { std::vector<int> cur(woff.begin(), woff.end() - 1);
          for (int q = 0; q < nw; q++) for (int k = R.off[wide[q]]; k < R.off[wide[q] + 1]; k++) { const int j = R.col[k]; wrow[cur[j]] = q; wcoef[cur[j]++] = R.coef[k]; } }
        std::vector<double> h(nw), pmax(nw, 0.0);
        for (int q = 0; q < nw; q++) h[q] = R.hi[wide[q]];
        for (int j = 0; j < n; j++) for (int k = woff[j]; k < woff[j + 1]; k++) if (wcoef[k] > 1e-12 && c[j] > 0.0) pmax[wrow[k]] = std::max(pmax[wrow[k]], c[j] / wcoef[k]);
        // one evaluation: value of the Lagrangian at pi, the activity of the wide rows and c.x at its maximiser
        std::vector<double> act(nw), xcur(n, 0.0);
        auto eval = [&](const std::vector<double> &pi, double *cx) -> double {
            std::fill(act.begin(), act.end(), 0.0);
            double total = 0.0, cxs = 0.0;
            for (int q = 0; q < nw; q++) total += pi[q] * h[q];
            for (auto &B : blocks) {
                const int nb = (int)B.cols.size();
                for (int l = 0; l < nb; l++) { const int j = B.cols[l]; double cj = c[j]; for (int k = woff[j]; k < woff[j + 1]; k++) cj -= pi[wrow[k]] * wcoef[k]; B.cost[l] = cj; }
                B.tab.init(&B.rows, B.cost, B.lb, B.ub);
                const double ops0 = B.tab.ops;
                if (B.tab.solve(100000) != LP_OPT) return INF;
                work += B.tab.ops - ops0;
                total += B.tab.objective();
                for (int l = 0; l < nb; l++) { const double xv = B.tab.x[l]; const int j = B.cols[l]; xcur[j] = xv; if (xv == 0.0) continue; cxs += c[j] * xv; for (int k = woff[j]; k < woff[j + 1]; k++) act[wrow[k]] += wcoef[k] * xv; }
            }
            *cx = cxs;
            return total;
        };
        std::vector<double> pi(nw, 0.0), pi_best(nw, 0.0);
        double cx = 0.0;
        double ub_best = eval(pi, &cx);
        if (ub_best >= INF) return false;
        const double theta_scale = std::max(ub_best, 1e-9);
        // master:  max  -pi.h - theta   s.t.  theta + pi.act_k >= cx_k  for every evaluated point k;   variables [pi (nw) | theta / theta_scale]
        Rows M; M.n = nw + 1;
        std::vector<double> mc(nw + 1), mlb(nw + 1, 0.0), mub(nw + 1);
        for (int q = 0; q < nw; q++) { mc[q] = -h[q] / theta_scale; mub[q] = pmax[q] > 0.0 ? pmax[q] : 0.0; }
        mc[nw] = -1.0; mub[nw] = 2.0;
        std::vector<std::vector<float>> cut_x; std::vector<double> cut_scale;  // the maximisers behind the cuts: their convex combination by the master's
                                                                                 // duals is (nearly) an optimal point of the LP relaxation — what the LP-guided windows compare with
        auto add_cut = [&]() {
            cut_x.emplace_back(xcur.begin(), xcur.end());
            terms.clear(); double sc = 1.0;
            for (int q = 0; q < nw; q++) if (act[q] != 0.0) { terms.push_back({q, act[q] / theta_scale}); sc = std::max(sc, std::fabs(act[q] / theta_scale)); }
            terms.push_back({nw, 1.0});
            for (auto &t : terms) t.second /= sc;
            M.add(terms, cx / theta_scale / sc, INF);
            cut_scale.push_back(sc);
        };
        add_cut();
        int it = 0;
        std::vector<double> lambda;
        for (; it < 600 && wall() < until; it++) {
            Tab mt; mt.init(&M, mc, mlb, mub);
            if (mt.solve(100000) != LP_OPT) break;
            const double lb_master = -mt.objective() * theta_scale;  // no pi does better than this
            lambda.assign(M.m, 0.0);
            for (int k = 0; k < M.m; k++) { const int a = mt.where[k]; if (a >= 0 && mt.st[M.n + a] != BASIC) lambda[k] = std::fabs(mt.d[M.n + a]) / cut_scale[k]; }
            if (ub_best - lb_master <= 1e-5 * std::fabs(ub_best)) break;
            for (int q = 0; q < nw; q++) pi[q] = mt.x[q];
            const double v = eval(pi, &cx);
            if (v >= INF) break;
            if (v < ub_best) { ub_best = v; pi_best = pi; }
            add_cut();
        }
        if (tracing) fprintf(stderr, "[milp] n=%d lagrangian bound over %d wide rows, %d blocks: %.9f after %d evaluations, t=%.3fs\n", n, nw, n_blocks, ub_best, it + 1, wall() - t_begin);
        // Price-directed construction: block after block (the model's column order: worker index), each takes the integer optimum of its own block at the
        // Lagrangian's prices — slightly discounted, so that a block which is indifferent about a scarce column takes it — within what the wide rows
        // still have left.  The prices keep the early, well-paid blocks from hogging what later ones need, which is where the plain greedy start loses.
        if (have && wall() < until) {
            std::vector<double> xi(n, 0.0), left(h);
            bool ok = true;
            for (auto &B : blocks) {
                const int nb = (int)B.cols.size();
                CompSolver sub; sub.n = nb; sub.in_lns = true; sub.deadline = until; sub.node_cap = 2000;
                sub.c.resize(nb); sub.lb.assign(nb, 0.0); sub.ub = B.ub; sub.R = B.rows;
                for (int l = 0; l < nb; l++) {
                    const int j = B.cols[l]; double cj = c[j];
                    for (int k = woff[j]; k < woff[j + 1]; k++) {
                        cj -= 0.999 * pi_best[wrow[k]] * wcoef[k];
                        if (wcoef[k] > 1e-12) sub.ub[l] = std::min(sub.ub[l], std::floor(std::max(0.0, left[wrow[k]]) / wcoef[k] + 1e-9));
                    }
                    sub.c[l] = cj;
                }
                std::vector<double> xb;
                const int st = sub.run(false, xb);
                nodes += sub.nodes;
                if (st == 0 || (int)xb.size() != nb) { ok = false; break; }
                for (int l = 0; l < nb; l++) {
                    const int j = B.cols[l]; if (xb[l] <= 0.0 || sub.c[l] <= 0.0) continue;  // columns the prices make worthless stay out
                    xi[j] = xb[l];
                    for (int k = woff[j]; k < woff[j + 1]; k++) left[wrow[k]] -= wcoef[k] * xb[l];
                }
            }
            for (int q = 0; q < nw; q++) if (left[q] < -FEAS_TOL) ok = false;  // several wide rows on one column could overshoot: then the point is simply not used
            if (ok) {
                const double before = best;
                greedy_from(xi);
                if (tracing) fprintf(stderr, "[milp] n=%d price-directed construction: incumbent %.9f -> %.9f, t=%.3fs\n", n, before, best, wall() - t_begin);
            }
        }
        double lsum = 0.0; for (double l : lambda) lsum += l;
        if (lsum > 0.0 && lambda.size() <= cut_x.size()) {
            lp_x.assign(n, 0.0);
            for (size_t k = 0; k < lambda.size(); k++) { if (lambda[k] == 0.0) continue; const double wgt = lambda[k] / lsum; const std::vector<float> &xk = cut_x[k]; for (int j = 0; j < n; j++) lp_x[j] += wgt * xk[j]; }
        }
        // what the guided windows rank blocks by: the block's Lagrangian value at the best multipliers, and the reduced costs c - pi A
        {
            double cx2 = 0.0;
            eval(pi_best, &cx2);
            lag_value.assign(n_blocks, 0.0);
            for (int b = 0; b < n_blocks; b++) lag_value[b] = blocks[b].tab.objective();
            lag_rcost.assign(n, 0.0);
            for (int j = 0; j < n; j++) { double cj = c[j]; for (int k = woff[j]; k < woff[j + 1]; k++) cj -= pi_best[wrow[k]] * wcoef[k]; lag_rcost[j] = cj; }
        }
        *bound_out = ub_best * (1.0 + 1e-9) + 1e-12;
        return true;
    }
    // Cheap windows first (1000 nodes each: most windows close far below that), and only when a whole round finds nothing the cap goes up 8x —
    // the few windows that hold the last improvements are plateaus of their own.
    long lns_cap = 1000;
    std::vector<double> lp_x;  // root LP optimum
    void lns_schedule(double until, bool escalate = true) {
        lns_cap = 1000;
        for (int round = 0; round < 16 && wall() < until && !timed_out && !certified(); round++) {
            bool any = false;
            any |= lns_windows(until, 32);
            any |= lns_windows(until, 64);
            for (int stride = n / 2; stride >= 64 && wall() < until && !timed_out; stride /= 2) any |= lns_windows(until, 32, stride);
            if (!any && n > 1024) any |= lns_windows(until, 128);  // nothing left for the small windows: 16 workers at a time
            if (!any && n > 1024) any |= lns_windows(until, 8, n / 8, 8);   // eight small groups spread over the whole index range
            if (!any && n > 1024) any |= lns_windows(until, 16, n / 4, 4);
            if (!any) any |= lns_guided(until);
            if (!any) { if (!escalate || lns_cap >= 64000) break; lns_cap *= 8; }
        }
    }
    long node_cap = -1;  // hard cap on the nodes of this solver (window sub-problems)

    // The price sweeps as a LATE phase (round 6): a model below the sweeps' own threshold (Sweeper::min_cols: a sweep costs ~80 us whatever the block count, and the host
    // tree is usually done first) that the host's search has NOT closed after its root cuts and a first proving phase goes to the sweeps after all — their
    // branch-and-price bound is the decomposition bound per node, which on clusters mid-run closes what the LP + cut bound of the tree leaves open (tools/price_fuzz.py
    // seeds 2005 / 2045: certified with the sweeps forced on, NeedMoreCompute on the product's default path).  Deterministic like the early phase: no clock inside.
    bool swept = false;
    bool late_sweeps() {
        if (swept || !sweeper || in_lns || rel_gap <= 0.0 || work_limit >= 0 || n < 16 || (int)col_group.size() != n) return false;
        for (int j = 0; j < n; j++) if (lb[j] != 0.0) return false;
        swept = true;
        const double tp0 = wall();
        hqprice::Request rq;
        rq.n = n; rq.m = R.m; rq.roff = R.off.data(); rq.rcol = R.col.data(); rq.rcoef = R.coef.data(); rq.rlo = R.lo.data(); rq.rhi = R.hi.data();
        rq.row_scale = row_scale.data(); rq.row_implied = (int)row_implied.size() == R.m ? row_implied.data() : nullptr; rq.col_group = col_group.data();
        rq.c = c.data(); rq.ub = ub.data(); rq.incumbent = have ? bx.data() : nullptr; rq.incumbent_value = best; rq.rel_gap = rel_gap; rq.trace = tracing; rq.time_limit_s = time_limit_s; rq.deadline_s = deadline;
        rq.polish = [this](std::vector<double> &x, double &value) { return polish_point(x, value); };
        const uint32_t keep_min = sweeper->min_cols;
        sweeper->min_cols = 1;   // (the threshold was the reason this model came to the host first)
        hqprice::Answer pa = hqprice::solve(rq, *sweeper);
        sweeper->min_cols = keep_min;
        price_us += (wall() - tp0) * 1e6;
        if (tracing) fprintf(stderr, "[milp] n=%d late price sweeps: ran %d (%s) sweeps %u rounds %u bound %.9f point %.9f incumbent %.9f, %.3f ms\n", n, (int)pa.ran, pa.why, pa.sweeps, pa.rounds, pa.ran ? pa.bound : -1.0, pa.x.empty() ? -1.0 : pa.x_value, have ? best : -1.0, (wall() - tp0) * 1e3);
        if (!pa.ran) return false;
        price_sweeps += (int)pa.sweeps; price_rounds += (int)pa.rounds;
        nodes += pa.sweeps;
        if (!pa.x.empty() && (!have || pa.x_value > best)) { bx = pa.x; best = pa.x_value; have = true; }
        root_bound = std::min(root_bound, pa.bound * (1.0 + 1e-9) + 1e-12);
        return certified();
    }

    // returns: 0 infeasible, 1 optimal, 2 incumbent only (time limit)
    int run(bool canonical, std::vector<double> &xout) {
        // (first: a model the sweeps certify needs neither the root tableau nor the greedy pass from zero below — 1.6 ms of a 3 ms tick at 8 k columns)
        if (sweeper && !in_lns && rel_gap > 0.0 && n >= (int)sweeper->min_cols && (int)col_group.size() == n) {
            // The coupled model by price sweeps (csrc/price.h): the wide rows priced out, every worker's block solved exactly per set of prices on the
            // MI355X, the incumbent rounded from the sweeps' own integer patterns.  Deterministic (no clock inside): replicas decide alike.
            const double tp0 = wall();
            hqprice::Request rq;
            rq.n = n; rq.m = R.m; rq.roff = R.off.data(); rq.rcol = R.col.data(); rq.rcoef = R.coef.data(); rq.rlo = R.lo.data(); rq.rhi = R.hi.data();
            rq.row_scale = row_scale.data(); rq.row_implied = (int)row_implied.size() == R.m ? row_implied.data() : nullptr; rq.col_group = col_group.data();
            rq.c = c.data(); rq.ub = ub.data(); rq.incumbent = have ? bx.data() : nullptr; rq.incumbent_value = best; rq.rel_gap = rel_gap; rq.trace = tracing; rq.time_limit_s = time_limit_s; rq.deadline_s = deadline;
            rq.polish = [this](std::vector<double> &x, double &value) { return polish_point(x, value); };
            bool lbzero = true; for (int j = 0; j < n && lbzero; j++) lbzero = lb[j] == 0.0;
            hqprice::Answer pa;
            if (lbzero) pa = hqprice::solve(rq, *sweeper);
            price_us = (wall() - tp0) * 1e6;
            if (tracing) fprintf(stderr, "[milp] n=%d price sweeps: ran %d (%s) sweeps %u rounds %u bound %.9f point %.9f incumbent %.9f, %.3f ms of which %.3f ms inside the sweeps\n", n, (int)pa.ran, pa.why, pa.sweeps, pa.rounds, pa.ran ? pa.bound : -1.0, pa.x.empty() ? -1.0 : pa.x_value, have ? best : -1.0, price_us / 1e3, sweeper->stat_sweep_us / 1e3);
            swept = true;
            if (pa.ran) {
                price_sweeps = (int)pa.sweeps; price_rounds = (int)pa.rounds;
                nodes += pa.sweeps;
                if (!pa.x.empty() && (!have || pa.x_value > best)) { bx = pa.x; best = pa.x_value; have = true; }
                root_bound = std::min(root_bound, pa.bound * (1.0 + 1e-9) + 1e-12);
                if (certified()) { canonical_done = false; xout = bx; trace("certified by the price sweeps"); return 1; }
                if (lazy_incumbent) { lazy_incumbent(); lazy_incumbent = nullptr; }
                if (have && n > 2000) {  // what the sweeps leave open goes to the host's window search, against their bound (smaller models: the tree below)
                    trace("window search against the price bound");
                    lns_schedule(deadline - 0.05);
                    xout = bx;
                    if (certified()) { canonical_done = false; trace("certificate only"); return 1; }
                    timed_out = true;
                    return 2;
                }
            }
        }
        if (lazy_incumbent && !in_lns) { lazy_incumbent(); lazy_incumbent = nullptr; }  // the sweeps did not take the component (or left it open): the search below wants the heuristic's point
        row_unit.assign((size_t)R.m, 0.0);
        for (int i = 0; i < R.m && !in_lns && n <= 2000; i++) {  // (not inside the windows of the large-model search: their sub-models are tuned as they are)
            const int a = R.off[i], b = R.off[i + 1];
            if (b - a < 3 || !(R.coef[a] > 0.0) || R.lo[i] > -INF || !(R.hi[i] < INF)) continue;  // `<=` rows only (the tick's batch-size and resource rows)
            bool same = true;
            for (int k = a + 1; k < b && same; k++) same = R.coef[k] == R.coef[a];
            if (same) row_unit[i] = R.coef[a];
        }
        Tab root; root.init(&R, c, lb, ub); root.deadline = deadline;
        greedy_from(lb);
        find_quantum();
        // Portfolio over restarts from the root (the incumbent carries over): a plain dive, then strong branching, each with a node budget that
        // quadruples per pair.  The dive finds incumbents and closes easy trees; strong branching proves plateaus the dive would need millions of
        // nodes for; neither dominates, and a problem that needs N nodes of its better strategy is done after < 3 N.
        long budget = std::min<long>(20000, std::max<long>(2000, 1300000 / std::max(1, n)));
        const double hard_deadline = deadline;
        const bool reserve_tail = n > 2000 && !in_lns;  // large model: the search below gets 70 % of the time, the window improvement the rest
        if (reserve_tail) { const double t0 = wall(); deadline = t0 + 0.7 * (hard_deadline - t0); root.deadline = deadline; }
        if (n > 2000 && have && !in_lns) {
            // Large model with block structure (an unsaturated tick of the whole cluster: one block per worker, the batch-size rows across): its LP bound
            // comes from the Lagrangian over the wide rows in a fraction of a second, and the rest of the time belongs to the window search, which stops
            // as soon as the incumbent is within rel_gap of that bound.
            const double tl0 = wall();
            double lag = INF;
            if (rel_gap > 0.0 && lagrangian_bound(tl0 + 0.3 * (hard_deadline - tl0), &lag)) {
                root_bound = lag;
                nodes++;
                deadline = hard_deadline;
                trace("window search against the Lagrangian bound");
                lns_schedule(deadline - 0.05);
                xout = bx;
                if (certified()) { canonical_done = false; trace("certificate only"); return 1; }
                timed_out = true;
                return 2;
            }
        }
        if (n > 2000 && have && !in_lns) {
            // Large model: the root LP gets half of the time.  From an all-at-upper start the dual simplex needs about one pivot per column, and when
            // most rows are violated there (unsaturated ticks: every resource row) each pivot touches the whole tableau — minutes at 8 k columns.  If
            // the LP is not done by then, the rest of the time improves the incumbent window by window instead (the answer is an incumbent either way).
            const double t0 = wall();
            root.deadline = t0 + 0.5 * (hard_deadline - t0);
            const int s0 = solve_counted(root);
            root.deadline = deadline;
            if (tracing) fprintf(stderr, "[milp] n=%d m=%d large-model root LP: status %d after %.3fs, %ld iterations, %d active rows, value %.6f incumbent %.6f\n", n, R.m, s0, wall() - t0, (long)root.iters, root.ma, s0 == LP_OPT ? root.objective() : -1.0, best);
            if (s0 == LP_LIMIT && hard_deadline - wall() > 0.2) {
                nodes++;
                deadline = hard_deadline;
                const double until = deadline - 0.05;
                lns_schedule(until);
                timed_out = true;
                lp_iters += root.iters;
                xout = bx;
                return 2;
            }
        }
        // Larger models: the window search BEFORE the tree.  A node of a 1000-column tableau costs a millisecond (a dive of 2000 nodes: seconds), the
        // windows take the greedy incumbent to within 1e-4 of the root LP bound in a fraction of that — and then the root closes the search.
        bool lns_done = false;
        if (!in_lns && have && n >= LNS_FIRST_COLS) {
            if (solve_counted(root) == LP_OPT) {
                root_bound = std::min(root_bound, root.objective()); lp_x.assign(root.x.begin(), root.x.begin() + n);  // the bound the windows work towards (dfs_opt finds the tableau solved)
                if (cuts_on && !certified()) { root_cuts(root); trace("root cuts done"); }
            }
            trace("window search first");
            lns_schedule(deadline, false);  // cheap windows only (what they leave open the tree below usually closes faster than bigger windows would), until they
                                            // stall or the incumbent is certified — not until a clock says so: replicas of a sharded scheduler walk the same sequence
            lns_done = true;
            if (!certified()) rins_at_cut_point();
        }
        const int first_strong = lns_done ? 1 : 0;  // with the windows' incumbent in hand the proof comes first, the dive (an incumbent finder) second
        auto search = [&]() {
            const double t_search = wall();
            long bud = budget;
            for (int phase = 0;; phase++) {
                strong = ((phase + first_strong) & 1) != 0;
                trace(strong ? "strong-branching phase" : "dive phase");
                // (with the sweeps still to come for a small model — late_sweeps() — the first phase runs on an eighth of its budget: a model it does not close by then goes to
                // the branch-and-price, whose per-node bound is the decomposition's; easy models never get there)
                const bool sweeps_ahead = phase == 0 && !swept && sweeper && !in_lns && rel_gap > 0.0 && work_limit < 0 && n >= 16 && (int)col_group.size() == n;
                aborted = false; node_budget = nodes + (sweeps_ahead ? std::max<long>(250, bud / 8) : bud);
                if (node_cap >= 0) node_budget = std::min(node_budget, node_cap);
                const bool on_cuts = rc_valid && rel_gap > 0.0 && work_limit < 0 && !in_lns;   // a certification phase with root cuts at hand: the tree works on the tightened rows
                if (phase > 0 || (on_cuts && root.R != &RCm)) { lp_iters += root.iters; root = Tab(); root.init(on_cuts ? &RCm : &R, c, lb, ub); root.deadline = deadline; }
                if (root_bound == INF && solve_counted(root) == LP_OPT) root_bound = root.objective();  // dfs_opt finds the tableau solved
                dfs_opt(root);
                if (cert_stop) break;
                if (!aborted || timed_out) break;
                if (node_cap >= 0 && nodes >= node_cap) { timed_out = true; break; }
                if (phase == 0 && !in_lns && !rc_valid && cuts_on && rel_gap > 0.0 && work_limit < 0 && n < LNS_FIRST_COLS && !certified()) {
                    // a small model the first dive did not close (the larger ones had their cut rounds before the windows): GMI rounds at the root now — on the 100-column
                    // clusters mid-run of tools/price_fuzz.py they take the LP bound from 6-7 % above the optimum down to the optimum itself
                    Tab r0; r0.init(&R, c, lb, ub); r0.deadline = deadline;
                    if (solve_counted(r0) == LP_OPT) { root_bound = std::min(root_bound, r0.objective()); root_cuts(r0); trace("root cuts done (after the first dive)"); }
                    if (certified()) { cert_stop = true; break; }
                }
                if (phase == 0 && !in_lns && !lns_done) {  // the dive did not finish: improve its incumbent before the expensive phases
                    // at most 30 % of what is left, and not more than three times what the dive itself took: a model that strong branching proves
                    // in a second must not spend six in here first
                    const double now = wall(), left = deadline - now, dive = now - t_search;
                    trace("window search");
                    lns_schedule(now + std::min(0.3 * left, std::max(0.05, 3.0 * dive)));
                    lns_done = true;
                    if (rel_gap > 0.0 && work_limit < 0 && certified()) { cert_stop = true; break; }
                }
                if (phase == 0 && !certified() && rins_at_cut_point() && certified()) { cert_stop = true; break; }
                if (phase == 0 && !swept && late_sweeps()) { cert_stop = true; break; }   // the first phase (on a shortened budget, below) has failed: the sweeps' branch-and-price before the tree gets its full budgets
                if (phase & 1) bud *= 4;
            }
            aborted = false; strong = false; node_budget = -1;
        };
        // A node of the tree costs a copy of the tableau: above ~8 M elements (10 ms) the windows are the better use of the time that is left,
        // and their incumbent is checked against the root bound like any other.
        const bool tree_affordable = in_lns || (double)std::max(root.ma, 1) * (double)root.width() <= 8.0e6;
        const double work0 = work;
        bool by_root_bound = false;  // certified by the root LP bound alone (no tree, or before it)
        if (certified()) by_root_bound = true;
        else if (tree_affordable) { search(); if (cert_stop) { by_root_bound = true; cert_stop = false; } }
        else timed_out = true;  // straight to the window search below
        const double work_search = work - work0;  // the tree search alone: the windows before it are not a measure of how hard the proof is
        lp_iters += root.iters;
        deadline = hard_deadline;
        if (timed_out && have && !in_lns && deadline - wall() > 0.2) {
            // out of its share of the time (large model), too large for a tree, or the LP gave up on size (tableau budget): what is left goes to the
            // window improvement — which stops as soon as the incumbent is within rel_gap of the root bound
            timed_out = false;
            trace("window search for the rest of the time");
            lns_schedule(deadline - 0.05);
            if (certified()) by_root_bound = true; else timed_out = true;
        }
        if (!timed_out && (gap_pruned || by_root_bound) && have && !in_lns && !canonical) { canonical_done = false; trace("certificate only (asked for)"); xout = bx; return 1; }  // the reference's own stopping rule
        if (!timed_out && (gap_pruned || by_root_bound) && have && !in_lns) {
            // Certified within rel_gap, which is all the reference asks of its solver.  The canonical answer needs the EXACT optimum: one more search
            // from the root with the exact pruning rule only, on a deterministic work budget (element updates, not seconds: every replica of a
            // sharded scheduler takes the same decision) — small coupled ticks finish it in milliseconds, the plateaus do not and keep the certificate.
            trace("certified; exact pass");
            bool exact = false;
            if (tree_affordable) {
                const double keep_gap = rel_gap; rel_gap = 0.0;
                work_limit = work + std::max(EXACT_PASS_WORK, work_search);
                root = Tab(); root.init(&R, c, lb, ub); root.deadline = deadline;
                lns_done = true;
                search();
                lp_iters += root.iters;
                work_limit = -1.0; rel_gap = keep_gap;
                exact = !timed_out;
            }
            if (!exact) {  // budget (or clock) ran out: the certificate stands, the answer is not canonical
                timed_out = false; canonical_done = false;
                trace("certificate only");
                xout = bx;
                return 1;
            }
        }
        if (tracing && !in_lns) fprintf(stderr, "[milp] n=%d t=%.6fs final incumbent %.9f timed_out %d nodes %ld (drifted nodes re-solved cold: %ld)\n", n, wall() - t_begin, have ? best : -1.0, (int)timed_out, nodes, drift_resolves);
        if (!have) return 0;
        xout = bx;
        if (timed_out) return 2;
        if (!canonical || n == 0) return 1;
        // Before any probe: is the optimum UNIQUE?  Solve the plain LP once more (tens of microseconds at the sizes this matters for).  If its vertex is integral, worth
        // exactly the incumbent, and dual non-degenerate by a margin — every nonbasic column's reduced cost, times the smallest step the column can take between integer
        // points (1 for a structural column, the row's own grid for a slack), exceeds the canonical rule's window of 1e-9 |best| — then every OTHER integer point
        // loses more than the window: x* is the only point the rule can return, and the probes below (one tableau copy + LP per placed column: 0.6 ms on an 80-column
        // tick of a few dozen ready tasks, 15 probes) have nothing to decide.  The (W - idx) / W factor of the tick's objective makes this the common case.
        if (!in_lns && unique_optimum(xout)) { trace("unique optimum: no tie-break probes"); return 1; }
        xout = bx;
        // phase 2: among vectors with c.x >= best - tol, minimise the LAST column, then the one before it, ... (bound probing,
        // one feasibility B&B per probe)
        double tol = 1e-9 * std::fabs(best);
        Rows R2 = R;
        double cs = 0.0; for (int k = 0; k < n; k++) cs = std::max(cs, std::fabs(c[k]));
        if (cs > 0.0) {
            std::vector<std::pair<int, double>> terms;
            for (int k = 0; k < n; k++) if (c[k] != 0.0) terms.push_back({k, c[k] / cs});
            R2.add(terms, (best - tol) / cs, INF);
        }
        std::vector<double> cur(bx);
        // One tableau carries the columns fixed so far; every probe is a copy of it with one tightened bound, re-optimised by
        // the dual simplex from the parent basis (a handful of pivots) instead of a cold start.
        work_limit = work + std::max(2.0e8, work);
        Tab warm; warm.init(&R2, c, lb, ub); warm.deadline = deadline;
        if (solve_counted(warm) != LP_OPT) { xout = cur; return 1; }  // cannot happen: `cur` is feasible for it
        {   // every column above its lower bound costs at least one probe = one tableau copy: skip the whole phase when that alone
            // exceeds the budget (large models keep the optimum they found — decided by size, identically on every replica)
            double probes = 0.0; for (int j = 0; j < n; j++) if (cur[j] > lb[j]) probes += 1.0;
            if (work + probes * (double)(warm.ma + 1) * (double)warm.width() > work_limit) { canonical_done = false; xout = cur; return 1; }
        }
        for (int j = n - 1; j >= 0; j--) {  // last column first
            double lo = lb[j], hi = cur[j];
            bool first = true;
            while (lo < hi) {
                // first probe just below the current value: most columns fail it immediately
                double mid = first ? hi - 1 : std::floor((lo + hi) / 2);
                first = false;
                {   // most probes fail, and most of those fail at the first step of their LP: the ratio test on the parent tableau bounds what the tightened bound costs
                    // (Tab::loss_if_ub) — below the threshold by a margin: no copy, no pivot, the probe has failed
                    const double loss = warm.loss_if_ub(j, mid);
                    if (loss >= INF || warm.objective() - loss < best - 2.0 * tol - 1e-12) { lo = mid + 1; continue; }
                }
                Tab t = warm;
                work += (double)(warm.ma + 1) * (double)warm.width();
                t.set_ub(j, mid);
                std::vector<double> sol;
                bool ok = dfs_feas(t, sol);
                lp_iters += t.iters - warm.iters;
                if (timed_out) { xout = cur; canonical_done = false; return 1; }  // optimal (phase 1 proved it) but the tie-break ran out of budget
                if (ok) { cur = sol; hi = sol[j]; } else lo = mid + 1;
            }
            if (tracing && !in_lns && cur[j] > lb[j]) fprintf(stderr, "[milp]   t=%.6fs column %d settled at %.0f (nodes %ld)\n", wall() - t_begin, j, hi, nodes);
            const bool unmoved = std::fabs(warm.x[j] - hi) <= FEAS_TOL;  // already sitting there: fixing it changes neither the point nor its optimality
            warm.set_lb(j, hi); warm.set_ub(j, hi);
            if (unmoved) continue;
            if (solve_counted(warm) != LP_OPT) {  // numerically lost the basis: rebuild it with the bounds fixed so far
                std::vector<double> flb(lb), fub(ub);
                for (int k = n - 1; k >= j; k--) flb[k] = fub[k] = cur[k];
                warm = Tab(); warm.init(&R2, c, flb, fub); warm.deadline = deadline;
                if (solve_counted(warm) != LP_OPT) { xout = cur; return 1; }
            }
        }
        xout = cur;
        trace("tie-break done");
        return 1;
    }
};

// Columns by descending cost, ties by ascending index (= a stable sort of the indices).  Large models: stable 11-bit counting passes over an order-preserving
// image of the doubles — the 2048 counters stay in L1, and a pass in which every key has the same digit (the costs of a tick span a few binades: most of the
// exponent digits) is skipped; small ones: the comparison sort.  (65 536 columns: the comparison sort took 5.5 ms, twice per coupled solve.)
void order_by_cost_desc(const double *c, int n, std::vector<int> &out) {
    out.resize((size_t)n);
    if (n < 4096) {
        std::vector<std::pair<double, int>> key((size_t)n);
        for (int j = 0; j < n; j++) key[(size_t)j] = {-c[j], j};
        std::sort(key.begin(), key.end());
        for (int j = 0; j < n; j++) out[(size_t)j] = key[(size_t)j].second;
        return;
    }
    std::vector<uint64_t> ka((size_t)n), kb((size_t)n); std::vector<int> ib((size_t)n);
    for (int j = 0; j < n; j++) {
        double v = c[j] == 0.0 ? 0.0 : c[j];  // (-0.0 and 0.0 are one cost)
        uint64_t u; memcpy(&u, &v, 8);
        u = (u >> 63) ? ~u : (u | 0x8000000000000000ull);  // ascending in the value
        ka[(size_t)j] = ~u;                                  // ascending in -value
        out[(size_t)j] = j;
    }
    const int BITS = 11, NB = 1 << BITS;
    uint32_t cnt[NB];
    uint64_t *src = ka.data(), *dst = kb.data(); int *isrc = out.data(), *idst = ib.data();
    for (int sh = 0; sh < 64; sh += BITS) {
        memset(cnt, 0, sizeof cnt);
        for (int j = 0; j < n; j++) cnt[(src[j] >> sh) & (NB - 1)]++;
        if (cnt[(src[0] >> sh) & (NB - 1)] == (uint32_t)n) continue;  // one digit for all: the pass would copy the arrays as they are
        uint32_t run = 0; for (int d = 0; d < NB; d++) { const uint32_t t = cnt[d]; cnt[d] = run; run += t; }
        for (int j = 0; j < n; j++) { const uint32_t p = cnt[(src[j] >> sh) & (NB - 1)]++; dst[p] = src[j]; idst[p] = isrc[j]; }
        std::swap(src, dst); std::swap(isrc, idst);
    }
    if (isrc != out.data()) memcpy(out.data(), isrc, (size_t)n * sizeof(int));  // (an odd number of passes ran)
}

struct DSU {
    std::vector<int> p;
    explicit DSU(int n) : p(n) { std::iota(p.begin(), p.end(), 0); }
    int find(int a) { while (p[a] != a) { p[a] = p[p[a]]; a = p[a]; } return a; }
    void unite(int a, int b) { a = find(a); b = find(b); if (a != b) p[std::max(a, b)] = std::min(a, b); }
};

// Sparse primal heuristic on the whole model (any size; O(passes x nnz)).  Start: every count column 0, every zero-cost bool that a
// `>=` row needs at x = 0 (the "blocker short" flags, scheduler/solver.rs:233-253) at 1 — i.e. every lower-priority batch capped at
// its cut — then raise columns greedily, most valuable first, as far as the rows allow; drop the flags whose `>=` row meanwhile
// holds without them (that lifts the caps they imposed) and raise again.  The result is a priority-respecting maximal packing.
// Returns false when the start point is not feasible (then there is simply no incumbent from here).
bool sparse_greedy(const Model &mdl, const std::vector<double> &ub, std::vector<double> &x) {
    const int n = mdl.ncols(), m = mdl.nrows();
    std::vector<int> coff(n + 1, 0), crow; std::vector<double> ccoef;
    for (int k = 0; k < (int)mdl.rcol.size(); k++) coff[mdl.rcol[k] + 1]++;
    for (int j = 0; j < n; j++) coff[j + 1] += coff[j];
    crow.resize(mdl.rcol.size()); ccoef.resize(mdl.rcol.size());
    { std::vector<int> cur(coff.begin(), coff.end() - 1);
      for (int i = 0; i < m; i++) for (int k = mdl.roff[i]; k < mdl.roff[i + 1]; k++) { int j = mdl.rcol[k]; crow[cur[j]] = i; ccoef[cur[j]++] = mdl.rcoef[k]; } }
    x.assign(n, 0.0);
    for (int i = 0; i < m; i++) {
        if (mdl.rtype[i] == ROW_MAX || mdl.rhs[i] <= 1e-9) continue;  // a `>=`/`==` row violated at 0: switch on a zero-cost bool with a large enough coefficient
        for (int k = mdl.roff[i]; k < mdl.roff[i + 1]; k++) {
            int j = mdl.rcol[k];
            if (mdl.kind[j] == COL_BOOL && mdl.obj[j] == 0.0 && mdl.rcoef[k] >= mdl.rhs[i] - 1e-9) { x[j] = 1.0; break; }
        }
    }
    std::vector<double> act(m, 0.0);
    for (int i = 0; i < m; i++) { double a = 0.0; for (int k = mdl.roff[i]; k < mdl.roff[i + 1]; k++) a += mdl.rcoef[k] * x[mdl.rcol[k]]; act[i] = a; }
    auto row_ok = [&](int i, double a) {
        const double tol = 1e-7 * (1.0 + std::fabs(mdl.rhs[i]));
        return mdl.rtype[i] == ROW_MAX ? a <= mdl.rhs[i] + tol : (mdl.rtype[i] == ROW_MIN ? a >= mdl.rhs[i] - tol : std::fabs(a - mdl.rhs[i]) <= tol);
    };
    for (int i = 0; i < m; i++) if (!row_ok(i, act[i])) return false;
    std::vector<int> order;
    order_by_cost_desc(mdl.obj.data(), n, order);
    for (int pass = 0; pass < 64; pass++) {
        bool changed = false;
        for (int j : order) {
            if (mdl.obj[j] <= 0.0) break;
            double step = ub[j] - x[j];
            for (int k = coff[j]; k < coff[j + 1] && step >= 1.0; k++) {
                const int i = crow[k]; const double a = ccoef[k];
                if (mdl.rtype[i] == ROW_EQ) { step = 0.0; break; }
                if (a > 0.0 && mdl.rtype[i] == ROW_MAX) { const double room = mdl.rhs[i] - act[i]; if (room < 0.5 * a) { step = 0.0; break; } step = std::min(step, std::floor(room / a + 1e-9)); }
                else if (a < 0.0 && mdl.rtype[i] == ROW_MIN) { const double room = act[i] - mdl.rhs[i]; if (room < -0.5 * a) { step = 0.0; break; } step = std::min(step, std::floor(room / -a + 1e-9)); }
            }
            if (step < 1.0) continue;
            x[j] += step; changed = true;
            for (int k = coff[j]; k < coff[j + 1]; k++) act[crow[k]] += ccoef[k] * step;
        }
        bool dropped = false;
        for (int j = 0; j < n; j++) {  // flags that are no longer needed
            if (mdl.kind[j] != COL_BOOL || mdl.obj[j] != 0.0 || x[j] != 1.0) continue;
            bool ok = true;
            for (int k = coff[j]; k < coff[j + 1] && ok; k++) ok = row_ok(crow[k], act[crow[k]] - ccoef[k]);
            if (!ok) continue;
            x[j] = 0.0; dropped = true;
            for (int k = coff[j]; k < coff[j + 1]; k++) act[crow[k]] -= ccoef[k];
        }
        if (!changed && !dropped) break;
        if (!dropped) break;  // nothing was unlocked: another raise pass cannot move anything
    }
    return true;
}

}  // namespace

void columns_by_cost_desc(const double *c, int n, std::vector<int> &out) { order_by_cost_desc(c, n, out); }

thread_local int g_fast_path = -1;
void set_fast_path(int on) { g_fast_path = on; }

// Row feasibility follows the solver the reference uses: HiGHS accepts a MIP solution whose rows are violated by at most
// mip_feasibility_tolerance = 1e-6 (absolute, original units).  That matters for one kind of row: the min_utilization pair of
// scheduler/solver.rs:501-540 multiplies its on/off flag by a value derived from an f32 (0.3 -> 0.30000001192...), e.g.
// "cpus >= 3.00000012 * flag", where every other term lives on the 1/10000 grid of ResourceAmount.  HiGHS takes 3 cpus as enough; a strict
// solver would demand 4 (found by the CPU campaign of tools/host_fuzz.py, 1 scenario in 6000).  Relaxing the row bounds by the tolerance is not
// an option for an LP-based search (every vertex then sits 1e-6 off the integers); instead the coefficient of a BOOL column is snapped to that
// grid when it is within the tolerance of it — for a 0/1 column this changes the row activity by at most the tolerance, i.e. it accepts
// exactly what HiGHS accepts there.
const double ROW_TOL = 1e-6;

namespace {
// true: `res` is the model's certified answer
bool solve_fast(const Model &mdl_in, double time_limit_s, double rel_gap, hqprice::Sweeper *sweeper, const double ts0, Result &res_out) {
    static const bool tracing_solve = getenv("HQMILP_TRACE") != nullptr;
    auto tmark = [&](const char *what) { if (tracing_solve && mdl_in.ncols() > 1000) fprintf(stderr, "[milp] solve(): %s at %.3f ms\n", what, (wall() - ts0) * 1e3); };
    // ---- the coupled tick's fast path: a large model whose builder said how it is made — one block of columns per worker (col_group), rows that share their leading
    // terms (row_lhs) — goes to the price sweeps AS IT IS (hqprice::solve_model): no snapped copy, no integer-hull pass, no components, no scaled row copy, no
    // column-wise copy for the greedy raise; the sweeps' own exact block solves need none of them.  What comes back certified is checked against every row of the model
    // right here and returned; anything else (a shape the flattening refuses, no certificate) takes the classic path below from the start, as if this had not run.
    // (c3p at BASELINE size: 1.7 ms of presolve + components + row copy and 1.0 ms of flattening on the build container -> 0.5 ms.)
    static const bool fast_env = !(getenv("HQMILP_FAST") && atoi(getenv("HQMILP_FAST")) == 0);  // (A/B switch)
    const bool fast_on = g_fast_path < 0 ? fast_env : g_fast_path != 0;                          // (... and the tests' own: tests/test_price.py compares the two paths)
    const int FAST_MIN_COLS = 2048;   // (= LAZY_GREEDY_COLS below: the models that go to the sweeps without an incumbent of the host's)
    if (fast_on && sweeper && rel_gap > 0.0 && mdl_in.ncols() >= FAST_MIN_COLS && mdl_in.ncols() >= (int)sweeper->min_cols && (int)mdl_in.col_group.size() == mdl_in.ncols() &&
        (int)mdl_in.row_lhs.size() == mdl_in.nrows() && (int)mdl_in.row_lhs_len.size() == mdl_in.nrows() && (int)mdl_in.start.size() != mdl_in.ncols()) {
        const int n = mdl_in.ncols(), m = mdl_in.nrows();
        hqprice::ModelView mv;
        mv.n = n; mv.m = m; mv.obj = mdl_in.obj.data(); mv.kind = mdl_in.kind.data(); mv.rtype = mdl_in.rtype.data(); mv.rhs = mdl_in.rhs.data();
        mv.roff = mdl_in.roff.data(); mv.rcol = mdl_in.rcol.data(); mv.rcoef = mdl_in.rcoef.data(); mv.col_group = mdl_in.col_group.data();
        mv.row_implied = (int)mdl_in.row_implied.size() == m ? mdl_in.row_implied.data() : nullptr; mv.row_lhs = mdl_in.row_lhs.data(); mv.row_lhs_len = mdl_in.row_lhs_len.data(); mv.list_off = mdl_in.list_off.data(); mv.list_col = mdl_in.list_col.data(); mv.n_lists = (int)mdl_in.list_off.size() - 1;
        if ((int)mdl_in.row_block.size() == m && (int)mdl_in.col_ub.size() == n) { mv.row_block = mdl_in.row_block.data(); mv.col_ub = mdl_in.col_ub.data(); }
        double cost_scale = 1.0;
        const double tf0 = wall();
        hqprice::Answer pa = hqprice::solve_model(mv, rel_gap, time_limit_s, ts0 + (time_limit_s > 0 ? time_limit_s : 1e18), tracing_solve, *sweeper, &cost_scale);
        if (tracing_solve) fprintf(stderr, "[milp] fast path: ran %d (%s) sweeps %u rounds %u bound %.9f point %.9f, %.3f ms of which %.3f ms inside the sweeps\n", (int)pa.ran, pa.why, pa.sweeps, pa.rounds, pa.ran ? pa.bound : -1.0, pa.x.empty() ? -1.0 : pa.x_value, (wall() - tf0) * 1e3, sweeper->stat_sweep_us / 1e3);
        if (pa.ran && (int)pa.x.size() == n && pa.bound * (1.0 + 1e-9) + 1e-12 <= pa.x_value + rel_gap * std::fabs(pa.x_value)) {
            // the point against EVERY row and bound of the model, in the model's own units
            bool ok = true;
            for (int j = 0; j < n && ok; j++) { const double v = pa.x[(size_t)j]; if (v < -1e-9 || std::fabs(v - std::floor(v + 0.5)) > 1e-9 || (mdl_in.kind[j] == COL_BOOL && v > 1.0 + 1e-9)) ok = false; }
            std::vector<double> lhs_act; std::vector<uint8_t> lhs_seen;   // the activity of a shared list of leading terms: once per list
            for (int i = 0; i < m && ok; i++) {
                double a = 0.0;
                int k0 = mdl_in.roff[i];
                const int L = mdl_in.row_lhs[i];
                if (L >= 0) {
                    if ((size_t)L >= lhs_seen.size()) { lhs_seen.resize((size_t)L + 1, 0); lhs_act.resize((size_t)L + 1, 0.0); }
                    if (!lhs_seen[(size_t)L]) { double s = 0.0; for (int k = mdl_in.list_off[L]; k < mdl_in.list_off[L + 1]; k++) s += pa.x[(size_t)mdl_in.list_col[k]]; lhs_act[(size_t)L] = s; lhs_seen[(size_t)L] = 1; }
                    a = lhs_act[(size_t)L];   // (the list's terms: coefficient 1 each, not among the row's stored terms)
                }
                for (int k = k0; k < mdl_in.roff[i + 1]; k++) a += mdl_in.rcoef[k] * pa.x[(size_t)mdl_in.rcol[k]];
                const double tol = std::max(ROW_TOL, 1e-9 * std::max(1.0, std::fabs(mdl_in.rhs[i])));   // (HiGHS's own mip_feasibility_tolerance, see ROW_TOL below)
                if (mdl_in.rtype[i] != ROW_MIN && a > mdl_in.rhs[i] + tol) ok = false;
                if (mdl_in.rtype[i] != ROW_MAX && a < mdl_in.rhs[i] - tol) ok = false;
            }
            double z_pt = 0.0;
            if (ok) {   // the certificate once more, on the objective of the ROUNDED point in the model's own costs — not on the value the sweeps reported
                for (int j = 0; j < n; j++) z_pt += mdl_in.obj[j] * std::floor(pa.x[(size_t)j] + 0.5);
                const double zs = cost_scale > 0.0 ? z_pt / cost_scale : z_pt;
                if (!(pa.bound * (1.0 + 1e-9) + 1e-12 <= zs + rel_gap * std::fabs(zs))) ok = false;
            }
            if (ok) {
                Result res;
                res.x = std::move(pa.x);
                for (double &v : res.x) v = std::floor(v + 0.5);   // (non-negative integers up to rounding noise: checked above)
                res.feasible = true; res.optimal = true; res.canonical = false;   // certified within rel_gap; which of the tied optima it is stays the sweeps' choice
                res.n_components = 1; res.nodes = pa.sweeps; res.price_sweeps = (int)pa.sweeps; res.price_rounds = (int)pa.rounds; res.price_total_us = (wall() - tf0) * 1e6;
                res.objective = z_pt;
                tmark("fast path certified");
                res_out = std::move(res);
                return true;
            }
            if (tracing_solve) fprintf(stderr, "[milp] fast path: the certified point fails a row of the model (or its own objective misses the certificate): classic path\n");
        }
    }
    return false;
}
}  // namespace

static Result solve_classic(const Model &mdl_in, double time_limit_s, bool canonical, double rel_gap, hqprice::Sweeper *sweeper, const double ts0);

Result solve(const Model &mdl_arg, double time_limit_s, bool canonical, double rel_gap, hqprice::Sweeper *sweeper) {
    const double ts_entry = wall();
    { Result r; if (solve_fast(mdl_arg, time_limit_s, rel_gap, sweeper, ts_entry, r)) return r; }
    if (mdl_arg.has_lists()) {   // every other consumer works on the plain form
        Model plain = mdl_arg;
        plain.expand_lists();
        return solve_classic(plain, time_limit_s, canonical, rel_gap, sweeper, ts_entry);
    }
    return solve_classic(mdl_arg, time_limit_s, canonical, rel_gap, sweeper, ts_entry);
}

static Result solve_classic(const Model &mdl_in, double time_limit_s, bool canonical, double rel_gap, hqprice::Sweeper *sweeper, const double ts0) {
    static const bool tracing_solve = getenv("HQMILP_TRACE") != nullptr;
    auto tmark = [&](const char *what) { if (tracing_solve && mdl_in.ncols() > 1000) fprintf(stderr, "[milp] solve(): %s at %.3f ms\n", what, (wall() - ts0) * 1e3); };
    Model snapped;  // a copy of the model only when a coefficient really has to be snapped (a block of a worker class has no BOOL column at all)
    bool need_snap = false;
    for (size_t k = 0; k < mdl_in.rcoef.size() && !need_snap; k++) {
        if (mdl_in.kind[mdl_in.rcol[k]] != COL_BOOL) continue;
        const double c = mdl_in.rcoef[k], g = std::round(c * 10000.0) / 10000.0;
        if (g != c && std::fabs(g - c) <= ROW_TOL) need_snap = true;
    }
    if (need_snap) {
        snapped = mdl_in;
        for (size_t k = 0; k < snapped.rcoef.size(); k++) {
            if (snapped.kind[snapped.rcol[k]] != COL_BOOL) continue;
            const double c = snapped.rcoef[k], g = std::round(c * 10000.0) / 10000.0;
            if (g != c && std::fabs(g - c) <= ROW_TOL) snapped.rcoef[k] = g;
        }
    }
    // ---- capacities on the integer hull of their row: a `<=` row whose coefficients are positive multiples of the ResourceAmount grid (1/10000) can only be
    // filled to a sum of its amounts.  5 cpus with requests of 2 and 4 cpus hold 4, never 5 — the LP thinks 5, and on a cluster of such workers its bound sits
    // 13 % above the integer optimum (20 workers, 19 tasks: 19 M branch-and-bound nodes, 19 s, against milliseconds with the row cut down to 4; HiGHS gets there
    // with its own coefficient tightening).  The new right-hand side is the largest reachable sum not above the old one: gcd first, then — while the row is
    // short in units — an unbounded-knapsack reachability pass.  No integer point is lost, so optimum and canonical optimum are unchanged.
    {
        const Model &src = need_snap ? snapped : mdl_in;
        const double GRID = 10000.0;
        std::vector<long long> ci;
        std::vector<uint8_t> reach;
        // the rows of identical workers are identical — one pass for all of them — but they alternate (a worker's cpu row, its gpu row, its memory row, the next worker's
        // cpu row ...): the last few distinct rows are kept, not only the last one
        struct Seen { std::vector<long long> ci; double rhs, new_rhs; };
        std::vector<Seen> seen_rows; size_t seen_next = 0; const size_t SEEN_CAP = 8;
        struct Cut { int row; long long d, rhs; };
        std::vector<Cut> cuts; std::vector<long long> cut_d;
        for (int i = 0; i < src.nrows(); i++) {
            if (src.rtype[i] != ROW_MAX) continue;
            const int a = src.roff[i], b = src.roff[i + 1];
            if (b - a < 1 || !(src.rhs[i] > 0.0) || src.rhs[i] > 1e11) continue;
            ci.clear();
            bool ok = true;
            for (int k = a; k < b && ok; k++) {
                const double c = src.rcoef[k] * GRID, r = std::round(c);
                if (!(src.rcoef[k] > 0.0) || r < 1.0 || std::fabs(c - r) > 1e-6 * std::max(1.0, r) || r > 9e15) ok = false; else ci.push_back((long long)r);
            }
            if (!ok) continue;
            double new_rhs;
            const Seen *hit = nullptr;
            for (const Seen &sr : seen_rows) if (sr.rhs == src.rhs[i] && sr.ci == ci) { hit = &sr; break; }
            if (hit) new_rhs = hit->new_rhs;
            else {
                long long g = 0;
                for (long long c : ci) { long long x = c, y = g; while (y) { const long long t = x % y; x = y; y = t; } g = x; }
                const long long cap = (long long)std::floor(src.rhs[i] * GRID + 1e-6);
                long long units = cap / g;  // the capacity in units of the gcd
                long long cmin = ci[0]; for (long long c : ci) cmin = std::min(cmin, c);
                if (cmin != g && units <= 65536) {  // (a request of one gcd unit reaches every multiple: nothing to search)
                    reach.assign((size_t)units + 1, 0); reach[0] = 1;
                    for (long long sidx = 1; sidx <= units; sidx++) for (long long c : ci) { const long long st = c / g; if (st <= sidx && reach[(size_t)(sidx - st)]) { reach[(size_t)sidx] = 1; break; } }
                    while (units > 0 && !reach[(size_t)units]) units--;
                }
                new_rhs = (double)(units * g) / GRID;
                if (seen_rows.size() < SEEN_CAP) seen_rows.push_back({ci, src.rhs[i], new_rhs});
                else { seen_rows[seen_next % SEEN_CAP] = {ci, src.rhs[i], new_rhs}; seen_next++; }
            }
            if (new_rhs < src.rhs[i] - 1e-9) {
                if (!need_snap) { snapped = mdl_in; need_snap = true; }
                snapped.rhs[i] = new_rhs;
            }
            // Rounding cuts of the row (Chvatal-Gomory, one divisor each): for any d > 0,  sum_j floor(a_j / d) x_j <= floor(cap / d)  holds for every integer
            // point of the row.  With d = one of the row's own amounts this is the statement "at most k requests of that size or larger fit" — e.g. one task per
            // worker where any two exceed it — which the LP of a bin-packing-like tick (a handful of ready tasks, idle workers a little larger than one request)
            // cannot see: its bound sat 4-13 % above the optimum there and the tree took 15-30 s where HiGHS needs 0.01-0.5 s.  Only for rows that hold few
            // requests (at most CUT_ITEMS of the smallest one): a 128-core worker with 1-core requests gains nothing and would carry thousands of idle rows.  The
            // cuts are ordinary rows: the LP activates them when its point violates them.
            {
                const long long cap_new = (long long)std::llround(new_rhs * GRID);
                long long cmin = ci[0]; for (long long c : ci) cmin = std::min(cmin, c);
                const int CUT_ITEMS = 32;
                if (cap_new / cmin <= CUT_ITEMS && b - a >= 2 && src.ncols() <= 512) {
                    cut_d.assign(ci.begin(), ci.end());
                    std::sort(cut_d.begin(), cut_d.end()); cut_d.erase(std::unique(cut_d.begin(), cut_d.end()), cut_d.end());
                    for (long long d : cut_d) {
                        if (d == cmin && cmin * (cap_new / cmin) == cap_new) { bool all_mult = true; for (long long c : ci) if (c % d) all_mult = false; if (all_mult) continue; }  // the row itself
                        const long long rhs_c = cap_new / d;
                        long long lhs_max = 0; int nz = 0;
                        for (long long c : ci) { if (c / d) nz++; lhs_max += (c / d) * (cap_new / c); }  // (every column at most floor(cap / a_j) by this row alone)
                        if (nz < 2 || lhs_max <= rhs_c) continue;  // never binding
                        cuts.push_back({i, d, rhs_c});
                    }
                }
            }
        }
        if (!cuts.empty()) {
            if (!need_snap) { snapped = mdl_in; need_snap = true; }
            snapped.row_implied.resize((size_t)snapped.nrows(), 0);
            for (const auto &cu : cuts) {
                const int a = mdl_in.roff[cu.row], b = mdl_in.roff[cu.row + 1];
                snapped.row_implied.push_back(1);  // holds at every integer point of the row it was rounded from
                snapped.begin_row(ROW_MAX, (double)cu.rhs);
                for (int k = a; k < b; k++) { const long long c = (long long)std::llround(mdl_in.rcoef[k] * GRID) / cu.d; if (c) snapped.term(mdl_in.rcol[k], (double)c); }
                snapped.end_row();
            }
        }
    }
    tmark("presolve done");
    const Model &mdl = need_snap ? snapped : mdl_in;
    Result res;
    int n = mdl.ncols(), m = mdl.nrows();
    res.x.assign(n, 0.0);
    res.feasible = true; res.optimal = true; res.canonical = canonical;
    if (n == 0) return res;
    double deadline = wall() + (time_limit_s > 0 ? time_limit_s : 1e18);

    // ---- column upper bounds implied by <=/== rows with non-negative coefficients (all columns have lb 0) ----
    std::vector<double> ub(n, INF);
    for (int j = 0; j < n; j++) if (mdl.kind[j] == COL_BOOL) ub[j] = 1.0;
    bool chained = false;  // some bound was derived from other columns' bounds: those may have tightened later in the pass, so a second pass follows (else it would repeat the first)
    for (int pass = 0; pass < 2 && (pass == 0 || chained); pass++) {
        for (int i = 0; i < m; i++) {
            if (mdl.rtype[i] == ROW_MIN) continue;
            int a = mdl.roff[i], b = mdl.roff[i + 1];
            bool nonneg = true; int neg = -1; double negc = 0;
            for (int k = a; k < b; k++) if (mdl.rcoef[k] < 0) { if (neg >= 0) nonneg = false; neg = mdl.rcol[k]; negc = mdl.rcoef[k]; }
            if (neg < 0) {
                if (mdl.rhs[i] < -1e-9) { res.feasible = false; res.optimal = false; return res; }
                for (int k = a; k < b; k++) if (mdl.rcoef[k] > 0) ub[mdl.rcol[k]] = std::min(ub[mdl.rcol[k]], std::floor(mdl.rhs[i] / mdl.rcoef[k] + 1e-9));
            } else if (nonneg && mdl.rtype[i] == ROW_EQ && mdl.rhs[i] == 0.0) {
                // sum(pos) == |negc| * y  (MN group rows, solver.rs:211-218): y <= sum(ub pos)/|negc|
                double s = 0; bool fin = true;
                for (int k = a; k < b; k++) if (mdl.rcoef[k] > 0) { if (ub[mdl.rcol[k]] >= INF) fin = false; else s += mdl.rcoef[k] * ub[mdl.rcol[k]]; }
                chained = true;
                if (fin) ub[neg] = std::min(ub[neg], std::floor(s / -negc + 1e-9));
            }
        }
    }
    std::vector<char> capped(n, 0);
    for (int j = 0; j < n; j++) if (ub[j] >= INF) { ub[j] = UB_CAP; capped[j] = 1; }

    // ---- a feasible point for the whole model, before any LP: seeds every component's search and is the answer for components the
    // dense method cannot take ----
    tmark("bounds done");
    // Computed at its first use: a component that goes to the price sweeps (csrc/price.h) does not need it — the sweeps build their own incumbent, and its only
    // other role there, naming the flag configuration to start from, "every flag on" plays as well (c3p: 10 sweeps instead of 11) — so such a component asks for
    // it only when the sweeps leave it open (CompSolver::lazy_incumbent); at 8 k columns the pass is a tenth of the coupled tick's host time.
    // (Tried on a second thread instead — std::async: it finished LATER than this thread would have, here and on the MI355X host.)
    std::vector<double> hx;
    bool have_hx = false, hx_joined = false;
    auto join_hx = [&]() {
        if (hx_joined) return;
        hx_joined = true;
        have_hx = sparse_greedy(mdl, ub, hx);
        tmark("sparse greedy done");
        if ((int)mdl.start.size() == n) {  // caller's starting point: taken if feasible and better
            bool ok = true; double zs = 0.0, zh = 0.0;
            for (int j = 0; j < n && ok; j++) { double v = mdl.start[j]; if (v < -1e-9 || v > ub[j] + 1e-9 || std::fabs(v - std::round(v)) > 1e-9) ok = false; zs += mdl.obj[j] * v; }
            for (int i = 0; i < m && ok; i++) {
                double a = 0.0, sc = 0.0;
                for (int k = mdl.roff[i]; k < mdl.roff[i + 1]; k++) { a += mdl.rcoef[k] * mdl.start[mdl.rcol[k]]; sc = std::max(sc, std::fabs(mdl.rcoef[k])); }
                const double tol = 1e-9 * std::max(1.0, sc);
                if (mdl.rtype[i] != ROW_MIN && a > mdl.rhs[i] + tol) ok = false;
                if (mdl.rtype[i] != ROW_MAX && a < mdl.rhs[i] - tol) ok = false;
            }
            if (ok) {
                if (have_hx) for (int j = 0; j < n; j++) zh += mdl.obj[j] * hx[j];
                if (!have_hx || zs > zh) { hx = mdl.start; have_hx = true; }
            }
        }
    };
    static const bool lazy_greedy = !(getenv("HQMILP_LAZY_GREEDY") && atoi(getenv("HQMILP_LAZY_GREEDY")) == 0);  // (A/B switch)
    const int LAZY_GREEDY_COLS = 2048;

    // ---- connected components ----
    DSU dsu(n);
    for (int i = 0; i < m; i++) for (int k = mdl.roff[i] + 1; k < mdl.roff[i + 1]; k++) dsu.unite(mdl.rcol[mdl.roff[i]], mdl.rcol[k]);
    std::vector<int> comp_of(n, -1); std::vector<std::vector<int>> ccols, crows;
    for (int j = 0; j < n; j++) {
        int r = dsu.find(j);
        if (comp_of[r] < 0) { comp_of[r] = (int)ccols.size(); ccols.emplace_back(); crows.emplace_back(); }
        comp_of[j] = comp_of[r]; ccols[comp_of[j]].push_back(j);
    }
    for (int i = 0; i < m; i++) {
        if (mdl.roff[i] == mdl.roff[i + 1]) {  // empty row: 0 {>=,<=,==} rhs
            double b = mdl.rhs[i];
            bool ok = mdl.rtype[i] == ROW_MAX ? b >= -1e-9 : (mdl.rtype[i] == ROW_MIN ? b <= 1e-9 : std::fabs(b) <= 1e-9);
            if (!ok) { res.feasible = false; res.optimal = false; return res; }
            continue;
        }
        crows[comp_of[mdl.rcol[mdl.roff[i]]]].push_back(i);
    }
    res.n_components = (int)ccols.size();
    tmark("components done");

    std::vector<int> local(n, -1);
    std::unordered_map<std::string, std::pair<int, std::vector<double>>> memo;
    for (size_t ci = 0; ci < ccols.size(); ci++) {
        auto &cols = ccols[ci]; auto &rows = crows[ci];
        CompSolver cs; cs.n = (int)cols.size(); cs.deadline = deadline; cs.time_limit_s = time_limit_s;
        cs.rel_gap = rel_gap;
        const int cm = (int)rows.size();
        for (int k = 0; k < cs.n; k++) local[cols[k]] = k;
        cs.c.resize(cs.n); cs.lb.assign(cs.n, 0.0); cs.ub.resize(cs.n);
        double cmax = 0.0;
        for (int k = 0; k < cs.n; k++) cmax = std::max(cmax, std::fabs(mdl.obj[cols[k]]));
        if (cmax == 0.0) cmax = 1.0;
        for (int k = 0; k < cs.n; k++) { cs.c[k] = mdl.obj[cols[k]] / cmax; cs.ub[k] = ub[cols[k]]; }  // costs O(1): scale-free pivoting
        if (cm == 0) {  // free columns: at their upper bound if it pays (or for the lexicographic rule)
            for (int k = 0; k < cs.n; k++) {
                if (capped[cols[k]]) { res.feasible = false; res.optimal = false; return res; }
                res.x[cols[k]] = cs.ub[k];
            }
            continue;
        }
        cs.R.n = cs.n;
        {   // one allocation per array instead of a doubling series
            size_t nnz = 0; for (int r = 0; r < cm; r++) nnz += (size_t)(mdl.roff[rows[r] + 1] - mdl.roff[rows[r]]);
            cs.R.col.reserve(nnz); cs.R.coef.reserve(nnz); cs.R.off.reserve((size_t)cm + 2); cs.R.lo.reserve((size_t)cm + 1); cs.R.hi.reserve((size_t)cm + 1);  // (+1: the tie-break's objective row)
        }
        std::vector<std::pair<int, double>> terms;
        std::vector<int> term_row((size_t)cs.n, -1), term_pos((size_t)cs.n, 0);
        for (int r = 0; r < cm; r++) {
            int i = rows[r]; double sc = 0.0;
            terms.clear();
            for (int k = mdl.roff[i]; k < mdl.roff[i + 1]; k++) {  // duplicate columns of one row are summed (stamp per column: a wide row has a thousand terms)
                const int lc = local[mdl.rcol[k]];
                if (term_row[(size_t)lc] == r) terms[(size_t)term_pos[(size_t)lc]].second += mdl.rcoef[k];
                else { term_row[(size_t)lc] = r; term_pos[(size_t)lc] = (int)terms.size(); terms.push_back({lc, mdl.rcoef[k]}); }
            }
            for (auto &t : terms) sc = std::max(sc, std::fabs(t.second));
            if (sc == 0.0) sc = 1.0;
            for (auto &t : terms) t.second /= sc;
            double b = mdl.rhs[i] / sc;
            cs.R.add(terms, mdl.rtype[i] == ROW_MAX ? -INF : b, mdl.rtype[i] == ROW_MIN ? INF : b);
            cs.row_scale.push_back(sc);
            if (sweeper) cs.row_implied.push_back((size_t)i < mdl.row_implied.size() ? mdl.row_implied[(size_t)i] : 0);
        }
        if ((int)mdl.col_group.size() == n) { cs.col_group.resize(cs.n); for (int k = 0; k < cs.n; k++) cs.col_group[k] = mdl.col_group[cols[k]]; }   // (the block-hull cuts of the root use it too)
        if (sweeper && (int)mdl.col_group.size() == n) cs.sweeper = sweeper;
        // identical components (same rows, bounds and — up to 2^-40 relative — the same normalised costs) share one solve:
        // workers with equal free/total vectors produce them by the hundred (solver.rs:95-192 builds one block per worker)
        std::string sig;
        bool memo_ok = ccols.size() > 1 && cs.R.col.size() + (size_t)cs.n + (size_t)cm <= 8192;  // (a single component has nobody to share its solve with)
        if (memo_ok) {
            auto put = [&](const void *p, size_t nb) { sig.append(reinterpret_cast<const char *>(p), nb); };
            int dims[2] = {cs.n, cm}; put(dims, sizeof dims);
            put(cs.ub.data(), cs.ub.size() * 8); put(cs.R.off.data(), cs.R.off.size() * 4); put(cs.R.col.data(), cs.R.col.size() * 4);
            put(cs.R.coef.data(), cs.R.coef.size() * 8); put(cs.R.lo.data(), cs.R.lo.size() * 8); put(cs.R.hi.data(), cs.R.hi.size() * 8);
            for (int k = 0; k < cs.n; k++) { long long qv = (long long)std::llround(cs.c[k] * 1099511627776.0); put(&qv, 8); }
            auto hit = memo.find(sig);
            if (hit != memo.end()) {
                const std::vector<double> &xo = hit->second.second;
                if (hit->second.first == 0) { res.feasible = false; res.optimal = false; return res; }
                for (int k = 0; k < cs.n; k++) {
                    if (capped[cols[k]] && xo[k] >= UB_CAP - 0.5) { res.feasible = false; res.optimal = false; return res; }
                    res.x[cols[k]] = xo[k];
                }
                continue;
            }
        }
        auto seed = [&]() {  // incumbent from the sparse heuristic (restricted to this component it is feasible for the component)
            join_hx();
            if (!have_hx) return;
            std::vector<double> bx((size_t)cs.n); double z = 0.0;
            for (int k = 0; k < cs.n; k++) { bx[(size_t)k] = hx[cols[k]]; z += cs.c[k] * bx[(size_t)k]; }
            if (!cs.have || z > cs.best) { cs.bx = std::move(bx); cs.have = true; cs.best = z; }
        };
        // (CompSolver::run's own condition, and only LARGE components: below ~2 k columns the pass costs tens of microseconds and the flag configuration it names
        // is sometimes the better start — GPU seed 2011 of tests/test_gpu_price.py is certified from it and not from "every flag on"; a caller's starting point is looked at now)
        const bool to_sweeps = lazy_greedy && cs.sweeper && rel_gap > 0.0 && cs.n >= (int)sweeper->min_cols && cs.n >= LAZY_GREEDY_COLS && (int)mdl.start.size() != n;
        if (to_sweeps) cs.lazy_incumbent = seed; else seed();
        tmark("component rows built");
        std::vector<double> xo;
        int st = cs.run(canonical, xo);
        tmark("component solved");
        if (memo_ok && !cs.timed_out && cs.canonical_done && (st == 0 || st == 1)) memo.emplace(std::move(sig), std::make_pair(st, xo));
        if (!cs.canonical_done || st == 2) res.canonical = false;
        res.nodes += cs.nodes; res.lp_iters += cs.lp_iters;
        res.price_sweeps += cs.price_sweeps; res.price_rounds += cs.price_rounds; res.price_total_us += cs.price_us;
        if (st == 0) {
            if (cs.timed_out) {  // nothing found in time: all-zero placement with every blocker flag on (feasible for the tick's models)
                res.optimal = false;
                for (int i : rows) if (mdl.rtype[i] == ROW_MIN && mdl.rhs[i] > 0) { int last = mdl.rcol[mdl.roff[i + 1] - 1]; if (mdl.kind[last] == COL_BOOL) res.x[last] = 1.0; }
                continue;
            }
            res.feasible = false; res.optimal = false; return res;
        }
        if (st == 2) res.optimal = false;
        for (int k = 0; k < cs.n; k++) {
            if (capped[cols[k]] && xo[k] >= UB_CAP - 0.5) { res.feasible = false; res.optimal = false; return res; }
            res.x[cols[k]] = xo[k];
        }
    }
    double z = 0.0; for (int j = 0; j < n; j++) z += mdl.obj[j] * res.x[j];
    res.objective = z;
    return res;
}

}  // namespace hqmilp
