// Exact small-MILP solver (see milp.h).  Branch-and-bound over a dense bounded-variable DUAL simplex that is
// warm-started from the parent node, followed by a lexicographic canonicalisation pass.
//
// Why dual simplex: every objective coefficient of the tick's model is >= 0 (scheduler/solver.rs:542-597) and every
// placement column has a finite upper bound implied by its worker's resource rows, so "all logicals basic, costly
// columns at their upper bound" is dual feasible from the start, and a B&B child differs from its parent by one
// bound — exactly the case the dual method re-optimises in a handful of pivots.
#include "milp.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <numeric>
#include <string>
#include <unordered_map>

namespace hqmilp {
namespace {

const double INF = 1e300;
const double FEAS_TOL = 1e-9;   // primal bound violation (rows are scaled to max |coef| = 1)
const double PIV_TOL = 1e-9;
const double INT_TOL = 1e-7;
const double UB_CAP = 1048576.0;  // columns with no derivable bound (unbounded models => `None`, highs.rs:82)

double wall() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

enum { BASIC = 0, AT_LO = 1, AT_UP = 2 };
enum { LP_OPT = 0, LP_INFEAS = 1, LP_LIMIT = 2 };

// A component's LP in "rows are logical variables" form:  A x - s = 0,  lo <= s <= hi,  lb <= x <= ub.
struct Tab {
    int n = 0, m = 0, N = 0;
    std::vector<double> T, d, x, lb, ub, cost;
    std::vector<int> B;
    std::vector<uint8_t> st;
    long iters = 0;

    void init(int n_, int m_, const std::vector<double> &A, const std::vector<double> &c, const std::vector<double> &clb,
              const std::vector<double> &cub, const std::vector<double> &rlo, const std::vector<double> &rhi) {
        n = n_; m = m_; N = n + m;
        T.assign((size_t)m * N, 0.0);
        for (int i = 0; i < m; i++) {
            for (int j = 0; j < n; j++) T[(size_t)i * N + j] = -A[(size_t)i * n + j];
            T[(size_t)i * N + n + i] = 1.0;
        }
        cost.assign(N, 0.0);
        for (int j = 0; j < n; j++) cost[j] = c[j];
        d = cost;
        lb.assign(N, 0.0); ub.assign(N, 0.0);
        for (int j = 0; j < n; j++) { lb[j] = clb[j]; ub[j] = cub[j]; }
        for (int i = 0; i < m; i++) { lb[n + i] = rlo[i]; ub[n + i] = rhi[i]; }
        st.assign(N, AT_LO); x.assign(N, 0.0); B.resize(m);
        for (int j = 0; j < n; j++) {
            if (cost[j] > 0.0) { st[j] = AT_UP; x[j] = ub[j]; } else { st[j] = AT_LO; x[j] = lb[j]; }
        }
        for (int i = 0; i < m; i++) {
            B[i] = n + i; st[n + i] = BASIC;
            double s = 0.0;
            for (int j = 0; j < n; j++) s += A[(size_t)i * n + j] * x[j];
            x[n + i] = s;
        }
    }
    double objective() const { double z = 0.0; for (int j = 0; j < n; j++) z += cost[j] * x[j]; return z; }

    // move a nonbasic variable to a new value, updating the basic ones
    void shift_nonbasic(int j, double nv) {
        double dl = nv - x[j];
        if (dl == 0.0) return;
        for (int i = 0; i < m; i++) { double t = T[(size_t)i * N + j]; if (t != 0.0) x[B[i]] -= t * dl; }
        x[j] = nv;
    }
    void set_lb(int j, double v) { lb[j] = v; if (st[j] == AT_LO) shift_nonbasic(j, v); else if (st[j] == AT_UP && ub[j] < v) shift_nonbasic(j, v); }
    void set_ub(int j, double v) { ub[j] = v; if (st[j] == AT_UP) shift_nonbasic(j, v); else if (st[j] == AT_LO && lb[j] > v) shift_nonbasic(j, v); }

    int solve(long max_iters) {
        for (long it = 0; it < max_iters; it++) {
            int r = -1; double best = FEAS_TOL; bool below = false;
            for (int i = 0; i < m; i++) {
                int k = B[i]; double v = x[k];
                if (v < lb[k] - FEAS_TOL) { double inf = lb[k] - v; if (inf > best) { best = inf; r = i; below = true; } }
                else if (v > ub[k] + FEAS_TOL) { double inf = v - ub[k]; if (inf > best) { best = inf; r = i; below = false; } }
            }
            if (r < 0) return LP_OPT;
            int k = B[r];
            if (lb[k] > ub[k] + FEAS_TOL) return LP_INFEAS;
            const double *row = &T[(size_t)r * N];
            int q = -1; double bratio = INF, babs = 0.0;
            for (int j = 0; j < N; j++) {
                if (st[j] == BASIC) continue;
                if (lb[j] == ub[j]) continue;  // fixed: cannot move
                double a = row[j];
                bool elig;
                if (below) elig = (st[j] == AT_LO && a < -PIV_TOL) || (st[j] == AT_UP && a > PIV_TOL);
                else elig = (st[j] == AT_LO && a > PIV_TOL) || (st[j] == AT_UP && a < -PIV_TOL);
                if (!elig) continue;
                double ratio = std::fabs(d[j]) / std::fabs(a);
                if (ratio < bratio - 1e-13 || (ratio <= bratio + 1e-13 && std::fabs(a) > babs)) { bratio = ratio; babs = std::fabs(a); q = j; }
            }
            if (q < 0) return LP_INFEAS;
            iters++;
            double target = below ? lb[k] : ub[k];
            double piv = row[q];
            double dq = (x[k] - target) / piv;
            for (int i = 0; i < m; i++) { double t = T[(size_t)i * N + q]; if (t != 0.0) x[B[i]] -= t * dq; }
            x[q] += dq;
            x[k] = target;
            // pivot
            double *prow = &T[(size_t)r * N];
            double inv = 1.0 / piv;
            for (int j = 0; j < N; j++) prow[j] *= inv;
            prow[q] = 1.0;
            for (int i = 0; i < m; i++) {
                if (i == r) continue;
                double f = T[(size_t)i * N + q];
                if (f == 0.0) continue;
                double *ri = &T[(size_t)i * N];
                for (int j = 0; j < N; j++) ri[j] -= f * prow[j];
                ri[q] = 0.0;
            }
            double f = d[q];
            if (f != 0.0) { for (int j = 0; j < N; j++) d[j] -= f * prow[j]; d[q] = 0.0; }
            st[k] = below ? AT_LO : AT_UP;
            st[q] = BASIC; B[r] = q;
        }
        return LP_LIMIT;
    }
};

struct CompSolver {
    int n = 0, m = 0;
    std::vector<double> A, c, lb, ub, rlo, rhi;
    double deadline = 0; bool timed_out = false;
    long nodes = 0, lp_iters = 0;
    // incumbent
    bool have = false; double best = -INF; std::vector<double> bx;
    bool canonical_done = true;

    long pivots = 0, pivot_limit = -1;  // deterministic work budget of the tie-break phase (wall clock stays the backstop)
    bool time_up() { if (timed_out) return true; if ((pivot_limit >= 0 && pivots > pivot_limit) || ((nodes & 31) == 0 && wall() > deadline)) timed_out = true; return timed_out; }
    int solve_counted(Tab &t) { long before = t.iters; int r = t.solve(200000); pivots += t.iters - before; return r; }

    // Primal heuristic: from an integer point inside the bounds (e.g. the floor of an LP solution), keep it only if every row holds,
    // then raise columns greedily — most valuable first — as far as the rows allow.  The placement models are packing
    // problems whose LP bound is usually attained, so a maximal point found here often closes the search at the root.
    std::vector<int> by_cost;
    void greedy_from(std::vector<double> x) {
        std::vector<double> act(m, 0.0);
        for (int i = 0; i < m; i++) { double a = 0.0; const double *row = &A[(size_t)i * n]; for (int j = 0; j < n; j++) a += row[j] * x[j]; act[i] = a; }
        for (int i = 0; i < m; i++) if (act[i] < rlo[i] - FEAS_TOL || act[i] > rhi[i] + FEAS_TOL) return;
        if (by_cost.empty()) {
            by_cost.resize(n); std::iota(by_cost.begin(), by_cost.end(), 0);
            std::stable_sort(by_cost.begin(), by_cost.end(), [&](int a, int b) { return c[a] > c[b]; });
        }
        for (int j : by_cost) {
            if (c[j] <= 0.0) break;
            double step = ub[j] - x[j];
            for (int i = 0; i < m && step >= 1.0; i++) {
                double a = A[(size_t)i * n + j];
                if (a > 0.0 && rhi[i] < INF) step = std::min(step, std::floor((rhi[i] - act[i]) / a + 1e-9));
                else if (a < 0.0 && rlo[i] > -INF) step = std::min(step, std::floor((act[i] - rlo[i]) / -a + 1e-9));
            }
            if (step < 1.0) continue;
            x[j] += step;
            for (int i = 0; i < m; i++) { double a = A[(size_t)i * n + j]; if (a != 0.0) act[i] += a * step; }
        }
        double z = 0.0; for (int j = 0; j < n; j++) z += c[j] * x[j];
        if (!have || z > best + 1e-12 * std::fabs(best)) { have = true; best = z; bx = x; }
    }
    void round_and_repair(const Tab &t) {
        std::vector<double> x(n);
        for (int j = 0; j < n; j++) x[j] = std::min(ub[j], std::max(lb[j], std::floor(t.x[j] + INT_TOL)));
        greedy_from(std::move(x));
    }

    static int pick_fractional(const Tab &t) {
        int j = -1; double bd = INT_TOL;
        for (int k = 0; k < t.n; k++) {
            double v = t.x[k], fr = std::fabs(v - std::round(v));
            if (fr > bd) { bd = fr; j = k; }
        }
        return j;
    }

    // phase 1: maximise
    void dfs_opt(Tab &t) {
        nodes++;
        if (time_up()) return;
        int s = solve_counted(t);
        if (s != LP_OPT) { if (s == LP_LIMIT) timed_out = true; return; }
        double z = t.objective();
        if (have && z <= best + 1e-12 * std::fabs(best)) return;
        int j = pick_fractional(t);
        if (j >= 0 && (nodes == 1 || (nodes & 63) == 0)) {  // root and every 64th node: try to close the gap from this LP point
            round_and_repair(t);
            if (have && z <= best + 1e-12 * std::fabs(best)) return;
        }
        if (j < 0) {
            have = true; best = z; bx.assign(t.x.begin(), t.x.begin() + n);
            for (auto &v : bx) v = std::round(v);
            double zz = 0.0; for (int k = 0; k < n; k++) zz += c[k] * bx[k];
            best = zz;
            return;
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
        double v = t.x[j];
        {
            Tab up = t;
            up.set_lb(j, std::ceil(v - INT_TOL));
            if (dfs_feas(up, out)) return true;
        }
        t.set_ub(j, std::floor(v + INT_TOL));
        return dfs_feas(t, out);
    }

    // returns: 0 infeasible, 1 optimal, 2 incumbent only (time limit)
    int run(bool canonical, std::vector<double> &xout) {
        Tab root; root.init(n, m, A, c, lb, ub, rlo, rhi);
        greedy_from(lb);
        dfs_opt(root);
        lp_iters += root.iters;
        if (!have) return 0;
        xout = bx;
        if (timed_out) return 2;
        if (!canonical || n == 0) return 1;
        // phase 2: among vectors with c.x >= best - tol, minimise the LAST column, then the one before it, ... (bound probing,
        // one feasibility B&B per probe)
        double tol = 1e-9 * std::fabs(best);
        std::vector<double> A2(A), rlo2(rlo), rhi2(rhi);
        double cs = 0.0; for (int k = 0; k < n; k++) cs = std::max(cs, std::fabs(c[k]));
        if (cs > 0.0) {
            for (int k = 0; k < n; k++) A2.push_back(c[k] / cs);
            rlo2.push_back((best - tol) / cs); rhi2.push_back(INF);
        }
        int m2 = (int)rlo2.size();
        std::vector<double> cur(bx);
        // One tableau carries the columns fixed so far; every probe is a copy of it with one tightened bound, re-optimised by
        // the dual simplex from the parent basis (a handful of pivots) instead of a cold start.
        // Budget in simplex pivots, not seconds: every replica of a sharded scheduler must take the same decision here.
        pivot_limit = pivots + std::max<long>(3000, 4 * pivots);
        Tab warm; warm.init(n, m2, A2, c, lb, ub, rlo2, rhi2);
        if (warm.solve(200000) != LP_OPT) { xout = cur; return 1; }  // cannot happen: `cur` is feasible for it
        for (int j = n - 1; j >= 0; j--) {  // last column first
            double lo = lb[j], hi = cur[j];
            bool first = true;
            while (lo < hi) {
                // first probe just below the current value: most columns fail it immediately
                double mid = first ? hi - 1 : std::floor((lo + hi) / 2);
                first = false;
                Tab t = warm;
                t.set_ub(j, mid);
                std::vector<double> sol;
                bool ok = dfs_feas(t, sol);
                lp_iters += t.iters - warm.iters;
                if (timed_out) { xout = cur; canonical_done = false; return 1; }  // optimal (phase 1 proved it) but the tie-break ran out of time
                if (ok) { cur = sol; hi = sol[j]; } else lo = mid + 1;
            }
            warm.set_lb(j, hi); warm.set_ub(j, hi);
            if (warm.solve(200000) != LP_OPT) {  // numerically lost the basis: rebuild it with the bounds fixed so far
                std::vector<double> flb(lb), fub(ub);
                for (int k = n - 1; k >= j; k--) flb[k] = fub[k] = cur[k];
                warm = Tab(); warm.init(n, m2, A2, c, flb, fub, rlo2, rhi2);
                if (warm.solve(200000) != LP_OPT) { xout = cur; return 1; }
            }
        }
        xout = cur;
        return 1;
    }
};

struct DSU {
    std::vector<int> p;
    explicit DSU(int n) : p(n) { std::iota(p.begin(), p.end(), 0); }
    int find(int a) { while (p[a] != a) { p[a] = p[p[a]]; a = p[a]; } return a; }
    void unite(int a, int b) { a = find(a); b = find(b); if (a != b) p[std::max(a, b)] = std::min(a, b); }
};

}  // namespace

Result solve(const Model &mdl, double time_limit_s, bool canonical) {
    Result res;
    int n = mdl.ncols(), m = mdl.nrows();
    res.x.assign(n, 0.0);
    res.feasible = true; res.optimal = true;
    if (n == 0) return res;
    double deadline = wall() + (time_limit_s > 0 ? time_limit_s : 1e18);

    // ---- column upper bounds implied by <=/== rows with non-negative coefficients (all columns have lb 0) ----
    std::vector<double> ub(n, INF);
    for (int j = 0; j < n; j++) if (mdl.kind[j] == COL_BOOL) ub[j] = 1.0;
    for (int pass = 0; pass < 2; pass++) {
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
                if (fin) ub[neg] = std::min(ub[neg], std::floor(s / -negc + 1e-9));
            }
        }
    }
    std::vector<char> capped(n, 0);
    for (int j = 0; j < n; j++) if (ub[j] >= INF) { ub[j] = UB_CAP; capped[j] = 1; }

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

    std::vector<int> local(n, -1);
    std::unordered_map<std::string, std::pair<int, std::vector<double>>> memo;
    for (size_t ci = 0; ci < ccols.size(); ci++) {
        auto &cols = ccols[ci]; auto &rows = crows[ci];
        CompSolver cs; cs.n = (int)cols.size(); cs.m = (int)rows.size(); cs.deadline = deadline;
        for (int k = 0; k < cs.n; k++) local[cols[k]] = k;
        cs.c.resize(cs.n); cs.lb.assign(cs.n, 0.0); cs.ub.resize(cs.n);
        double cmax = 0.0;
        for (int k = 0; k < cs.n; k++) cmax = std::max(cmax, std::fabs(mdl.obj[cols[k]]));
        if (cmax == 0.0) cmax = 1.0;
        for (int k = 0; k < cs.n; k++) { cs.c[k] = mdl.obj[cols[k]] / cmax; cs.ub[k] = ub[cols[k]]; }  // costs O(1): scale-free pivoting
        if (cs.m == 0) {  // free columns: at their upper bound if it pays (or for the lexicographic rule)
            for (int k = 0; k < cs.n; k++) {
                if (capped[cols[k]]) { res.feasible = false; res.optimal = false; return res; }
                res.x[cols[k]] = cs.ub[k];
            }
            continue;
        }
        if ((double)cs.m * (double)(cs.n + cs.m) > 4.0e7) {  // too large for the dense exact method in this round
            res.optimal = false;  // all-zero placement (every blocker flag = 1) is feasible: keep x = 0, beta = 1
            for (int k = 0; k < cs.n; k++) res.x[cols[k]] = 0.0;
            // satisfy Min rows of the form sum + s*beta >= s
            for (int i : rows) if (mdl.rtype[i] == ROW_MIN && mdl.rhs[i] > 0) { int last = mdl.rcol[mdl.roff[i + 1] - 1]; res.x[last] = 1.0; }
            continue;
        }
        cs.A.assign((size_t)cs.m * cs.n, 0.0); cs.rlo.resize(cs.m); cs.rhi.resize(cs.m);
        for (int r = 0; r < cs.m; r++) {
            int i = rows[r]; double sc = 0.0;
            for (int k = mdl.roff[i]; k < mdl.roff[i + 1]; k++) sc = std::max(sc, std::fabs(mdl.rcoef[k]));
            if (sc == 0.0) sc = 1.0;
            for (int k = mdl.roff[i]; k < mdl.roff[i + 1]; k++) cs.A[(size_t)r * cs.n + local[mdl.rcol[k]]] += mdl.rcoef[k] / sc;
            double b = mdl.rhs[i] / sc;
            cs.rlo[r] = mdl.rtype[i] == ROW_MAX ? -INF : b;
            cs.rhi[r] = mdl.rtype[i] == ROW_MIN ? INF : b;
        }
        // identical components (same rows, bounds and — up to 2^-40 relative — the same normalised costs) share one solve:
        // workers with equal free/total vectors produce them by the hundred (solver.rs:95-192 builds one block per worker)
        std::string sig;
        bool memo_ok = (size_t)cs.m * cs.n <= 4096;
        if (memo_ok) {
            auto put = [&](const void *p, size_t nb) { sig.append(reinterpret_cast<const char *>(p), nb); };
            int dims[2] = {cs.n, cs.m}; put(dims, sizeof dims);
            put(cs.ub.data(), cs.ub.size() * 8); put(cs.A.data(), cs.A.size() * 8); put(cs.rlo.data(), cs.rlo.size() * 8); put(cs.rhi.data(), cs.rhi.size() * 8);
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
        std::vector<double> xo;
        int st = cs.run(canonical, xo);
        if (memo_ok && !cs.timed_out && (st == 0 || st == 1)) memo.emplace(std::move(sig), std::make_pair(st, xo));
        res.nodes += cs.nodes; res.lp_iters += cs.lp_iters;
        if (st == 0) {
            if (cs.timed_out) { res.optimal = false; continue; }  // nothing found in time: leave zeros
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
