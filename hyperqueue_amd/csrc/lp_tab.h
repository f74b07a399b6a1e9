// The LP engine of the exact solver (csrc/milp.cpp) and of the pricing master (csrc/price.cpp): a bounded-variable DUAL simplex on a dense tableau
// that holds only the ACTIVE rows — constraints enter when the current point violates them.  See milp.cpp for why the dual method fits the tick's models.
#pragma once
#include <algorithm>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace hqmilp {
namespace lp {

constexpr double INF = 1e300;
constexpr double FEAS_TOL = 1e-9;   // primal bound violation (rows are scaled to max |coef| = 1)
constexpr double PIV_TOL = 1e-7;   // (rows are scaled to max |coef| = 1: a tableau entry below this is rounding noise, not a coefficient.  1e-9 until the last session of round 6:
                                   //  pivots on such entries were what made cold solves of several hundred cut rows fail their own consistency test — tools/exp/README.md)
constexpr double INT_TOL = 1e-7;
constexpr int LNS_FIRST_COLS = 128;          // models from this size on improve the greedy incumbent by window search before any tree search
constexpr double EXACT_PASS_WORK = 5.0e7;    // floor of the exact pass's budget after a gap certificate, in tableau element updates (~50 ms)
constexpr double UB_CAP = 1048576.0;  // columns with no derivable bound (unbounded models => `None`, highs.rs:82)
constexpr double TAB_LIMIT = 6.0e7;   // doubles in one tableau (480 MB): beyond it the LP gives up (reported like a time limit)

// y[0..n) -= f * x[0..n): the row update of a pivot, where the solver spends its time.  Compiled for AVX2 on the host pass (every x86-64 server
// CPU of the last decade has it); no FMA contraction, so the arithmetic is the same mul + sub as the plain loop.
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__)
__attribute__((target("avx2")))
#endif
inline void axpy_neg(double *__restrict__ y, const double *__restrict__ x, double f, int n) {
    for (int j = 0; j < n; j++) y[j] -= f * x[j];
}

inline double wall() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

enum { BASIC = 0, AT_LO = 1, AT_UP = 2 };
enum { LP_OPT = 0, LP_INFEAS = 1, LP_LIMIT = 2 };

// Sparse rows of one component (local column ids), scaled to max |coef| = 1:  lo <= a.x <= hi
struct Rows {
    int n = 0, m = 0;
    std::vector<int> off{0}, col;
    std::vector<double> coef, lo, hi;
    void add(const std::vector<std::pair<int, double>> &terms, double lo_, double hi_) {
        for (auto &t : terms) { col.push_back(t.first); coef.push_back(t.second); }
        off.push_back((int)col.size()); lo.push_back(lo_); hi.push_back(hi_); m++;
    }
    double activity(int i, const double *x) const { double a = 0.0; for (int k = off[i]; k < off[i + 1]; k++) a += coef[k] * x[col[k]]; return a; }
};

// LP  max c.x,  lb <= x <= ub,  rows of R — as a tableau over the rows activated so far.  Columns: [0, n) structural,
// [n, n + ma) the slacks s_a = a_row.x of the active rows.  Every tableau row r reads  x_B[r] + sum_{j nonbasic} T[r][j] x_j = 0.
struct Tab {
    const Rows *R = nullptr;
    int n = 0, ma = 0, cap = 0, stride = 0;
    std::vector<double> T, d, x, lb, ub, cost;
    std::vector<int> B, arow, where;
    std::vector<uint8_t> st;
    long iters = 0;
    double ops = 0.0;         // tableau elements touched so far (pivots, row activations, separation scans): the deterministic work measure
    double deadline = 1e300;  // wall-clock backstop inside long re-optimisations

    // Tableau storage is recycled through a small per-thread pool: a B&B node copies its parent's tableau, and a fresh std::vector of a megabyte
    // is an mmap + page faults + munmap per node — more than the copy itself.
    static std::vector<std::vector<double>> &pool() { static thread_local std::vector<std::vector<double>> p; return p; }
    static std::vector<double> take_buffer(size_t elems) {
        auto &p = pool();
        for (size_t i = p.size(); i-- > 0;) if (p[i].size() >= elems) { std::vector<double> b = std::move(p[i]); p.erase(p.begin() + (long)i); return b; }
        return std::vector<double>(elems);
    }
    static bool poolable(const std::vector<double> &b) { return b.size() >= 4096 && b.size() <= (1u << 20) && pool().size() < 64; }  // <= 8 MB each, <= 64 of them
    // Small tableaus (the 8-column block of a worker class: ~40 copies per solve, ten vectors each) recycle ALL their vectors: a retired Tab leaves them in a
    // per-thread list of shells and the next copy / init adopts one, so that its assign()s find the capacity in place.  (Of the 337 heap allocations of a
    // C3-block solve 200 were these; the solve is 18 us on the build container, 10 us on the MI355X box's host.)
    struct Shell { std::vector<double> T, d, x, lb, ub, cost; std::vector<int> B, arow, where; std::vector<uint8_t> st; };
    static std::vector<Shell> &shells() { static thread_local std::vector<Shell> s; return s; }
    void swap_with(Shell &b) { T.swap(b.T); d.swap(b.d); x.swap(b.x); lb.swap(b.lb); ub.swap(b.ub); cost.swap(b.cost); B.swap(b.B); arow.swap(b.arow); where.swap(b.where); st.swap(b.st); }
    void adopt_shell() { auto &s = shells(); if (s.empty()) return; swap_with(s.back()); s.pop_back(); }
    void retire() {  // give the storage away (the object is about to be destroyed or overwritten)
        if (poolable(T)) { pool().push_back(std::move(T)); T = std::vector<double>(); }
        if (d.capacity() == 0 || T.capacity() > 4096) return;
        auto &s = shells();
        if (s.size() >= 64) return;
        s.emplace_back();
        swap_with(s.back());
    }
    ~Tab() { retire(); }
    Tab() = default;
    Tab(Tab &&) = default;
    Tab &operator=(Tab &&o) {
        if (this != &o) {
            retire();
            R = o.R; n = o.n; ma = o.ma; cap = o.cap; stride = o.stride;
            T = std::move(o.T); d = std::move(o.d); x = std::move(o.x); lb = std::move(o.lb); ub = std::move(o.ub); cost = std::move(o.cost);
            B = std::move(o.B); arow = std::move(o.arow); where = std::move(o.where); st = std::move(o.st);
            iters = o.iters; ops = o.ops; deadline = o.deadline;
        }
        return *this;
    }
    // B&B children and tie-break probes copy their parent: copy the active rows only, into a tableau with a little headroom
    Tab(const Tab &o) : R(o.R), n(o.n), ma(o.ma), cap(o.ma + 16), stride(o.n + o.ma + 16), iters(o.iters), ops(o.ops + (double)(o.ma + 1) * (double)(o.n + o.ma)), deadline(o.deadline) {
        adopt_shell();
        B = o.B; arow = o.arow; where = o.where;
        const size_t need = (size_t)cap * stride;
        if (need >= 4096) { T = take_buffer(need); }  // contents unspecified: rows < ma are written below, rows >= ma by activate() before any use
        else if (T.size() < need) T.resize(need);
        const int N = o.width();
        for (int r = 0; r < ma; r++) {
            double *dst = &T[(size_t)r * stride];
            memcpy(dst, &o.T[(size_t)r * o.stride], sizeof(double) * N);
            std::fill(dst + N, dst + stride, 0.0);  // the columns later slacks will take
        }
        auto cp = [&](std::vector<double> &dst, const std::vector<double> &src) { dst.assign(stride, 0.0); memcpy(dst.data(), src.data(), sizeof(double) * N); };
        cp(d, o.d); cp(x, o.x); cp(lb, o.lb); cp(ub, o.ub); cp(cost, o.cost);
        st.assign(stride, AT_LO); memcpy(st.data(), o.st.data(), N);
    }
    Tab &operator=(const Tab &o) { if (this != &o) { Tab t(o); *this = std::move(t); } return *this; }

    void init(const Rows *rows, const std::vector<double> &c, const std::vector<double> &clb, const std::vector<double> &cub) {
        R = rows; n = rows->n; ma = 0; cap = 32; stride = n + cap;
        if (d.capacity() == 0) adopt_shell();
        T.assign((size_t)cap * stride, 0.0);
        cost.assign(stride, 0.0); d.assign(stride, 0.0); x.assign(stride, 0.0); lb.assign(stride, 0.0); ub.assign(stride, 0.0); st.assign(stride, AT_LO);
        B.clear(); arow.clear(); where.assign(rows->m, -1);
        for (int j = 0; j < n; j++) {
            cost[j] = d[j] = c[j]; lb[j] = clb[j]; ub[j] = cub[j];
            if (c[j] > 0.0) { st[j] = AT_UP; x[j] = cub[j]; } else { st[j] = AT_LO; x[j] = clb[j]; }
        }
    }
    double objective() const { double z = 0.0; for (int j = 0; j < n; j++) z += cost[j] * x[j]; return z; }
    int width() const { return n + ma; }

    void grow() {
        int ncap = cap * 2, nstride = n + ncap;
        std::vector<double> nT((size_t)ncap * nstride, 0.0);
        for (int r = 0; r < ma; r++) memcpy(&nT[(size_t)r * nstride], &T[(size_t)r * stride], sizeof(double) * width());
        T.swap(nT);
        for (auto *v : {&d, &x, &lb, &ub, &cost}) v->resize(nstride, 0.0);
        st.resize(nstride, AT_LO);
        cap = ncap; stride = nstride;
    }

    // constraint i of R enters the tableau with its slack basic
    bool activate(int i) {
        if (ma == cap) { if ((double)cap * 2.0 * (double)(n + cap * 2) > TAB_LIMIT) return false; grow(); }
        const int a = ma, k = n + a, N = width();
        double *v = &T[(size_t)a * stride];
        std::fill(v, v + stride, 0.0);  // the whole row: recycled storage is not zero beyond what the copy constructor wrote
        double act = 0.0;
        for (int t = R->off[i]; t < R->off[i + 1]; t++) { v[R->col[t]] -= R->coef[t]; act += R->coef[t] * x[R->col[t]]; }
        for (int r = 0; r < a; r++) {  // express the row in the current nonbasic columns
            const int kb = B[r];
            if (kb >= n) continue;
            const double f = v[kb];
            if (f == 0.0) continue;
            const double *row = &T[(size_t)r * stride];
            axpy_neg(v, row, f, N);
            v[kb] = 0.0;
            ops += N;
        }
        v[k] = 1.0;
        B.push_back(k); arow.push_back(i); where[i] = a;
        st[k] = BASIC; lb[k] = R->lo[i]; ub[k] = R->hi[i]; cost[k] = 0.0; d[k] = 0.0; x[k] = act;
        ma++;
        return true;
    }

    // move a nonbasic variable to a new value, updating the basic ones
    void shift_nonbasic(int j, double nv) {
        double dl = nv - x[j];
        if (dl == 0.0) return;
        for (int r = 0; r < ma; r++) { double t = T[(size_t)r * stride + j]; if (t != 0.0) x[B[r]] -= t * dl; }
        x[j] = nv;
    }
    void set_lb(int j, double v) { lb[j] = v; if (st[j] == AT_LO) shift_nonbasic(j, v); else if (st[j] == AT_UP && ub[j] < v) shift_nonbasic(j, v); }
    void set_ub(int j, double v) { ub[j] = v; if (st[j] == AT_UP) shift_nonbasic(j, v); else if (st[j] == AT_LO && lb[j] > v) shift_nonbasic(j, v); }

    // A lower bound of how much the LP optimum drops when column j's upper bound is tightened to nv < x[j] — the first step of the dual simplex that would follow
    // (the leaving variable is j itself, the objective moves by the smallest ratio times the infeasibility, and never back): one ratio test, no pivot, no copy.
    // INF: no entering column, the tightened LP is infeasible.  The tableau must be optimal.
    double loss_if_ub(int j, double nv) const {
        if (x[j] <= nv + FEAS_TOL) return 0.0;
        const double delta = x[j] - nv;
        if (st[j] != BASIC) return std::fabs(d[j]) * delta;  // nonbasic (at its upper bound): moving it costs its reduced cost per unit, at least
        int r = -1;
        for (int i = 0; i < ma; i++) if (B[i] == j) { r = i; break; }
        if (r < 0) return 0.0;
        const double *prow = &T[(size_t)r * stride];
        const int N = width();
        double bratio = INF;
        for (int q = 0; q < N; q++) {
            if (st[q] == BASIC || lb[q] == ub[q]) continue;
            const double a = prow[q];
            if (!((st[q] == AT_LO && a > PIV_TOL) || (st[q] == AT_UP && a < -PIV_TOL))) continue;
            const double ratio = std::fabs(d[q]) / std::fabs(a);
            if (ratio < bratio) bratio = ratio;
        }
        return bratio >= INF ? INF : bratio * delta;
    }

    // dual simplex over the active rows
    int reoptimise(long max_iters) {
        const int N = width();
        const long bland_after = 400 + 8L * (ma + 8);  // a re-optimisation normally takes a handful of pivots; far beyond that it is stalling on
                                                        // degenerate ties: switch to smallest-index choices (Bland), which cannot cycle
        for (long it = 0; it < max_iters; it++) {
            const bool bland = it > bland_after;
            int r = -1; double best = FEAS_TOL; bool below = false; int rk = INT32_MAX;
            for (int i = 0; i < ma; i++) {
                int k = B[i]; double v = x[k];
                double inf = 0.0; bool bl = false;
                if (v < lb[k] - FEAS_TOL) { inf = lb[k] - v; bl = true; }
                else if (v > ub[k] + FEAS_TOL) inf = v - ub[k];
                else continue;
                if (bland ? k < rk : inf > best) { best = inf; r = i; below = bl; rk = k; }
            }
            if (r < 0) return LP_OPT;
            int k = B[r];
            if (lb[k] > ub[k] + FEAS_TOL) return LP_INFEAS;
            double *prow = &T[(size_t)r * stride];
            int q = -1; double bratio = INF, babs = 0.0;
            for (int j = 0; j < N; j++) {
                if (st[j] == BASIC) continue;
                if (lb[j] == ub[j]) continue;  // fixed: cannot move
                double a = prow[j];
                bool elig;
                if (below) elig = (st[j] == AT_LO && a < -PIV_TOL) || (st[j] == AT_UP && a > PIV_TOL);
                else elig = (st[j] == AT_LO && a > PIV_TOL) || (st[j] == AT_UP && a < -PIV_TOL);
                if (!elig) continue;
                double ratio = std::fabs(d[j]) / std::fabs(a);
                if (ratio < bratio - 1e-13 || (!bland && ratio <= bratio + 1e-13 && std::fabs(a) > babs)) { bratio = ratio; babs = std::fabs(a); q = j; }
            }
            if (q < 0) return LP_INFEAS;
            iters++;
            if (((iters & 63) == 0 || (double)ma * (double)N > 2.0e5) && wall() > deadline) return LP_LIMIT;  // a pivot of a large tableau costs milliseconds
            double target = below ? lb[k] : ub[k];
            double piv = prow[q];
            double dq = (x[k] - target) / piv;
            for (int i = 0; i < ma; i++) { double t = T[(size_t)i * stride + q]; if (t != 0.0) x[B[i]] -= t * dq; }
            x[q] += dq;
            x[k] = target;
            double inv = 1.0 / piv;
            for (int j = 0; j < N; j++) prow[j] *= inv;
            prow[q] = 1.0;
            ops += 3.0 * N + ma;
            for (int i = 0; i < ma; i++) {
                if (i == r) continue;
                double *ri = &T[(size_t)i * stride];
                double f = ri[q];
                if (f == 0.0) continue;
                axpy_neg(ri, prow, f, N);
                ri[q] = 0.0;
                ops += N;
            }
            double f = d[q];
            if (f != 0.0) { axpy_neg(d.data(), prow, f, N); d[q] = 0.0; }
            st[k] = below ? AT_LO : AT_UP;
            st[q] = BASIC; B[r] = q;
        }
        return LP_LIMIT;
    }

    // LP optimum over ALL rows of R: re-optimise, bring in the rows the point violates, repeat
    int solve(long max_iters) {
        std::vector<std::pair<double, int>> bad;
        for (;;) {
            int s = reoptimise(max_iters);
            if (s != LP_OPT) return s;
            bad.clear();
            ops += (double)R->col.size();
            for (int i = 0; i < R->m; i++) {
                if (where[i] >= 0) continue;
                double a = R->activity(i, x.data());
                double v = std::max(R->lo[i] - a, a - R->hi[i]);
                if (v > FEAS_TOL) bad.push_back({-v, i});
            }
            if (bad.empty()) return LP_OPT;
            std::sort(bad.begin(), bad.end());  // most violated first; ties by row index: deterministic
            size_t take = std::min<size_t>(bad.size(), std::max<size_t>(32, bad.size() / 4));
            for (size_t t = 0; t < take; t++) {
                if (!activate(bad[t].second)) return LP_LIMIT;
                if ((t & 15) == 15 && wall() > deadline) return LP_LIMIT;
            }
        }
    }
};

}  // namespace lp
}  // namespace hqmilp
