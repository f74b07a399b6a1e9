// Exact small-MILP solver of the MI355X tick (host side of the placement stage).
//
// Replaces what the reference delegates to HiGHS: `LpSolver::solve_bounded` / `solve`
// (/root/reference/crates/tako/src/internal/solver/mod.rs:200-219, solver/highs.rs:51-88): maximise c.x over
// integer columns (`0..` nat, `0..=1` bool) subject to Min/Max/Eq rows.
//
// Result convention ("canonical optimum"): the model is split into the connected components of
// its row/column incidence graph; inside every component, among the feasible integer vectors whose objective is
// within 1e-9 (relative) of the component's optimum, the returned one minimises the LAST column, then the one before
// it, and so on (lexicographically smallest read from the last column backwards).  Where the optimum is unique
// (almost all of the reference's pinned unit tests) this is simply the optimum HiGHS returns; where it is not, it
// makes the answer a function of the model alone.  Why this order: auxiliary columns are created last
// (scheduler/solver.rs:233-253 blocker flags), so "blocker fully placed" (flag = 0) wins ties — the one pinned test
// with tied optima, test_schedule_gap_filling3 (tests/test_scheduler_sn.rs:497-525), is satisfied only by that choice,
// which is also what HiGHS returns there — and among placement columns ties go to earlier batches / lower worker ids,
// the same direction the objective's (W - idx) factor pushes.
#pragma once
#include <cstdint>
#include <algorithm>
#include <vector>

namespace hqprice { struct Sweeper; }

namespace hqmilp {

enum { COL_NAT = 0, COL_BOOL = 1 };
enum { ROW_MIN = 0, ROW_MAX = 1, ROW_EQ = 2 };  // ConstraintType  solver/mod.rs:22-27

struct Model {
    std::vector<double> obj;
    std::vector<uint8_t> kind;
    std::vector<uint8_t> rtype;
    std::vector<double> rhs;
    std::vector<int> roff{0};
    std::vector<int> rcol;
    std::vector<double> rcoef;
    std::vector<double> start;  // optional integral starting point (ncols values); used as incumbent if it satisfies every row
    // Optional structure hints of the model's builder (both empty = none; the solver's answer never depends on them, only how fast it gets there):
    //   col_group[j] >= 0: column j belongs to that block (the tick: a worker's placement columns, solver.rs:95-192); -1: a column of the whole model
    //                      (the "blocker short" flags, solver.rs:233-253).  What the price sweeps of csrc/price.cpp decompose along.
    //   row_implied[i] != 0: row i is implied, for INTEGER points, by the other rows of its block (a cut from the block's own integer optimum)
    //   row_lhs[i] >= 0: the left-hand side of row i BEGINS with shared list row_lhs[i] (below), every column of it with coefficient 1 — a batch's cut rows against
    //                      its blockers differ in their flag and right-hand side only.  Those terms are NOT in rcol / rcoef (the row's stored terms are what follows
    //                      the list: a flag, usually): a three-level C3 tick has 70 such rows over 16 lists of ~1000 columns — written once per list instead of once per
    //                      row, read once per list by the coupled solve.  row_lhs_len[i] = the list's length.  -1: an ordinary row.  expand_lists() turns the model
    //                      into its plain form (what every consumer other than the coupled solve's fast path works on).
    std::vector<int32_t> col_group;
    std::vector<uint8_t> row_implied;
    std::vector<int32_t> row_lhs, row_lhs_len;
    std::vector<int> list_off{0}, list_col;   // shared lists: list l = list_col[list_off[l] .. list_off[l + 1])
    int add_list(const int *cols, size_t n) { list_col.insert(list_col.end(), cols, cols + n); list_off.push_back((int)list_col.size()); return (int)list_off.size() - 2; }
    bool has_lists() const { return list_off.size() > 1; }
    void expand_lists() {   // the plain form: every row carries all its terms (a list's first, in the list's order, coefficient 1), no list left
        if (!has_lists()) return;
        const int m = nrows();
        std::vector<int> noff; noff.reserve((size_t)m + 1); noff.push_back(0);
        size_t total = rcol.size();
        for (int i = 0; i < m; i++) if (i < (int)row_lhs.size() && row_lhs[i] >= 0) total += (size_t)(list_off[row_lhs[i] + 1] - list_off[row_lhs[i]]);
        std::vector<int> ncol; std::vector<double> ncoef; ncol.reserve(total); ncoef.reserve(total);
        for (int i = 0; i < m; i++) {
            if (i < (int)row_lhs.size() && row_lhs[i] >= 0) {
                const int l = row_lhs[i];
                ncol.insert(ncol.end(), list_col.begin() + list_off[l], list_col.begin() + list_off[l + 1]);
                ncoef.resize(ncol.size(), 1.0);
            }
            ncol.insert(ncol.end(), rcol.begin() + roff[i], rcol.begin() + roff[i + 1]);
            ncoef.insert(ncoef.end(), rcoef.begin() + roff[i], rcoef.begin() + roff[i + 1]);
            noff.push_back((int)ncol.size());
        }
        roff.swap(noff); rcol.swap(ncol); rcoef.swap(ncoef);
        list_off.assign(1, 0); list_col.clear();
        std::fill(row_lhs.begin(), row_lhs.end(), -1); std::fill(row_lhs_len.begin(), row_lhs_len.end(), 0);
    }
    //   row_block[i] >= 0: every column of row i belongs to that block (col_group value) and to nothing else — a worker's resource rows; -1: no statement
    //   col_ub[j] != UINT32_MAX: column j cannot exceed this value in any integer point (what the worker's free resources allow: min over the request's entries of
    //                            floor(free / amount), in exact integers) — a bound the rows imply, handed over so that nobody has to derive it again
    std::vector<int32_t> row_block;
    std::vector<uint32_t> col_ub;
    int ncols() const { return (int)obj.size(); }
    int nrows() const { return (int)rhs.size(); }
    int add_col(double w, uint8_t k) {
        obj.push_back(w);
        kind.push_back(k);
        return (int)obj.size() - 1;
    }
    void begin_row(uint8_t t, double b) {
        rtype.push_back(t);
        rhs.push_back(b);
    }
    void term(int c, double v) {
        rcol.push_back(c);
        rcoef.push_back(v);
    }
    void end_row() { roff.push_back((int)rcol.size()); }
};

struct Result {
    std::vector<double> x;   // integral values
    double objective = 0.0;  // c.x in the model's own (unscaled) coefficients
    bool feasible = false;   // false => the reference's `None` (infeasible / unbounded)   highs.rs:82
    bool optimal = false;    // false with feasible => time limit hit, incumbent returned   highs.rs:73-80; true = certified within rel_gap (below)
    bool canonical = true;   // false: some component's tie-break phase was skipped / cut short (or not requested): x is an optimum, not THE canonical one
    long nodes = 0, lp_iters = 0;
    int n_components = 0;
    int price_sweeps = 0, price_rounds = 0;      // block sweeps of the coupled solve / flag configurations tried (0: the host-only search ran)
    double price_bound_us = 0, price_total_us = 0;
};

// What the reference calls optimal: solve_bounded sets `time_limit` and nothing else (solver/highs.rs:65-68), so HiGHS runs with its default
// mip_rel_gap = 1e-4 and reports HighsModelStatus::Optimal as soon as (dual bound - incumbent) <= 1e-4 |incumbent|.
const double REFERENCE_MIP_REL_GAP = 1e-4;

// canonical=true applies the lexicographic tie-break (always on in the product; off only in solver unit tests).
// rel_gap: `optimal` = every component's incumbent is certified within rel_gap of its bound (0: proven exact only).  A certified component gets one
// more, work-budgeted search with the exact rule; only where that finishes does the tie-break run, otherwise `canonical` comes back false.
// sweeper (optional): the block sweeps of the coupled solve (csrc/price.h) — on the MI355X in the tick (k_price_sweep), the emulated wavefront in the CPU
// tests; nullptr: the host-only search.  Used for large components whose columns carry `col_group`.
Result solve(const Model &m, double time_limit_s, bool canonical = true, double rel_gap = REFERENCE_MIP_REL_GAP, hqprice::Sweeper *sweeper = nullptr);

// columns by descending cost, ties by ascending index (a stable order of the indices): the order the solver's greedy raises go through the columns in
void columns_by_cost_desc(const double *c, int n, std::vector<int> &out);
// the coupled tick's fast path (solve(): large structured models straight to the price sweeps) for this thread: 1 on, 0 off, -1 the default (on unless HQMILP_FAST=0)
void set_fast_path(int on);

}  // namespace hqmilp
