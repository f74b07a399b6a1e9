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
    //   row_lhs[i] >= 0: the first row_lhs_len[i] terms of row i are the SAME list (columns, coefficients, order) in every row that carries this id — a batch's cut rows
    //                      against its blockers differ in their flag and right-hand side only; -1: no statement.  What lets the coupled solve read such a list once.
    std::vector<int32_t> col_group;
    std::vector<uint8_t> row_implied;
    std::vector<int32_t> row_lhs, row_lhs_len;
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
