// The coupled placement solve by price sweeps: host side.
//
// run_scheduling_solver's model (/root/reference/crates/tako/src/internal/scheduler/solver.rs:95-430) is one small block per worker plus a few
// WIDE rows across the workers (batch sizes :264-271, "blocker short" flags :233-253, priority cuts :274-429).  The reference gives it to HiGHS
// (solver/highs.rs:65-88) and accepts an incumbent proven within mip_rel_gap = 1e-4.  Here:
//   bound    Dantzig-Wolfe / Lagrangian over the wide rows with INTEGER blocks:  pi.h + sum_w V_w(pi)  >= optimum for every pi >= 0; a cutting-plane
//            master of K + 1 variables on the host picks the prices, every evaluation is one sweep over all blocks on the MI355X (price_core.h);
//   primal   at the master's optimum the maximisers of the active cuts are all optimal at the final prices: choosing ONE of them per worker by error
//            diffusion over the wide rows' running totals gives an integer point whose wide-row activities are a few tasks away from the LP's, a
//            short repair makes it feasible — typically 1e-5 from the bound, where the window search of csrc/milp.cpp needed seconds for 1e-4;
//   flags    the model's global 0/1 columns ("blocker short") are fixed per round, all on at first, and dropped while their `>=` row holds
//            without them (the rule of milp.cpp's sparse_greedy); the bound keeps them relaxed in [0, 1].
// The caller (CompSolver::run, csrc/milp.cpp) verifies the point against ITS rows and certifies it against the bound; what is not certified
// goes on to the host's window search with this bound and this incumbent.
#pragma once
#include <cstdint>
#include <functional>
#include <vector>

namespace hqprice {

// the flattened blocks (host copy of price_core.h's Tables)
struct HostTables {
    uint32_t n_blocks = 0, n_cols = 0, K = 0;
    std::vector<uint32_t> blk_off; std::vector<uint8_t> blk_m; std::vector<double> blk_cap;
    std::vector<double> col_cost, col_a; std::vector<int32_t> col_cap; std::vector<uint32_t> col_woff; std::vector<uint16_t> w_row; std::vector<int32_t> w_coef;
};

struct SweepTotals {
    double cx = 0, rc = 0, bnd = 0;     // sums over the blocks, in block order
    uint32_t n_budget = 0, max_steps = 0;
    std::vector<long long> act;         // [K]
    // the same per PART of the model (PARTS contiguous ranges of blocks, price_core.h's ASLOTS): the master models every part's value function on its own
    std::vector<double> part_cx;        // [PARTS]
    std::vector<long long> part_act;    // [PARTS * K]
};
constexpr int PARTS = 16;

// Where the sweeps run: the MI355X (csrc/price.hip) in the tick, the emulated wavefront (libhqtick_test.so) in the CPU tests.
struct Sweeper {
    virtual ~Sweeper() {}
    virtual bool begin(const HostTables &t, uint32_t max_sweeps) = 0;       // false: cannot run this model (the host search takes over)
    virtual bool set_caps(const int32_t *col_cap) = 0;                      // new column bounds for the following sweeps
    virtual bool set_block_caps(const double *blk_cap) = 0;                 // new row capacities [n_blocks * 4] (a branch that fixes part of a block's content)
    virtual bool sweep(const double *pi, SweepTotals &out) = 0;             // sweep number = count of sweeps since begin()
    virtual const uint16_t *patterns(uint32_t first, uint32_t count) = 0;   // host pointer to the patterns of sweeps [first, first + count): [count][n_cols]
    virtual void end() = 0;
    uint32_t min_cols = 256;    // components below this many columns stay with the host search (a sweep is ~80 us whatever the block count: below ~250 columns the host tree is usually done first)
    uint32_t budget = 4096;     // search steps per block and sweep
    // statistics of the last solve
    uint32_t stat_sweeps = 0;
    double stat_sweep_us = 0;   // wall time inside sweep()
};

// The component as CompSolver holds it: rows scaled to max |coef| = 1 (row_scale gives the original back), costs scaled to max 1, lower bounds 0.
struct Request {
    int n = 0, m = 0;
    const int *roff = nullptr, *rcol = nullptr; const double *rcoef = nullptr, *rlo = nullptr, *rhi = nullptr;
    const double *row_scale = nullptr; const uint8_t *row_implied = nullptr;  // row_implied may be nullptr
    const int32_t *col_group = nullptr;
    const double *c = nullptr, *ub = nullptr;
    const double *incumbent = nullptr; double incumbent_value = 0.0;  // nullptr: none
    double rel_gap = 1e-4;
    double time_limit_s = 5.0;  // the caller's CONFIGURED time limit: the deterministic work caps of the branch-and-price phase scale with it
    double deadline_s = 1e300;  // ... and the wall-clock point (steady clock, seconds) at which that limit runs out: the one clock in this path — branch-and-price stops at
                                // 70 % of the limit whatever its counts say (like the host search it hands over to, a tick that gets there may differ between replicas)
    // checks a point against ALL rows of the component, raises what can still be raised, returns the value; false: the point violates a row
    std::function<bool(std::vector<double> &x, double &value)> polish;
    bool trace = false;
};

struct Answer {
    bool ran = false;              // the model has the supported shape and the sweeps ran
    double bound = 1e300;          // upper bound of the model's optimum (flags relaxed)
    std::vector<double> x;         // a candidate point (empty: none); the caller checks it against its own rows
    double x_value = 0.0;
    uint32_t sweeps = 0, rounds = 0;
    const char *why = "";          // ran == false: what kept the model on the host
    // for the guided windows of the host's search: block of every column (-1: global), the blocks' values at the final prices and the reduced costs
    std::vector<int> block_of; std::vector<double> block_value, rcost;
};

Answer solve(const Request &rq, Sweeper &sw);

}  // namespace hqprice
