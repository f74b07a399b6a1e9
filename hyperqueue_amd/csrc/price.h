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

// What a sweep over a RANGE of blocks leaves (worker-range shards): arrays indexed by ABSOLUTE block number, valid inside the range; the partial
// activity vectors of all PARTS (those outside the range zero).  Host memory, valid until the next call.
struct RangeValues { const double *cx = nullptr, *rc = nullptr, *bnd = nullptr; const uint32_t *steps = nullptr; const long long *part_act = nullptr; };

// The sweep's totals from the per-block values, in the ORDER THE KERNEL's last workgroup adds them (csrc/price.hip): lane l takes blocks l, l + 64, ..., lane 0
// then adds the 64 partial sums in lane order; c.x per part by four lanes, every fourth block each.  One definition for the emulation, for the shards' merge
// and (restated in device code) the kernel: a sharded tick, a plain tick and the emulated tick walk the same sequence of prices.
void totals_from_blocks(uint32_t n_blocks, uint32_t K, const double *cx, const double *rc, const double *bnd, const uint32_t *steps, const long long *part_act, SweepTotals &out);

// Where the sweeps run: the MI355X (csrc/price.hip) in the tick, the emulated wavefront (libhqtick_test.so) in the CPU tests.
struct Sweeper {
    virtual ~Sweeper() {}
    virtual bool begin(const HostTables &t, uint32_t max_sweeps) = 0;       // false: cannot run this model (the host search takes over)
    virtual bool set_caps(const int32_t *col_cap) = 0;                      // new column bounds for the following sweeps
    virtual bool set_block_caps(const double *blk_cap) = 0;                 // new row capacities [n_blocks * 4] (a branch that fixes part of a block's content)
    virtual bool sweep(const double *pi, SweepTotals &out) = 0;             // sweep number = count of sweeps since begin()
    // the same in two halves, for a caller with host work that does not depend on the sweep's outcome (the first sweep of a solve runs at zero prices): launch, ..., finish.
    // The default runs the whole sweep in finish().
    virtual bool sweep_launch(const double *pi) { pending_pi.assign(pi, pi + pending_k); return true; }
    virtual bool sweep_finish(SweepTotals &out) { return sweep(pending_pi.data(), out); }
    std::vector<double> pending_pi; uint32_t pending_k = 0;   // (set by whoever calls sweep_launch: the number of prices)
    virtual const uint16_t *patterns(uint32_t first, uint32_t count) = 0;   // host pointer to the patterns of sweeps [first, first + count): [count][n_cols]
    virtual void end() = 0;
    // one sweep over the blocks [b0, b1) only — this rank's worker range of a sharded scheduler; the patterns of the other blocks' columns are left untouched
    virtual bool sweep_range(const double *pi, uint32_t b0, uint32_t b1, RangeValues &out) { (void)pi; (void)b0; (void)b1; (void)out; return false; }
    // The one clock of the price path (Request::deadline_s) is read HERE, at a sweep, and nowhere else: `time_up` turns true (and stays) once a sweep ends past the
    // guard.  A sharded sweeper ORs the ranks' readings inside the sweep's exchange, so that every replica leaves the sweeps at the same sweep.
    double guard_s = 1e300;     // steady-clock second at which the sweeps stop being worth starting (set by solve())
    bool time_up = false;
    virtual bool merges_clock() const { return false; }   // true: sweep() itself sets time_up (from every rank's reading)
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

// The same solve straight from the model as its builder wrote it (hqmilp::Model's arrays, csrc/milp.h) — no component copy, no row scaling, and every list of
// leading terms that several rows share (row_lhs) read ONCE: the coupled tick's fast path (csrc/milp.cpp, solve()).  Column bounds are derived here the way
// hqmilp::solve derives them (BOOL: 1; every `<=` / `==` row without a negative coefficient bounds its columns).  Answer::x is over the model's columns,
// Answer::bound / x_value in units of obj / *cost_scale (the largest |obj|).  The candidate point was checked against the blocks' rows, the wide rows and the
// conditional bounds and raised greedily; the caller still checks it against ALL rows of the model.
struct ModelView {
    int n = 0, m = 0;
    const double *obj = nullptr; const uint8_t *kind = nullptr;         // kind: 0 = nat, 1 = bool (hqmilp::COL_*)
    const uint8_t *rtype = nullptr; const double *rhs = nullptr;        // rtype: 0 = `>=`, 1 = `<=`, 2 = `==` (hqmilp::ROW_*)
    const int *roff = nullptr, *rcol = nullptr; const double *rcoef = nullptr;
    const int32_t *col_group = nullptr; const uint8_t *row_implied = nullptr;   // row_implied may be nullptr
    const int32_t *row_lhs = nullptr, *row_lhs_len = nullptr;
    const int *list_off = nullptr, *list_col = nullptr; int n_lists = 0;   // the shared lists row_lhs names (milp.h: Model::list_off / list_col): coefficient 1 each, not among the rows' stored terms
    const int32_t *row_block = nullptr; const uint32_t *col_ub = nullptr;   // optional (milp.h: Model::row_block / col_ub)
};
// tests: also check the builder's column bounds against the rows they stand for (this thread); mismatches since the last call that switched it on
void set_check_hints(bool on);
int hint_mismatches();
Answer solve_model(const ModelView &mv, double rel_gap, double time_limit_s, double deadline_s, bool trace, Sweeper &sw, double *cost_scale);

// All-gather of small host buffers between the ranks of a sharded scheduler: librccl inside the library (hqtick_comm_init) or a callback of the host
// (hqtick_set_exchange) — csrc/hqtick.cpp.  Every rank contributes `bytes` (the same everywhere); recv gets world x bytes, rank-major.
struct Exchange {
    uint32_t rank = 0, world = 1;
    virtual ~Exchange() {}
    virtual bool allgather(const void *send, void *recv, size_t bytes) = 0;
    uint64_t n_calls = 0, n_bytes = 0; double us = 0;   // statistics
};

// A sweeper that runs only THIS rank's blocks — the contiguous range made of its share of the master's 16 PARTS (worker ranges) — and completes every sweep with
// one small all-gather: per block (c.x, reduced value, bound, steps), per part the activity vector, and the rank's reading of the clock (the blocks
// are independent per worker; what couples them is the master, which stays replicated).  The totals are then added up in the kernel's order from the complete
// per-block arrays, so every rank — and the unsharded tick — sees bit-identical cuts.  Patterns cross once, when the master asks for them.
struct ShardedSweeper : Sweeper {
    Sweeper &inner; Exchange &ex;
    const HostTables *T = nullptr;
    std::vector<uint32_t> rank_b0, rank_b1, rank_p0, rank_p1;   // blocks / parts of every rank
    uint32_t max_blocks = 0, max_parts = 0, max_cols = 0;
    std::vector<unsigned char> send, recv;
    std::vector<double> cx, rc, bnd; std::vector<uint32_t> steps; std::vector<long long> part_act;
    std::vector<uint16_t> pats;
    uint32_t n_sweeps = 0;
    uint32_t min_blocks = 1025;   // models with fewer blocks are swept whole by every rank (one round of resident workgroups: a sweep costs ~70 us whatever its block count up to 1024)
    bool pass = false;            // ... decided in begin(), the same on every rank
    ShardedSweeper(Sweeper &in, Exchange &e, uint32_t min_blocks_) : inner(in), ex(e), min_blocks(min_blocks_) { min_cols = in.min_cols; budget = in.budget; }
    bool begin(const HostTables &t, uint32_t max_sweeps) override;
    bool set_caps(const int32_t *col_cap) override { return inner.set_caps(col_cap); }
    bool set_block_caps(const double *blk_cap) override { return inner.set_block_caps(blk_cap); }
    bool sweep(const double *pi, SweepTotals &out) override;
    const uint16_t *patterns(uint32_t first, uint32_t count) override;
    void end() override { inner.end(); T = nullptr; }
    bool merges_clock() const override { return ex.world > 1; }   // (also when every rank sweeps the whole of a small model: one word per rank and sweep)
};

}  // namespace hqprice
