// libhqtick_test.so only: the price sweeps of the coupled solve (csrc/price_core.h = the algorithm of k_price_sweep) with the wavefront emulated on
// the CPU — the 64 lanes of every lane-parallel step run in a loop — so that the CPU test suite executes the kernel's logic and the host side of
// csrc/price.cpp without a GPU.  Not a product path: libhqtick.so has no CPU sweeper, its coupled solve needs the MI355X (csrc/price.hip).
#include "price_emul.h"

#include <cstring>

#include "price_core.h"

namespace hqprice {

static_assert(PARTS == ASLOTS, "the master's parts are the kernel's activity slots");

bool EmulatedSweeper::begin(const HostTables &t, uint32_t max_sweeps) {
    T = &t; caps = t.col_cap; bcaps = t.blk_cap; n_sweeps = 0; cap_sweeps = max_sweeps;
    pats.clear();
    blk_cx.assign(t.n_blocks, 0.0); blk_rc.assign(t.n_blocks, 0.0); blk_bnd.assign(t.n_blocks, 0.0); blk_steps.assign(t.n_blocks, 0);
    return t.K <= (uint32_t)KMAX && fail_at != 0;
}
bool EmulatedSweeper::set_caps(const int32_t *c) { caps.assign(c, c + T->n_cols); return true; }
bool EmulatedSweeper::set_block_caps(const double *c) { bcaps.assign(c, c + (size_t)T->n_blocks * MMAX); return true; }
bool EmulatedSweeper::sweep_range(const double *pi, uint32_t b0, uint32_t b1, RangeValues &rv) {
    if (n_sweeps >= cap_sweeps || !T || b1 > T->n_blocks || b0 > b1) return false;
    if (fail_at >= 1 && n_sweeps == (uint32_t)(fail_at - 1)) return false;
    static thread_local hqblock::Shared *S = new hqblock::Shared();
    hqblock::HostWave wv;
    const HostTables &t = *T;
    Tables tv{t.n_blocks, t.n_cols, t.K, t.blk_off.data(), t.blk_m.data(), bcaps.data(), t.col_cost.data(), t.col_a.data(), caps.data(), t.col_woff.data(), t.w_row.data(), t.w_coef.data()};
    pats.resize((size_t)(n_sweeps + 1) * t.n_cols);
    slots.assign((size_t)ASLOTS * t.K, 0);
    SweepOut so{pats.data() + (size_t)n_sweeps * t.n_cols, blk_cx.data(), blk_rc.data(), blk_bnd.data(), slots.data(), blk_steps.data(), nullptr};
    for (uint32_t b = b0; b < b1; b++) solve_priced_block(wv, *S, tv, pi, b, so, budget);
    rv = RangeValues{blk_cx.data(), blk_rc.data(), blk_bnd.data(), blk_steps.data(), slots.data()};
    n_sweeps++;
    return true;
}
bool EmulatedSweeper::sweep(const double *pi, SweepTotals &out) {
    RangeValues rv;
    if (!sweep_range(pi, 0, T ? T->n_blocks : 0, rv)) return false;
    // the device's order of additions (price.h: totals_from_blocks), so that a GPU tick and the emulation walk the same sequence of prices
    totals_from_blocks(T->n_blocks, T->K, rv.cx, rv.rc, rv.bnd, rv.steps, rv.part_act, out);
    return true;
}
const uint16_t *EmulatedSweeper::patterns(uint32_t first, uint32_t count) { return first + count <= n_sweeps ? pats.data() + (size_t)first * T->n_cols : nullptr; }
void EmulatedSweeper::end() {}

}  // namespace hqprice
