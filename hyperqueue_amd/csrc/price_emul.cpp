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
    return t.K <= (uint32_t)KMAX;
}
bool EmulatedSweeper::set_caps(const int32_t *c) { caps.assign(c, c + T->n_cols); return true; }
bool EmulatedSweeper::set_block_caps(const double *c) { bcaps.assign(c, c + (size_t)T->n_blocks * MMAX); return true; }
bool EmulatedSweeper::sweep(const double *pi, SweepTotals &out) {
    if (n_sweeps >= cap_sweeps) return false;
    static thread_local hqblock::Shared *S = new hqblock::Shared();
    hqblock::HostWave wv;
    const HostTables &t = *T;
    Tables tv{t.n_blocks, t.n_cols, t.K, t.blk_off.data(), t.blk_m.data(), bcaps.data(), t.col_cost.data(), t.col_a.data(), caps.data(), t.col_woff.data(), t.w_row.data(), t.w_coef.data()};
    pats.resize((size_t)(n_sweeps + 1) * t.n_cols);
    std::vector<long long> slots((size_t)ASLOTS * t.K, 0);
    SweepOut so{pats.data() + (size_t)n_sweeps * t.n_cols, blk_cx.data(), blk_rc.data(), blk_bnd.data(), slots.data(), blk_steps.data(), nullptr};
    for (uint32_t b = 0; b < t.n_blocks; b++) solve_priced_block(wv, *S, tv, pi, b, so, budget);
    out.act.assign(t.K, 0);
    for (int sl = 0; sl < ASLOTS; sl++) for (uint32_t k = 0; k < t.K; k++) out.act[k] += slots[(size_t)sl * t.K + k];
    out.part_act = slots;
    out.part_cx.assign(ASLOTS, 0.0);
    {   // the device's order again: four lanes per part, every fourth block each, then the four partial sums
        const uint32_t per = part_size(t.n_blocks);
        for (uint32_t g = 0; g < (uint32_t)ASLOTS; g++) {
            const uint32_t b0 = g * per, b1 = std::min(t.n_blocks, b0 + per);
            double s4[4] = {0, 0, 0, 0};
            for (uint32_t p = 0; p < 4; p++) for (uint32_t b = b0 + p; b < b1; b += 4) s4[p] += blk_cx[b];
            out.part_cx[g] = ((s4[0] + s4[1]) + s4[2]) + s4[3];
        }
    }
    // the device's order: lane l of the last workgroup adds blocks l, l + 64, ...; lane 0 then adds the 64 partial sums in lane order — the same
    // floating-point sums here, so that a GPU tick and the emulation walk the same sequence of prices
    double pcx[WAVE] = {0}, prc[WAVE] = {0}, pb[WAVE] = {0};
    out.n_budget = 0; out.max_steps = 0;
    for (uint32_t b = 0; b < t.n_blocks; b++) {
        pcx[b % WAVE] += blk_cx[b]; prc[b % WAVE] += blk_rc[b]; pb[b % WAVE] += blk_bnd[b];
        if (blk_steps[b] & 0x80000000u) out.n_budget++;
        out.max_steps = std::max(out.max_steps, blk_steps[b] & 0x7FFFFFFFu);
    }
    out.cx = out.rc = out.bnd = 0.0;
    for (int l = 0; l < WAVE; l++) { out.cx += pcx[l]; out.rc += prc[l]; out.bnd += pb[l]; }
    n_sweeps++;
    return true;
}
const uint16_t *EmulatedSweeper::patterns(uint32_t first, uint32_t count) { return first + count <= n_sweeps ? pats.data() + (size_t)first * T->n_cols : nullptr; }
void EmulatedSweeper::end() {}

}  // namespace hqprice
