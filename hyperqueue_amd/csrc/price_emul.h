// libhqtick_test.so only (see price_emul.cpp).
#pragma once
#include <algorithm>
#include <vector>

#include "price.h"

namespace hqprice {

struct EmulatedSweeper : Sweeper {
    const HostTables *T = nullptr;
    std::vector<int32_t> caps;
    std::vector<double> bcaps;
    std::vector<uint16_t> pats;
    std::vector<double> blk_cx, blk_rc, blk_bnd;
    std::vector<uint32_t> blk_steps;
    std::vector<long long> slots;
    uint32_t n_sweeps = 0, cap_sweeps = 0;
    // fault injection (hqtick_debug_set_price_fault): begin() refuses the model (fail_at == 0) / the sweep number fail_at - 1 fails (fail_at >= 1); -1: off
    int fail_at = -1;
    bool begin(const HostTables &t, uint32_t max_sweeps) override;
    bool set_caps(const int32_t *col_cap) override;
    bool set_block_caps(const double *blk_cap) override;
    bool sweep(const double *pi, SweepTotals &out) override;
    bool sweep_range(const double *pi, uint32_t b0, uint32_t b1, RangeValues &out) override;
    const uint16_t *patterns(uint32_t first, uint32_t count) override;
    void end() override;
};

}  // namespace hqprice
