// The price sweeps of the coupled solve on the MI355X (csrc/price.hip): k_price_sweep, one workgroup (four wavefronts) per worker block.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "devbuf.h"
#include "price.h"

namespace hqprice {

struct DeviceSweeper : Sweeper {
    hipStream_t stream = nullptr;
    hqbuf::PinBuf h_stage, h_res, h_pats, h_pin, h_prof, h_blkv;
    static constexpr uint32_t PIN_SWEEPS = 64;   // sweeps whose patterns the kernel writes into pinned memory (price.hip: launch_only)
    bool sync_clean = false;   // d_sync is all zero (price.hip: begin)
    bool in_flight = false; double flight_t0 = 0; uint32_t flight_blocks = 0;   // a sweep launched and not yet waited for
    bool profile = getenv("HQTICK_PRICE_PROFILE") != nullptr; double prof_med[16] = {0}, prof_max[16] = {0}, prof_steps = 0, prof_span = 0, prof_tail = 0; int prof_n = 0;
    hqbuf::DevBuf d_tab, d_pats, d_blk, d_sync, d_prof;
    const HostTables *T = nullptr;
    size_t o_off = 0, o_m = 0, o_cap = 0, o_cost = 0, o_a = 0, o_ccap = 0, o_woff = 0, o_wrow = 0, o_wcoef = 0, tab_bytes = 0;
    uint32_t n_sweeps = 0, cap_sweeps = 0, seq = 0;
    uint32_t max_block_cols = 0;   // widest block of the model at hand: picks the kernel's working-set size (price.hip)
    bool force_nmax = getenv("HQTICK_PRICE_NMAX") != nullptr;
    int force_waves = getenv("HQTICK_PRICE_WAVES") ? atoi(getenv("HQTICK_PRICE_WAVES")) : 0;   // A/B switch: wavefronts per block (1 / 2; anything else: 4, price.hip: launch)
    uint32_t dbg = getenv("HQTICK_PRICE_DBG") ? (uint32_t)atoi(getenv("HQTICK_PRICE_DBG")) : 0u;   // experiments (price_core.h: SweepOut::dbg)
    double last_kernel_us = 0;   // duration of the last sweep as the host saw it (launch -> result visible)
    // statistics for the bench line
    uint64_t total_sweeps = 0, total_block_solves = 0; double total_us = 0;
    explicit DeviceSweeper(hipStream_t s) : stream(s) {}
    ~DeviceSweeper() override;
    bool begin(const HostTables &t, uint32_t max_sweeps) override;
    bool set_caps(const int32_t *col_cap) override;
    bool set_block_caps(const double *blk_cap) override;
    bool sweep(const double *pi, SweepTotals &out) override;
    bool sweep_range(const double *pi, uint32_t b0, uint32_t b1, RangeValues &out) override;   // this rank's blocks of a sharded sweep (price.h: ShardedSweeper)
    bool launch(const double *pi, uint32_t b0, uint32_t b1, bool local, SweepTotals *out);
    bool launch_only(const double *pi, uint32_t b0, uint32_t b1, bool local);
    bool wait_done(SweepTotals *out);
    bool sweep_launch(const double *pi) override;
    bool sweep_finish(SweepTotals &out) override;
    const uint16_t *patterns(uint32_t first, uint32_t count) override;
    void end() override;
};

}  // namespace hqprice
