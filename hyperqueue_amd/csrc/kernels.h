// Launch wrappers of the gfx950 kernels of the tick (kernels.hip).  Host-callable, plain pointers (device memory).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace hqk {

// Capacity of the distinct-priority hash set (power of two).  More distinct priority levels than
// HQK_PRIO_SET_CAP/2 in one ready set is reported as HQTICK_E_CAPACITY.
static const uint32_t PRIO_SET_CAP = 1u << 15;
static const uint64_t PRIO_EMPTY = 0xFFFFFFFFFFFFFFFFull;  // Priority u64::MAX is representable; it is tracked by a side flag
static const uint32_t MAX_GROUPS_4W = 2048;                // G <= this: 4 waves per block share the LDS
static const uint32_t MAX_GROUPS = 16384;                  // hard cap on G = levels x requests
static const uint32_t MAX_LEVELS = 4096;                   // distinct priority levels per tick (LDS sort)

struct WaveGeom {
    uint32_t tasks_per_wave;  // contiguous ready-set slice a wavefront owns (multiple of 64)
    uint32_t n_waves;         // number of slices
    uint32_t waves_per_block; // 4 or 1
};

// K0: insert every task priority into the open-addressing set `set` (PRIO_SET_CAP slots, pre-filled with PRIO_EMPTY).
// flags[0] |= 1 when some priority equals PRIO_EMPTY itself, flags[1] = 1 on overflow.
hipError_t distinct_priorities(const uint64_t *prio, uint64_t n, uint64_t *set, uint32_t *flags, hipStream_t s);
// K0b: compact the set and sort it descending into levels[]; n_levels[0] = L.  Single workgroup.
hipError_t sort_levels(const uint64_t *set, const uint32_t *flags, uint64_t *levels, uint32_t *n_levels, hipStream_t s);

// K1: per-wave-slice histogram of (level, rq) groups.  wave_cnt is [n_waves][G] (G = L*Q, g = level*Q + rq).
hipError_t level_hist(const uint64_t *prio, const uint32_t *rq, uint64_t n, const uint64_t *levels, uint32_t L, uint32_t Q,
                WaveGeom geom, uint32_t *wave_cnt, uint32_t *err_flag, hipStream_t s);
// K1b: exclusive scan of wave_cnt over the wave axis (in place -> offsets) and totals into hist[G].
hipError_t scan_waves(uint32_t *wave_cnt, uint32_t n_waves, uint32_t G, uint32_t *hist, hipStream_t s);

// K2: per (worker, variant) capability flags and task_max_count (server/workerload.rs:77-83,121-145).
//   flags bit0: free resources cover the variant (have_immediate_resources_for_rq)
//         bit1: total resources cover it (is_capable_to_run_request)      bit2: has_time_to_run(min_time)
struct RequestTable {
    const uint32_t *variant_entry_off;  // [NV+1]
    const uint32_t *entry_resource;
    const uint8_t *entry_kind;
    const uint64_t *entry_amount;
    const uint64_t *variant_min_time_ns;
    uint32_t n_variants;
};
hipError_t worker_eval(const uint64_t *total, const uint64_t *free_, const int64_t *remaining_ns, uint32_t W, uint32_t R,
                 RequestTable rt, uint8_t *flags, uint32_t *tmc, hipStream_t s);

// K4: select the first take[g] tasks (ascending id) of every group and scatter them to sel_task/sel_level at
// base[g] + rank.  wave_off = output of scan_waves.
hipError_t select_scatter(const uint64_t *task_id, const uint64_t *prio, const uint32_t *rq, uint64_t n, const uint64_t *levels,
                    uint32_t L, uint32_t Q, WaveGeom geom, const uint32_t *wave_off, const uint32_t *take,
                    const uint32_t *base, uint64_t *sel_task, uint16_t *sel_level, hipStream_t s);

// K5: expand per-(request,variant,worker) counts into the per-worker assignment records, in the order
// WorkerTaskMapping::send_messages emits them (scheduler/mapping.rs:36-131,259-282).
struct MapKeys {
    uint32_t n_keys;
    const uint32_t *key_rq;        // [n_keys]
    const uint8_t *key_variant;    // [n_keys]
    const uint32_t *key_seg_start; // [n_keys] position of the key's first task in its queue's logical sequence
    const uint32_t *key_ord_off;   // [n_keys+1] into ord_cnt (workers with a count, in Map iteration order)
    const uint32_t *ord_cnt;       // counts in iteration order
    const uint32_t *key_t_off;     // [n_keys+1] into t_sweep
    const uint32_t *t_sweep;       // T_i(s) = sum_j min(c_j, s), s = 0..maxc_i
    // per worker CSR of the keys it takes part in
    const uint32_t *wk_off;        // [W+1]
    const uint32_t *wk_key;        // key index
    const uint32_t *wk_pos;        // position of the worker in that key's iteration order
    // per request: where the selected queue tasks of rq start in sel_*, and the prefilled block of its logical sequence
    const uint32_t *rq_sel_base;   // [Q]
    const uint32_t *rq_pf_start;   // [Q]
    const uint32_t *rq_pf_n;       // [Q]
    // new prefills (process_proactive_filling): per worker CSR of (absolute sel index, count)
    const uint32_t *pfl_off;       // [W+1]
    const uint32_t *pfl_src;
    const uint32_t *pfl_cnt;
    // output placement
    const uint32_t *out_off;       // [W+1]
};
hipError_t expand_mapping(MapKeys mk, uint32_t W, const uint64_t *sel_task, const uint16_t *sel_level, const uint64_t *levels,
                    uint32_t max_items, uint32_t max_count, uint64_t *rec_task, uint8_t *rec_variant, uint8_t *rec_kind, uint32_t *err_flag,
                    hipStream_t s);

}  // namespace hqk
