// Launch wrappers of the gfx950 kernels of the tick (kernels.hip).  Host-callable, plain pointers (device memory).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace hqk {

// Kernel timing: the NEXT launch of one of the measured kernels (K1, K1b, K4, K5a, K5b, k_block_solve) on this thread is bracketed by these two
// events at the dispatch itself (hipExtLaunchKernelGGL: the events take their timestamps from the kernel's own start / completion signal — the
// source rocprofv3's kernel trace reads — not from markers queued around it, which add the event's own dispatch latency).
void time_next_launch(hipEvent_t start, hipEvent_t stop);
struct LaunchTimer { hipEvent_t start = nullptr, stop = nullptr; };
LaunchTimer take_launch_timer();

// Capacity of the distinct-priority hash set (power of two).  More distinct priority levels than
// HQK_PRIO_SET_CAP/2 in one ready set is reported as HQTICK_E_CAPACITY.
static const uint32_t PRIO_SET_CAP = 1u << 15;
static const uint64_t PRIO_EMPTY = 0xFFFFFFFFFFFFFFFFull;  // Priority u64::MAX is representable; it is tracked by a side flag
static const uint32_t MAX_GROUPS_4W = 2048;                // G <= this: 4 waves per block share the LDS
static const uint32_t MAX_GROUPS = 16384;                  // hard cap on G = levels x requests
static const uint32_t MAX_LEVELS = 4096;                   // distinct priority levels per tick (LDS sort)

struct WaveGeom {
    uint32_t tasks_per_wave;  // contiguous ready-set slice a wavefront owns (multiple of 256)
    uint32_t n_waves;         // number of slices
    uint32_t waves_per_block; // 4 or 1
    uint32_t tab_stride;      // row stride of wave_tab (n_waves rounded up to 16 entries)
};
static const uint16_t GKEY_INVALID = 0xFFFFu;
static const uint32_t RQ_TOMBSTONE = 0xFFFFFFFFu;  // rq column value of a task that left the resident ready set

// K0: insert every task priority into the open-addressing set `set` (PRIO_SET_CAP slots, pre-filled with PRIO_EMPTY).
// flags[0] |= 1 when some priority equals PRIO_EMPTY itself, flags[1] = 1 on overflow.
hipError_t distinct_priorities(const uint64_t *prio, const uint32_t *rq, uint64_t n, uint64_t *set, uint32_t *flags, hipStream_t s);  // rq (may be NULL): tasks with rq == RQ_TOMBSTONE do not count
// K0b: compact the set and sort it descending into levels[]; n_levels[0] = L.  Single workgroup.
hipError_t sort_levels(uint64_t *set, uint32_t *flags, uint64_t *levels, uint32_t *n_levels, uint64_t *host_out, uint32_t seq, hipStream_t s);  // set: PRIO_SET_CAP slots + the compact list of MAX_LEVELS values behind them; host_out (may be NULL): pinned, device-mapped, [n_levels u32 | flags0 u32 | flags1 u32 | seq u32 (written last, system-scope release)][levels ...]; clears flags[0..3] AND the set's slots of the listed values (the set is EMPTY again unless it reports more than MAX_LEVELS values)

// K1: per-slice histogram of (level, rq) groups.  wave_tab is [G][tab_stride] (G = L*Q, g = level*Q + rq); also writes
// the per-task group key gkey[i] = g (GKEY_INVALID for a task whose priority / rq is not in the tables) that K4 re-reads
// instead of the 12 B/task priority + rq columns.  levels = descending table in HBM, levels_host = the same on the host (passed in
// the kernel arguments when L <= 4, so the common case needs neither a staging load nor a barrier).
struct WorkerEvalArgs;  // K2 riding along (below)
hipError_t level_hist(const uint64_t *prio, const uint32_t *rq, uint64_t n, const uint64_t *levels, const uint64_t *levels_host, uint32_t L,
                uint32_t Q, WaveGeom geom, uint32_t *wave_tab, uint16_t *gkey, uint32_t *err_flag, const WorkerEvalArgs *ride_along, hipStream_t s,
                const uint32_t *n_levels_dev = nullptr);
// n_levels_dev (device, may be NULL): the launch does not know the level table yet — k_sort_levels is still ahead of it on the stream.  L is then the bound the launch
// is sized for (<= 4), the kernel reads the count and the table from HBM and sets bit 4 of err_flag if there are more levels than that (or none).
// K1b: exclusive scan of every wave_tab row (in place -> offsets) and the row totals into hist[G].
// err_in (device) is forwarded to err_out (may be pinned host memory) by the same launch.
hipError_t empty_like_level_hist(WaveGeom geom, hipStream_t s);  // an empty kernel of K1's grid (calibration of the per-dispatch timing)
hipError_t scan_waves(uint32_t *wave_tab, WaveGeom geom, uint32_t G, uint32_t *hist, uint32_t *err_in, uint32_t *err_out, hipStream_t s,
                      const WorkerEvalArgs *ride_along = nullptr);  // ride_along: K2 as extra workgroups of this launch

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
                 RequestTable rt, uint32_t n_entries, uint8_t *flags, uint32_t *tmc, hipStream_t s);
size_t worker_eval_lds(uint32_t R, uint32_t n_variants, uint32_t n_entries);  // LDS the launch needs (request table + 32 worker rows)
// The same evaluation as extra workgroups of the K1 launch (level_hist's `ride_along`): one launch and one kernel boundary less per tick.
struct WorkerEvalArgs {
    const uint64_t *total, *free_; const int64_t *remaining_ns; uint32_t W, R; RequestTable rt; uint32_t n_entries; uint8_t *flags; uint32_t *tmc;
};

// K4: select the first take[g] tasks (ascending id) of every group and scatter them to sel_task/sel_key at base[g] + rank
// (sel_key = the task's group key; its priority level is key / Q).  wave_off = output of scan_waves; slices whose groups are
// all exhausted exit early.  The selection plan is `take[G] | base[G]`: take_host points at it in host memory (passed in the
// kernel arguments when G <= 64), take_dev at its place inside the HBM copy of the plan.  The same launch copies the whole
// plan (plan_bytes from plan_src, pinned, to plan_dst, HBM) with ride-along workgroups when G <= 64, else with its own launch.
hipError_t select_scatter(const uint64_t *task_id, const uint16_t *gkey, uint64_t n, uint32_t Q, uint32_t G, WaveGeom geom,
                    const uint32_t *wave_off, const uint32_t *take_host, const uint32_t *take_dev, uint64_t *sel_task, uint16_t *sel_key,
                    const void *plan_src, void *plan_dst, size_t plan_bytes, uint32_t *mark_rq, hipStream_t s, uint32_t mark_and_select = 0);  // mark_rq alone: tombstones instead of the selection; with mark_and_select: both
// mark_rq != NULL: consume mode — instead of scattering, every selected task gets rq = RQ_TOMBSTONE in that column.
// A small table from pinned (device-mapped) host memory into HBM by a kernel of the stream (16-byte loads across PCIe): a few microseconds where the copy
// engine's hipMemcpyAsync costs 10-15 on its own.  bytes is rounded up to 16: both buffers must have that slack.
hipError_t copy_pinned_to_hbm(const void *src_device_ptr, void *dst, size_t bytes, hipStream_t s);

// K5: expand per-(request,variant,worker) counts into the per-worker assignment records, in the order
// WorkerTaskMapping::send_messages emits them (scheduler/mapping.rs:36-131,259-282).
// The round-robin of mapping.rs:42-124 hands the task at index  T_k(s) + #{j < p : c_j > s}  of key k's take_tasks()
// vector to the worker at position p (Map iteration order) in sweep s, with T_k(s) = sum_j min(c_j, s).  K5a builds, per
// (key, sweep), the bit row [c_j > s] with per-word prefix popcounts and T_k(s); K5b gathers per worker.
struct MapKeys {
    uint32_t n_keys;
    const uint32_t *key_rq;        // [n_keys]
    const uint8_t *key_variant;    // [n_keys]
    const uint32_t *key_seg_start; // [n_keys] position of the key's first task in its queue's logical sequence
    const uint32_t *key_ord_off;   // [n_keys+1] into ord_cnt (workers with a count, in Map iteration order)
    const uint32_t *ord_cnt;       // counts in iteration order
    const uint32_t *key_t_off;     // [n_keys+1] sweep units: key k owns units [key_t_off[k], key_t_off[k+1]) = sweeps 0..maxc_k
    const uint32_t *key_bits_off;  // [n_keys] first word of key k's bit rows (row s at + s * words_k, words_k = ceil(n_k / 64))
    const uint32_t *key_tr;        // [n_keys] c > 0: every worker of the key takes exactly c tasks and K4 stored the key's tasks WORKER-MAJOR (the task of the worker at
                                   // position j and sweep s at rq_sel_base + j * c + s): K5b reads its c ids contiguously and needs no round-robin cell; 0: queue order
    uint32_t *t_sweep;             // [n_units]  T_k(s)                     (written by K5a)
    uint64_t *bits;                // bit rows                               (written by K5a)
    uint32_t *pre;                 // per-word exclusive prefix popcounts    (written by K5a)
    const uint32_t *wpos;          // [n_keys * W] position of worker w in key k's iteration order, 0xFFFFFFFF = none
    const uint32_t *wcnt;          // [n_keys * W] count of worker w in key k (0 = none)
    // per request: where the selected queue tasks of rq start in sel_*, and the prefilled block of its logical sequence
    const uint32_t *rq_sel_base;   // [Q]
    const uint32_t *rq_pf_start;   // [Q]
    const uint32_t *rq_pf_n;       // [Q]
    // new prefills (process_proactive_filling): per prefilling request pi, worker w takes chunk pfl_j[pi*W+w] (or none)
    uint32_t n_pfq;
    const uint32_t *pfq_src;       // [n_pfq] absolute sel index of chunk 0
    const uint32_t *pfq_size;      // [n_pfq] chunk length
    const uint32_t *pfl_j;         // [n_pfq * W]
    // logical queue positions whose task is Retracting (sorted (rq << 32) | position): no record for them
    uint32_t n_holes;
    const uint64_t *holes;
    // output placement
    const uint32_t *out_off;       // [W+1]
};
hipError_t sweep_bits(MapKeys mk, uint32_t max_count, uint32_t max_workers_per_key, hipStream_t s);
static const uint32_t SWEEP_MAX_WORKERS = 24576;  // workers per key the round-robin kernel stages in LDS
// Compact emission (hqtick.h, HQTICK_FLAG_COMPACT_RECORDS): rec_lo == nullptr switches it off.
struct CompactOut {
    uint32_t *rec_lo;       // [n_rec] low 32 bits of every record's task id, per-worker CSR order (rec_off)
    uint2 *run_span;        // [W] (first run of the worker = its first record's offset, number of runs)    = hqtick_run_span
    uint32_t *runs;         // [n_rec * 3] per run: index of its first record inside the worker's range, high 32 bits of its task ids, variant | kind << 8   = hqtick_rec_run
    uint16_t *units;        // HQTICK_FLAG_COMPACT_DELTA16 (else nullptr): [n_rec * 4] 16-bit unit streams, worker w's at unit 4 * rec_off[w]; rec_lo is unused and
                            // the runs are 4 words (+ the low id of the run's first record) = hqtick_rec_run16
};
hipError_t expand_mapping(MapKeys mk, uint32_t W, const uint64_t *sel_task, const uint16_t *sel_key, uint32_t Q, uint32_t max_items,
                    uint64_t *rec_task, uint8_t *rec_variant, uint8_t *rec_kind, uint32_t *err_flag, CompactOut co, uint32_t max_out, bool may_reorder, hipStream_t s);
// may_reorder: the tick has more than one priority level, Retracting holes or prefilled tasks in the queues — a worker's records then need the stable
// sort of mapping.rs:128-131 (an LDS key array of the next power of two above max_items); without it the items are emitted in gather order
size_t expand_mapping_lds(uint32_t max_items, uint32_t n_keys, uint32_t max_out, bool may_reorder);

// Resident cluster tables (f1): rows of the worker table that changed, scattered into the HBM copy (inputs may sit in pinned host memory)
hipError_t scatter_worker_rows(uint64_t *free_, int64_t *rem, uint32_t R, uint32_t n, const uint32_t *idx, const uint64_t *rows, const int64_t *new_rem, hipStream_t s);

// ... and the whole table re-packed when workers join or leave: new row i = old row src[i], or staged new worker src[i] - W_old
hipError_t repack_worker_rows(const uint64_t *old_total, const uint64_t *old_free, const int64_t *old_rem, uint32_t W_old, uint32_t R, uint32_t W_new, const uint32_t *src,
                              const uint64_t *add_total, const uint64_t *add_free, const int64_t *add_rem, uint64_t *new_total, uint64_t *new_free, int64_t *new_rem, hipStream_t s);

// Resident ready-set deltas: tombstone the given ids (sorted id column, binary search), count live tasks per
// 256-task slice, and rebuild the columns dropping tombstones while merging a sorted batch of new tasks.
hipError_t ready_restore_consumed(const uint16_t *gkey, uint32_t *rq, uint64_t n, uint32_t Q, uint32_t *n_done, hipStream_t s);
hipError_t ready_mark_removed(const uint64_t *ids, uint32_t *rq, uint64_t n, const uint64_t *rm, uint32_t n_rm, uint32_t *n_done, hipStream_t s);
// hqtick_ready_add_packed: the packed batch (pinned, device-mapped memory) -> the id / priority / rq columns of the batch in HBM
hipError_t ready_unpack_adds(uint32_t n, uint32_t n_id_runs, const uint64_t *id_start, const uint32_t *id_first, const uint32_t *id_off, uint32_t n_prio_runs, const uint64_t *prio_value,
                             const uint32_t *prio_first, const uint16_t *rq, uint64_t *aid, uint64_t *aprio, uint32_t *arq, uint64_t last_resident_id, uint32_t *err_flag, hipStream_t s);
// a batch whose ids all sort behind the resident set: appended at the tail of the columns (no rebuild)
hipError_t ready_append(const uint64_t *aid, const uint64_t *aprio, const uint32_t *arq, uint32_t n_add, uint64_t last_resident_id, uint64_t *nid, uint64_t *nprio, uint32_t *nrq,
                        uint32_t *err_flag, hipStream_t s);
hipError_t ready_live_count(const uint32_t *rq, uint64_t n, uint32_t *slice_cnt, hipStream_t s);
hipError_t ready_rebuild(const uint64_t *oid, const uint64_t *oprio, const uint32_t *orq, uint64_t n, uint32_t n_live, const uint32_t *slice_off, const uint64_t *aid,
                   const uint64_t *aprio, const uint32_t *arq, uint32_t n_add, uint64_t *nid, uint64_t *nprio, uint32_t *nrq, uint8_t *pre8, uint32_t *err_flag, hipStream_t s);

// group key and rank-in-group of the wanted ids (0xFFFFFFFF key = not in the set); after scan_waves, same geometry
hipError_t rank_of(const uint64_t *ids, const uint16_t *gkey, uint64_t n, const uint32_t *wave_off, WaveGeom geom, const uint64_t *want, uint32_t n_want,
             uint32_t *out_key, uint32_t *out_rank, hipStream_t s);

// hqtick_upload_ready(sorted = 0): in-place bitonic sort of the three columns by id (buffers sized for n_pow2 elements, the next
// power of two >= n; the padding is filled with sentinels here).  dup_flag |= 1 when two ids are equal.
hipError_t sort_ready(uint64_t *id, uint64_t *prio, uint32_t *rq, uint64_t n, uint64_t n_pow2, uint32_t *dup_flag, hipStream_t s);

}  // namespace hqk
