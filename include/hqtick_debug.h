/*
 * hqtick_debug.h — test hooks exported by libhqtick_test.so ONLY (the product library libhqtick.so exports include/hqtick.h and
 * include/hqwire.h and nothing else; the hooks below are compiled under -DHQTICK_TEST_HOOKS, hyperqueue_amd/build.py::build_test).
 *
 * These expose HOST-side building blocks of the tick so they can be unit-tested on a machine without a GPU.
 * They are not part of the reference's surface and are not a CPU path of the tick: hqtick_run() has no CPU
 * implementation and fails with HQTICK_E_NO_DEVICE when no gfx950 device is present — in either library.
 */
#ifndef HQTICK_DEBUG_H
#define HQTICK_DEBUG_H
#include <stdint.h>
#include "hqtick.h"
#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: what this header declares is its whole dynamic symbol table. */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

/* The exact MILP solver standing where the reference calls HiGHS (solver/highs.rs:51-88).  Maximise obj.x,
 * col_kind 0 = nat (0..), 1 = bool (0..=1); row_type 0 = Min (>=), 1 = Max (<=), 2 = Eq.  Returns 1 when a solution
 * was written (0 == the reference's `None`). */
int hqtick_debug_milp_solve(int ncols, const double *obj, const uint8_t *col_kind, int nrows, const uint8_t *row_type,
                            const double *rhs, const int *row_off, const int *row_col, const double *row_coef,
                            double time_limit_s, int canonical, double *x_out, double *obj_out, int *is_optimal,
                            long *nodes_out);

/* The HOST stages of a tick — create_task_batches + run_scheduling_solver (scheduler/batches.rs:42-181, scheduler/solver.rs:36-483) — on
 * caller-supplied outputs of the GPU scans, so that this logic can be unit-tested on a machine without a GPU.  This is not a tick: the scans
 * (K0/K1/K2), the selection and the mapping have no CPU implementation, and nothing in the product calls this.
 *   vflags[W * NV], vtmc[W * NV]   what K2 (k_worker_eval) writes per (worker, variant slot): bit 0 fits the free resources now, bit 1 fits the
 *                                  total resources, bit 2 the worker's remaining time covers min_time; task_max_count_for_request
 *   levels[L] (descending), hist[L * Q]   what K0/K1/K1b produce: the distinct priorities of the ready set and the task count per (level, rq)
 * Fills status, is_optimal, is_canonical, the batch arrays and the count arrays of `out` (valid until the next call on this thread). */
int hqtick_debug_host_stages(const hqtick_config *config, const hqtick_snapshot *snapshot, const uint8_t *vflags, const uint32_t *vtmc,
                             uint32_t n_levels, const uint64_t *levels, const uint32_t *hist, hqtick_result *out);

/* The host stages of hqtick_query (compute_new_worker_query, scheduler/query.rs:12-131) on caller-supplied scan outputs: as above, plus
 * fake_vflags / fake_vtmc [fake->n_workers * NV] = what K2 writes for the fake workers (free == total).  Fills `out` (valid until the next call
 * on this thread). */
int hqtick_debug_host_query(const hqtick_config *config, const hqtick_snapshot *snapshot, const hqtick_query_workers *fake,
                            const uint8_t *vflags, const uint32_t *vtmc, const uint8_t *fake_vflags, const uint32_t *fake_vtmc,
                            uint32_t n_levels, const uint64_t *levels, const uint32_t *hist, hqtick_query_result *out);

/* 1 if the last hqtick_debug_milp_solve on this thread completed its tie-break phase (Result::canonical, milp.h). */
int hqtick_debug_milp_was_canonical(void);

/* Iteration order of a hashbrown Map<WorkerId,_> built by inserting `keys` (distinct u32) in the given order:
 * out_pos[i] = index into `keys` of the i-th element visited (scheduler/mapping.rs:43). */
void hqtick_debug_map_order_u32(const uint32_t *keys, uint32_t n, uint32_t *out_pos);

/* The per-worker-class block solver of the separable placement (csrc/block_core.h = the algorithm of k_block_solve) with the wavefront
 * emulated on the CPU: the 64 lanes of every lane-parallel step run in a loop.  Arguments = the kernel's tables (hqblock::ColTable /
 * ClassTable / Output) as host arrays: the tick's (batch, variant) columns as a CSR of request entries + weight + resource pool sums, per class
 * free[R] / total[R] / eligibility mask; out: x[n_classes * n_cols], status[n_classes] (0 ok, 1 step budget exhausted, 2 block shape not
 * supported), steps[n_classes]. */
int hqtick_debug_block_solve_host(uint32_t n_cols, uint32_t n_resources, const uint32_t *ent_off, const uint32_t *ent_res, const uint8_t *ent_kind,
                                  const uint64_t *ent_amount, const uint32_t *weight, const double *pool, uint32_t n_classes, const uint64_t *free_,
                                  const uint64_t *total, const uint64_t *elig, uint32_t budget, uint32_t *x, uint32_t *status, uint32_t *steps);
/* on != 0: hqtick_debug_host_stages / _host_query (this thread) solve the class blocks of the separable path with that emulation instead
 * of the host MILP solver — the code path of a GPU tick, minus the hardware.  budget = search steps per class (0 keeps the current one). */
void hqtick_debug_set_block_emulation(int on, uint32_t budget);
/* host wall clock of the separable placement's stages in the last hqtick_debug_host_stages call: worker classes, block solves, counts into Map order (us) */
void hqtick_debug_last_stage_us(double *out3);
/* classes the last hqtick_debug_host_stages call solved through the emulation / with the host solver */
void hqtick_debug_last_blocks(uint32_t *n_emulated, uint32_t *n_host);

/* Measurement hooks of the tools under tools/ (round 3: moved here from the product's header; libhqtick_test.so is the same objects + these).  GPU only. */
/* Re-launches one streaming kernel of the last resident tick `iters` times back to back between two HIP events on the ctx's
 * stream and returns the average launch duration (which: 0 = K1 level_hist, 1 = K4 select_scatter).  which = 2: an EMPTY kernel of K1's grid, every launch
 * bracketed by its own dispatch events as in hqtick_set_kernel_timing — what that measure records for a kernel that does nothing.  GPU only. */
int hqtick_time_kernel(hqtick_ctx *ctx, int which, int iters, double *avg_us);
/* Host wall-clock marks (microseconds since the start of the last tick) at the internal stage boundaries of the last tick;
 * returns the number of marks written (bench tooling). */
int hqtick_timeline(const hqtick_ctx *ctx, double *out, int cap);
/* With HQTICK_BLOCK_PROFILE=1 in the environment at hqtick_create: 8 u64 per class of the last k_block_solve launch — 100 MHz timestamps
 * at start / block built / duals / greedy / phase 1 / phase 2 done, then phase-1 steps and dual-pool size.  NULL when off. */
const uint64_t *hqtick_block_profile_last(const hqtick_ctx *ctx, uint32_t *n_classes);


/* The coupled placement by price sweeps (csrc/price.cpp; k_price_sweep's algorithm in csrc/price_core.h) with the wavefront emulated on the CPU.
 * on != 0: hqtick_debug_host_stages (this thread) hands coupled models of at least min_cols columns (0: the default) to the sweeps — the code path
 * of a GPU tick, minus the hardware. */
void hqtick_debug_set_price_emulation(int on, uint32_t min_cols);
/* fault injection into this thread's emulated sweeps: fail_at = 0: the sweeper refuses the model at begin(); k >= 1: its k-th sweep fails; -1: off.  (A rank of a
 * sharded solve whose device fails must still take part in the exchange the others wait in: tests/test_shard_solve.py) */
void hqtick_debug_set_price_fault(int fail_at);
/* The coupled tick's fast path (csrc/milp.cpp, solve(): a large model with the builder's structure hints goes to the sweeps as it is) for this thread:
 * 1 on, 0 off (every model takes the classic path: presolve, components, scaled row copy), -1 the default. */
void hqtick_debug_set_fast_path(int on);
/* The fast path reads the model builder's structure hints (csrc/milp.h: Model::col_group / row_lhs / row_block / col_ub).  Shared lists and row_block rows are
 * always checked against the rows themselves (a wrong hint sends the model down the classic path); on != 0 makes this thread also recompute every hinted column bound
 * from the rows it stands for.  _mismatches: models refused for that reason since the check was switched on. */
void hqtick_debug_check_model_hints(int on);
int hqtick_debug_model_hint_mismatches(void);
/* sweeps over all blocks / flag configurations of the last hqtick_debug_host_stages call (0: the host search ran alone) */
void hqtick_debug_last_price(uint32_t *sweeps, uint32_t *rounds);
/* hqtick_debug_milp_solve on a model that carries the builder's structure hints (col_group: block of every column, -1 = a column of the whole model;
 * row_implied: rows implied for integer points by their block's other rows; both may be NULL), with the price sweeps run through the emulated
 * wavefront when use_sweeps != 0.  stats_out (optional, 4 doubles): sweeps, rounds, microseconds inside the price solve, canonical flag. */
int hqtick_debug_milp_solve_priced(int ncols, const double *obj, const uint8_t *col_kind, const int32_t *col_group, int nrows, const uint8_t *row_type,
                                   const uint8_t *row_implied, const double *rhs, const int *row_off, const int *row_col, const double *row_coef,
                                   double time_limit_s, int use_sweeps, uint32_t min_cols, double *x_out, double *obj_out, int *is_optimal, double *stats_out);

/* The wire encoding of include/hqwire.h on HOST memory: the same phase functions the three kernels run (csrc/wire_core.h), executed for
 * thread 0..255 in turn with a loop end standing in for each workgroup barrier.  Same arguments as hqwire_encode_device, all pointers host
 * pointers.  Lets the CPU test suite execute the encoder's logic; it is not a product path (hqwire_encode_device has no CPU fallback). */
struct hqwire_tables;
struct hqwire_records;
struct hqwire_output;
int hqwire_debug_encode_host(const struct hqwire_tables *tables, const struct hqwire_records *records, const struct hqwire_output *out);
/* As above with the emulated threads of every phase run in another sequence (0 ascending, 1 descending, 2 a fixed permutation): the bytes
 * must not depend on it -- a phase that did would be a data race on the GPU. */
int hqwire_debug_encode_host_order(const struct hqwire_tables *tables, const struct hqwire_records *records, const struct hqwire_output *out, int order);

/* The guard on the class blocks' answers (csrc/host_model.cpp; hqtick_kernel_stats.n_classes_verified / _mismatch / _rejected) under fault injection: `verify` classes
 * of the launch are re-solved by the host (window starting at tick_seq * verify; UINT32_MAX = all), and the emulated blocks corrupt the answer of launch position
 * corrupt_class: mode 1 = one task short (feasible, not maximal), 2 = does not fit the rows, 3 = everything on ONE column, corrupt_fill = column << 16 | count (a count that
 * exactly fills a row every column needs is feasible and maximal, yet not the optimum), 0 = off. */
void hqtick_debug_set_block_guard(uint32_t verify, uint32_t tick_seq, int corrupt_mode, uint32_t corrupt_class, uint32_t corrupt_fill);
void hqtick_debug_last_block_guard(uint32_t *verified, uint32_t *mismatch, uint32_t *rejected);
/* The table of earlier class-block answers (hqtick.h: HQTICK_FLAG_NO_BLOCK_MEMO) for hqtick_debug_host_stages: on = 1 keeps one per thread from call to call,
 * 0 drops it (the default: every block is solved).  _last_block_memo: host blocks of the last call answered from it. */
void hqtick_debug_set_block_memo(int on);
uint32_t hqtick_debug_last_block_memo(void);
/* hqtick_debug_host_stages as ONE RANK of a sharded scheduler (include/hqtick.h: hqtick_set_exchange): the emulated sweeps / class blocks run over this rank's
 * share and are completed through `fn`; min_blocks / min_classes = the thresholds below which every rank solves the whole model.  fn = NULL: off. */
void hqtick_debug_set_exchange(hqtick_exchange_fn fn, void *user, uint32_t rank, uint32_t world, uint32_t min_blocks, uint32_t min_classes);
uint32_t hqtick_debug_last_exchange_calls(void);
/* one exchange of the sharded solve through the library's RCCL communicator (hqtick_comm_init): recv gets world x bytes_per_rank bytes */
int hqtick_debug_exchange(hqtick_ctx *ctx, const void *send, void *recv, size_t bytes_per_rank);
#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
