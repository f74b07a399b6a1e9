/*
 * hqtick_debug.h — test hooks exported by libhqtick.so next to the product ABI (include/hqtick.h).
 *
 * These expose two HOST-side building blocks of the tick so they can be unit-tested on a machine without a GPU.
 * They are not part of the reference's surface and are not a CPU path of the tick: hqtick_run() has no CPU
 * implementation and fails with HQTICK_E_NO_DEVICE when no gfx950 device is present.
 */
#ifndef HQTICK_DEBUG_H
#define HQTICK_DEBUG_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* The exact MILP solver standing where the reference calls HiGHS (solver/highs.rs:51-88).  Maximise obj.x,
 * col_kind 0 = nat (0..), 1 = bool (0..=1); row_type 0 = Min (>=), 1 = Max (<=), 2 = Eq.  Returns 1 when a solution
 * was written (0 == the reference's `None`). */
int hqtick_debug_milp_solve(int ncols, const double *obj, const uint8_t *col_kind, int nrows, const uint8_t *row_type,
                            const double *rhs, const int *row_off, const int *row_col, const double *row_coef,
                            double time_limit_s, int canonical, double *x_out, double *obj_out, int *is_optimal,
                            long *nodes_out);

/* 1 if the last hqtick_debug_milp_solve on this thread completed its tie-break phase (Result::canonical, milp.h). */
int hqtick_debug_milp_was_canonical(void);

/* Iteration order of a hashbrown Map<WorkerId,_> built by inserting `keys` (distinct u32) in the given order:
 * out_pos[i] = index into `keys` of the i-th element visited (scheduler/mapping.rs:43). */
void hqtick_debug_map_order_u32(const uint32_t *keys, uint32_t n, uint32_t *out_pos);

/* Host wall-clock marks (microseconds since the start of the last tick) at the internal stage boundaries of
 * hqtick_run(); bench tooling only.  Returns the number of marks written. */
struct hqtick_ctx;
/* Re-launches one streaming kernel of the last resident tick `iters` times back to back between two HIP events on the ctx's
 * stream and returns the average launch duration (which: 0 = K1 level_hist, 1 = K4 select_scatter).  Amortises the ~2 us of
 * event/dispatch latency that a single bracketed launch inside a tick carries. */
int hqtick_debug_time_kernel(struct hqtick_ctx *ctx, int which, int iters, double *avg_us);
int hqtick_debug_timeline(const struct hqtick_ctx *ctx, double *out, int cap);

#ifdef __cplusplus
}
#endif
#endif
