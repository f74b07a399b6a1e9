/*
 * hqtick_records.h — walking the records of a tick in any of the three emission forms of include/hqtick.h (header-only, C99; host code, no HIP).
 *
 * The reactor applies a tick's records worker by worker, in order (scheduler/mapping.rs:259-292: prefills with variant None, then the assigned tasks).
 * hqtick_worker_records() hands them to a callback in exactly that order whatever form they arrived in:
 *   full      rec_task / rec_variant / rec_kind                                   10 B per record over PCIe
 *   compact   rec_task_lo + runs of equal (job, variant, kind)        (ABI 4/5)    4 B per record
 *   delta16   16-bit differences of the low halves + runs16           (ABI 6)      2 B per record (6 B where a difference does not fit)
 * so a shim written against this header can switch forms with hqtick_config.flags alone.  Returns the number of records visited, or -1 when the result
 * carries none of the three forms for a worker that has records (a tick into a device record sink: the records are in HBM, include/hqtick.h).
 */
#ifndef HQTICK_RECORDS_H
#define HQTICK_RECORDS_H
#include <stddef.h>
#include <stdint.h>

#include "hqtick.h"

#ifdef __cplusplus
extern "C" {
#endif

/* task = packed TaskId (job_id << 32 | job_task_id); variant 0xFF = None (a prefill); kind = HQ_REC_PREFILL / HQ_REC_ASSIGN */
typedef void (*hqtick_record_fn)(void *user, uint32_t worker_index, uint64_t task, uint8_t variant, uint8_t kind);

static inline int64_t hqtick_worker_records(const hqtick_result *res, uint32_t w, hqtick_record_fn fn, void *user) {
    const uint32_t a = res->rec_off[w], b = res->rec_off[w + 1], tot = b - a;
    if (tot == 0) return 0;
    if (res->rec_task) {  /* full records */
        for (uint32_t i = 0; i < tot; i++) fn(user, w, res->rec_task[a + i], res->rec_variant[a + i], res->rec_kind[a + i]);
        return tot;
    }
    if (!res->run_span) return -1;
    const uint32_t r0 = res->run_span[w].start, nr = res->run_span[w].count;
    if (res->rec_task_lo && res->runs) {  /* compact: u32 low halves */
        for (uint32_t r = 0; r < nr; r++) {
            const hqtick_rec_run run = res->runs[r0 + r];
            const uint32_t end = r + 1 < nr ? res->runs[r0 + r + 1].first : tot;
            for (uint32_t i = run.first; i < end; i++) fn(user, w, ((uint64_t)run.job << 32) | res->rec_task_lo[a + i], (uint8_t)(run.meta & 0xFFu), (uint8_t)(run.meta >> 8));
        }
        return tot;
    }
    if (res->rec_delta16 && res->runs16) {  /* 16-bit differences */
        const uint16_t *u = res->rec_delta16 + (size_t)4 * a;  /* worker w's unit stream */
        for (uint32_t r = 0; r < nr; r++) {
            const hqtick_rec_run16 run = res->runs16[r0 + r];
            const uint32_t end = r + 1 < nr ? res->runs16[r0 + r + 1].first : tot;
            uint32_t lo = run.first_lo;  /* the record that opens a run consumes no unit */
            for (uint32_t i = run.first; i < end; i++) {
                if (i != run.first) {
                    const uint16_t d = *u++;
                    if (d != 0xFFFFu) lo += d;
                    else { lo = (uint32_t)u[0] | ((uint32_t)u[1] << 16); u += 2; }  /* a difference that did not fit: the low half itself */
                }
                fn(user, w, ((uint64_t)run.job << 32) | lo, (uint8_t)(run.meta & 0xFFu), (uint8_t)(run.meta >> 8));
            }
        }
        return tot;
    }
    return -1;
}

/* every worker in index order; stops at the first worker whose records cannot be walked (returns -1), else the total */
static inline int64_t hqtick_all_records(const hqtick_result *res, uint32_t n_workers, hqtick_record_fn fn, void *user) {
    int64_t total = 0;
    for (uint32_t w = 0; w < n_workers; w++) {
        const int64_t n = hqtick_worker_records(res, w, fn, user);
        if (n < 0) return -1;
        total += n;
    }
    return total;
}

#ifdef __cplusplus
}
#endif
#endif
