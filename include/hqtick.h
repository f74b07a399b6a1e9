/*
 * hqtick.h — C ABI of the MI355X-native tako scheduling tick (libhqtick.so).
 *
 * This is the drop-in boundary for ONE hot path of It4innovations/hyperqueue: the server-side
 * scheduling tick of the `tako` crate.  The reference has no FFI seam today; the seam this ABI
 * replaces is (all paths relative to /root/reference/crates/tako/src/internal/):
 *
 *   run_scheduling_inner(core, comm, now)            scheduler/main.rs:50-72
 *     = create_task_batches(core, now, None)         scheduler/batches.rs:42-181
 *     + run_scheduling_solver(core, now, batches,..) scheduler/solver.rs:36-483
 *     + create_task_mapping(core, solution)          scheduler/mapping.rs:23-157
 *     + WorkerTaskMapping::send_messages order       scheduler/mapping.rs:259-292
 *   compute_new_worker_query (stages 1-2 only)       scheduler/query.rs:12-131
 *
 * The Rust host (tako) keeps `Core`, `Comm` and the wire layer; it flattens `Core` into the
 * SoA snapshot below, calls hqtick_run(), and applies the returned assignment vector exactly the way
 * create_task_mapping()/send_messages() would (see INTEGRATION.md for the extern "C" binding).
 *
 * Conventions
 *   - plain C, no torch / HIP types in any signature; all pointers are HOST pointers unless the
 *     function name says `_device`.
 *   - amounts are tako `ResourceAmount`: u64 fixed point, 10 000 fractions per unit,
 *     UINT64_MAX = ResourceAmount::MAX ("unknown / unbounded")   common/resources/amount.rs:7,26-31
 *   - task ids are packed `(job_id << 32) | job_task_id` which preserves `TaskId: Ord`  common/ids.rs:17-21
 *   - priorities are the raw `Priority(u64)`                       common/priority.rs:43-47
 *   - one hqtick_ctx per thread; a ctx owns its device buffers, streams and result storage;
 *     no global state; the library never aborts the host: every failure is a negative return code.
 */
#ifndef HQTICK_H
#define HQTICK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: what this header declares is its whole dynamic symbol table. */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define HQTICK_ABI_VERSION 10u  /* 10: HQTICK_FLAG_NO_TICK_CACHES, hqtick_set_kernel_timing takes any non-zero value but 2 as 1, hqtick_graph_blevel leaves a pending hqtick_ready_consume_last alone; 9: hqtick_graph_blevel / _priorities (extension), hqtick_set_kernel_timing(ctx, 2), a failed CONSUME_IN_TICK tick restores its tasks; 8: HQ_WORKERS_RESIDENT, hqtick_cluster_last_reassigned; 7: hqtick_kernel_stats carries the price-sweep figures of coupled ticks; cluster membership deltas */

/* ResourceAmount::MAX                                    common/resources/amount.rs:31 */
#define HQ_AMOUNT_MAX UINT64_MAX
/* FRACTIONS_PER_UNIT                                     common/resources/amount.rs:7 */
#define HQ_FRACTIONS_PER_UNIT 10000u
/* MAX_TASK_PER_WORKER                                    server/workerload.rs:12 */
#define HQ_MAX_TASK_PER_WORKER 1024u
/* Worker without termination_time                        server/worker.rs:320-326 */
#define HQ_NO_TIME_LIMIT INT64_MAX
/* PriorityCut blocker size == None                       scheduler/batches.rs:15 */
#define HQ_BLOCKER_UNBOUNDED UINT32_MAX
/* "no worker" in records                                 */
#define HQ_NO_WORKER UINT32_MAX

/* AllocationRequest kinds; only Amount-vs-All matters to the tick   common/resources/request.rs:14-21 */
enum { HQ_ENTRY_AMOUNT = 0, HQ_ENTRY_ALL = 1 };

/* worker_flags bits */
enum {
    HQ_WORKER_SN = 1u,       /* WorkerAssignment::Sn (not holding a multi-node task)  server/worker.rs:48-51 */
    HQ_WORKER_STOPPING = 2u  /* stop_reason.is_some()                                 server/worker.rs:312-314 */
};

/* hqtick_run return codes == SchedulerResult              scheduler/main.rs:44-48 */
enum { HQTICK_DONE = 0, HQTICK_NEED_MORE_COMPUTE = 1, HQTICK_NO_PROGRESS = 2 };

/* error codes (negative) */
enum {
    HQTICK_E_INVALID = -1,      /* malformed snapshot (NULL pointer, unsorted ids, bad index)         */
    HQTICK_E_NO_DEVICE = -2,    /* HIP runtime / gfx950 device / kernels not available: NO CPU fallback */
    HQTICK_E_DEVICE = -3,       /* a HIP call failed; hqtick_last_error() has the text                */
    HQTICK_E_CAPACITY = -4,     /* a documented capacity limit was exceeded (levels, groups, ...)     */
    HQTICK_E_QUEUE_UNDERFLOW = -5, /* solver asked for more tasks than a queue holds: the reference
                                      panics here (taskqueue.rs:327 unwrap)                          */
    HQTICK_E_UNSUPPORTED = -6
};

/* assignment record kinds, in the order WorkerTaskMapping::send_messages emits them per worker
 * (prefills first with variant None, then assigned)          scheduler/mapping.rs:268-279 */
enum { HQ_REC_PREFILL = 0, HQ_REC_ASSIGN = 1 };

/* hqtick_config.flags: skip the HIP events that feed hqtick_kernel_stats_last() (saves ~6 API calls per tick) */
#define HQTICK_FLAG_NO_KERNEL_TIMING 1u
/* hqtick_config.flags: compact record emission.  The records of a tick cross PCIe as 4 bytes each (rec_task_lo = job_task_id, the low half of the
 * packed task id) plus one 10-byte RUN per maximal stretch of a worker's records that share (job_id, variant, kind) — instead of 10 bytes per
 * record.  result.rec_task / rec_variant / rec_kind are NULL then; see the run_* fields of hqtick_result.  (Ignored while a device record sink is
 * set: the sink keeps its own layout.) */
#define HQTICK_FLAG_COMPACT_RECORDS 2u
/* hqtick_config.flags, on top of COMPACT_RECORDS (ABI 6): the low halves travel as 16-bit DIFFERENCES.  The records of one (worker, request) come in
 * ascending id order, a few thousand ids apart on a large cluster, so a record that does not open a run is one 16-bit unit — its job_task_id minus its
 * predecessor's — or three units (0xFFFF, low 16 bits, high 16 bits of its job_task_id) when that difference is negative or above 0xFFFE; the record
 * that opens a run has its job_task_id in the run record (hqtick_rec_run16.first_lo).  2 bytes per record instead of 4: the mapping kernel's launch is
 * bound by these bytes crossing PCIe (C3: 25.6 -> ~20 us).  result.rec_task_lo / runs are NULL then; see rec_delta16 / runs16. */
#define HQTICK_FLAG_COMPACT_DELTA16 4u
/* hqtick_config.flags (ABI 8): do not keep class-block answers from tick to tick.  By default a context remembers the last 64 class blocks its HOST solver
 * answered with a certified canonical optimum, keyed by the block's whole model (every coefficient, byte for byte), and answers an identical block from that
 * table: between two ticks of a steady cluster the blocks rarely change while the code that solves them has gone cold.  The answer is the one the solver would
 * give again; kernel_stats.n_classes_memo counts the hits.  bench.py's headline sets this flag (nothing cached inside its timed region). */
#define HQTICK_FLAG_NO_BLOCK_MEMO 8u
/* hqtick_config.flags (ABI 8): hqtick_run_resident takes what it hands out out of the resident ready set ITSELF — as take_tasks / take_tasks_for_prefill / take_one do
 * inside the reference's tick (scheduler/taskqueue.rs:304-373): the selection kernel writes the tombstones while it selects, one kernel and one call per tick less
 * than hqtick_run_resident + hqtick_ready_consume_last (which is a no-op then).  A tick that FAILS after its selection was launched (a record sink too small, a
 * capacity exceeded, a failed exchange) puts back what it took — the group keys its scan left behind tell which tombstones are its own — and the set is as it
 * was before the call; only if that restore itself fails (a device error) does the context drop the resident set (hqtick_last_error says which happened) and the
 * host uploads it again.  Without the flag a tick changes nothing until hqtick_ready_consume_last says so: the two-call form stays the default and is what a
 * drop-in should start from; the flag is for a host whose ticks are short enough for one kernel and one call to matter (bench.py's add -> tick loops). */
#define HQTICK_FLAG_CONSUME_IN_TICK 16u
/* hqtick_config.flags (ABI 9): stop the coupled model's solve where the reference's solver stops — at the certificate.  HiGHS under solve_bounded returns as soon as
 * its incumbent is within mip_rel_gap = 1e-4 of the dual bound (solver/highs.rs:65-88); this library by default goes on to the EXACT optimum and then to the canonical
 * one among the tied optima (result.is_canonical = 1: the answer is a function of the snapshot alone, what parity tier T1 and replicas of a sharded scheduler compare
 * byte for byte).  That proof is most of a small coupled tick (an 80-column model of a few dozen ready tasks: certified after 0.1 ms, canonical after 0.8 ms; a
 * 128-column one: 0.3 ms against 26 ms).  With the flag the tick returns the certified point: is_optimal = 1, is_canonical = 0, the objective within 1e-4 of the
 * optimum exactly as the reference's — for a single scheduler whose ticks must be short.  (Ticks the class blocks or the price sweeps settle are not affected:
 * separable placements stay exact and canonical, swept ones were certificates already.)  IGNORED (ABI 10) on a context that is one of several replicas — after
 * hqtick_set_shard with more than one shard, with a record sink, an exchange callback or a communicator set — whose answers are compared and merged byte for byte. */
#define HQTICK_FLAG_CERTIFICATE_ONLY 32u
/* hqtick_config.flags (ABI 10): nothing DERIVED survives from one tick to the next.  By default a context keeps, next to the resident inputs (ready set, cluster
 * tables), three things it worked out in earlier ticks: the table of host-solved class blocks (NO_BLOCK_MEMO above), the priority-level table of the resident ready
 * set (re-validated by the scan kernel of every tick, rebuilt when a new priority shows up) and the hashbrown iteration orders of the worker-id lists it has seen
 * (Map<WorkerId, _> of scheduler/mapping.rs:36-43, memoised on the whole id list).  All three are pure functions of the tick's inputs, so the results do not change;
 * with this flag every tick works them out again.  For measurements: a loop that repeats one identical tick would otherwise hit all three on every iteration
 * (bench.py's headline sets the flag and says so in its line).  Implies HQTICK_FLAG_NO_BLOCK_MEMO. */
#define HQTICK_FLAG_NO_TICK_CACHES 64u

/* redirect_kind of a result entry (scheduler/mapping.rs:66-101):
 *   FROM_PREFILL  the task sat in a prefill set: Prefilled{old} -> Retracting{old}, retract sent to `old`, redirects.insert(task, (worker, v))
 *   RETARGET      the task was already Retracting{old} in its queue and goes to another worker: redirects.insert(task, (worker, v)); a
 *                 previous target, if any, gets its resources back (remove_sn_task)
 *   SAME_WORKER   the task was Retracting{old} and the tick put it on `old` itself: insert_sn_task(old) only, the redirect table is untouched
 */
enum { HQ_REDIRECT_FROM_PREFILL = 0, HQ_REDIRECT_RETARGET = 1, HQ_REDIRECT_SAME_WORKER = 2 };

/* SchedulerConfig                                           scheduler/state.rs:5-27 */
typedef struct hqtick_config {
    uint32_t abi_version;               /* HQTICK_ABI_VERSION */
    uint32_t proactive_filling_reserve; /* default 16 */
    uint32_t proactive_filling_max;     /* default 40 */
    double mip_time_limit_s;            /* default 5.0 (60.0 under the reference's cfg(test)) */
    int32_t device_index;               /* HIP device ordinal of this ctx */
    uint32_t flags;                     /* HQTICK_FLAG_* */
} hqtick_config;

/*
 * Snapshot of `Core` as the tick reads it.  Caller-owned, read-only during the call.
 *
 * Workers (W): ALL workers of core.worker_map, sorted by ascending WorkerId.  Per-worker resource
 * vectors are padded with zeros to R = n_resources (server/workerload.rs:25-31: missing => 0).
 * Ready set (N): every task sitting in a TaskQueue's `queue` (not its `prefill` set), as SoA,
 * sorted by ascending task id (the order TaskIds are minted in; hqtick_upload_ready() can sort
 * an unsorted set on the device once, outside the tick).
 */
typedef struct hqtick_snapshot {
    /* --- resources ------------------------------------------------------------------------- */
    uint32_t n_resources; /* R = GlobalResourceMapping::n_resources()   common/resources/map.rs:76 */

    /* --- workers --------------------------------------------------------------------------- */
    uint32_t n_workers;                 /* W */
    const uint32_t *worker_id;          /* [W] ascending                                         */
    const uint64_t *worker_total;       /* [W*R] row-major, Worker::resources    server/worker.rs:69 */
    const uint64_t *worker_free;        /* [W*R] SingleNodeTaskAssignment::free_resources   :44  */
    const int64_t *worker_remaining_ns; /* [W] termination_time - now (may be negative) or HQ_NO_TIME_LIMIT */
    const float *worker_min_utilization; /* [W] configuration.min_utilization                   */
    const uint8_t *worker_flags;        /* [W] HQ_WORKER_* bits                                  */
    const uint32_t *worker_group;       /* [W] dense group index in [0, n_groups)  (configuration.group) */
    uint32_t n_groups;
    const uint32_t *worker_map_rank;    /* [W] position of the worker in core.worker_map iteration
                                           (hashbrown order; the Rust shim reads it off `values()`),
                                           or NULL => emulate a map built by inserting ascending ids */

    /* Worker::blocked_requests as (worker index, rq, variant) triples   server/worker.rs:70 */
    uint32_t n_blocked;
    const uint32_t *blocked_worker; /* index into worker arrays */
    const uint32_t *blocked_rq;
    const uint8_t *blocked_variant;

    /* SingleNodeTaskAssignment::assigned_tasks as CSR of (rq, variant) per worker — what
     * GapCache::get_gap subtracts (scheduler/solver.rs:296-303) and what Worker::is_free tests */
    const uint32_t *assigned_off; /* [W+1] */
    const uint32_t *assigned_rq;
    const uint8_t *assigned_variant;

    /* SingleNodeTaskAssignment::prefilled_tasks as CSR of rq per worker (mapping.rs:199-205) */
    const uint32_t *prefilled_off; /* [W+1] */
    const uint32_t *prefilled_rq;

    /* --- resource requests: ResourceRqMap as CSR rq -> variants -> entries ------------------ */
    uint32_t n_requests;              /* Q (= number of TaskQueues, taskqueue.rs:32-35)       */
    const uint32_t *rq_variant_off;   /* [Q+1] into variant_* arrays                           */
    const uint32_t *variant_entry_off; /* [NV+1] into entry_* arrays; entries sorted by resource id */
    const uint32_t *variant_n_nodes;  /* [NV] 0 = single node                                  */
    const uint64_t *variant_min_time_ns; /* [NV]                                                */
    const uint32_t *variant_weight;   /* [NV] ResourceWeight(u32), 10 000 = 1.0                */
    const uint32_t *entry_resource;   /* [NE] */
    const uint8_t *entry_kind;        /* [NE] HQ_ENTRY_* */
    const uint64_t *entry_amount;     /* [NE] (ignored for HQ_ENTRY_ALL) */

    /* --- ready set SoA (TaskQueue::queue of every rq, flattened) ----------------------------- */
    uint64_t n_ready;            /* N */
    const uint64_t *task_id;     /* [N] ascending */
    const uint64_t *task_priority; /* [N] */
    const uint32_t *task_rq;     /* [N] */

    /* --- TaskQueue::prefill of every rq (taskqueue.rs:118) ------------------------------------ */
    const uint32_t *prefill_off;     /* [Q+1] */
    const uint64_t *prefill_priority; /* [Q] valid when the rq's prefill set is non-empty        */
    const uint64_t *prefill_task;    /* ids in the Set<TaskId>'s iteration order                  */
    const uint32_t *prefill_worker;  /* worker INDEX holding the prefilled task                   */

    /* --- ready tasks in state Retracting{worker}: put back into their queue when a higher-priority arrival dissolved the
     * prefill set (check_dispose_prefill, scheduler/taskqueue.rs:148-154).  They are ordinary members of the ready set (they appear in
     * the task_* columns / the resident set), but create_task_mapping treats them differently (scheduler/mapping.rs:66-80): no
     * `assigned` record, a redirect instead.  Ascending task id. */
    uint32_t n_retracting;
    const uint64_t *retracting_task;
    const uint32_t *retracting_worker;           /* worker INDEX the task is being retracted from                         */
    const uint32_t *retracting_redirect_worker;  /* current scheduler_state.redirects target (worker INDEX) or HQ_NO_WORKER */
    const uint8_t *retracting_redirect_variant;  /* its variant (ignored when there is no redirect)                       */
} hqtick_snapshot;

/* What-if query input: fake workers appended after the real ones (scheduler/query.rs:20-56).
 * Same layout as the worker arrays above; ids must be above every real id. */
typedef struct hqtick_query_workers {
    uint32_t n_workers;
    const uint32_t *worker_id;
    const uint64_t *worker_total; /* [n*R]; free == total for a fresh fake worker */
    const int64_t *worker_remaining_ns;
    const float *worker_min_utilization;
} hqtick_query_workers;

/* Compact emission (HQTICK_FLAG_COMPACT_RECORDS): one run = a maximal stretch of a worker's records that share (job id, variant, kind) */
typedef struct hqtick_rec_run {
    uint32_t first; /* index of the run's first record inside its worker's range */
    uint32_t job;   /* high 32 bits of the task ids of the run (the job id of HyperQueue's TaskId packing) */
    uint32_t meta;  /* variant | kind << 8 (low 16 bits; the rest is zero) */
} hqtick_rec_run;
typedef struct hqtick_run_span { uint32_t start, count; } hqtick_run_span;
/* HQTICK_FLAG_COMPACT_DELTA16: the run record also carries the low half of its first record's task id */
typedef struct hqtick_rec_run16 { uint32_t first, job, meta, first_lo; } hqtick_rec_run16;

/*
 * Result view.  All pointers are owned by the ctx and stay valid until the next call on it.
 */
typedef struct hqtick_result {
    int32_t status;     /* HQTICK_DONE / NEED_MORE_COMPUTE / NO_PROGRESS */
    uint8_t is_optimal; /* SchedulingSolution::is_optimal  scheduler/solver.rs:14-16 — with the reference's meaning: solve_bounded sets only
                           `time_limit` (solver/highs.rs:65-68), so HiGHS reports Optimal once the incumbent is within its default
                           mip_rel_gap = 1e-4 of the dual bound.  1 here = certified within that same 1e-4 (or proven exact). */
    uint8_t is_canonical; /* 1: the placement is the EXACT optimum and, among the optimal placements, the canonical one — the answer
                             is a function of the snapshot alone.  0: certified-optimal (or an incumbent) but the exact pass / the tie-break phase
                             was skipped or ran out of its work budget: only the objective value can be compared with another solver's answer.
                             (ABI 3; uses former padding.) */

    /* TaskBatch list (scheduler/batches.rs:18-27) — exposed so parity tier T1 is testable */
    uint32_t n_batches;
    const uint32_t *batch_rq;
    const uint32_t *batch_size;
    const uint32_t *batch_limit;
    const uint8_t *batch_limit_reached;
    const uint8_t *batch_is_blocker;
    const uint32_t *batch_cut_off;     /* [n_batches+1] */
    const uint32_t *cut_size;          /* [n_cuts] */
    const uint32_t *cut_blocker_off;   /* [n_cuts+1] */
    const uint32_t *blocker_rq;
    const uint32_t *blocker_size;      /* HQ_BLOCKER_UNBOUNDED = None */

    /* SchedulingSolution::sn_counts (solver.rs:12) flattened in the reference's iteration order:
     * outer = sn_counts.into_iter() order, inner = counts.iter() order (mapping.rs:36,43) */
    uint32_t n_counts;
    const uint32_t *count_rq;
    const uint8_t *count_variant;
    const uint32_t *count_worker; /* worker INDEX */
    const uint32_t *count_value;

    /* WorkerTaskMapping (mapping.rs:11-21) as CSR over worker index.  Per worker the records are in
     * the exact order send_messages() walks them: prefills, then assigned sorted by priority desc
     * (stable).  Retracts are a separate CSR (they are sent first, as one RetractTasks message). */
    const uint32_t *rec_off;    /* [W+1] */
    const uint64_t *rec_task;
    const uint8_t *rec_variant; /* 0xFF for prefills (variant None) */
    const uint8_t *rec_kind;    /* HQ_REC_* */
    const uint32_t *retract_off; /* [W+1] */
    const uint64_t *retract_task;
    /* Compact emission (HQTICK_FLAG_COMPACT_RECORDS; NULL otherwise).  Worker w's records are rec_off[w] .. rec_off[w + 1] as before; record i of
     * that range has task id (run.job << 32) | rec_task_lo[rec_off[w] + i], variant run.meta & 0xFF (0xFF = None: a prefill) and kind run.meta >> 8,
     * where `run` is the one of runs[run_span[w].start .. + run_span[w].count) with the largest run.first <= i (runs ascend by `first`; the first one
     * starts at 0).  run_span is defined only for workers with records; a worker's runs sit in the run slots of its own records (run_span[w].start ==
     * rec_off[w]: at most one run per record, so no allocation is needed on the device).  (ABI 5: the runs are 12-byte records and the span one 8-byte pair — every
     * store of the mapping kernel crosses PCIe as a write of its own, and three arrays of 4 + 4 + 2 bytes per run cost a third of the launch.) */
    const uint32_t *rec_task_lo;        /* [n_records] */
    const struct hqtick_run_span *run_span; /* [W] */
    const struct hqtick_rec_run *runs;
    /* HQTICK_FLAG_COMPACT_DELTA16 (ABI 6; NULL otherwise, and rec_task_lo / runs are NULL then).  run_span as above, indexing runs16.  Worker w's unit
     * stream starts at rec_delta16[4 * rec_off[w]] (at most 3 units per record; the slack keeps every stream 8-byte aligned).  Decoding the records of w
     * in order: a record that opens a run (i == run.first) has job_task_id = run.first_lo and consumes no unit; any other record reads one unit u:
     * u != 0xFFFF -> job_task_id = previous job_task_id + u;  u == 0xFFFF -> job_task_id = next unit | next-but-one unit << 16 (three units consumed). */
    const uint16_t *rec_delta16;
    const struct hqtick_rec_run16 *runs16;

    /* scheduler_state.redirects insertions made by this tick (mapping.rs:78-100) */
    uint32_t n_redirects;
    const uint64_t *redirect_task;
    const uint32_t *redirect_worker; /* worker INDEX of the new target */
    const uint8_t *redirect_variant;
    const uint8_t *redirect_kind;    /* HQ_REDIRECT_* */

    /* multi-node placements (solver.rs:442-464, mapping.rs:133-154): task + its workers, root first */
    uint32_t n_mn;
    const uint64_t *mn_task;
    const uint32_t *mn_worker_off; /* [n_mn+1] */
    const uint32_t *mn_worker;     /* worker INDEX */

    /* free resources of every worker after the tick (Worker::insert_sn_task, server/worker.rs:188-196) */
    const uint64_t *new_free; /* [W*R] */

    /* timing of the last call, microseconds (host wall clock around each stage) */
    double t_total_us, t_scan_us, t_batches_us, t_solve_us, t_mapping_us;
} hqtick_result;

/* compute_new_worker_query result: for every fake worker, 1 if it received any count
 * (query.rs:73-95), plus the batches for inspection */
typedef struct hqtick_query_result {
    uint32_t n_workers;
    const uint8_t *is_loaded; /* [n_workers] */
    uint8_t is_optimal;
} hqtick_query_result;

typedef struct hqtick_ctx hqtick_ctx;

/* Create a context bound to one HIP device.  Fails with HQTICK_E_NO_DEVICE when no gfx950
 * device / runtime is usable — there is deliberately no CPU implementation behind this ABI. */
int hqtick_create(const hqtick_config *config, hqtick_ctx **out_ctx);
void hqtick_destroy(hqtick_ctx *ctx);

/* One scheduling tick == run_scheduling_inner().  Copies the snapshot's columns to the device,
 * runs the tick, fills `out`.  Returns HQTICK_DONE/NEED_MORE_COMPUTE/NO_PROGRESS or a negative error. */
int hqtick_run(hqtick_ctx *ctx, const hqtick_snapshot *snapshot, hqtick_result *out);

/* Device-resident variant: keep the ready-set columns in HBM across ticks.
 * hqtick_upload_ready() copies (and, when `sorted`==0, sorts by task id on the device: bitonic, once per upload) the ready
 * set into ctx-owned HBM buffers; hqtick_run_resident() then runs ticks against them, taking every
 * other field (workers, requests, prefill sets) from `snapshot` and ignoring its task_* pointers. */
int hqtick_upload_ready(hqtick_ctx *ctx, uint64_t n_ready, const uint64_t *task_id,
                        const uint64_t *task_priority, const uint32_t *task_rq, int sorted);
int hqtick_run_resident(hqtick_ctx *ctx, const hqtick_snapshot *snapshot, hqtick_result *out);

/*
 * Resident ready-set deltas: what the reactor's event handlers do to TaskQueues between two ticks, applied to
 * the HBM-resident columns instead of re-uploading them.
 *   hqtick_ready_consume_last  every task the last hqtick_run_resident handed out (assigned, newly prefilled, multi-node)
 *                              leaves the set — take_tasks / take_tasks_for_prefill / take_one   scheduler/taskqueue.rs:304-373
 *   hqtick_ready_remove        ids (any order) leave the set — TaskQueue::remove on cancel        scheduler/taskqueue.rs:146-217
 *                              returns how many of them were present
 *   hqtick_ready_add           new ready tasks, ids strictly ascending — TaskQueues::add_ready_task scheduler/taskqueue.rs:37-43
 *   hqtick_ready_add_stage /   the same without a copy on the host (ABI 6): _stage hands out the three columns of the library's pinned staging buffer for
 *   hqtick_ready_add_staged    n tasks (valid until the next ready_add* call), the reactor writes the new tasks there, _staged(n' <= n) merges the first n'
 *   hqtick_ready_compact       drop the tombstones now (done automatically when they outnumber the live tasks, and by every add that merges)
 *   hqtick_ready_count         live tasks in the set
 * Removal writes a tombstone into the rq column (4 B per task).  An add whose first id lies behind every resident id (freshly minted ids: the usual batch) and
 * that finds room behind the columns is APPENDED there by one kernel, which also validates it (ABI 8; hqtick_kernel_stats.ready_appends counts); any other add, and
 * compact, stream the columns once (20 B read + 20 B written per live task) through a merge kernel and leave room for as many tasks again.  A refused batch
 * (ids not ascending, an id already resident, a reserved request id) leaves the set as it was.  Request id 0xFFFFFFFF is reserved for the tombstone.
 */
int hqtick_ready_consume_last(hqtick_ctx *ctx);
int hqtick_ready_remove(hqtick_ctx *ctx, uint64_t n, const uint64_t *task_id);
int hqtick_ready_add(hqtick_ctx *ctx, uint64_t n, const uint64_t *task_id, const uint64_t *task_priority, const uint32_t *task_rq);
int hqtick_ready_add_stage(hqtick_ctx *ctx, uint64_t n, uint64_t **task_id, uint64_t **task_priority, uint32_t **task_rq);
/* The same delta in the form a batch of newly ready tasks has anyway (ABI 8): 2-6 bytes per task over PCIe instead of 20 — at 188 k arrivals per tick the
 * transfer is what hqtick_ready_add costs (3.8 MB: ~70 us of a 134 us call).
 *   ids         n_id_runs runs of ascending ids: run r starts at id_run_start[r] and holds id_run_len[r] tasks; id_off[j] (u32, per task, in batch order) is the task's
 *               offset from its run's start — or id_off = NULL: the ids of a run are consecutive (a submit's array, a finished wave of one job).  The batch as a
 *               whole must ascend strictly, like hqtick_ready_add's.
 *   priorities  n_prio_runs runs over the same batch order: prio_run_value[r] for the next prio_run_len[r] tasks
 *   requests    task_rq u16 per task (request ids >= 65535: use hqtick_ready_add)
 * Expanded on the device into the columns hqtick_ready_add would have sent; everything else (merge, validation, errors) is hqtick_ready_add's. */
int hqtick_ready_add_packed(hqtick_ctx *ctx, uint64_t n, uint32_t n_id_runs, const uint64_t *id_run_start, const uint32_t *id_run_len, const uint32_t *id_off,
                            uint32_t n_prio_runs, const uint64_t *prio_run_value, const uint32_t *prio_run_len, const uint16_t *task_rq);
int hqtick_ready_add_staged(hqtick_ctx *ctx, uint64_t n);
int hqtick_ready_compact(hqtick_ctx *ctx);
uint64_t hqtick_ready_count(const hqtick_ctx *ctx);

/*
 * Cluster tables resident in HBM (ABI 5): the worker rows (total, free, remaining lifetime) and the request tables K2 reads every
 * tick stay on the device, and the reactor sends what its event handlers change instead of the library re-packing W rows per tick:
 *   hqtick_cluster_upload          the snapshot's worker_total / worker_free / worker_remaining_ns and request tables -> HBM.  Again whenever
 *                                  workers join or leave (on_new_worker / on_remove_worker, server/reactor.rs:20-186: the row count changes).
 *   hqtick_cluster_update_workers  n rows whose free resources (and, with remaining_ns != NULL, remaining lifetime) changed since the last call —
 *                                  tasks started (the previous tick's assignments, Worker::insert_sn_task) or finished (on_task_finished),
 *                                  time limits running down.  worker_index[n] are row numbers of the uploaded set, free_rows[n * n_resources].
 *                                  Applied in stream order before the next tick; returns without synchronising.
 *   hqtick_cluster_drop            back to per-tick packing.
 * While the tables are resident, hqtick_run / hqtick_run_resident read the worker rows from HBM (the snapshot's worker arrays still feed the host-side
 * model build and MUST hold the same values: the caller owns that invariant; HQTICK_CHECK_CLUSTER=1 in the environment makes every tick verify it and
 * fail with HQTICK_E_INVALID).  New request classes in a snapshot are detected and re-sent by the tick itself (a few hundred bytes).  A snapshot with
 * another n_workers / n_resources than the uploaded set is refused (HQTICK_E_INVALID).
 */
int hqtick_cluster_upload(hqtick_ctx *ctx, const hqtick_snapshot *snapshot);
int hqtick_cluster_update_workers(hqtick_ctx *ctx, uint32_t n, const uint32_t *worker_index, const uint64_t *free_rows, const int64_t *remaining_ns);
int hqtick_cluster_drop(hqtick_ctx *ctx);
/*
 * Cluster MEMBERSHIP and blocked requests as deltas (ABI 7): the worker set itself lives in the library between ticks — rows in HBM for the scans, a
 * mirror in host memory for the host stages — and the reactor forwards what its handlers do:
 *   hqtick_cluster_add_workers     on_new_worker  server/reactor.rs:20-32 (Core::new_worker): n workers with ids ascending and above every id present
 *                                  (WorkerIds are minted ascending), their total / free rows [n * R], and (NULL = the defaults of a fresh single-node
 *                                  worker) remaining lifetime, min_utilization, HQ_WORKER_* flags, group.  They take the row indices W .. W + n - 1.
 *   hqtick_cluster_remove_workers  on_remove_worker  server/reactor.rs:64-186: the rows of these ids leave; later rows move up (the row index of a
 *                                  worker is its position in ascending id order, as in a snapshot).  Unknown ids are an error (HQTICK_E_INVALID).
 *   hqtick_cluster_set_blocked     Worker::blocked_requests  server/worker.rs:70 of ONE worker, replaced by the n given (rq, variant) pairs:
 *                                  task_reject adds a pair (reactor.rs:365-445), EnableRequest removes it (request_enabled, reactor.rs:447-460),
 *                                  n = 0 clears.
 *   hqtick_cluster_workers         the current ids in row order (valid until the next membership call).
 * One re-pack kernel per membership call (rows read from HBM, new rows from pinned staging); nothing is re-uploaded.
 * A tick whose snapshot has worker_id == NULL takes the WHOLE worker side from the library: ids, total / free rows (as kept current by
 * hqtick_cluster_update_workers), remaining lifetime, min_utilization, flags, groups and the blocked pairs.  Say so with n_workers = HQ_WORKERS_RESIDENT: such a
 * tick FAILS (HQTICK_E_INVALID) when the library holds no worker set (never uploaded, or hqtick_cluster_drop) instead of running as a legitimate tick of
 * zero workers, which is what n_workers = 0 with NULL arrays means without a resident set (with one, 0 and the current count are accepted as before).
 * worker_map_rank is emulated, and the per-worker CSRs assigned_off / prefilled_off (if given) must have current-count + 1 entries.  The reactor then
 * no longer flattens W x R worker arrays per tick.
 */
int hqtick_cluster_add_workers(hqtick_ctx *ctx, uint32_t n, const uint32_t *worker_id, const uint64_t *total_rows, const uint64_t *free_rows,
                               const int64_t *remaining_ns, const float *min_utilization, const uint8_t *flags, const uint32_t *group);
int hqtick_cluster_remove_workers(hqtick_ctx *ctx, uint32_t n, const uint32_t *worker_id);
int hqtick_cluster_set_blocked(hqtick_ctx *ctx, uint32_t worker_id, uint32_t n, const uint32_t *rq, const uint8_t *variant);
int hqtick_cluster_workers(const hqtick_ctx *ctx, uint32_t *n_workers, const uint32_t **worker_id);
/*
 * Retracting tasks as deltas (ABI 7; on_retract_response, process_retracted — server/reactor.rs:34-62,462-508).  The table of tasks in state
 * Retracting{worker} — with their scheduler_state.redirects entry — can live in the library instead of travelling in every snapshot:
 *   hqtick_retracting_add      process_retracted outside a tick: a higher-priority arrival dissolved a prefill set (check_dispose_prefill,
 *                              scheduler/taskqueue.rs:148-154) and these tasks went back into their queue as Retracting{worker}.  worker = worker id.
 *   (the tick itself)          a tick run with snapshot->n_retracting == HQ_RETRACTING_RESIDENT takes the in-queue entries of the table as its
 *                              retracting_* arrays (worker ids -> row indices of the resident worker set: needs hqtick_cluster_upload) and, when it
 *                              succeeds, applies to the table what create_task_mapping does to task states and redirects (scheduler/mapping.rs:66-101):
 *                              a prefilled task it hands to another worker enters as Retracting{old} with a redirect, a Retracting task it takes gets its
 *                              redirect (re)targeted, or none when it lands on the worker it is retracting from.
 *   hqtick_retract_response    on_retract_response for one worker: every listed task that is Retracting{that worker} leaves the table — with a redirect
 *                              it is now Assigned to the target (reported in the three output arrays, valid until the next call: the host sends the
 *                              ComputeTasks message), without one it is an ordinary Waiting task of its queue again.  Other ids are ignored, as in the
 *                              reference ("retracted task in invalid state").  Returns the number of tasks that left the table.
 *   hqtick_retracting_count    entries in the table.
 * hqtick_cluster_remove_workers applies on_remove_worker (server/reactor.rs:86-147) to the table: an entry whose OLD worker is removed leaves it — with a
 * redirect to a worker that stays the task is Assigned{target} from now on and the host owes the target its ComputeTasks message: those (task, target id,
 * variant) triples are read with hqtick_cluster_last_reassigned (valid until the next hqtick_cluster_remove_workers / hqtick_retract_response call); without
 * one it is an ordinary Waiting task of its queue.  An entry whose redirect TARGET is removed loses the redirect and is in its queue again, still
 * Retracting{old} (redirects.remove + add_ready_task): the host re-adds the task to the resident ready set (hqtick_ready_add) with the lost worker's other tasks.
 */
#define HQ_RETRACTING_RESIDENT 0xFFFFFFFFu
/* hqtick_snapshot.n_workers of a tick whose worker side is the library's resident worker set (worker_id == NULL) */
#define HQ_WORKERS_RESIDENT 0xFFFFFFFFu
int hqtick_retracting_add(hqtick_ctx *ctx, uint32_t n, const uint64_t *task_id, const uint32_t *worker_id);
int hqtick_retract_response(hqtick_ctx *ctx, uint32_t worker_id, uint32_t n, const uint64_t *task_id, uint32_t *n_assigned, const uint64_t **assigned_task,
                            const uint32_t **assigned_worker_id, const uint8_t **assigned_variant);
uint32_t hqtick_retracting_count(const hqtick_ctx *ctx);
int hqtick_cluster_last_reassigned(const hqtick_ctx *ctx, uint32_t *n, const uint64_t **task_id, const uint32_t **worker_id, const uint8_t **variant);

/*
 * Device-resident dependency graph (BASELINE config 5): the `Waiting{unfinished_deps}` counters and the consumer
 * sets of tako's task map live in HBM next to the ready set, so a batch of `Finished` updates releases its consumers into the
 * resident ready set on the device.  A resident ready set must exist (hqtick_upload_ready with n = 0 creates an empty one).
 *   hqtick_graph_add_tasks   on_new_tasks  server/reactor.rs:188-220.  Tasks in submission order; dep_off[n + 1] / dep_task_id is
 *                            the CSR of Task::task_deps (each list free of duplicates, as hyperqueue's submit guarantees,
 *                            crates/hyperqueue/src/server/client/submit.rs:446).  A dependency is kept only if the task map holds
 *                            it when the consumer is processed: ids that are not in the graph (finished tasks), the task itself and
 *                            tasks LATER in the same batch are dropped, exactly as `find_task_mut` misses them.  Tasks with no kept
 *                            dependency are merged into the ready set (Core::add_task, server/core.rs:213-218).
 *                            Returns how many tasks of the batch are ready now.  HQTICK_E_INVALID (graph unchanged) if an id is
 *                            already in the graph or listed twice (the reference asserts, core.rs:217).
 *   hqtick_graph_finish      task_finished  server/reactor.rs:510-590: every consumer's counter is decremented
 *                            (Task::decrease_unfinished_deps, server/task.rs:207-216); consumers reaching 0 are merged into the ready
 *                            set (add_ready_task); the finished tasks leave the graph (Core::remove_task).  Ids that are not in the
 *                            graph are counted (hqtick_graph_last_unknown; "Unknown task finished", reactor.rs:565-567).  The tasks
 *                            must not be in the ready set any more (they were handed out by a tick).  Returns the number of
 *                            released tasks.
 *   hqtick_graph_remove      Core::remove_task  server/core.rs:222-240 for cancel (on_cancel_tasks, reactor.rs:706-780) and failed
 *                            dependencies (task_failed, reactor.rs:606-704): the tasks leave the graph and, if they are there, the
 *                            ready set.  recursive != 0 also removes their transitive consumers
 *                            (Task::collect_recursive_consumers, server/task.rs:235-250).  Returns the number of removed tasks.
 *   hqtick_graph_last_ids    ids the last graph call produced (ready-now / released / removed tasks), ASCENDING, in pinned host
 *                            memory owned by the ctx; valid until the next graph call.  The host shim uses them to keep its Task
 *                            objects in step (state Waiting{0}, client notifications).
 *   hqtick_graph_unfinished  Task::get_unfinished_deps for each id; 0xFFFFFFFF for ids that are not in the graph (test accessor).
 * Task ids >= 0xFFFFFFFFFFFFFFFE are reserved.  Memory: 52 B per task slot + 24 B per hash bucket (2-4 buckets per task) + 20 B per
 * dependency edge; pools grow on demand and are compacted when the dead edges of finished producers fill them.
 */
typedef struct hqtick_graph_stats {
    uint64_t n_tasks;          /* tasks in the graph (waiting, ready, assigned, running)   */
    uint64_t n_slots;          /* task slots ever used (high-water mark)                    */
    uint64_t n_edges_live, n_edges_pool, n_runs;
    uint64_t hash_capacity, hash_tombstones;
    uint64_t bytes_hbm;
    double last_kernel_us;     /* HIP-event time of the dominant kernel(s) of the last graph call */
} hqtick_graph_stats;
int hqtick_graph_add_tasks(hqtick_ctx *ctx, uint64_t n, const uint64_t *task_id, const uint64_t *task_priority,
                           const uint32_t *task_rq, const uint32_t *dep_off, const uint64_t *dep_task_id);
int hqtick_graph_finish(hqtick_ctx *ctx, uint64_t n, const uint64_t *task_id);
int hqtick_graph_remove(hqtick_ctx *ctx, uint64_t n, const uint64_t *task_id, int recursive);
const uint64_t *hqtick_graph_last_ids(const hqtick_ctx *ctx, uint64_t *n);
uint64_t hqtick_graph_last_unknown(const hqtick_ctx *ctx);
int hqtick_graph_unfinished(hqtick_ctx *ctx, uint64_t n, const uint64_t *task_id, uint32_t *out);
int hqtick_graph_get_stats(const hqtick_ctx *ctx, hqtick_graph_stats *out);
/* EXTENSION — no reference counterpart, parity unpinned, OFF unless called (BASELINE.json configs[4] names a "dynamic b-level recompute"; the reference has none:
 * the low 32 bits of Priority, "scheduler priority", are never written — common/priority.rs:43-66, server/task.rs:175-177).
 * hqtick_graph_blevel: b-level(t) = 0 for a task no task still in the graph depends on, else 1 + the largest b-level among its consumers — the longest path
 * from t to a sink — for every task of the device-resident graph, by level-synchronous sweeps over its consumer lists; the value REPLACES the low 32 bits of the
 * task's priority (what Priority::add_priority_u32 would add to a priority whose low bits are zero), so that tasks released by later hqtick_graph_finish calls
 * carry it into the ready set.  HQTICK_BLEVEL_UPDATE_READY: the tasks already in the resident ready set get their graph priorities too (one pass over the ready
 * columns, ids -> graph slots through the hash table; *n_ready_updated = how many).  Returns the number of sweeps (>= 0) or a negative error; *max_level = the
 * largest b-level (the depth of the graph).  A host that wants the reference's behaviour never calls it; no parity test does.  It may be called between
 * hqtick_run_resident and hqtick_ready_consume_last: the pending consume is not disturbed (it replays the last tick's selection, which does not read priorities).
 * hqtick_graph_priorities: the priorities the graph holds for the given ids (0 for an id it does not hold): test accessor. */
#define HQTICK_BLEVEL_UPDATE_READY 1u
int hqtick_graph_blevel(hqtick_ctx *ctx, uint32_t flags, uint32_t *max_level, uint32_t *n_ready_updated);
int hqtick_graph_priorities(hqtick_ctx *ctx, uint64_t n, const uint64_t *task_id, uint64_t *out);

/*
 * Multi-GPU: worker sharding.  Every rank (one ctx per GPU) runs the tick on the SAME snapshot — scans, batches and the
 * placement are replicated and deterministic — but expands and emits records only for the workers it owns:
 *     FxHash(worker_id) % shard_count == shard_index       (FxHash = fxhash 0.2.1 of the u32 id, as tako's Map uses)
 * rec_off[] of the result then holds zero-length ranges for the other workers.  shard_count <= 1 switches sharding off.
 */
int hqtick_set_shard(hqtick_ctx *ctx, uint32_t shard_index, uint32_t shard_count);

/*
 * Record sink in DEVICE memory (for the RCCL all-gather that merges the shards' assignment vectors).  While a sink is set,
 * the records of a tick are written into it instead of host memory (result.rec_task/rec_variant/rec_kind are NULL;
 * result.rec_off stays valid).  Layout, all little-endian, with cap = hqtick_sink_capacity_records(W, capacity_bytes):
 *     u32 header[4] = { n_records, placement_checksum, HQTICK_SINK_MAGIC, cap }
 *         placement_checksum: FNV-1a over the tick's counts (rq, variant, worker, count), multi-node sets and is_optimal — the same on
 *         every rank unless their replicated placements diverged (only a tick that ran into its time limit can do that: its incumbent
 *         depends on the clock); the merging side compares the values and falls back to one rank's placement if they differ.
 *     u32 rec_off[W + 1]        (+ 4 bytes of padding when W is even, so that the next array is 8-byte aligned)
 *     u64 task[cap]   u8 variant[cap]   u8 kind[cap]
 * A tick whose records do not fit returns HQTICK_E_CAPACITY.  device_ptr == NULL removes the sink.
 */
#define HQTICK_SINK_MAGIC 0x48515354u
int hqtick_set_record_sink(hqtick_ctx *ctx, void *device_ptr, size_t capacity_bytes);
size_t hqtick_sink_bytes(uint32_t n_workers, uint32_t capacity_records);
uint32_t hqtick_sink_capacity_records(uint32_t n_workers, size_t capacity_bytes);

/*
 * The merge of the shards inside the boundary: one RCCL all-gather (over xGMI on one node) of every rank's record sink, so that a host
 * that loads this library needs nothing else for the multi-GPU path.  librccl is loaded on first use (dlopen).
 *   hqtick_comm_unique_id   rank 0 creates the 128-byte communicator id; the host carries it to the other ranks by its own means
 *   hqtick_comm_init        every rank, collectively: ncclCommInitRank on the ctx's device; also hqtick_set_shard(rank, world)
 *   hqtick_shard_allgather  after a tick with a record sink set: recv_device (device memory, >= world x sink capacity_bytes) receives the
 *                           sinks of ranks 0..world-1 back to back, each in the layout documented above; returns when the data is there
 *                           (ncclAllGather on the ctx's stream + one synchronisation).  Every rank must use the same capacity_bytes.
 *   hqtick_comm_destroy     ncclCommDestroy (also done by hqtick_destroy)
 */
#define HQTICK_COMM_ID_BYTES 128
/*
 * The placement itself over the ranks (ABI 8; SURVEY.md §8e: the per-worker blocks are independent, what couples them stays replicated).  With
 * shard_count > 1 and an exchange — the library's RCCL communicator (hqtick_comm_init with world == shard_count) or a callback of the host — every rank
 *   * runs only ITS worker range of a coupled tick's price sweeps (k_price_sweep over its share of the master's 16 worker ranges) and completes each sweep
 *     with one small all-gather: per block (c.x, reduced value, bound, steps), per range the wide rows' activities, and the rank's reading of the clock, so
 *     that every replica sees bit-identical cuts and leaves the sweeps at the same sweep; the patterns cross once, when the master asks for them;
 *   * solves only every shard_count-th class block of a separable tick (k_block_solve) and completes the launch with one all-gather of the answers.
 * Results are the unsharded tick's, bit for bit (tests/test_sharded.py, tests/test_gpu_multi.py).  Models under HQTICK_SHARD_MIN_BLOCKS blocks / launches
 * under HQTICK_SHARD_MIN_CLASSES classes (default 1025 both: what one MI355X holds resident at once) are solved whole by every rank, without an exchange;
 * HQTICK_SHARD_SOLVE=0 switches the split off.  EVERY rank must run the same ticks: a rank that stops calling leaves the others in the collective.
 *   hqtick_exchange_fn    all-gather of host memory: this rank contributes bytes_per_rank bytes at `send`; on return `recv` holds shard_count x bytes_per_rank
 *                         bytes, rank-major.  0 = ok.  Called from inside hqtick_run* on the calling thread, a few KB per call.
 *   hqtick_set_exchange   install (fn != NULL) or remove the callback; it takes precedence over the RCCL communicator.  rank / world are those of
 *                         hqtick_set_shard.
 */
typedef int (*hqtick_exchange_fn)(void *user, const void *send, void *recv, size_t bytes_per_rank);
int hqtick_set_exchange(hqtick_ctx *ctx, hqtick_exchange_fn fn, void *user);
int hqtick_comm_unique_id(void *id_out);
int hqtick_comm_init(hqtick_ctx *ctx, const void *id, uint32_t rank, uint32_t world);
int hqtick_shard_allgather(hqtick_ctx *ctx, void *recv_device, size_t recv_bytes);
int hqtick_comm_destroy(hqtick_ctx *ctx);

/* compute_new_worker_query(): batches + solver on fake workers, no mapping  scheduler/query.rs:70-71.  Takes its ready set from the
 * snapshot's task_* columns and runs on a private sub-context: a resident ready set, the last tick's selection (hqtick_ready_consume_last)
 * and the dependency graph of `ctx` are left untouched, so a query may be issued between any two resident ticks. */
int hqtick_query(hqtick_ctx *ctx, const hqtick_snapshot *snapshot, const hqtick_query_workers *fake,
                 hqtick_query_result *out);

/* Text of the last error on this ctx (never NULL). */
const char *hqtick_last_error(const hqtick_ctx *ctx);

/* Library/ABI version and the gfx arch the kernels were built for ("gfx950"). */
uint32_t hqtick_abi_version(void);
const char *hqtick_build_arch(void);

/*
 * Measurement hooks (used by bench.py; not part of the reference surface).
 * hqtick_kernel_stats(): HIP-event time of the ready-set streaming kernels of the last tick,
 * measured on the ctx's own stream, and the algorithmic bytes they moved.
 */
typedef struct hqtick_kernel_stats {
    double level_hist_us;   /* K1: per-(rq,priority) histogram over the ready set   */
    double select_us;       /* K4: selection + scatter of the taken tasks            */
    double distinct_us;     /* K0: distinct-priority discovery                       */
    double other_us;        /* K5b: per-worker expansion into records               */
    double scan_us;         /* K1b: scan of the per-slice counts                     */
    double sweep_us;        /* K5a: round-robin bit rows                             */
    double tick_gpu_us;     /* first kernel start -> last kernel end                 */
    uint64_t algorithmic_bytes; /* N*20 + W*R*16 + Q*V*R*9 + A*13 + P*12 */
    uint64_t n_assigned, n_prefilled;
    double block_solve_us;  /* k_block_solve: the per-worker-class blocks of the separable placement (one wavefront per class) */
    uint32_t n_classes_device, n_classes_host; /* worker classes solved by k_block_solve / by the host solver in the last tick */
    uint32_t block_steps_max, n_classes;       /* most search steps any class took; worker classes of the separable placement  */
    double solve_classify_us, solve_blocks_us, solve_decode_us; /* host wall clock inside the placement stage: worker classes, block solves
                                                  (launch + wait included), counts into the reference's Map iteration order */
    /* coupled ticks (priority cuts, unsaturated batches): the placement by price sweeps (k_price_sweep, csrc/price.hip) */
    uint32_t price_sweeps, price_rounds;   /* sweeps over all worker blocks / flag configurations of the last tick (0: the host search ran alone)   */
    uint32_t milp_cols, milp_rows;         /* size of the coupled model                                                                              */
    double price_us, price_sweep_us;       /* host wall clock inside the price solve / inside the sweeps (launch -> totals visible in pinned memory) */
    double milp_us, model_us;              /* host wall clock of the whole solve of the coupled model / of building it (solver.rs:95-430)            */
    double solve_pre_us;                   /* ... and of what the placement stage did before it (worker classes, the separable attempt)               */
    /* the guard on k_block_solve's answers: classes of the launch the host solved itself while the kernel ran (HQTICK_BLOCK_VERIFY, default 2 per launch, a window that
     * moves with the tick count) / of those: answers that differed (any: the whole launch is re-solved on the host) / answers thrown out by the per-class checks
     * (fits the rows, no room left for another task) and re-solved on the host */
    uint32_t n_classes_verified, n_classes_mismatch, n_classes_rejected;
    uint32_t n_classes_memo;               /* host class blocks of the last tick answered from the context's table of earlier identical blocks (HQTICK_FLAG_NO_BLOCK_MEMO: 0) */
    /* sharded placement solve (hqtick_set_exchange / hqtick_comm_init): the ranks' exchanges inside the last tick */
    uint32_t exchange_calls;               /* all-gathers of host buffers (one per sharded sweep, one per pattern fetch, one per class-block launch)          */
    uint32_t ready_appends;                /* NOT per tick: hqtick_ready_add* batches of this context that were appended behind the resident columns (fresh ids,
                                            * room at the tail: one kernel) instead of merged into new ones (HQTICK_APPEND=0: never)                       */
    uint64_t exchange_bytes;               /* bytes received by this rank in them                                                                              */
    double exchange_us;                    /* host wall clock inside them                                                                                      */
} hqtick_kernel_stats;
int hqtick_kernel_stats_last(const hqtick_ctx *ctx, hqtick_kernel_stats *out);
/* Switch the per-kernel timing on / off at run time (HQTICK_FLAG_NO_KERNEL_TIMING sets the initial state).  When on, the measured kernels are
 * bracketed by start / stop events AT THE DISPATCH (hipExtLaunchKernel): the duration is the kernel's own, the figure rocprofv3's kernel trace
 * reports, without the latency of markers queued around it.  on = 2: the events go around K1 alone (k_level_hist, the launch that streams the ready
 * set: the kernel the roofline figure is quoted on) — what bench.py's timed region runs with, so that every K1 launch of the run is measured the same
 * way and the live figure and a rocprofv3 kernel trace of the same command describe the same launches.  0 = off; any other value = every measured kernel. */
int hqtick_set_kernel_timing(hqtick_ctx *ctx, int on);
#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* HQTICK_H */
