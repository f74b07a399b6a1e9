/* hqwire.h -- worker-message wire encoding on the device, part of libhqtick.so.
 *
 * Replaces, for the messages one scheduling tick emits, what the reference does on the host between
 * `WorkerTaskMapping::send_messages` and the socket:
 *   /root/reference/crates/tako/src/internal/scheduler/mapping.rs:259-292   per worker RetractTasks + ComputeTasks, per multi-node task
 *                                                                          a single-task message with its node list
 *   .../server/task.rs:315-445       ComputeTasksBuilder (shared data deduplicated per message, size estimate, 32 MiB cut)
 *   .../messages/worker.rs:27-57,76-88   the serde structs
 *   .../transfer/auth.rs:253-263     bincode DefaultOptions + fixint: little endian, u64 lengths, u32 enum tags, u8 Option tags
 * Input: the tick's records where they already are -- in HBM (the record sink of hqtick_set_record_sink, include/hqtick.h) -- plus a
 * task-attribute table and a configuration table the host keeps resident in HBM (it knows both at submit time).  Output: one byte
 * buffer in HBM holding every message of the tick back to back, and per message slot its byte range; the host (or a NIC) reads ranges,
 * it never touches a task.  Sealing (orion AEAD) and length-delimited framing stay the reference's (network layer, out of scope).
 *
 * Message slots: slot w < n_workers = worker index w (its RetractTasks message, then its ComputeTasks message);
 * slot n_workers + k = multi-node task k (one ComputeTasks message for worker_id[mn_worker[mn_worker_off[k]]], the root).
 *
 * Fragmentation (ABI 2).  ComputeTasksBuilder cuts a worker's message whenever its size estimate passes MAX_TASK_MSG_SIZE = 32 MiB
 * (create_message_on_overflow, task.rs:388-400): the message so far is sent, the configuration index starts afresh.  With
 * hqwire_output.slot_nfrag / frag_end set the device does the same: the ComputeTasks range of a slot then holds slot_nfrag[s] messages back to
 * back, message f ending at frag_end[s * HQWIRE_MAX_FRAGMENTS + f] (absolute offset; message 0 starts at slot_off[2s + 1]), each with its own
 * shared-data list and shared_index numbering.
 *
 * Not covered on the device, reported per slot so the host builds the slot's ComputeTasks messages itself (they are rare).  The slot's
 * RetractTasks message IS emitted in every case — only the ComputeTasks part falls back:
 *   HQWIRE_SLOT_OVERSIZE  more than HQWIRE_MAX_FRAGMENTS messages for one slot (> 512 MiB for one worker in one tick), or an estimate above the
 *                         limit while the caller passed no fragment arrays
 *   HQWIRE_SLOT_TOO_MANY  more than HQWIRE_MAX_RECORDS records for one worker in one tick (device-side dedup table)
 *   HQWIRE_SLOT_UNKNOWN   a record names a task id that is not in the attribute table
 * No CPU implementation behind this entry point: without a gfx950 device it returns HQTICK_E_NO_DEVICE.
 */
#ifndef HQWIRE_H
#define HQWIRE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: what this header declares is its whole dynamic symbol table. */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define HQWIRE_ABI_VERSION 2u
#define HQWIRE_MAX_RECORDS 2048u                 /* records per worker message handled on the device */
#define HQWIRE_MAX_TASK_MSG_SIZE (32u << 20)     /* MAX_FRAME_SIZE / 4 (crates/tako/src/lib.rs:31, server/task.rs:315) */
#define HQWIRE_MAX_FRAGMENTS 16u                 /* ComputeTasks messages one slot may be cut into on the device */

enum { HQWIRE_SLOT_OK = 0, HQWIRE_SLOT_OVERSIZE = 1, HQWIRE_SLOT_UNKNOWN = 2, HQWIRE_SLOT_TOO_MANY = 3 };
enum { HQWIRE_OK = 0, HQWIRE_CAPACITY = 1 };     /* header[0] */

/* Task attributes (what ComputeTaskSeparateData takes from `Task`, messages/worker.rs:27-39) and the interned
 * `TaskConfiguration`s (server/task.rs:95-101; equal configurations share one index: the builder's `configuration_index` key). */
typedef struct hqwire_tables {
    uint64_t n_tasks;
    const uint64_t *task_id;       /* ascending; job_id << 32 | job_task_id                      */
    const uint32_t *task_rq;       /* ResourceRqId                                               */
    const uint32_t *task_instance; /* InstanceId                                                 */
    const uint64_t *task_priority; /* Task::priority() raw                                       */
    const uint32_t *task_config;   /* index into the configuration table                         */
    const uint8_t *entry_some;     /* 1 = Some(entry)                                            */
    const uint64_t *entry_off;     /* [n_tasks + 1] into entry_blob                              */
    const uint8_t *entry_blob;
    uint32_t n_configs;
    const uint8_t *config_time_some;  /* time_limit: Option<Duration>                            */
    const uint64_t *config_time_secs;
    const uint32_t *config_time_nanos;
    const uint64_t *body_off;      /* [n_configs + 1] into body_blob                             */
    const uint8_t *body_blob;
} hqwire_tables;

/* The tick's mapping (hqtick_result / record sink layout, include/hqtick.h). */
typedef struct hqwire_records {
    uint32_t n_workers;
    uint32_t n_records;          /* rec_off[n_workers] (known on the host: result.rec_off stays valid in sink mode) */
    const uint32_t *worker_id;   /* [n_workers]                                                 */
    const uint32_t *rec_off;     /* [n_workers + 1]                                             */
    const uint64_t *rec_task;
    const uint8_t *rec_variant;
    const uint8_t *rec_kind;     /* HQ_REC_PREFILL = 0 -> variant None, HQ_REC_ASSIGN = 1 -> Some(rec_variant) */
    const uint32_t *retract_off; /* [n_workers + 1] or NULL                                     */
    const uint64_t *retract_task;
    uint32_t n_mn;
    const uint64_t *mn_task;
    const uint32_t *mn_worker_off; /* [n_mn + 1] */
    const uint32_t *mn_worker;     /* worker INDEX, root first */
} hqwire_records;

typedef struct hqwire_output {
    uint8_t *bytes;        /* message bytes, back to back                                                          */
    uint64_t capacity;     /* of `bytes`                                                                           */
    uint64_t *slot_off;    /* [2 * n_slots + 1]: RetractTasks of slot s = [off[2s], off[2s+1]), ComputeTasks = [off[2s+1], off[2s+2]) */
    uint8_t *slot_status;  /* [n_slots] HQWIRE_SLOT_*                                                               */
    uint32_t *header;      /* [4] = { HQWIRE_OK / HQWIRE_CAPACITY, n_slots, total bytes low, total bytes high }     */
    void *scratch;         /* hqwire_scratch_bytes(n_records + n_mn, n_slots); 8-byte aligned                       */
    uint64_t scratch_bytes;
    /* ABI 2: fragmentation.  Both NULL = off (an estimate above the limit then marks the slot HQWIRE_SLOT_OVERSIZE). */
    uint32_t *slot_nfrag;  /* [n_slots] ComputeTasks messages of the slot (0 = none)                                */
    uint64_t *frag_end;    /* [n_slots * HQWIRE_MAX_FRAGMENTS] absolute end offset of message f of slot s           */
    uint64_t msg_size_limit; /* 0 = HQWIRE_MAX_TASK_MSG_SIZE; the builder's estimate limit (tests use small values)  */
} hqwire_output;

uint64_t hqwire_scratch_bytes(uint64_t n_records_incl_mn, uint64_t n_slots);

/* Encodes every message of one tick.  ALL pointers of the three structs are DEVICE pointers (HBM); the three kernels are enqueued on
 * `hip_stream` (a hipStream_t, NULL = the null stream) and the call returns without synchronising: header[0] / slot_status / slot_off are
 * valid once the stream has drained.  On HQWIRE_CAPACITY nothing is written to `bytes` (header[2..3] still hold the size needed).
 * Returns 0, HQTICK_E_INVALID (-1), HQTICK_E_NO_DEVICE (-2) or HQTICK_E_DEVICE (-3) (include/hqtick.h). */
int hqwire_encode_device(const hqwire_tables *tables, const hqwire_records *records, const hqwire_output *out, void *hip_stream);

uint32_t hqwire_abi_version(void);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
