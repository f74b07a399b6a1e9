// Device-resident dependency graph of the task map: the counters `Waiting{unfinished_deps}` and the
// consumer sets of tako's tasks, kept in HBM so that a batch of `Finished` updates releases its consumers into the resident
// ready set without a host round trip per task.
//
// Reference behaviour restated (crates/tako/src/internal):
//   on_new_tasks      server/reactor.rs:188-220   deps naming tasks that are not in the task map (finished, or later in the same
//                                                 batch) are dropped; unfinished_deps = the rest; 0 => ready queue (core.rs:213-218)
//   task_finished     server/reactor.rs:570-581   every consumer's counter is decremented; reaching 0 => add_ready_task
//   remove_task       server/core.rs:222-240      cancel / failed dependency: the task leaves the map (and the ready queue)
//   collect_recursive_consumers  server/task.rs:235-250   transitive consumers of a failed / cancelled task
//
// Layout (all arrays indexed by SLOT, a dense index that is recycled through a free stack):
//   id u64 | priority u64 | rq u32 | unfinished u32 (0xFFFFFFFF = free slot) | gen u32 (bumped when the slot is freed)
//   order u64 = batch number << 32 | position in its batch   (visibility rule of on_new_tasks)
//   head u32 -> chain of RUNS; a run = (next, offset, length) into the edge pool; one run per (producer, add batch)
//   edge = (consumer slot u32, consumer gen u32); an edge whose gen no longer matches is stale and skipped
//   id -> slot: open-addressing hash table in HBM (linear probing, 2x..4x slots)
// Everything here is integer work bound by HBM latency/bandwidth; no MFMA.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "devbuf.h"

namespace hqgraph {

constexpr uint32_t NONE = 0xFFFFFFFFu;
constexpr uint32_t ST_FREE = 0xFFFFFFFFu;
constexpr uint64_t HT_EMPTY = ~0ull, HT_TOMB = ~0ull - 1;  // task ids must be below HT_TOMB

enum ErrBits : uint32_t {
    ERR_EXISTS = 1u,        // add: a task id is already in the graph (reference: assert in Core::add_task, core.rs:217)
    ERR_DUP_IN_BATCH = 2u,  // add: the same id twice in one batch
    ERR_NOT_READY = 4u,     // finish: the task still had unfinished dependencies (reference: unreachable!, reactor.rs:551-555)
    ERR_CAPACITY = 8u,      // an internal pool overflowed (host sizing bug)
};

struct Ctl {  // device-resident counters; published to pinned host memory at the end of every operation
    uint32_t run_top;     // run records in use            } one 8-byte word: k_g_link_alloc bumps both with a single 64-bit atomic
    uint32_t edge_top;    // edge pool entries in use      }   (live + dead)
    uint32_t free_top;    // entries on the free-slot stack
    uint32_t edges_dead;  // edges of producers that have left the graph
    uint32_t n_out;       // entries of the operation's output list (released / removed / ready-now tasks)
    uint32_t n_unknown;   // ids of the operation that are not in the graph
    uint32_t err;         // ErrBits
    uint32_t n_big;       // runs deferred to the wide kernel
    uint32_t lev_begin;   // BFS window of remove(recursive)
    uint32_t lev_end;
    uint32_t pad[6];
};

struct View {  // device pointers, passed by value to the kernels
    uint64_t *id, *prio, *order;
    uint32_t *rq, *unfinished, *gen, *head;
    uint32_t *free_slot;
    uint64_t *ht_key; uint32_t *ht_val; uint32_t ht_mask;
    uint32_t *run_next, *run_off, *run_len;
    uint2 *edge;
    uint32_t *tmp_cnt, *tmp_base;  // per-slot scratch of add (tmp_cnt is all-zero between operations)
    Ctl *ctl;
};

struct Stats {
    uint64_t n_tasks, n_slots, n_edges_live, n_edges_pool, n_runs, hash_capacity, hash_tombstones, bytes_hbm;
};

class Graph {
  public:
    // All operations enqueue on `s` and synchronise it before they return.  A negative return is an HQTICK_E_* code with `err` set.
    // add: returns the number of tasks of the batch that are ready now; their (id, priority, rq) sorted by id are in out_*() (device).
    int add(uint64_t n, const uint64_t *id, const uint64_t *prio, const uint32_t *rq, const uint32_t *dep_off, const uint64_t *dep_id, hipStream_t s);
    // finish: returns the number of released consumers (sorted by id in out_*()); n_unknown() = ids that were not in the graph.
    int finish(uint64_t n, const uint64_t *id, hipStream_t s);
    // remove: the tasks (and, if recursive, their transitive consumers) leave the graph; returns how many left (sorted ids in out_id()).
    int remove(uint64_t n, const uint64_t *id, bool recursive, hipStream_t s);
    // test accessor: Task::get_unfinished_deps; 0xFFFFFFFF for an id that is not in the graph
    int unfinished(uint64_t n, const uint64_t *id, uint32_t *out, hipStream_t s);
    // EXTENSION (no reference counterpart, parity unpinned): the b-level of every task — longest path to a sink of what is still in the graph — into the low 32
    // bits of its priority (common/priority.rs:43-66: the "scheduler priority" bits the reference never writes).  Returns the sweeps it took.
    int blevel(uint32_t *max_level, const uint64_t *ready_id, const uint32_t *ready_rq, uint64_t *ready_prio, uint64_t n_ready, uint32_t *n_ready_updated, hipStream_t s);
    int priorities(uint64_t n, const uint64_t *id, uint64_t *out, hipStream_t s);  // test accessor: the priorities the graph holds (0 for an unknown id)
    void clear();
    void release();

    const uint64_t *out_id() const { return d_out_id.as<uint64_t>(); }        // device, ascending ids
    const uint64_t *out_prio() const { return d_out_prio.as<uint64_t>(); }    // device
    const uint32_t *out_rq() const { return d_out_rq.as<uint32_t>(); }        // device
    const uint64_t *out_id_host() const { return h_out.as<uint64_t>(); }      // pinned host copy of out_id()
    uint32_t n_out() const { return n_out_; }
    uint32_t n_unknown() const { return n_unknown_; }
    uint64_t n_tasks() const { return n_live_; }
    double last_kernel_us() const { return last_us_; }  // GPU time of the dominant kernel of the last operation (HIP events)
    Stats stats() const;
    std::string err;

  private:
    int fail(int code, const std::string &m) { err = m; return code; }
    bool grow_slots(uint64_t want, hipStream_t s);
    bool rebuild_hash(uint64_t want_entries, hipStream_t s);
    int ensure_edges(uint64_t extra, hipStream_t s);
    int stage(uint64_t n_ids, const uint64_t *ids, hipStream_t s);
    int finish_output(hipStream_t s, bool with_payload);
    View view() const;
    bool init(hipStream_t s);

    hqbuf::DevBuf d_id, d_prio, d_order, d_rq, d_unf, d_gen, d_head, d_free, d_tmpc, d_tmpb;
    hqbuf::DevBuf d_htk, d_htv;
    hqbuf::DevBuf d_rn, d_ro, d_rl, d_edge, d_rn2, d_ro2, d_rl2, d_edge2;
    hqbuf::DevBuf d_bl;
    hqbuf::DevBuf d_ctl, d_stage, d_eds, d_erk, d_okey, d_oval, d_out_id, d_out_prio, d_out_rq, d_big;
    hqbuf::PinBuf h_ctl, h_stage, h_out;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    uint64_t cap_slots = 0, n_slots = 0, n_live_ = 0, cap_edges = 0, ht_cap = 0, ht_used = 0;  // ht_used = keys + tombstones
    uint32_t free_top = 0, run_top = 0, edge_top = 0, edges_dead = 0, batch_no = 0;
    uint32_t n_out_ = 0, n_unknown_ = 0;
    double last_us_ = 0.0;
    bool ready_ = false;
};

// Sorts (key, value) pairs by key ascending on the device: LDS-local bitonic network for the low stages, global steps above.
// keys beyond n (up to the next power of two) are filled with UINT64_MAX.  Buffers must hold n_pow2 entries.
hipError_t sort_pairs(uint64_t *key, uint32_t *val, uint64_t n, hipStream_t s);
uint64_t sort_capacity(uint64_t n);  // entries a sort buffer for n elements needs

}  // namespace hqgraph
