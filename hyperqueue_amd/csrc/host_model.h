// Host-side stages of the tick that are small and serial by nature: batch merge, MILP construction, gap cache.
// They consume what the GPU scans produced (level histogram, per-worker capability flags) and produce the
// per-(request, variant, worker) counts the GPU mapping stage expands.
//
// Reference (paths relative to /root/reference/crates/tako/src/internal/):
//   scheduler/batches.rs:42-217   create_task_batches + prune_progressive
//   scheduler/solver.rs:36-483    run_scheduling_solver (model + decode)
//   scheduler/gap.rs:38-147       GapCache::get_gap / compute_gap_resources
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/hqtick.h"
#include "block_core.h"
#include "price.h"
#include "milp.h"

namespace hqhost {

struct VariantView {
    const uint32_t *res;   // entries, sorted by resource id
    const uint8_t *kind;
    const uint64_t *amount;
    uint32_t n_entries, n_nodes, weight;
    uint64_t min_time_ns;
    bool multi_node() const { return n_nodes > 0; }
};

struct RequestView {
    uint32_t first_variant, n_variants;  // global variant slots [first_variant, first_variant + n_variants)
};

// (priority, number of tasks) of one TaskQueue, highest priority first — TaskQueue::iter_priority_sizes
struct QueueLevels {
    std::vector<std::pair<uint64_t, uint32_t>> levels;
};

struct PriorityCut { uint32_t size; std::vector<std::pair<uint32_t, uint32_t>> blockers; };  // blocker size HQ_BLOCKER_UNBOUNDED = None
struct TaskBatch {
    uint32_t rq = 0, size = 0, limit = 0;
    bool limit_reached = false, is_blocker = false;
    std::vector<PriorityCut> cuts;
};

// Workers whose snapshot rows are identical: same total / free / remaining time / min_utilization / flags and no blocked request.  K2's row of a
// worker (capability bits, task_max_count) is a function of exactly these columns and the request tables, so the groups can be formed from the
// snapshot alone — while the GPU still runs phase A — and the per-worker loops of create_task_batches (limits) and of the separable placement
// (worker classes) run once per group.  Groups are numbered in the order of their first worker.
struct WorkerGroups {
    std::vector<uint32_t> of, rep, count;  // [n] group of worker w; [n_groups] first worker / number of workers
    std::vector<uint32_t> solver_workers;  // the workers run_scheduling_solver models (solver.rs:57-66), ascending
    std::vector<double> pool;              // resource_sums over them (solver.rs:56,68-82), summed in worker order like the reference
    bool valid = false;
};

// Everything the host stages need to know about workers, as plain views over the snapshot + the GPU's K2 output.
struct WorkerSet {
    WorkerGroups rows;
    uint32_t n = 0, R = 0;
    const uint32_t *id = nullptr;
    const uint64_t *total = nullptr, *free_ = nullptr;  // [n*R]
    const int64_t *remaining_ns = nullptr;
    const float *min_util = nullptr;
    const uint8_t *flags = nullptr;    // HQ_WORKER_*; nullptr => all SN
    const uint32_t *group = nullptr;
    // K2 output, indexed [w * n_variant_slots + slot]: bit0 immediate, bit1 total-capable, bit2 time ok
    const uint8_t *vflags = nullptr;
    const uint32_t *vtmc = nullptr;
    uint32_t n_variant_slots = 0;
    // sparse per-worker state
    std::vector<std::vector<std::pair<uint32_t, uint8_t>>> blocked;   // [n]
    uint32_t n_blocked_lists = 0;   // entries in `blocked` (0: every list is empty and the vector can be reused as it is)
    // SingleNodeTaskAssignment::assigned_tasks as the snapshot's CSR (views, not copies; nullptr = nothing assigned)
    const uint32_t *assigned_off = nullptr, *assigned_rq = nullptr; const uint8_t *assigned_variant = nullptr;
    uint32_t n_assigned(uint32_t w) const { return assigned_off ? assigned_off[w + 1] - assigned_off[w] : 0; }
    bool is_sn(uint32_t w) const { return flags ? (flags[w] & HQ_WORKER_SN) != 0 : true; }
    bool stopping(uint32_t w) const { return flags ? (flags[w] & HQ_WORKER_STOPPING) != 0 : false; }
    bool is_free(uint32_t w) const { return is_sn(w) && n_assigned(w) == 0 && !stopping(w); }
    uint8_t vf(uint32_t w, uint32_t slot) const { return vflags[(size_t)w * n_variant_slots + slot]; }
    uint32_t tmc(uint32_t w, uint32_t slot) const { return vtmc[(size_t)w * n_variant_slots + slot]; }
};

// Batch solver of the per-class blocks of the separable path (run_scheduling_solver below): the tick hands in the launcher of
// k_block_solve (csrc/block_solve.hip).  All pointers of the tables are host memory; false = not solved (the host solver takes over).
struct BlockSolver {
    virtual ~BlockSolver() {}
    virtual bool solve(const hqblock::ColTable &cols, const hqblock::ClassTable &classes, const hqblock::Output &out) = 0;
    // The same in two halves, so that the host can work while the kernel runs (it solves the classes it held back: the hardest ones, which would otherwise be the
    // launch's span).  begin() returns once the launch is enqueued, finish() when the answers are in `out`.  Default: everything in begin().
    virtual bool begin(const hqblock::ColTable &cols, const hqblock::ClassTable &classes, const hqblock::Output &out) { begun_ok = solve(cols, classes, out); return begun_ok; }
    virtual bool finish() { return begun_ok; }
    virtual bool overlaps() const { return false; }  // begin() really returns before the answers are there
    bool begun_ok = false;
};

// Forget what earlier ticks left behind on this thread (the memoised Map iteration orders): HQTICK_FLAG_NO_TICK_CACHES.
void flush_tick_caches();

// The last class blocks the host solver answered with a certified canonical optimum, keyed by the block's whole model: an identical block of a later tick is
// answered from here (hqtick.h: HQTICK_FLAG_NO_BLOCK_MEMO).  Small on purpose — a steady cluster has a handful of host-solved classes; the device solves the many.
struct BlockMemo {
    struct Entry { uint64_t hash = 0; std::vector<unsigned char> key; std::vector<double> x; long nodes = 0; int n_components = 0; };
    static constexpr size_t CAP = 64;
    std::vector<Entry> entries;
    size_t next = 0;
    std::vector<unsigned char> scratch;  // the key of the model being looked up (kept between find and put)
    uint64_t scratch_hash = 0;
    const Entry *find(const hqmilp::Model &m);
    void put(const hqmilp::Result &r);    // under the key of the last find()
};

struct Problem {
    BlockMemo *memo = nullptr;           // nullptr: every host block is solved
    BlockSolver *blocks = nullptr;       // nullptr: every block on the host
    hqprice::Sweeper *pricer = nullptr;  // the block sweeps of the coupled solve (csrc/price.h): k_price_sweep in the tick; nullptr: host-only search
    uint32_t block_min_classes = 1;      // fewer device-eligible classes than this: not worth a launch
    uint32_t block_verify = 2;           // classes of every device launch the host solves itself while the kernel runs, to compare (0: none, UINT32_MAX: all)
    uint32_t tick_seq = 0;               // moves the sample window from tick to tick (the ctx's tick counter: the same on every replica)
    uint32_t R = 0, n_groups = 0;
    std::vector<RequestView> rqs;
    std::vector<VariantView> variants;  // all variant slots
    WorkerSet real;                      // core.worker_map
    const WorkerSet *custom = nullptr;   // what-if query: fake workers replace the worker list of the solver
    double time_limit_s = 5.0;
    bool certificate_only = false;       // HQTICK_FLAG_CERTIFICATE_ONLY: the coupled model's solve stops at the rel_gap certificate, as the reference's does
    bool rq_multi_node(uint32_t rq) const { return variants[rqs[rq].first_variant].multi_node(); }
    // Worker::is_capable_to_run_rqv  server/worker.rs:277-296
    bool capable_rqv(const WorkerSet &ws, uint32_t w, uint32_t rq) const {
        for (uint32_t v = 0; v < rqs[rq].n_variants; v++) {
            uint32_t slot = rqs[rq].first_variant + v;
            uint8_t f = ws.vf(w, slot);
            if ((f & 4) && (variants[slot].multi_node() || (f & 2))) return true;
        }
        return false;
    }
};

void group_equal_rows(WorkerSet &ws, bool all_solver);  // fills ws.rows (needs id / total / free_ / remaining_ns / min_util / flags / blocked; not K2's output);
                                                       // all_solver: every worker is a solver worker (the fake workers of a what-if query), else the SN ones

std::vector<TaskBatch> create_task_batches(const Problem &pb, const std::vector<QueueLevels> &queues);

struct Counts {
    // sn_counts in the reference's iteration order: keys, and per key (worker index, count) in `counts` iteration order
    std::vector<std::pair<uint32_t, uint8_t>> keys;
    std::vector<std::vector<std::pair<uint32_t, uint32_t>>> per_key;
    std::vector<uint32_t> mn_rq;                               // mn_workers keys (variant 0) in iteration order
    std::vector<std::vector<std::vector<uint32_t>>> mn_sets;   // per key: worker sets
    // Separable ticks also say how the counts came about — per worker class — so that the mapping plan can fill its per-(key, worker) tables with
    // sequential passes over the workers instead of scattering the pairs above: count of key k on worker w = class_x[wclass[w] * n_cols + key_col[k]]
    // (workers outside the solver carry the all-zero class n_classes), and keys whose worker lists are equal share one list (workers in Map order).
    struct WorkerList { std::vector<uint32_t> widx; };
    bool by_class = false;
    bool one_class = false;  // by_class with a single class that holds every worker: the count of a key is one number
    uint32_t n_cols = 0;
    std::vector<uint32_t> wclass, class_x, key_col, key_list;
    std::vector<WorkerList> lists;
    // by_class results leave per_key EMPTY (building 8 x 1024 pairs costs the cold tick 3 us nobody needs): pairs() builds them on demand
    bool pairs_built = true;
    void pairs() {
        if (pairs_built) return;
        per_key.assign(keys.size(), {});
        for (size_t k = 0; k < keys.size(); k++) {
            const std::vector<uint32_t> &wi = lists[key_list[k]].widx;
            auto &out = per_key[k]; out.resize(wi.size());
            for (size_t i = 0; i < wi.size(); i++) out[i] = {wi[i], class_x[(size_t)wclass[wi[i]] * n_cols + key_col[k]]};
        }
        pairs_built = true;
    }
    size_t key_size(size_t k) const { return pairs_built ? per_key[k].size() : lists[key_list[k]].widx.size(); }  // workers of key k
    bool is_optimal = true;
    bool is_canonical = true;  // every solve completed its tie-break phase
    bool empty() const {
        if (!pairs_built) { for (size_t k = 0; k < keys.size(); k++) if (key_size(k)) return false; }
        for (auto &k : per_key) if (!k.empty()) return false;
        for (auto &k : mn_sets) if (!k.empty()) return false;
        return true;
    }
    uint32_t checksum() const {  // FNV-1a over the placement: replicas of a sharded scheduler compare it (include/hqtick.h, record sink)
        uint32_t h = 2166136261u;
        auto mix = [&](uint32_t v) { for (int i = 0; i < 4; i++) { h ^= (v >> (8 * i)) & 0xFFu; h *= 16777619u; } };
        mix(is_optimal ? 1u : 0u);
        for (size_t k = 0; k < keys.size(); k++) {
            mix(keys[k].first); mix(keys[k].second);
            if (pairs_built) for (auto &wc : per_key[k]) { mix(wc.first); mix(wc.second); }
            else for (uint32_t w : lists[key_list[k]].widx) { mix(w); mix(class_x[(size_t)wclass[w] * n_cols + key_col[k]]); }  // the same (worker, count) sequence
        }
        for (size_t k = 0; k < mn_rq.size(); k++) { mix(mn_rq[k]); for (auto &set : mn_sets[k]) { mix(0xFFFFFFFFu); for (uint32_t w : set) mix(w); } }
        return h;
    }
    int error = 0; std::string errmsg;
    long milp_nodes = 0; int milp_cols = 0, milp_rows = 0, milp_components = 0;
    int price_sweeps = 0, price_rounds = 0; double price_us = 0, milp_us = 0, model_us = 0, pre_us = 0;  // coupled path: block sweeps / flag rounds of csrc/price.cpp, time inside them, inside hqmilp::solve, building the model
    uint32_t blocks_device = 0, blocks_host = 0, block_steps_max = 0, n_classes = 0;
    uint32_t blocks_verified = 0, blocks_mismatch = 0, blocks_rejected = 0;  // device block answers re-solved by the host while the kernel ran / of those: different / answers the O(columns) checks threw out
    uint32_t blocks_memo = 0;  // host blocks answered from Problem::memo
    double t_classify_us = 0, t_blocks_us = 0, t_decode_us = 0;  // separable path: worker classes / block solves (device wait included) / counts in Map order  // separable path: classes solved by k_block_solve / by the host solver
};

Counts run_scheduling_solver(const Problem &pb, const std::vector<TaskBatch> &batches);

}  // namespace hqhost
