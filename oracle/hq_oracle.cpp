// =====================================================================================================
// TEST INFRASTRUCTURE — CPU ORACLE.  NOT PRODUCT CODE.
//
// A plain C++17 restatement of the tako scheduling tick of It4innovations/hyperqueue
// (reference mounted read-only at /root/reference; every function cites the file:line it follows, paths
// relative to crates/tako/src/internal/).  Only tests/, __graft_entry__.smoke() and bench.py's
// `cpu_baseline` leg may load this library; the product (libhqtick.so) never links or calls it.
//
// The placement MILP is solved by the reference with HiGHS (third party, `highs` 2.4.0 / `highs-sys` 1.15.0,
// Cargo.lock:1107-1122, sources not under /root/reference).  This oracle builds exactly the reference's
// model (same variables, rows and f64 coefficients in the same order — scheduler/solver.rs:36-430) and hands
// it to a solver callback; oracle/oracle.py plugs in HiGHS 1.8.0 (scipy.optimize.milp), i.e. the same
// third-party solver family the reference uses.
//
// Pinning: the restatement is checked against the reference's own unit-test vectors, transcribed in
// tests/test_oracle_golden.py.  The hashbrown/fxhash iteration-order emulation (hb_emul.h)
// is validated only on those small pinned cases => "parity unpinned" for Map iteration order at scale.
// =====================================================================================================
#include "../include/hqtick.h"
#include "hb_emul.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace {

typedef uint64_t u64;
typedef uint32_t u32;
typedef uint8_t u8;

double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ---------------------------------------------------------------------------------------------------
// LP model handed to the solver callback (solver/mod.rs:29-54 LpInnerSolver, solver/highs.rs:19-45)
// ---------------------------------------------------------------------------------------------------
enum { COL_NAT = 0, COL_BOOL = 1 };
enum { ROW_MIN = 0, ROW_MAX = 1, ROW_EQ = 2 };          // ConstraintType  solver/mod.rs:22-27
enum { CT_SN = 0, CT_MN = 1, CT_RESERVE = 2, CT_MU = 3, CT_GROUP = 4, CT_BETA = 5 };

struct Model {
    std::vector<double> obj;
    std::vector<u8> kind;
    std::vector<u8> ctype;       // CT_* (metadata for tests)
    std::vector<u32> cworker;    // worker index or HQ_NO_WORKER
    std::vector<u32> crq;
    std::vector<u8> cvariant;
    std::vector<u8> rtype;
    std::vector<double> rhs;
    std::vector<int> roff{0};
    std::vector<int> rcol;
    std::vector<double> rcoef;
    int add_col(double w, u8 k, u8 ct, u32 worker, u32 rq, u8 v) {
        obj.push_back(w); kind.push_back(k); ctype.push_back(ct);
        cworker.push_back(worker); crq.push_back(rq); cvariant.push_back(v);
        return (int)obj.size() - 1;
    }
    void add_row(u8 t, double b, const std::vector<std::pair<int, double>> &terms) {
        rtype.push_back(t); rhs.push_back(b);
        for (auto &p : terms) { rcol.push_back(p.first); rcoef.push_back(p.second); }
        roff.push_back((int)rcol.size());
    }
    void add_row_extra(u8 t, double b, const std::vector<int> &vars, int var, double coef) {  // constraint_extra_var solver.rs:599-612
        rtype.push_back(t); rhs.push_back(b);
        for (int v : vars) { rcol.push_back(v); rcoef.push_back(1.0); }
        rcol.push_back(var); rcoef.push_back(coef);
        roff.push_back((int)rcol.size());
    }
    int ncols() const { return (int)obj.size(); }
    int nrows() const { return (int)rhs.size(); }
};

}  // namespace

extern "C" {
// returns 1 when a solution was written to x_out (objective in *obj_out), 0 otherwise
typedef int (*oracle_solve_fn)(void *user, int ncols, const double *obj, const uint8_t *col_kind, int nrows,
                               const uint8_t *row_type, const double *rhs, const int *row_off, const int *row_col,
                               const double *row_coef, double time_limit_s, double *x_out, double *obj_out,
                               int *is_optimal);
}

namespace {

// ---------------------------------------------------------------------------------------------------
// a0: types.  ResourceAmount arithmetic (common/resources/amount.rs), requests (common/resources/request.rs)
// ---------------------------------------------------------------------------------------------------
inline double as_f64(u64 a) { return (double)a / 10000.0; }                       // amount.rs:91-93
inline u64 sat_sub(u64 a, u64 b) { return a > b ? a - b : 0; }                    // amount.rs:95-98
inline u64 amount_from_float(float v) { return (u64)std::ceil(v * 10000.0f); }   // amount.rs:41-43

struct Entry { u32 res; u8 kind; u64 amount; };
struct Variant {
    std::vector<Entry> entries;  // sorted by resource id (request.rs:144)
    u32 n_nodes; u64 min_time_ns; u32 weight;
    bool is_mn() const { return n_nodes > 0; }                                    // request.rs:157-159
    double weight_f64() const { return (double)weight / 10000.0; }                // request.rs:122-124
};
struct Rqv {
    std::vector<Variant> v;
    bool is_mn() const { return v[0].is_mn(); }                                   // request.rs:322-324
};

struct Res {  // WorkerResources padded to R   server/workerload.rs:17-31
    std::vector<u64> a;
    u64 get(u32 r) const { return r < a.size() ? a[r] : 0; }
    // is_capable_to_run_request  workerload.rs:77-83  (All => min_amount 1 fraction, request.rs:34-36)
    bool capable(const Variant &rq) const {
        for (auto &e : rq.entries) {
            u64 ask = e.kind == HQ_ENTRY_ALL ? 1 : e.amount;
            if (ask > get(e.res)) return false;
        }
        return true;
    }
    // task_max_count_for_request  workerload.rs:121-145
    u32 tmc(const Variant &rq) const {
        bool any = false; u64 best = 0;
        for (auto &e : rq.entries) {
            u64 c;
            if (e.kind != HQ_ENTRY_ALL) c = std::min<u64>(get(e.res) / e.amount, HQ_MAX_TASK_PER_WORKER);
            else c = get(e.res) == 0 ? 0 : 1;
            if (!any || c < best) best = c;
            any = true;
        }
        return any ? (u32)best : 0;
    }
    u32 tmc_rqv(const Rqv &rqv) const {  // task_max_count  workerload.rs:147-154
        u32 s = 0;
        for (auto &v : rqv.v) s += tmc(v);
        return s;
    }
    void remove(const Variant &rq) {  // workerload.rs:156-165
        for (auto &e : rq.entries) a[e.res] = e.kind != HQ_ENTRY_ALL ? sat_sub(a[e.res], e.amount) : 0;
    }
    void add(const Variant &rq, const Res &all) {  // workerload.rs:194-202
        for (auto &e : rq.entries) a[e.res] = e.kind != HQ_ENTRY_ALL ? a[e.res] + e.amount : all.get(e.res);
    }
    void remove_multiple(const Variant &rq, u32 n) {  // workerload.rs:167-177
        for (auto &e : rq.entries) a[e.res] = e.kind != HQ_ENTRY_ALL ? sat_sub(a[e.res], e.amount * (u64)n) : 0;
    }
};

struct WorkerS {
    u32 id; Res total, free; int64_t remaining_ns; float min_util; u8 flags; u32 group;
    std::vector<std::pair<u32, u8>> blocked;
    std::vector<std::pair<u32, u8>> assigned;   // (rq, variant) of assigned_tasks
    std::vector<u32> prefilled_rq;              // rq of prefilled_tasks
    bool is_sn() const { return flags & HQ_WORKER_SN; }
    bool stopping() const { return flags & HQ_WORKER_STOPPING; }
    bool is_free() const { return is_sn() && assigned.empty() && !stopping(); }   // server/worker.rs:181-186
    bool has_time(u64 min_time_ns) const {                                        // server/worker.rs:320-326
        if (remaining_ns == HQ_NO_TIME_LIMIT) return true;
        return remaining_ns >= 0 && (u64)remaining_ns >= min_time_ns;
    }
    bool capable_rq(const Variant &rq) const {                                    // server/worker.rs:277-286
        if (!has_time(rq.min_time_ns)) return false;
        return rq.is_mn() ? true : total.capable(rq);
    }
    bool capable_rqv(const Rqv &rqv) const {                                      // server/worker.rs:288-296
        for (auto &v : rqv.v) if (capable_rq(v)) return true;
        return false;
    }
    bool blocked_rq(u32 rq, u8 v) const {                                         // server/worker.rs:332-338
        for (auto &b : blocked) if (b.first == rq && b.second == v) return true;
        return false;
    }
};

// ---------------------------------------------------------------------------------------------------
// a1: TaskQueue (scheduler/taskqueue.rs:115-373) over sorted arrays.
// ---------------------------------------------------------------------------------------------------
struct Level { u64 priority; std::vector<u64> ids; size_t head = 0; size_t size() const { return ids.size() - head; } };
struct Queue {
    u32 rq;
    std::vector<Level> levels;   // priority descending; ids ascending; emptied levels are skipped via `first`
    size_t first = 0;
    bool has_prefill = false; u64 prefill_priority = 0;
    std::vector<u64> prefill_ids; std::vector<u32> prefill_workers; size_t prefill_head = 0;
    std::vector<u64> prefill_added;  // ids appended by take_tasks_for_prefill in this tick (not drained in-tick)
    void skip_empty() { while (first < levels.size() && levels[first].size() == 0) first++; }
    bool queue_empty() { skip_empty(); return first >= levels.size(); }
    size_t prefill_size() const { return has_prefill ? prefill_ids.size() - prefill_head + prefill_added.size() : 0; }
    bool is_empty() { return queue_empty() && prefill_size() == 0; }              // taskqueue.rs:227-230
    bool top_priority(u64 *p) { if (queue_empty()) return false; *p = levels[first].priority; return true; }  // :233-238
    u32 top_size_no_prefill() {                                                   // :241-253
        if (queue_empty()) return 0;
        if (has_prefill && prefill_priority != levels[first].priority) return 0;
        return (u32)levels[first].size();
    }
    // iter_priority_sizes  taskqueue.rs:273-302
    std::vector<std::pair<u64, u32>> priority_sizes() {
        skip_empty();
        std::vector<std::pair<u64, u32>> r;
        size_t i = first;
        if (has_prefill) {
            u32 ps = (u32)prefill_size();
            if (i < levels.size() && levels[i].priority == prefill_priority) { r.push_back({prefill_priority, (u32)levels[i].size() + ps}); i++; }
            else r.push_back({prefill_priority, ps});
        }
        for (; i < levels.size(); i++) if (levels[i].size()) r.push_back({levels[i].priority, (u32)levels[i].size()});
        return r;
    }
    bool take_from_entry(u32 &count, std::vector<u64> &out) {                     // :406-420 (+ first_entry().unwrap())
        if (queue_empty()) return false;
        Level &l = levels[first];
        while (count > 0 && l.size() > 0) { out.push_back(l.ids[l.head++]); count--; }
        return true;
    }
    void drain_prefill(u32 &count, std::vector<u64> &out, std::vector<u32> &out_old_worker) {  // :381-397
        if (!has_prefill) return;
        while (count > 0 && prefill_head < prefill_ids.size()) {
            out.push_back(prefill_ids[prefill_head]); out_old_worker.push_back(prefill_workers[prefill_head]);
            prefill_head++; count--;
        }
        if (prefill_size() == 0) has_prefill = false;
    }
    // take_tasks  taskqueue.rs:320-355.  old_worker[i] = HQ_NO_WORKER for Waiting tasks, else the worker the
    // task was Prefilled on.  Returns false when the reference would panic (queue exhausted).
    bool take_tasks(u32 count, std::vector<u64> &out, std::vector<u32> &old_worker) {
        auto pad = [&]() { old_worker.resize(out.size(), HQ_NO_WORKER); };
        if (!has_prefill) {
            while (count > 0) { if (!take_from_entry(count, out)) return false; }
            pad(); return true;
        }
        u64 tp; bool has_top = top_priority(&tp);
        if (has_top && tp == prefill_priority) {
            if (count > 0) { take_from_entry(count, out); }
            pad();
            drain_prefill(count, out, old_worker);
            while (count > 0) { if (!take_from_entry(count, out)) return false; }
            pad();
        } else {
            drain_prefill(count, out, old_worker);
            while (count > 0) { if (!take_from_entry(count, out)) return false; }
            pad();
        }
        return true;
    }
    bool take_one(u64 *id) {                                                      // :357-373
        if (queue_empty()) return false;
        Level &l = levels[first]; *id = l.ids[l.head++]; return true;
    }
    std::vector<u64> take_for_prefill(u32 count) {                                // :304-318
        std::vector<u64> r;
        u64 p = levels[first].priority;
        take_from_entry(count, r);
        if (!has_prefill) { has_prefill = true; prefill_priority = p; }
        for (u64 t : r) prefill_added.push_back(t);
        return r;
    }
};

// ---------------------------------------------------------------------------------------------------
// a2: batches (scheduler/batches.rs)
// ---------------------------------------------------------------------------------------------------
struct Cut { u32 size; std::vector<std::pair<u32, u32>> blockers; };             // batches.rs:13-16 (size UINT32_MAX = None)
struct Batch { u32 rq; std::vector<Cut> cuts; u32 size = 0, limit = 0; bool limit_reached = false, is_blocker = false; };

// prune_progressive  batches.rs:183-217
template <typename T> void prune_progressive(std::vector<T> &vec, size_t prefix, size_t limit) {
    size_t n = vec.size();
    if (n <= limit) return;
    size_t remaining = limit - prefix, pool = n - prefix, last = prefix - 1;
    std::vector<size_t> idx;
    for (size_t i = 0; i < prefix; i++) idx.push_back(i);
    for (size_t i = 0; i < remaining; i++) {
        double t = (double)i / (double)(remaining - 1);
        size_t index = prefix + (size_t)std::round(t * t * (double)(pool - 1));
        if (index <= last) index = last + 1;
        idx.push_back(index); last = index;
    }
    for (size_t i = 0; i < idx.size() && i < limit; i++) std::swap(vec[i], vec[idx[i]]);
    vec.resize(limit);
}

struct State {
    u32 R = 0;
    std::vector<WorkerS> workers;         // ascending id
    std::vector<u32> worker_map_order;    // worker indices in core.worker_map iteration order
    std::vector<Rqv> rqs;
    std::vector<Queue> queues;
    u32 n_groups = 0;
    hqtick_config cfg;
    std::map<u64, u64> task_priority;     // for the final per-worker sort (mapping.rs:128-131)
};

// create_task_batches  batches.rs:42-181.  `custom` = fake workers of the what-if query (None => real ones).
std::vector<Batch> create_task_batches(State &st, const std::vector<WorkerS> *custom) {
    std::vector<Queue *> queues;
    for (auto &q : st.queues) if (!q.is_empty()) queues.push_back(&q);
    std::vector<Batch> batches;
    if (queues.empty()) return batches;
    for (Queue *q : queues) {
        const Rqv &rqv = st.rqs[q->rq];
        u32 limit;
        if (rqv.is_mn()) {  // :65-78
            u32 n_frees = 0;
            for (auto &w : st.workers) if (w.is_free()) n_frees++;
            limit = n_frees / rqv.v[0].n_nodes;
        } else {  // :79-92
            limit = 0;
            const std::vector<WorkerS> &ws = custom ? *custom : st.workers;
            for (auto &w : ws) {
                if (!w.capable_rqv(rqv)) continue;
                u32 runnable = w.is_sn() ? w.free.tmc_rqv(rqv) : 0;
                limit += runnable > 0 ? runnable : 1;
            }
        }
        Batch b; b.rq = q->rq; b.limit = limit;
        batches.push_back(b);
    }
    size_t nq = queues.size();
    std::vector<std::vector<std::pair<u64, u32>>> iters(nq);
    std::vector<size_t> pos(nq, 0);
    std::vector<bool> alive(nq);
    for (size_t i = 0; i < nq; i++) { iters[i] = queues[i]->priority_sizes(); alive[i] = !iters[i].empty(); }
    auto advance = [&](size_t i) { pos[i]++; alive[i] = pos[i] < iters[i].size(); };
    long unique = -1;
    std::vector<size_t> found;
    for (;;) {
        found.clear();
        u64 highest = 0;  // Priority::new(0)  :105
        for (size_t i = 0; i < nq; i++) {
            if (!alive[i]) continue;
            u64 p = iters[i][pos[i]].first;
            if (p == highest) found.push_back(i);
            else if (p > highest) { highest = p; found.clear(); found.push_back(i); }
        }
        auto add_level = [&](size_t idx) {
            batches[idx].size += iters[idx][pos[idx]].second;
            if (batches[idx].size > batches[idx].limit) {
                batches[idx].size = batches[idx].limit; batches[idx].limit_reached = true; alive[idx] = false;
            } else advance(idx);
        };
        if (found.size() == 1 && unique == (long)found[0]) {
            add_level(found[0]);
        } else if (found.empty()) {
            break;
        } else {
            for (size_t idx : found) {
                u32 size = batches[idx].size;
                std::vector<std::pair<u32, u32>> higher;
                for (size_t i = 0; i < nq; i++) {
                    Batch &b = batches[i];
                    if (i != idx && (b.size > 0 || b.limit_reached)) {
                        b.is_blocker = true;
                        higher.push_back({b.rq, b.limit_reached ? HQ_BLOCKER_UNBOUNDED : b.size});
                    }
                }
                if (!higher.empty()) batches[idx].cuts.push_back(Cut{size, higher});
            }
            for (size_t idx : found) add_level(idx);
            unique = found.size() == 1 ? (long)found[0] : -1;
        }
    }
    std::vector<Batch> out;
    for (auto &b : batches) {
        prune_progressive(b.cuts, 4, 32);  // BATCH_PRUNING_FIXED_PREFIX / MAX_SIZE  :8-9
        if (b.size > 0) out.push_back(b);
    }
    return out;
}

// ---------------------------------------------------------------------------------------------------
// a6: gap (scheduler/gap.rs)
// ---------------------------------------------------------------------------------------------------
struct Solver { oracle_solve_fn fn; void *user; double solve_us = 0; int calls = 0; };

bool run_solver(Solver &s, const Model &m, double time_limit, std::vector<double> &x, double *objv, int *is_opt) {
    x.assign(m.ncols(), 0.0);
    double t0 = now_us();
    int ok = s.fn(s.user, m.ncols(), m.obj.data(), m.kind.data(), m.nrows(), m.rtype.data(), m.rhs.data(), m.roff.data(),
                  m.rcol.data(), m.rcoef.data(), time_limit, x.data(), objv, is_opt);
    s.solve_us += now_us() - t0; s.calls++;
    return ok != 0;
}

// compute_gap_resources  gap.rs:95-147
Res compute_gap_resources(const Rqv &rqv, const Res &resources, Solver &solver) {
    long maxr = -1;
    for (auto &v : rqv.v) for (auto &e : v.entries) maxr = std::max<long>(maxr, e.res);
    Res out;
    if (maxr < 0) return out;
    size_t n_res = (size_t)maxr + 1;
    // `resources.iter_pairs()`: only non-zero amounts, and the result vector has one element per PAIR (gap.rs:107-145):
    // the collected Vec is indexed by position, not by resource id.  Restated literally.
    for (u32 r = 0; r < resources.a.size(); r++) {
        u64 r_amount = resources.a[r];
        if (r_amount == 0) continue;
        Model m;
        std::vector<int> vars;
        for (auto &v : rqv.v) {  // :115-121  weight = rq.get_amount(r_id).unwrap_or(resources.get(r_id))
            u64 a = 0; bool found = false, all = false;
            for (auto &e : v.entries) if (e.res == r) { found = true; all = e.kind == HQ_ENTRY_ALL; a = e.amount; }
            double w = !found ? 0.0 : (all ? as_f64(resources.get(r)) : as_f64(a));
            vars.push_back(m.add_col(w, COL_NAT, CT_SN, HQ_NO_WORKER, 0, 0));
        }
        std::vector<std::vector<std::pair<int, double>>> cst(n_res);
        for (size_t i = 0; i < rqv.v.size(); i++)
            for (auto &e : rqv.v[i].entries) {  // :122-131 (All => resources.get(r_id) of the OUTER r_id, as written)
                double a = e.kind == HQ_ENTRY_ALL ? as_f64(resources.get(r)) : as_f64(e.amount);
                cst[e.res].push_back({vars[i], a});
            }
        for (size_t idx = 0; idx < n_res; idx++) m.add_row(ROW_MAX, as_f64(resources.get((u32)idx)), cst[idx]);
        std::vector<double> x; double objv = 0; int is_opt = 0;
        if (!run_solver(solver, m, 1e30, x, &objv, &is_opt) || !is_opt) { out.a.push_back(0); continue; }  // :142-144
        out.a.push_back(r_amount - amount_from_float((float)std::round(objv)));  // :145 (u64 wrapping sub in release)
    }
    return out;
}

struct GapCache {  // gap.rs:8-35
    std::map<std::pair<u32, std::vector<u64>>, Res> m;
};

// GapCache::get_gap  gap.rs:38-93
u32 get_gap(GapCache &cache, const State &st, u32 high_rq, u32 low_rq, const Res &resources,
            const std::vector<std::pair<u32, u8>> &assigned, Solver &solver) {
    const Rqv &h = st.rqs[high_rq], &l = st.rqs[low_rq];
    if (h.is_mn() || l.is_mn()) return 0;
    Res free;
    if (h.v.size() == 1) {
        for (auto &e : h.v[0].entries) if (e.kind == HQ_ENTRY_ALL) return 0;
        u32 count = resources.tmc(h.v[0]);
        free = resources; free.remove_multiple(h.v[0], count);
    } else {
        auto key = std::make_pair(high_rq, resources.a);
        auto it = cache.m.find(key);
        if (it != cache.m.end()) free = it->second;
        else { free = compute_gap_resources(h, resources, solver); cache.m[key] = free; }
    }
    free.a.resize(std::max<size_t>(free.a.size(), st.R), 0);
    for (auto &a : assigned) if (a.first != high_rq) free.remove(st.rqs[a.first].v[a.second]);
    bool any = false; u32 best = 0;
    for (auto &v : l.v) { u32 c = free.tmc(v); if (!any || c < best) best = c; any = true; }
    return any ? best : 0;
}

// ---------------------------------------------------------------------------------------------------
// a4: model build (scheduler/solver.rs:36-430) and a7: decode (:439-481)
// ---------------------------------------------------------------------------------------------------
struct Solution {
    // sn_counts in the reference's iteration order (see decode): keys + per-key (worker idx, count) in counts order
    std::vector<std::pair<u32, u8>> keys;
    std::vector<std::vector<std::pair<u32, u32>>> counts;
    std::vector<std::pair<u32, u8>> mn_keys;
    std::vector<std::vector<std::vector<u32>>> mn_workers;
    bool is_optimal = true;
    bool empty() const {
        for (auto &c : counts) if (!c.empty()) return false;
        for (auto &m : mn_workers) if (!m.empty()) return false;
        return true;
    }
};

struct BuiltModel {
    Model m;
    std::map<std::tuple<u32, u32, u8>, int> placements;  // (worker idx, rq, variant) -> col   solver.rs:88
};

int g_error = 0;
std::string g_errmsg;

BuiltModel build_model(State &st, const std::vector<Batch> &batches, const std::vector<WorkerS> *custom, GapCache &gaps,
                       Solver &solver, std::vector<const WorkerS *> &workers_out) {
    BuiltModel bm; Model &m = bm.m;
    u32 R = st.R;
    std::vector<const WorkerS *> workers;  // :57-66
    if (custom) for (auto &w : *custom) workers.push_back(&w);
    else for (auto &w : st.workers) if (w.is_sn()) workers.push_back(&w);  // already ascending id
    workers_out = workers;
    std::vector<double> sums(R, 0.0);  // :56,68-82
    for (auto *w : workers) for (u32 r = 0; r < R; r++) { u64 c = w->free.get(r); sums[r] += c == HQ_AMOUNT_MAX ? 1.0 : as_f64(c); }
    size_t nw = workers.size();
    // worker index used in `placements` keys: position in st.workers (or in custom list)
    auto widx_of = [&](const WorkerS *w) -> u32 { return custom ? (u32)(w - custom->data()) : (u32)(w - st.workers.data()); };

    std::map<u32, std::vector<int>> count_vars;  // tasks_count_vars  :89
    std::vector<std::vector<std::pair<int, double>>> res_cst(R);  // worker_res_constraint :91
    std::vector<std::pair<int, double>> cpu_cst;  // worker_cpu_constraint_no_reserves :92
    // group capability for MN requests (server/workergroup.rs:35-52) uses ALL workers of the real worker map
    auto group_capable_rq = [&](u32 g, const Variant &rq) {
        u32 target = rq.is_mn() ? rq.n_nodes : 1;
        for (auto &w : st.workers) if (w.group == g && w.capable_rq(rq)) { if (--target == 0) return true; }
        return false;
    };
    auto group_capable_rqv = [&](u32 g, const Rqv &rqv) { for (auto &v : rqv.v) if (group_capable_rq(g, v)) return true; return false; };

    for (size_t wi = 0; wi < nw; wi++) {  // :95
        const WorkerS *w = workers[wi]; u32 widx = widx_of(w);
        cpu_cst.clear();
        for (auto &batch : batches) {
            const Rqv &rqv = st.rqs[batch.rq];
            bool has_variant = false;
            for (u8 vi = 0; vi < rqv.v.size(); vi++) {
                const Variant &rq = rqv.v[vi];
                if (rq.is_mn()) {  // :101-122
                    if (custom && w->is_free()) { g_error = HQTICK_E_UNSUPPORTED; g_errmsg = "reference panics: fake worker group lookup (solver.rs:104-106)"; return bm; }
                    if (w->is_free() && group_capable_rq(w->group, rq)) {
                        double s = 0.0;  // create_mn_var :573-597
                        for (u32 r = 0; r < R; r++) { u64 a = w->total.get(r); if (a == 0) continue; double g = sums[r]; s += g < 0.000001 ? 0.0 : as_f64(a) / g; }
                        double weight = s * (double)(nw - wi) * rq.weight_f64() / (double)nw;
                        int v = m.add_col(weight, COL_BOOL, CT_MN, widx, batch.rq, vi);
                        bm.placements[{widx, batch.rq, vi}] = v;
                        for (u32 r = 0; r < R; r++) { u64 a = w->total.get(r); if (a) res_cst[r].push_back({v, as_f64(a)}); }
                    }
                } else if (!w->blocked_rq(batch.rq, vi) && w->has_time(rq.min_time_ns) && w->is_sn() && w->free.capable(rq)) {  // :123-126
                    has_variant = true;
                    double s = 0.0;  // create_sn_var :542-571
                    for (auto &e : rq.entries) {
                        double g = sums[e.res];
                        s += g < 0.000001 ? 0.0 : as_f64(e.kind == HQ_ENTRY_ALL ? w->total.get(e.res) : e.amount) / g;
                    }
                    double weight = s * (double)(nw - wi) * rq.weight_f64() / (double)nw;
                    int v = m.add_col(weight, COL_NAT, CT_SN, widx, batch.rq, vi);
                    bm.placements[{widx, batch.rq, vi}] = v;
                    count_vars[batch.rq].push_back(v);
                    for (auto &e : rq.entries) {  // :137-149
                        double amount = as_f64(e.kind == HQ_ENTRY_ALL ? w->total.get(e.res) : e.amount);
                        res_cst[e.res].push_back({v, amount});
                        if (e.res == 0) cpu_cst.push_back({v, amount});  // CPU_RESOURCE_ID == 0  map.rs
                    }
                }
            }
            if (!has_variant && !rqv.is_mn() && batch.is_blocker && w->capable_rqv(rqv) && w->is_sn()) {  // :153-169
                double weight = (double)wi / (double)(nw * 100);
                int v = m.add_col(weight, COL_BOOL, CT_RESERVE, widx, batch.rq, 0);
                count_vars[batch.rq].push_back(v);
                for (u32 r = 0; r < R; r++) { u64 c = w->free.get(r); if (c) res_cst[r].push_back({v, as_f64(c)}); }
            }
        }
        if (w->min_util > 0.001f) {  // add_min_utilization :501-540
            u64 all_amount = w->total.get(0);
            if (w->is_sn() && all_amount != HQ_AMOUNT_MAX) {
                double all_cpus = as_f64(all_amount), free_cpus = as_f64(w->free.get(0));
                double min_cpus = all_cpus * ((double)w->min_util - 1.0) + free_cpus;
                if (!(min_cpus < 0.0001)) {
                    int v = m.add_col(0.0, COL_BOOL, CT_MU, widx, 0, 0);
                    cpu_cst.push_back({v, -min_cpus}); m.add_row(ROW_MIN, 0.0, cpu_cst); cpu_cst.pop_back();
                    cpu_cst.push_back({v, -all_cpus}); m.add_row(ROW_MAX, 0.0, cpu_cst); cpu_cst.pop_back();
                }
            }
        }
        for (u32 r = 0; r < R; r++) {  // :177-191 — note: a MAX free amount `continue`s WITHOUT clearing the row terms
            u64 free = w->free.get(r);
            if (free == HQ_AMOUNT_MAX) continue;
            if (!res_cst[r].empty()) m.add_row(ROW_MAX, as_f64(free), res_cst[r]);
            res_cst[r].clear();
        }
    }
    std::map<std::pair<u32, u32>, int> per_group;  // task_counts_per_group :193
    for (auto &batch : batches) {  // :195-227
        const Rqv &rqv = st.rqs[batch.rq];
        if (!rqv.is_mn()) continue;
        double n_nodes = (double)rqv.v[0].n_nodes;
        for (u32 g = 0; g < st.n_groups; g++) {  // worker_groups.iter(): group indices are given in the reference's iteration order
            std::vector<int> temp;
            // group.worker_ids() is a Set<WorkerId> in hash order; only the SET of terms matters for the row
            for (auto &w : st.workers) if (w.group == g) {
                auto it = bm.placements.find({(u32)(&w - st.workers.data()), batch.rq, (u8)0});
                if (it != bm.placements.end()) temp.push_back(it->second);
            }
            if (!temp.empty()) {
                int v = m.add_col(0.0, COL_NAT, CT_GROUP, HQ_NO_WORKER, batch.rq, 0);
                m.add_row_extra(ROW_EQ, 0.0, temp, v, -n_nodes);
                count_vars[batch.rq].push_back(v);
                per_group[{batch.rq, g}] = v;
            }
        }
    }
    std::map<std::pair<u32, u32>, int> bvars;  // blocked_priority_vars :231
    auto get_bvar = [&](u32 blocker_rq, u32 size) -> int {  // :233-253
        auto it = bvars.find({blocker_rq, size});
        if (it != bvars.end()) return it->second;
        auto cv = count_vars.find(blocker_rq);
        if (cv == count_vars.end()) return -1;
        int nv = m.add_col(0.0, COL_BOOL, CT_BETA, HQ_NO_WORKER, blocker_rq, 0);
        double bound = (double)size;
        m.add_row_extra(ROW_MIN, bound, cv->second, nv, bound);
        bvars[{blocker_rq, size}] = nv;
        return nv;
    };
    for (auto &batch : batches) {  // :258-430
        auto tc = count_vars.find(batch.rq);
        if (tc == count_vars.end()) continue;
        const Rqv &batch_rqv = st.rqs[batch.rq];
        if (!batch.limit_reached) {  // :264-271
            std::vector<std::pair<int, double>> t;
            for (int v : tc->second) t.push_back({v, 1.0});
            m.add_row(ROW_MAX, (double)batch.size, t);
        }
        double batch_size = (double)batch.size;
        std::vector<u32> blocked_by_unbounded;
        for (auto &cut : batch.cuts) {
            for (auto &bl : cut.blockers) {
                u32 blocker_rq = bl.first; bool has_s = bl.second != HQ_BLOCKER_UNBOUNDED; u32 s = bl.second;
                std::vector<int> zero_cond;
                const Rqv &blocker_rqv = st.rqs[blocker_rq];
                if (batch_rqv.is_mn()) {  // :279-289
                    for (u32 g = 0; g < st.n_groups; g++) {
                        auto it = per_group.find({batch.rq, g});
                        if (it != per_group.end() && group_capable_rqv(g, blocker_rqv)) zero_cond.push_back(it->second);
                    }
                } else {
                    for (auto *w : workers) {  // :291-347
                        if (!w->is_sn()) continue;
                        if (!w->capable_rqv(blocker_rqv)) continue;
                        u32 widx = widx_of(w);
                        u32 gap = get_gap(gaps, st, blocker_rq, batch.rq, w->total, w->assigned, solver);
                        std::vector<int> vars;
                        for (u8 vi = 0; vi < batch_rqv.v.size(); vi++) {
                            auto it = bm.placements.find({widx, batch.rq, vi});
                            if (it != bm.placements.end()) vars.push_back(it->second);
                        }
                        if (gap > 0) {
                            double cut_size = (double)cut.size;
                            int bv;
                            if (has_s && (bv = get_bvar(blocker_rq, s)) >= 0) {
                                m.add_row_extra(ROW_MAX, cut_size + batch_size + (double)gap, vars, bv, batch_size);
                            } else if (!has_s) {
                                std::vector<std::pair<int, double>> t;
                                for (int v : vars) t.push_back({v, 1.0});
                                m.add_row(ROW_MAX, cut_size + (double)gap, t);
                            }
                        } else {
                            for (int v : vars) zero_cond.push_back(v);
                        }
                    }
                }
                if (zero_cond.empty()) continue;
                int bv;
                if (has_s && (bv = get_bvar(blocker_rq, s)) >= 0) {  // :395-411
                    m.add_row_extra(ROW_MAX, batch_size + (double)cut.size, zero_cond, bv, batch_size);
                } else if (!has_s && std::find(blocked_by_unbounded.begin(), blocked_by_unbounded.end(), blocker_rq) == blocked_by_unbounded.end()) {  // :412-426
                    blocked_by_unbounded.push_back(blocker_rq);
                    std::vector<std::pair<int, double>> t;
                    for (int v : zero_cond) t.push_back({v, 1.0});
                    m.add_row(ROW_MAX, (double)cut.size, t);
                }
            }
        }
    }
    return bm;
}

// decode  solver.rs:439-481.  The Map iteration orders are emulated with hb::Table.
Solution decode(State &st, const std::vector<Batch> &batches, const BuiltModel &bm, const std::vector<const WorkerS *> &workers,
                const std::vector<WorkerS> *custom, const std::vector<double> &x, bool is_optimal) {
    Solution sol; sol.is_optimal = is_optimal;
    auto widx_of = [&](const WorkerS *w) -> u32 { return custom ? (u32)(w - custom->data()) : (u32)(w - st.workers.data()); };
    const std::vector<WorkerS> &wl = custom ? *custom : st.workers;
    hb::RqVTable key_table, mn_key_table;
    std::map<u64, std::vector<std::pair<u32, u32>>> by_key;
    std::map<u64, std::vector<std::vector<u32>>> mn_by_key;
    for (auto &batch : batches) {
        const Rqv &rqv = st.rqs[batch.rq];
        if (rqv.is_mn()) {  // :442-464
            size_t n_nodes = rqv.v[0].n_nodes;
            std::vector<std::vector<u32>> ws;
            for (auto *w : workers) {
                auto it = bm.placements.find({widx_of(w), batch.rq, (u8)0});
                if (it == bm.placements.end()) continue;
                u32 count = (u32)std::round(x[it->second]);
                if (count > 0) {
                    if (!ws.empty() && ws.back().size() < n_nodes) ws.back().push_back(widx_of(w));
                    else ws.push_back({widx_of(w)});
                }
            }
            if (!ws.empty()) { u64 k = ((u64)batch.rq << 8); mn_key_table.insert(k); mn_by_key[k] = ws; }
        } else {  // :465-478
            for (u8 vi = 0; vi < rqv.v.size(); vi++) {
                hb::WorkerIdTable counts;  // Map<WorkerId,u32> collected from workers in ascending id
                std::map<u32, std::pair<u32, u32>> by_id;
                for (auto *w : workers) {
                    auto it = bm.placements.find({widx_of(w), batch.rq, vi});
                    if (it == bm.placements.end()) continue;
                    u32 count = (u32)std::round(x[it->second]);
                    if (count > 0) { counts.insert(w->id); by_id[w->id] = {widx_of(w), count}; }
                }
                if (counts.items > 0) {
                    u64 k = ((u64)batch.rq << 8) | vi;
                    key_table.insert(k);
                    std::vector<std::pair<u32, u32>> ordered;
                    counts.for_each([&](u64 id) { ordered.push_back(by_id[(u32)id]); });
                    by_key[k] = ordered;
                }
            }
        }
    }
    (void)wl;
    key_table.for_each([&](u64 k) { sol.keys.push_back({(u32)(k >> 8), (u8)k}); sol.counts.push_back(by_key[k]); });
    mn_key_table.for_each([&](u64 k) { sol.mn_keys.push_back({(u32)(k >> 8), (u8)k}); sol.mn_workers.push_back(mn_by_key[k]); });
    return sol;
}

// ---------------------------------------------------------------------------------------------------
// result storage
// ---------------------------------------------------------------------------------------------------
struct Rec { u64 task; u8 variant; u8 kind; u64 priority; u32 rq; };
struct Out {
    std::vector<u32> batch_rq, batch_size, batch_limit; std::vector<u8> batch_lr, batch_blk;
    std::vector<u32> batch_cut_off, cut_size, cut_blocker_off, blocker_rq, blocker_size;
    std::vector<u32> count_rq; std::vector<u8> count_variant; std::vector<u32> count_worker, count_value;
    std::vector<u32> rec_off; std::vector<u64> rec_task; std::vector<u8> rec_variant, rec_kind;
    std::vector<u32> retract_off; std::vector<u64> retract_task;
    std::vector<u64> redirect_task; std::vector<u32> redirect_worker; std::vector<u8> redirect_variant, redirect_kind;
    std::vector<u64> mn_task; std::vector<u32> mn_worker_off, mn_worker;
    std::vector<u64> new_free;
    std::vector<u8> q_loaded;
    BuiltModel last_model;
    std::vector<double> last_x; double last_obj = 0;
};

void export_batches(Out &o, const std::vector<Batch> &batches) {
    o.batch_rq.clear(); o.batch_size.clear(); o.batch_limit.clear(); o.batch_lr.clear(); o.batch_blk.clear();
    o.batch_cut_off.assign(1, 0); o.cut_size.clear(); o.cut_blocker_off.assign(1, 0); o.blocker_rq.clear(); o.blocker_size.clear();
    for (auto &b : batches) {
        o.batch_rq.push_back(b.rq); o.batch_size.push_back(b.size); o.batch_limit.push_back(b.limit);
        o.batch_lr.push_back(b.limit_reached); o.batch_blk.push_back(b.is_blocker);
        for (auto &c : b.cuts) {
            o.cut_size.push_back(c.size);
            for (auto &bl : c.blockers) { o.blocker_rq.push_back(bl.first); o.blocker_size.push_back(bl.second); }
            o.cut_blocker_off.push_back((u32)o.blocker_rq.size());
        }
        o.batch_cut_off.push_back((u32)o.cut_size.size());
    }
}

bool load_state(State &st, const hqtick_snapshot *s, const hqtick_config *cfg) {
    st.cfg = *cfg; st.R = s->n_resources; st.n_groups = s->n_groups;
    u32 W = s->n_workers, R = st.R;
    st.workers.resize(W);
    for (u32 i = 0; i < W; i++) {
        WorkerS &w = st.workers[i];
        w.id = s->worker_id[i];
        if (i && s->worker_id[i - 1] >= w.id) return false;
        w.total.a.assign(s->worker_total + (size_t)i * R, s->worker_total + (size_t)(i + 1) * R);
        w.free.a.assign(s->worker_free + (size_t)i * R, s->worker_free + (size_t)(i + 1) * R);
        w.remaining_ns = s->worker_remaining_ns ? s->worker_remaining_ns[i] : HQ_NO_TIME_LIMIT;
        w.min_util = s->worker_min_utilization ? s->worker_min_utilization[i] : 0.0f;
        w.flags = s->worker_flags ? s->worker_flags[i] : HQ_WORKER_SN;
        w.group = s->worker_group ? s->worker_group[i] : 0;
        if (s->assigned_off) for (u32 k = s->assigned_off[i]; k < s->assigned_off[i + 1]; k++) w.assigned.push_back({s->assigned_rq[k], s->assigned_variant[k]});
        if (s->prefilled_off) for (u32 k = s->prefilled_off[i]; k < s->prefilled_off[i + 1]; k++) w.prefilled_rq.push_back(s->prefilled_rq[k]);
    }
    for (u32 k = 0; k < s->n_blocked; k++) st.workers[s->blocked_worker[k]].blocked.push_back({s->blocked_rq[k], s->blocked_variant[k]});
    st.worker_map_order.resize(W);
    if (s->worker_map_rank) {
        for (u32 i = 0; i < W; i++) st.worker_map_order[s->worker_map_rank[i]] = i;
    } else {  // emulate a Map<WorkerId, Worker> built by inserting ascending ids
        hb::WorkerIdTable t; std::map<u32, u32> idx;
        for (u32 i = 0; i < W; i++) { t.insert(st.workers[i].id); idx[st.workers[i].id] = i; }
        size_t k = 0; t.for_each([&](u64 id) { st.worker_map_order[k++] = idx[(u32)id]; });
    }
    u32 Q = s->n_requests;
    st.rqs.resize(Q);
    for (u32 q = 0; q < Q; q++) {
        for (u32 vi = s->rq_variant_off[q]; vi < s->rq_variant_off[q + 1]; vi++) {
            Variant v; v.n_nodes = s->variant_n_nodes[vi]; v.min_time_ns = s->variant_min_time_ns[vi]; v.weight = s->variant_weight[vi];
            for (u32 e = s->variant_entry_off[vi]; e < s->variant_entry_off[vi + 1]; e++) v.entries.push_back(Entry{s->entry_resource[e], s->entry_kind[e], s->entry_amount[e]});
            st.rqs[q].v.push_back(v);
        }
        if (st.rqs[q].v.empty()) return false;
    }
    st.queues.resize(Q);
    for (u32 q = 0; q < Q; q++) st.queues[q].rq = q;
    // TaskQueue per rq: BTreeMap<Reverse<Priority>, BTreeSet<TaskId>>  taskqueue.rs:115-119
    std::vector<std::vector<std::pair<u64, u64>>> per(Q);
    for (u64 i = 0; i < s->n_ready; i++) {
        if (s->task_rq[i] >= Q) return false;
        per[s->task_rq[i]].push_back({s->task_priority[i], s->task_id[i]});
        st.task_priority[s->task_id[i]] = s->task_priority[i];
    }
    for (u32 q = 0; q < Q; q++) {
        auto &v = per[q];
        std::sort(v.begin(), v.end(), [](const std::pair<u64, u64> &a, const std::pair<u64, u64> &b) { return a.first != b.first ? a.first > b.first : a.second < b.second; });
        for (auto &p : v) {
            if (st.queues[q].levels.empty() || st.queues[q].levels.back().priority != p.first) st.queues[q].levels.push_back(Level{p.first, {}, 0});
            st.queues[q].levels.back().ids.push_back(p.second);
        }
        if (s->prefill_off && s->prefill_off[q + 1] > s->prefill_off[q]) {
            Queue &qu = st.queues[q]; qu.has_prefill = true; qu.prefill_priority = s->prefill_priority[q];
            for (u32 k = s->prefill_off[q]; k < s->prefill_off[q + 1]; k++) {
                qu.prefill_ids.push_back(s->prefill_task[k]); qu.prefill_workers.push_back(s->prefill_worker[k]);
                st.task_priority[s->prefill_task[k]] = qu.prefill_priority;
            }
        }
    }
    return true;
}

struct Ctx {
    hqtick_config cfg; Out out; std::string err; double t_load = 0, t_batches = 0, t_model = 0, t_solve = 0, t_mapping = 0;
    // one-shot: the next oracle_tick takes its placement from here instead of the solver (parity tier T3: mapping given counts, scheduler/mapping.rs:23-234)
    bool given = false, g_optimal = true; std::vector<u32> g_rq, g_worker, g_value; std::vector<u8> g_variant;
};

}  // namespace

extern "C" {

void *oracle_create(const hqtick_config *cfg) { Ctx *c = new Ctx(); c->cfg = *cfg; return c; }
void oracle_destroy(void *p) { delete (Ctx *)p; }
const char *oracle_last_error(void *p) { return ((Ctx *)p)->err.c_str(); }

// create_task_batches only (tier T1)
// The next oracle_tick on this ctx skips the MILP: its solution is the given single-node counts (worker = index into the snapshot's worker arrays).
// Batches, model (for the column mapping), decode into the reference's Map orders, create_task_mapping, proactive filling and the record order are
// computed as always — what remains is exactly "the reference's mapping stage given these counts" (tier T3).
void oracle_set_given_counts(void *p, uint32_t n, const uint32_t *rq, const uint8_t *variant, const uint32_t *worker, const uint32_t *value, int is_optimal) {
    Ctx *c = (Ctx *)p;
    c->g_rq.assign(rq, rq + n); c->g_variant.assign(variant, variant + n); c->g_worker.assign(worker, worker + n); c->g_value.assign(value, value + n);
    c->g_optimal = is_optimal != 0; c->given = true;
}

int oracle_batches(void *p, const hqtick_snapshot *s, hqtick_result *res) {
    Ctx *c = (Ctx *)p; State st;
    if (!load_state(st, s, &c->cfg)) { c->err = "invalid snapshot"; return HQTICK_E_INVALID; }
    auto batches = create_task_batches(st, nullptr);
    export_batches(c->out, batches);
    memset(res, 0, sizeof(*res));
    Out &o = c->out;
    res->n_batches = (u32)o.batch_rq.size(); res->batch_rq = o.batch_rq.data(); res->batch_size = o.batch_size.data();
    res->batch_limit = o.batch_limit.data(); res->batch_limit_reached = o.batch_lr.data(); res->batch_is_blocker = o.batch_blk.data();
    res->batch_cut_off = o.batch_cut_off.data(); res->cut_size = o.cut_size.data(); res->cut_blocker_off = o.cut_blocker_off.data();
    res->blocker_rq = o.blocker_rq.data(); res->blocker_size = o.blocker_size.data();
    return 0;
}

// run_scheduling_inner  scheduler/main.rs:50-72
int oracle_tick(void *p, const hqtick_snapshot *s, oracle_solve_fn fn, void *user, hqtick_result *res) {
    Ctx *c = (Ctx *)p; Out &o = c->out; State st; g_error = 0;
    double t0 = now_us();
    if (!load_state(st, s, &c->cfg)) { c->err = "invalid snapshot"; return HQTICK_E_INVALID; }
    double t1 = now_us();
    u32 W = (u32)st.workers.size(), R = st.R;
    auto batches = create_task_batches(st, nullptr);
    double t2 = now_us();
    export_batches(o, batches);
    Solver solver{fn, user}; GapCache gaps;
    Solution sol;
    std::vector<const WorkerS *> workers;
    if (!st.rqs.empty()) {  // solver.rs:53-55
        BuiltModel bm = build_model(st, batches, nullptr, gaps, solver, workers);
        if (g_error) { c->err = g_errmsg; return g_error; }
        double gap_solve = solver.solve_us;
        std::vector<double> x; double objv = 0; int is_opt = 1;
        bool ok;
        if (c->given) {  // T3 "given counts" (oracle_set_given_counts): the placement comes from the caller, everything after it is the reference's
            x.assign((size_t)bm.m.ncols(), 0.0);
            ok = true;
            for (size_t i = 0; i < c->g_rq.size(); i++) {
                auto it = bm.placements.find({c->g_worker[i], c->g_rq[i], c->g_variant[i]});
                if (it == bm.placements.end()) { c->err = "given count for a (worker, rq, variant) without a placement column"; c->given = false; return HQTICK_E_INVALID; }
                x[(size_t)it->second] = (double)c->g_value[i];
            }
            for (int j = 0; j < bm.m.ncols(); j++) objv += bm.m.obj[(size_t)j] * x[(size_t)j];
            is_opt = c->g_optimal ? 1 : 0;
            c->given = false;
        } else
        ok = run_solver(solver, bm.m, st.cfg.mip_time_limit_s, x, &objv, &is_opt);  // :433-437
        (void)gap_solve;
        if (ok) sol = decode(st, batches, bm, workers, nullptr, x, is_opt != 0);
        o.last_x = x; o.last_obj = objv; o.last_model = std::move(bm);
    }
    double t3 = now_us();
    c->t_solve = solver.solve_us; c->t_model = (t3 - t2) - solver.solve_us;
    int status = HQTICK_DONE;  // main.rs:57-68
    if (!sol.is_optimal) status = sol.empty() ? HQTICK_NO_PROGRESS : HQTICK_NEED_MORE_COMPUTE;

    // ---- a8: create_task_mapping  mapping.rs:23-157 ----
    std::vector<std::vector<Rec>> assigned(W), prefills(W);
    std::vector<std::vector<u64>> retracts(W);
    o.redirect_task.clear(); o.redirect_worker.clear(); o.redirect_variant.clear(); o.redirect_kind.clear();
    // ready tasks in state Retracting{old}: task -> (old worker, current redirect target or none, its variant)   mapping.rs:66-80
    struct Retr { u32 old, target; u8 variant; };
    std::map<u64, Retr> retracting;
    for (u32 i = 0; i < s->n_retracting; i++)
        retracting[s->retracting_task[i]] = Retr{s->retracting_worker[i], s->retracting_redirect_worker ? s->retracting_redirect_worker[i] : HQ_NO_WORKER,
                                                 s->retracting_redirect_variant ? s->retracting_redirect_variant[i] : (u8)0};
    o.count_rq.clear(); o.count_variant.clear(); o.count_worker.clear(); o.count_value.clear();
    for (size_t ki = 0; ki < sol.keys.size(); ki++) {
        u32 rq = sol.keys[ki].first; u8 vi = sol.keys[ki].second;
        auto counts = sol.counts[ki];
        for (auto &cw : counts) { o.count_rq.push_back(rq); o.count_variant.push_back(vi); o.count_worker.push_back(cw.first); o.count_value.push_back(cw.second); }
        const Variant &rqd = st.rqs[rq].v[vi];
        u32 sum = 0; for (auto &cw : counts) sum += cw.second;
        std::vector<u64> tasks; std::vector<u32> oldw;
        if (!st.queues[rq].take_tasks(sum, tasks, oldw)) { c->err = "queue underflow (reference panics at taskqueue.rs:327)"; return HQTICK_E_QUEUE_UNDERFLOW; }
        size_t ti = 0;
        if (!tasks.empty()) {
            bool done = false;
            while (!done) {  // 'outer loop  :42-124
                for (auto &cw : counts) {
                    if (cw.second == 0) continue;
                    cw.second--;
                    u64 task = tasks[ti]; u32 w = cw.first;
                    st.workers[w].free.remove(rqd);                   // insert_sn_task  server/worker.rs:188-196
                    st.workers[w].assigned.push_back({rq, vi});
                    auto rt = oldw[ti] == HQ_NO_WORKER ? retracting.find(task) : retracting.end();
                    if (rt != retracting.end()) {                     // Retracting{old} stays Retracting  :66-80
                        Retr &r = rt->second;
                        if (r.old != w) {                             // redirects.insert(task, (w, v)); a previous target gives the task back
                            if (r.target != HQ_NO_WORKER) {
                                st.workers[r.target].free.add(st.rqs[rq].v[r.variant], st.workers[r.target].total);  // remove_sn_task  server/worker.rs:223-234
                                auto &as = st.workers[r.target].assigned;
                                auto it = std::find(as.begin(), as.end(), std::make_pair(rq, r.variant)); if (it != as.end()) as.erase(it);
                            }
                            r.target = w; r.variant = vi;
                            o.redirect_task.push_back(task); o.redirect_worker.push_back(w); o.redirect_variant.push_back(vi); o.redirect_kind.push_back(HQ_REDIRECT_RETARGET);
                        } else {
                            o.redirect_task.push_back(task); o.redirect_worker.push_back(w); o.redirect_variant.push_back(vi); o.redirect_kind.push_back(HQ_REDIRECT_SAME_WORKER);
                        }
                    } else if (oldw[ti] == HQ_NO_WORKER) {            // Waiting -> Assigned  :53-65
                        assigned[w].push_back(Rec{task, vi, HQ_REC_ASSIGN, st.task_priority[task], rq});
                    } else {                                          // Prefilled{old} -> Retracting  :81-101
                        u32 old = oldw[ti];
                        auto &pf = st.workers[old].prefilled_rq;       // remove_prefill_task
                        auto it = std::find(pf.begin(), pf.end(), rq); if (it != pf.end()) pf.erase(it);
                        retracts[old].push_back(task);
                        o.redirect_task.push_back(task); o.redirect_worker.push_back(w); o.redirect_variant.push_back(vi); o.redirect_kind.push_back(HQ_REDIRECT_FROM_PREFILL);
                    }
                    ti++;
                    if (ti >= tasks.size()) { done = true; break; }
                }
            }
        }
    }
    for (u32 w = 0; w < W; w++)  // :128-131 stable sort by Reverse(priority)
        std::stable_sort(assigned[w].begin(), assigned[w].end(), [](const Rec &a, const Rec &b) { return a.priority > b.priority; });
    o.mn_task.clear(); o.mn_worker_off.assign(1, 0); o.mn_worker.clear();
    for (size_t ki = 0; ki < sol.mn_keys.size(); ki++) {  // :133-154
        u32 rq = sol.mn_keys[ki].first;
        for (auto &ws : sol.mn_workers[ki]) {
            u64 task;
            if (!st.queues[rq].take_one(&task)) { c->err = "mn queue underflow"; return HQTICK_E_QUEUE_UNDERFLOW; }
            for (u32 w : ws) { st.workers[w].flags &= ~HQ_WORKER_SN; o.mn_worker.push_back(w); }   // set_mn_task
            o.mn_task.push_back(task); o.mn_worker_off.push_back((u32)o.mn_worker.size());
        }
    }
    // ---- a9: process_proactive_filling  mapping.rs:159-234 ----
    {
        u64 top = 0;  // task_queues.top_priority()  taskqueue.rs:62-68
        for (auto &q : st.queues) { u64 p; if (q.top_priority(&p)) top = std::max(top, p); }
        for (auto &q : st.queues) {
            u64 p; if (!q.top_priority(&p) || p != top) continue;
            u32 tsz = q.top_size_no_prefill();
            u32 size = tsz > st.cfg.proactive_filling_reserve ? tsz - st.cfg.proactive_filling_reserve : 0;
            if (size == 0) continue;
            std::vector<u32> elig;
            for (u32 wi : st.worker_map_order) {  // worker_map.values_mut(): hash order
                WorkerS &w = st.workers[wi];
                if (!w.is_sn()) continue;
                bool got = false;  // mapping.workers[w].assigned holds a task of this rq (assigned in THIS tick)  :184-193
                for (auto &r : assigned[wi]) if (r.rq == q.rq) { got = true; break; }
                if (!got) continue;
                if (std::find(w.prefilled_rq.begin(), w.prefilled_rq.end(), q.rq) != w.prefilled_rq.end()) continue;
                elig.push_back(wi);
            }
            if (elig.empty()) continue;
            u32 psz = std::min(size / (u32)elig.size(), st.cfg.proactive_filling_max);
            if (psz == 0) continue;
            for (u32 wi : elig) {
                auto tasks = q.take_for_prefill(psz);
                for (u64 t : tasks) if (retracting.count(t)) { c->err = "a Retracting task reached take_tasks_for_prefill: the reference asserts task.is_waiting() (mapping.rs:221)"; return HQTICK_E_UNSUPPORTED; }
                for (u64 t : tasks) { st.workers[wi].prefilled_rq.push_back(q.rq); prefills[wi].push_back(Rec{t, 0xFF, HQ_REC_PREFILL, 0, q.rq}); }
            }
        }
    }
    // ---- a10: per-worker record order of send_messages  mapping.rs:259-292 ----
    o.rec_off.assign(1, 0); o.rec_task.clear(); o.rec_variant.clear(); o.rec_kind.clear();
    o.retract_off.assign(1, 0); o.retract_task.clear();
    for (u32 w = 0; w < W; w++) {
        for (auto &r : prefills[w]) { o.rec_task.push_back(r.task); o.rec_variant.push_back(0xFF); o.rec_kind.push_back(HQ_REC_PREFILL); }
        for (auto &r : assigned[w]) { o.rec_task.push_back(r.task); o.rec_variant.push_back(r.variant); o.rec_kind.push_back(HQ_REC_ASSIGN); }
        o.rec_off.push_back((u32)o.rec_task.size());
        for (u64 t : retracts[w]) o.retract_task.push_back(t);
        o.retract_off.push_back((u32)o.retract_task.size());
    }
    o.new_free.resize((size_t)W * R);
    for (u32 w = 0; w < W; w++) for (u32 r = 0; r < R; r++) o.new_free[(size_t)w * R + r] = st.workers[w].free.get(r);
    double t4 = now_us();

    memset(res, 0, sizeof(*res));
    res->status = status; res->is_optimal = sol.is_optimal; res->is_canonical = sol.is_optimal;  // the Python driver applies the tie-break (oracle.py)
    res->n_batches = (u32)o.batch_rq.size(); res->batch_rq = o.batch_rq.data(); res->batch_size = o.batch_size.data();
    res->batch_limit = o.batch_limit.data(); res->batch_limit_reached = o.batch_lr.data(); res->batch_is_blocker = o.batch_blk.data();
    res->batch_cut_off = o.batch_cut_off.data(); res->cut_size = o.cut_size.data(); res->cut_blocker_off = o.cut_blocker_off.data();
    res->blocker_rq = o.blocker_rq.data(); res->blocker_size = o.blocker_size.data();
    res->n_counts = (u32)o.count_rq.size(); res->count_rq = o.count_rq.data(); res->count_variant = o.count_variant.data();
    res->count_worker = o.count_worker.data(); res->count_value = o.count_value.data();
    res->rec_off = o.rec_off.data(); res->rec_task = o.rec_task.data(); res->rec_variant = o.rec_variant.data(); res->rec_kind = o.rec_kind.data();
    res->retract_off = o.retract_off.data(); res->retract_task = o.retract_task.data();
    res->n_redirects = (u32)o.redirect_task.size(); res->redirect_task = o.redirect_task.data(); res->redirect_worker = o.redirect_worker.data(); res->redirect_variant = o.redirect_variant.data(); res->redirect_kind = o.redirect_kind.data();
    res->n_mn = (u32)o.mn_task.size(); res->mn_task = o.mn_task.data(); res->mn_worker_off = o.mn_worker_off.data(); res->mn_worker = o.mn_worker.data();
    res->new_free = o.new_free.data();
    res->t_total_us = t4 - t0; res->t_scan_us = t1 - t0; res->t_batches_us = t2 - t1; res->t_solve_us = t3 - t2; res->t_mapping_us = t4 - t3;
    c->t_load = t1 - t0; c->t_batches = t2 - t1; c->t_mapping = t4 - t3;
    return status;
}

// compute_new_worker_query  scheduler/query.rs:12-131 (stages only; fake workers supplied by the caller)
int oracle_query(void *p, const hqtick_snapshot *s, const hqtick_query_workers *fake, oracle_solve_fn fn, void *user, hqtick_query_result *res) {
    Ctx *c = (Ctx *)p; Out &o = c->out; State st; g_error = 0;
    if (!load_state(st, s, &c->cfg)) { c->err = "invalid snapshot"; return HQTICK_E_INVALID; }
    u32 R = st.R;
    std::vector<WorkerS> fw(fake->n_workers);
    for (u32 i = 0; i < fake->n_workers; i++) {
        WorkerS &w = fw[i]; w.id = fake->worker_id[i];
        w.total.a.assign(fake->worker_total + (size_t)i * R, fake->worker_total + (size_t)(i + 1) * R);
        w.free = w.total; w.remaining_ns = fake->worker_remaining_ns ? fake->worker_remaining_ns[i] : HQ_NO_TIME_LIMIT;
        w.min_util = fake->worker_min_utilization ? fake->worker_min_utilization[i] : 0.0f;
        w.flags = HQ_WORKER_SN; w.group = UINT32_MAX;
    }
    auto batches = create_task_batches(st, &fw);
    export_batches(o, batches);
    Solver solver{fn, user}; GapCache gaps; Solution sol;
    std::vector<const WorkerS *> workers;
    if (!st.rqs.empty()) {
        BuiltModel bm = build_model(st, batches, &fw, gaps, solver, workers);
        if (g_error) { c->err = g_errmsg; return g_error; }
        std::vector<double> x; double objv = 0; int is_opt = 1;
        if (run_solver(solver, bm.m, st.cfg.mip_time_limit_s, x, &objv, &is_opt)) sol = decode(st, batches, bm, workers, &fw, x, is_opt != 0);
        o.last_x = x; o.last_obj = objv; o.last_model = std::move(bm);
    }
    o.q_loaded.assign(fake->n_workers, 0);
    for (auto &cs : sol.counts) for (auto &cw : cs) if (cw.second > 0) o.q_loaded[cw.first] = 1;  // query.rs:73-81
    res->n_workers = fake->n_workers; res->is_loaded = o.q_loaded.data(); res->is_optimal = sol.is_optimal;
    return 0;
}

// ---- model / timing introspection for tests and the cpu_baseline leg ----
typedef struct oracle_model_view {
    int ncols, nrows;
    const double *obj; const uint8_t *col_kind; const uint8_t *col_type; const uint32_t *col_worker; const uint32_t *col_rq; const uint8_t *col_variant;
    const uint8_t *row_type; const double *rhs; const int *row_off; const int *row_col; const double *row_coef;
    const double *x; double objective;
} oracle_model_view;

void oracle_last_model(void *p, oracle_model_view *v) {
    Ctx *c = (Ctx *)p; Model &m = c->out.last_model.m;
    v->ncols = m.ncols(); v->nrows = m.nrows(); v->obj = m.obj.data(); v->col_kind = m.kind.data(); v->col_type = m.ctype.data();
    v->col_worker = m.cworker.data(); v->col_rq = m.crq.data(); v->col_variant = m.cvariant.data();
    v->row_type = m.rtype.data(); v->rhs = m.rhs.data(); v->row_off = m.roff.data(); v->row_col = m.rcol.data(); v->row_coef = m.rcoef.data();
    v->x = c->out.last_x.data(); v->objective = c->out.last_obj;
}
void oracle_stage_times(void *p, double *t5) {
    Ctx *c = (Ctx *)p; t5[0] = c->t_load; t5[1] = c->t_batches; t5[2] = c->t_model; t5[3] = c->t_solve; t5[4] = c->t_mapping;
}

// ---- unit-level entry points pinned by reference unit tests ----
// prune_progressive over 0..n  (scheduler/batches.rs:223-250 test_prune_progressive)
int oracle_prune_progressive(uint32_t n, uint32_t prefix, uint32_t limit, uint32_t *out) {
    std::vector<u32> v(n); for (u32 i = 0; i < n; i++) v[i] = i;
    prune_progressive(v, prefix, limit);
    for (size_t i = 0; i < v.size(); i++) out[i] = v[i];
    return (int)v.size();
}
// GapCache::get_gap on a worker's total resources with no assigned tasks (scheduler/gap.rs:176-246 test_compute_gap)
int oracle_gap(void *p, const hqtick_snapshot *s, uint32_t high_rq, uint32_t low_rq, uint32_t worker_index, oracle_solve_fn fn, void *user) {
    Ctx *c = (Ctx *)p; State st;
    if (!load_state(st, s, &c->cfg)) return -1;
    Solver solver{fn, user}; GapCache gaps;
    return (int)get_gap(gaps, st, high_rq, low_rq, st.workers[worker_index].total, {}, solver);
}
// hashbrown/fxhash iteration-order emulation: order of a Map<WorkerId,_> / Set<TaskId> built by inserting `keys` in order
void oracle_hb_order_u32(const uint32_t *keys, uint32_t n, uint32_t *out) {
    hb::WorkerIdTable t; for (u32 i = 0; i < n; i++) t.insert(keys[i]);
    size_t k = 0; t.for_each([&](u64 key) { out[k++] = (u32)key; });
}
void oracle_hb_order_taskid(const uint64_t *keys, uint32_t n, uint64_t *out) {
    hb::TaskIdTable t; for (u32 i = 0; i < n; i++) t.insert(keys[i]);
    size_t k = 0; t.for_each([&](u64 key) { out[k++] = key; });
}
void oracle_hb_order_rqv(const uint32_t *rq, const uint8_t *v, uint32_t n, uint32_t *out_rq, uint8_t *out_v) {
    hb::RqVTable t; for (u32 i = 0; i < n; i++) t.insert(((u64)rq[i] << 8) | v[i]);
    size_t k = 0; t.for_each([&](u64 key) { out_rq[k] = (u32)(key >> 8); out_v[k] = (u8)key; k++; });
}
uint64_t oracle_fx_u32(uint32_t v) { return hb::fx_u32(v); }

}  // extern "C"
