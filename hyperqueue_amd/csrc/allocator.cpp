// libhqalloc.so -- worker-side resource allocator behind include/hqalloc.h.
//
// Follows, under /root/reference/crates/tako/src/internal/worker/resources/: allocator.rs (has_resources_for_request,
// claim_resources, try_allocate, release_allocation), pool.rs (claim orders of the index / group / sum pools),
// concise.rs (concise free state) and groups.rs (the 0/1 group model, solved there by HiGHS and here by csrc/milp.cpp,
// whose canonical optimum makes the chosen groups a function of the model alone).  Host-only: no HIP.
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <iterator>
#include <map>
#include <stdexcept>
#include <string>
#include <tuple>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../include/hqalloc.h"
#include "hb_table.h"
#include "milp.h"

namespace {

constexpr uint64_t FPU = HQALLOC_FRACTIONS_PER_UNIT;

struct Bug : std::runtime_error {
    using std::runtime_error::runtime_error;
};
#define HQ_ASSERT(cond)                                                                          \
    do {                                                                                         \
        if (!(cond)) throw Bug(std::string("assertion failed: " #cond " (allocator.cpp:") + std::to_string(__LINE__) + ")"); \
    } while (0)

inline uint64_t make_amount(uint64_t units, uint64_t fractions) { return units * FPU + fractions; }

struct AIndex {
    uint32_t index, group, fractions;
};
struct RAlloc {
    uint32_t resource;
    uint64_t amount;
    std::vector<AIndex> indices;
};
using Alloc = std::vector<RAlloc>;

struct Entry {
    uint32_t resource;
    uint8_t kind;
    uint64_t amount;
    bool operator<(const Entry &o) const { return std::tie(resource, kind, amount) < std::tie(o.resource, o.kind, o.amount); }
};
inline bool coupling_kind(uint8_t k) { return k == HQALLOC_COMPACT || k == HQALLOC_TIGHT || k == HQALLOC_FORCE_COMPACT || k == HQALLOC_FORCE_TIGHT; }
inline bool forced_kind(uint8_t k) { return k == HQALLOC_FORCE_COMPACT || k == HQALLOC_FORCE_TIGHT; }

// ---- pool.rs ---------------------------------------------------------------------------------------------------
struct Pool {
    uint8_t kind = HQALLOC_POOL_EMPTY;
    uint64_t full_size = 0, free = 0;               // free: SUM only
    std::vector<std::vector<uint32_t>> indices;     // per group, popped from the back
    std::vector<hqhb::U32Map> fractions;            // per group: partially used index -> what is left of it

    // pool.rs:372-380: first minimum in iteration order among the entries with enough left
    static bool best_fraction_match(hqhb::U32Map &m, uint32_t fractions, uint32_t &key) {
        bool found = false;
        uint32_t best = 0;
        m.for_each([&](uint32_t k, uint32_t f) {
            if (f >= fractions && (!found || f < best)) {
                found = true;
                best = f;
                key = k;
            }
        });
        return found;
    }
    void take_indices(uint32_t g, uint64_t units, std::vector<AIndex> &out) {  // pool.rs:305-318
        for (uint64_t i = 0; i < units; i++) {
            HQ_ASSERT(!indices[g].empty());
            out.push_back({indices[g].back(), g, 0});
            indices[g].pop_back();
        }
    }
    bool try_take_fraction(uint32_t g, uint32_t fr, std::vector<AIndex> &out) {  // pool.rs:349-370
        uint32_t key;
        if (fr == 0 || !best_fraction_match(fractions[g], fr, key)) return false;
        *fractions[g].get(key) -= fr;
        out.push_back({key, g, fr});
        return true;
    }
    bool split_index(uint32_t g, uint32_t fr, std::vector<AIndex> &out) {
        if (indices[g].empty()) return false;
        uint32_t index = indices[g].back();
        indices[g].pop_back();
        fractions[g].insert(index, (uint32_t)FPU - fr);
        out.push_back({index, g, fr});
        return true;
    }
    void take_fraction_index_or_split(uint32_t g, uint32_t fr, std::vector<AIndex> &out) {  // pool.rs:320-347
        if (fr == 0) return;
        if (!try_take_fraction(g, fr, out)) HQ_ASSERT(split_index(g, fr, out));
    }
    std::vector<uint64_t> group_amounts() const {  // pool.rs:27-37
        std::vector<uint64_t> a(indices.size());
        for (size_t g = 0; g < indices.size(); g++) a[g] = make_amount(indices[g].size(), fractions[g].max_value());
        return a;
    }
    std::vector<AIndex> claim_all_from_groups() {  // pool.rs:164-178
        std::vector<AIndex> out;
        for (uint32_t g = 0; g < indices.size(); g++) {
            for (uint32_t i : indices[g]) out.push_back({i, g, 0});
            indices[g].clear();
        }
        return out;
    }
    std::vector<AIndex> claim_scatter_from_groups(uint64_t amount, const std::vector<uint32_t> *group_set) {  // pool.rs:180-232
        std::vector<AIndex> out;
        uint64_t units = amount / FPU;
        uint32_t fr = (uint32_t)(amount % FPU);
        const size_t n = group_set ? group_set->size() : indices.size();
        HQ_ASSERT(n > 0);
        // the reference loops until served; availability was checked before, but guard against an endless walk
        size_t idle_rounds = 0, pos = 0;
        while (units > 0 || fr > 0) {
            const uint32_t g = group_set ? (*group_set)[pos] : (uint32_t)pos;
            bool progress = false;
            if (units > 0) {
                if (!indices[g].empty()) {
                    out.push_back({indices[g].back(), g, 0});
                    indices[g].pop_back();
                    units--;
                    progress = true;
                }
            } else if (try_take_fraction(g, fr, out) || split_index(g, fr, out)) {
                fr = 0;
                progress = true;
            }
            idle_rounds = progress ? 0 : idle_rounds + 1;
            HQ_ASSERT(idle_rounds <= n);
            pos = (pos + 1) % n;
        }
        std::stable_sort(out.begin(), out.end(), [](const AIndex &a, const AIndex &b) {
            return std::tie(a.fractions, a.group, a.index) < std::tie(b.fractions, b.group, b.index);
        });
        return out;
    }
    std::vector<AIndex> claim_compact_from_groups(uint64_t amount, const std::vector<uint32_t> *group_set) {  // pool.rs:234-303
        std::vector<AIndex> out;
        uint64_t remaining = amount;
        long fraction_idx = -1;
        std::vector<uint64_t> amounts = group_amounts();
        auto allowed = [&](size_t g) { return !group_set || std::find(group_set->begin(), group_set->end(), (uint32_t)g) != group_set->end(); };
        for (;;) {
            long fit = -1;  // min_by_key: the first minimum
            for (size_t g = 0; g < amounts.size(); g++)
                if (amounts[g] >= remaining && allowed(g) && (fit < 0 || amounts[g] < amounts[fit])) fit = (long)g;
            if (fit >= 0) {
                take_indices((uint32_t)fit, remaining / FPU, out);
                take_fraction_index_or_split((uint32_t)fit, (uint32_t)(remaining % FPU), out);
                break;
            }
            long big = -1;  // max_by_key: the last maximum
            for (size_t g = 0; g < amounts.size(); g++)
                if (allowed(g) && (big < 0 || amounts[g] >= amounts[big])) big = (long)g;
            HQ_ASSERT(big >= 0 && amounts[big] > 0);  // nothing left in the allowed groups: the reference would spin here
            amounts[big] = 0;
            uint64_t units = remaining / FPU;
            uint32_t fr = (uint32_t)(remaining % FPU);
            const uint64_t size = indices[big].size();
            HQ_ASSERT(units >= size);
            units -= size;
            take_indices((uint32_t)big, size, out);
            if (try_take_fraction((uint32_t)big, fr, out)) {
                fraction_idx = (long)out.size() - 1;
                fr = 0;
            }
            remaining = make_amount(units, fr);
        }
        if (fraction_idx >= 0) std::swap(out[fraction_idx], out.back());
        return out;
    }
    RAlloc claim_with_group_mask(uint32_t resource, uint8_t kind, uint64_t amount, const std::vector<uint32_t> &gs) {  // pool.rs:382-405
        HQ_ASSERT(this->kind == HQALLOC_POOL_GROUPS);
        if (kind == HQALLOC_COMPACT || kind == HQALLOC_FORCE_COMPACT) return {resource, amount, claim_scatter_from_groups(amount, &gs)};
        HQ_ASSERT(kind == HQALLOC_TIGHT || kind == HQALLOC_FORCE_TIGHT);
        return {resource, amount, claim_compact_from_groups(amount, &gs)};
    }
    RAlloc claim(uint32_t resource, uint8_t rkind, uint64_t amount) {  // pool.rs:407-455
        if (kind == HQALLOC_POOL_INDICES) {
            const uint64_t a = rkind == HQALLOC_ALL ? full_size : amount;
            std::vector<AIndex> out;
            take_indices(0, a / FPU, out);
            take_fraction_index_or_split(0, (uint32_t)(a % FPU), out);
            return {resource, a, out};
        }
        if (kind == HQALLOC_POOL_GROUPS) {
            if (rkind == HQALLOC_SCATTER) return {resource, amount, claim_scatter_from_groups(amount, nullptr)};
            HQ_ASSERT(rkind == HQALLOC_ALL);  // the other kinds are claimed through the coupled solver
            return {resource, full_size, claim_all_from_groups()};
        }
        HQ_ASSERT(kind == HQALLOC_POOL_SUM);
        const uint64_t a = rkind == HQALLOC_ALL ? full_size : amount;
        HQ_ASSERT(free >= a);
        free -= a;
        return {resource, a, {}};
    }
    void release(const RAlloc &al) {  // pool.rs:457-501
        if (kind == HQALLOC_POOL_SUM) {
            free += al.amount;
            HQ_ASSERT(free <= full_size && al.indices.empty());
            return;
        }
        HQ_ASSERT(kind == HQALLOC_POOL_INDICES || kind == HQALLOC_POOL_GROUPS);
        for (auto it = al.indices.rbegin(); it != al.indices.rend(); ++it) {  // taken by pop(): returned in reverse
            HQ_ASSERT(it->group < indices.size());
            if (it->fractions == 0) {
                indices[it->group].push_back(it->index);
                continue;
            }
            uint32_t *f = fractions[it->group].get(it->index);
            HQ_ASSERT(f != nullptr);
            *f += it->fractions;
            if (*f == FPU) {
                fractions[it->group].remove(it->index);
                indices[it->group].push_back(it->index);
            }
        }
    }
    uint64_t current_free() const {  // pool.rs:555-566
        if (kind == HQALLOC_POOL_SUM) return free;
        uint64_t n = 0;
        for (auto &g : indices) n += g.size();
        return n * FPU;
    }
};

// ---- concise.rs ------------------------------------------------------------------------------------------------
struct CGroup {
    uint64_t units = 0;
    std::map<uint32_t, uint32_t> fractions;  // the order of this map is never observed (max / sum only)
};
using CState = std::vector<CGroup>;

CState concise_of(const Pool &p) {  // pool.rs:135-162
    CState st;
    if (p.kind == HQALLOC_POOL_EMPTY) return st;
    if (p.kind == HQALLOC_POOL_SUM) {
        CGroup g;
        g.units = p.free / FPU;
        if (p.free % FPU) g.fractions[0] = (uint32_t)(p.free % FPU);
        st.push_back(g);
        return st;
    }
    for (size_t i = 0; i < p.indices.size(); i++) {
        CGroup g;
        g.units = p.indices[i].size();
        p.fractions[i].for_each([&](uint32_t k, uint32_t v) { g.fractions[k] = v; });
        st.push_back(g);
    }
    return st;
}
void remove_fractions(CGroup &g, uint32_t index, uint32_t fr) {  // concise.rs:31-46
    uint32_t &old = g.fractions[index];
    if (old < fr) {
        old = (uint32_t)FPU + old - fr;
        HQ_ASSERT(g.units > 0);
        g.units--;
    } else old -= fr;
}
void add_fractions(CGroup &g, uint32_t index, uint32_t fr) {  // concise.rs:78-92
    uint32_t &old = g.fractions[index];
    old += fr;
    if (old >= FPU) {
        old -= (uint32_t)FPU;
        g.units++;
    }
}
void concise_apply(CState &st, const RAlloc &ra, bool add) {  // concise.rs:48-76, 94-119
    auto frac = [&](CGroup &g, uint32_t i, uint32_t f) { add ? add_fractions(g, i, f) : remove_fractions(g, i, f); };
    if (st.size() == 1) {
        const uint64_t units = ra.amount / FPU;
        const uint32_t fr = (uint32_t)(ra.amount % FPU);
        if (add) st[0].units += units;
        else {
            HQ_ASSERT(st[0].units >= units);
            st[0].units -= units;
        }
        if (fr > 0) {
            if (ra.indices.empty()) frac(st[0], 0, fr);
            else
                for (auto it = ra.indices.rbegin(); it != ra.indices.rend() && it->fractions != 0; ++it) frac(st[0], it->index, it->fractions);
        }
        return;
    }
    for (const AIndex &ai : ra.indices) {
        HQ_ASSERT(ai.group < st.size());
        if (ai.fractions == 0) {
            if (add) st[ai.group].units++;
            else {
                HQ_ASSERT(st[ai.group].units > 0);
                st[ai.group].units--;
            }
        } else frac(st[ai.group], ai.index, ai.fractions);
    }
}
uint32_t max_fraction(const CGroup &g) {
    uint32_t m = 0;
    for (auto &kv : g.fractions) m = std::max(m, kv.second);
    return m;
}
uint64_t amount_max_alloc(const CState &st) {  // concise.rs:130-134
    uint64_t units = 0;
    uint32_t fr = 0;
    for (auto &g : st) {
        units += g.units;
        fr = std::max(fr, max_fraction(g));
    }
    return make_amount(units, fr);
}
uint64_t amount_sum(const CState &st) {  // concise.rs:148-152
    uint64_t s = 0;
    for (auto &g : st) {
        s += g.units * FPU;
        for (auto &kv : g.fractions) s += kv.second;
    }
    return s;
}

struct Coupling {
    uint32_t r1, g1, r2, g2;
    double weight;
};

// ---- groups.rs:61-155 -------------------------------------------------------------------------------------------
bool group_solver(const std::vector<CState> &free, const std::vector<Entry> &entries, const std::vector<Coupling> &weights,
                  std::vector<std::vector<uint32_t>> &selected, double &objective) {
    hqmilp::Model m;
    std::vector<std::vector<int>> vars(entries.size());
    for (size_t e = 0; e < entries.size(); e++) {
        const CState &r = free[entries[e].resource];
        const uint64_t units = entries[e].amount / FPU;
        const uint32_t fr = (uint32_t)(entries[e].amount % FPU);
        if (fr == 0) {  // groups.rs:72-84
            for (auto &g : r) vars[e].push_back(m.add_col(-1024.0 - (double)g.units / 32.0, hqmilp::COL_BOOL));
            m.begin_row(hqmilp::ROW_MIN, (double)units);
            for (size_t g = 0; g < r.size(); g++) m.term(vars[e][g], (double)r[g].units);
            m.end_row();
        } else {  // groups.rs:85-118
            bool second = false;
            std::vector<uint32_t> maxf(r.size());
            for (size_t g = 0; g < r.size(); g++) {
                maxf[g] = max_fraction(r[g]);
                if (maxf[g] >= fr) {
                    second = true;
                    vars[e].push_back(m.add_col(-1024.0 + ((double)maxf[g] / ((double)FPU / 16.0)), hqmilp::COL_BOOL));
                } else vars[e].push_back(m.add_col(-1024.0, hqmilp::COL_BOOL));
            }
            m.begin_row(hqmilp::ROW_MIN, (double)(units + 1));
            for (size_t g = 0; g < r.size(); g++) m.term(vars[e][g], (double)(maxf[g] >= fr ? r[g].units + 1 : r[g].units));
            m.end_row();
            if (units > 0 && second) {
                m.begin_row(hqmilp::ROW_MIN, (double)units);
                for (size_t g = 0; g < r.size(); g++) m.term(vars[e][g], (double)r[g].units);
                m.end_row();
            }
        }
    }
    for (const Coupling &w : weights) {  // groups.rs:121-141
        long p1 = -1, p2 = -1;
        for (size_t e = 0; e < entries.size(); e++) {
            if (p1 < 0 && entries[e].resource == w.r1) p1 = (long)e;
            if (p2 < 0 && entries[e].resource == w.r2) p2 = (long)e;
        }
        if (p1 < 0 || p2 < 0) continue;
        HQ_ASSERT(w.g1 < vars[p1].size() && w.g2 < vars[p2].size());
        // the reference's u is continuous in [0, 1]; its weight is a u16 >= 0, so u = min(v1, v2) at every optimum and a
        // 0/1 column gives the same optimal set
        const int u = m.add_col(w.weight, hqmilp::COL_BOOL);
        for (int v : {vars[p1][w.g1], vars[p2][w.g2]}) {
            m.begin_row(hqmilp::ROW_MIN, 0.0);
            m.term(v, 1.0);
            m.term(u, -1.0);
            m.end_row();
        }
    }
    const hqmilp::Result res = hqmilp::solve(m, 60.0, true);
    if (!res.feasible) return false;
    HQ_ASSERT(res.optimal);
    selected.assign(entries.size(), {});
    for (size_t e = 0; e < entries.size(); e++)
        for (size_t g = 0; g < vars[e].size(); g++)
            if (res.x[vars[e][g]] > 0.5) selected[e].push_back((uint32_t)g);
    objective = res.objective;
    return true;
}

}  // namespace

// ---- allocator.rs ------------------------------------------------------------------------------------------------
struct hqalloc_ctx {
    std::vector<Pool> pools;
    std::vector<CState> free_resources, all_resources;
    std::vector<Coupling> coupling;
    std::map<std::vector<Entry>, double> optional_objectives;  // allocator.rs:17-20
    std::unordered_map<uint64_t, Alloc> live;
    uint64_t next_id = 1;
    std::string error;

    std::vector<Entry> coupled_entries(const std::vector<Entry> &rq) const {
        std::vector<Entry> c;
        for (const Entry &e : rq)
            if (pools[e.resource].kind == HQALLOC_POOL_GROUPS && coupling_kind(e.kind)) c.push_back(e);
        return c;
    }
    bool has_resources_for_request(const std::vector<Entry> &rq) {  // allocator.rs:115-167
        for (const Entry &e : rq) {
            if (e.resource >= pools.size()) return false;
            const uint64_t max_alloc = amount_max_alloc(free_resources[e.resource]);
            if (e.kind == HQALLOC_ALL ? max_alloc != pools[e.resource].full_size : e.amount > max_alloc) return false;
        }
        const std::vector<Entry> c = coupled_entries(rq);
        if (std::none_of(c.begin(), c.end(), [](const Entry &e) { return forced_kind(e.kind); })) return true;
        std::vector<std::vector<uint32_t>> groups;
        double objective;
        if (!group_solver(free_resources, c, coupling, groups, objective)) return false;
        auto it = optional_objectives.find(rq);
        if (it == optional_objectives.end()) {
            double best;
            HQ_ASSERT(group_solver(all_resources, c, coupling, groups, best));
            it = optional_objectives.emplace(rq, best - 0.1).first;
        }
        return objective >= it->second;
    }
    Alloc claim_resources(const std::vector<Entry> &rq) {  // allocator.rs:169-204
        Alloc al;
        std::vector<Entry> c;
        for (const Entry &e : rq) {
            Pool &p = pools[e.resource];
            if (p.kind == HQALLOC_POOL_GROUPS && coupling_kind(e.kind)) {
                c.push_back(e);
                continue;
            }
            al.push_back(p.claim(e.resource, e.kind, e.amount));
        }
        if (c.empty()) return al;
        std::vector<std::vector<uint32_t>> groups;
        double objective;
        HQ_ASSERT(group_solver(free_resources, c, coupling, groups, objective));
        for (size_t i = 0; i < c.size(); i++) al.push_back(pools[c[i].resource].claim_with_group_mask(c[i].resource, c[i].kind, c[i].amount, groups[i]));
        std::sort(al.begin(), al.end(), [](const RAlloc &a, const RAlloc &b) { return a.resource < b.resource; });
        return al;
    }
};

namespace {

int fail(hqalloc_ctx *ctx, int code, const char *fmt, ...) {
    if (ctx) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        ctx->error = buf;
    }
    return code;
}

int read_request(hqalloc_ctx *ctx, const hqalloc_request *rq, std::vector<Entry> &out) {
    if (!rq || (rq->n_entries && (!rq->resource_id || !rq->kind || !rq->amount))) return fail(ctx, HQALLOC_E_INVALID, "request: NULL array");
    for (uint32_t i = 0; i < rq->n_entries; i++) {
        if (rq->kind[i] > HQALLOC_ALL) return fail(ctx, HQALLOC_E_INVALID, "request: unknown kind %u", rq->kind[i]);
        if (i && rq->resource_id[i] <= rq->resource_id[i - 1]) return fail(ctx, HQALLOC_E_INVALID, "request: entries not sorted by resource id");
        if (rq->kind[i] != HQALLOC_ALL && rq->amount[i] == 0) return fail(ctx, HQALLOC_E_INVALID, "request: zero amount (request.rs:24-32)");
        out.push_back({rq->resource_id[i], rq->kind[i], rq->kind[i] == HQALLOC_ALL ? 0 : rq->amount[i]});
    }
    return 0;
}

int write_allocation(hqalloc_ctx *ctx, const Alloc &al, uint64_t id, hqalloc_allocation *out) {
    uint32_t k = 0, n = 0;
    out->idx_off[0] = 0;
    for (const RAlloc &ra : al) {
        out->resource_id[k] = ra.resource;
        out->amount[k] = ra.amount;
        for (const AIndex &ai : ra.indices) {
            out->index[n] = ai.index;
            out->group_idx[n] = ai.group;
            out->fractions[n] = ai.fractions;
            n++;
        }
        out->idx_off[++k] = n;
    }
    out->n_resources = k;
    out->n_indices = n;
    out->allocation_id = id;
    (void)ctx;
    return 1;
}

// upper bound of what an allocation of `rq` writes, known before anything is claimed
bool fits(const hqalloc_ctx *ctx, const std::vector<Entry> &rq, const hqalloc_allocation *out) {
    uint64_t n = 0;
    for (const Entry &e : rq) {
        if (e.resource >= ctx->pools.size()) continue;
        const Pool &p = ctx->pools[e.resource];
        if (p.kind == HQALLOC_POOL_SUM || p.kind == HQALLOC_POOL_EMPTY) continue;
        const uint64_t a = e.kind == HQALLOC_ALL ? p.full_size : e.amount;
        n += a / FPU + (a % FPU ? 1 : 0);
    }
    return rq.size() <= out->cap_resources && n <= out->cap_indices;
}

bool out_ok(const hqalloc_allocation *o) {
    return o && o->resource_id && o->amount && o->idx_off && (o->cap_indices == 0 || (o->index && o->group_idx && o->fractions));
}

}  // namespace

extern "C" {

uint32_t hqalloc_abi_version(void) { return HQALLOC_ABI_VERSION; }

const char *hqalloc_last_error(const hqalloc_ctx *ctx) { return ctx ? ctx->error.c_str() : "no context"; }

int hqalloc_create(const hqalloc_descriptor *d, hqalloc_ctx **out_ctx) {
    if (!d || !out_ctx || d->abi_version != HQALLOC_ABI_VERSION || d->n_resources == 0 || !d->pool_kind || !d->group_off) return HQALLOC_E_INVALID;
    auto ctx = new hqalloc_ctx();
    try {
        ctx->pools.resize(d->n_resources);
        for (uint32_t r = 0; r < d->n_resources; r++) {
            Pool &p = ctx->pools[r];
            p.kind = d->pool_kind[r];
            const uint32_t g0 = d->group_off[r], g1 = d->group_off[r + 1];
            if (p.kind > HQALLOC_POOL_SUM || g1 < g0) throw Bug("descriptor: bad pool kind or group offsets");
            if (p.kind == HQALLOC_POOL_SUM) {
                if (!d->sum_size) throw Bug("descriptor: sum_size missing");
                p.full_size = p.free = d->sum_size[r];
            } else if (p.kind != HQALLOC_POOL_EMPTY) {
                if (!d->index_off || (p.kind == HQALLOC_POOL_INDICES ? g1 - g0 != 1 : g1 - g0 < 1)) throw Bug("descriptor: an INDICES pool has one group, a GROUPS pool at least one");
                uint64_t n = 0;
                for (uint32_t g = g0; g < g1; g++) {
                    if (d->index_off[g + 1] < d->index_off[g] || (d->index_off[g + 1] > d->index_off[g] && !d->index)) throw Bug("descriptor: bad index offsets");
                    p.indices.emplace_back(d->index + d->index_off[g], d->index + d->index_off[g + 1]);
                    n += p.indices.back().size();
                }
                p.fractions.resize(p.indices.size());
                p.full_size = n * FPU;
            }
            ctx->free_resources.push_back(concise_of(p));
        }
        ctx->all_resources = ctx->free_resources;
        for (uint32_t i = 0; i < d->n_couplings; i++) {
            Coupling c{d->coupling_resource1[i], d->coupling_group1[i], d->coupling_resource2[i], d->coupling_group2[i], (double)d->coupling_weight[i]};
            if (c.r1 >= d->n_resources || c.r2 >= d->n_resources || c.g1 >= ctx->free_resources[c.r1].size() || c.g2 >= ctx->free_resources[c.r2].size())
                throw Bug("descriptor: coupling refers to an unknown resource or group");
            ctx->coupling.push_back(c);
        }
    } catch (const std::exception &) {
        delete ctx;
        return HQALLOC_E_INVALID;
    }
    *out_ctx = ctx;
    return 0;
}

void hqalloc_destroy(hqalloc_ctx *ctx) { delete ctx; }

int hqalloc_is_enabled(hqalloc_ctx *ctx, const hqalloc_request *rq) {
    if (!ctx) return HQALLOC_E_INVALID;
    std::vector<Entry> es;
    if (int rc = read_request(ctx, rq, es)) return rc;
    try {
        return ctx->has_resources_for_request(es) ? 1 : 0;
    } catch (const std::exception &e) {
        return fail(ctx, HQALLOC_E_INTERNAL, "%s", e.what());
    }
}

int hqalloc_try_allocate(hqalloc_ctx *ctx, const hqalloc_request *rq, hqalloc_allocation *out) {
    if (!ctx) return HQALLOC_E_INVALID;
    if (!out_ok(out)) return fail(ctx, HQALLOC_E_INVALID, "allocation: NULL output array");
    std::vector<Entry> es;
    if (int rc = read_request(ctx, rq, es)) return rc;
    try {
        if (!ctx->has_resources_for_request(es)) return 0;
        if (!fits(ctx, es, out)) return fail(ctx, HQALLOC_E_CAPACITY, "allocation does not fit the output arrays");
        Alloc al = ctx->claim_resources(es);
        for (const RAlloc &ra : al) concise_apply(ctx->free_resources[ra.resource], ra, false);
        const uint64_t id = ctx->next_id++;
        write_allocation(ctx, al, id, out);
        ctx->live.emplace(id, std::move(al));
        return 1;
    } catch (const std::exception &e) {
        return fail(ctx, HQALLOC_E_INTERNAL, "%s", e.what());
    }
}

int hqalloc_release(hqalloc_ctx *ctx, uint64_t allocation_id) {
    if (!ctx) return HQALLOC_E_INVALID;
    auto it = ctx->live.find(allocation_id);
    if (it == ctx->live.end()) return fail(ctx, HQALLOC_E_INVALID, "unknown allocation id %llu", (unsigned long long)allocation_id);
    try {
        for (const RAlloc &ra : it->second) concise_apply(ctx->free_resources[ra.resource], ra, true);
        for (const RAlloc &ra : it->second) ctx->pools[ra.resource].release(ra);
        ctx->live.erase(it);
        return 0;
    } catch (const std::exception &e) {
        return fail(ctx, HQALLOC_E_INTERNAL, "%s", e.what());
    }
}

int hqalloc_force_claim_from_groups(hqalloc_ctx *ctx, uint32_t resource, uint32_t n_groups, const uint32_t *groups, uint64_t amount, hqalloc_allocation *out) {
    if (!ctx) return HQALLOC_E_INVALID;
    if (!out_ok(out) || !groups || n_groups == 0 || resource >= ctx->pools.size() || ctx->pools[resource].kind != HQALLOC_POOL_GROUPS)
        return fail(ctx, HQALLOC_E_INVALID, "force_claim: bad argument");
    std::vector<uint32_t> gs(groups, groups + n_groups);
    for (uint32_t g : gs)
        if (g >= ctx->pools[resource].indices.size()) return fail(ctx, HQALLOC_E_INVALID, "force_claim: unknown group %u", g);
    if (out->cap_resources < 1 || out->cap_indices < amount / FPU + 1) return fail(ctx, HQALLOC_E_CAPACITY, "allocation does not fit the output arrays");
    try {
        Alloc al{ctx->pools[resource].claim_with_group_mask(resource, HQALLOC_COMPACT, amount, gs)};
        concise_apply(ctx->free_resources[resource], al[0], false);
        const uint64_t id = ctx->next_id++;
        write_allocation(ctx, al, id, out);
        ctx->live.emplace(id, std::move(al));
        return 1;
    } catch (const std::exception &e) {
        return fail(ctx, HQALLOC_E_INTERNAL, "%s", e.what());
    }
}

int hqalloc_pool_free(const hqalloc_ctx *ctx, uint32_t resource, uint64_t *out_amount) {
    if (!ctx || !out_amount || resource >= ctx->pools.size()) return HQALLOC_E_INVALID;
    *out_amount = ctx->pools[resource].kind == HQALLOC_POOL_EMPTY ? 0 : ctx->pools[resource].current_free();
    return 0;
}

int hqalloc_concise_sum(const hqalloc_ctx *ctx, uint32_t resource, int which, uint64_t *out_amount) {
    if (!ctx || !out_amount || resource >= ctx->pools.size()) return HQALLOC_E_INVALID;
    *out_amount = which == 0 ? amount_sum(ctx->free_resources[resource]) : amount_sum(concise_of(ctx->pools[resource]));
    return 0;
}

int hqalloc_free_groups(const hqalloc_ctx *ctx, uint32_t resource, uint32_t cap, uint32_t *out_units, uint32_t *out_n_fraction_entries) {
    if (!ctx || resource >= ctx->pools.size() || (cap && (!out_units || !out_n_fraction_entries))) return HQALLOC_E_INVALID;
    const CState &st = ctx->free_resources[resource];
    for (uint32_t g = 0; g < st.size() && g < cap; g++) {
        out_units[g] = (uint32_t)st[g].units;
        out_n_fraction_entries[g] = (uint32_t)st[g].fractions.size();
    }
    return (int)st.size();
}

int hqalloc_free_fractions(const hqalloc_ctx *ctx, uint32_t resource, uint32_t group, uint32_t cap, uint32_t *out_index, uint32_t *out_fractions) {
    if (!ctx || resource >= ctx->pools.size() || group >= ctx->free_resources[resource].size() || (cap && (!out_index || !out_fractions))) return HQALLOC_E_INVALID;
    uint32_t n = 0;
    for (auto &kv : ctx->free_resources[resource][group].fractions) {
        if (n < cap) {
            out_index[n] = kv.first;
            out_fractions[n] = kv.second;
        }
        n++;
    }
    return (int)n;
}

int hqalloc_validate(const hqalloc_ctx *cctx) {
    if (!cctx) return HQALLOC_E_INVALID;
    hqalloc_ctx *ctx = const_cast<hqalloc_ctx *>(cctx);  // only the error text is written
    auto strip = [](CState st) {
        for (auto &g : st)
            for (auto it = g.fractions.begin(); it != g.fractions.end();) it = it->second == 0 ? g.fractions.erase(it) : std::next(it);
        return st;
    };
    for (size_t r = 0; r < ctx->pools.size(); r++) {
        const Pool &p = ctx->pools[r];
        if (p.kind == HQALLOC_POOL_SUM) {
            if (p.free > p.full_size) return fail(ctx, HQALLOC_E_INTERNAL, "resource %zu: sum pool over-released", r);
        } else if (p.kind != HQALLOC_POOL_EMPTY) {  // pool.rs:507-540
            std::vector<uint32_t> flat;
            for (auto &g : p.indices) flat.insert(flat.end(), g.begin(), g.end());
            std::vector<uint32_t> sorted = flat;
            std::sort(sorted.begin(), sorted.end());
            if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end()) return fail(ctx, HQALLOC_E_INTERNAL, "resource %zu: duplicate free index", r);
            if (flat.size() > p.full_size / FPU) return fail(ctx, HQALLOC_E_INTERNAL, "resource %zu: more free indices than the pool holds", r);
            for (auto &f : p.fractions) {
                bool bad = false;
                f.for_each([&](uint32_t k, uint32_t v) { bad |= v >= FPU || std::binary_search(sorted.begin(), sorted.end(), k); });
                if (bad) return fail(ctx, HQALLOC_E_INTERNAL, "resource %zu: a partially used index is also free, or holds a whole unit", r);
            }
        }
        const CState a = strip(concise_of(p)), b = strip(ctx->free_resources[r]);  // allocator.rs:231-234
        bool same = a.size() == b.size();
        for (size_t g = 0; same && g < a.size(); g++) same = a[g].units == b[g].units && a[g].fractions == b[g].fractions;
        if (!same) return fail(ctx, HQALLOC_E_INTERNAL, "resource %zu: concise state differs from the pool", r);
    }
    return 0;
}

}  // extern "C"
