// CPU campaign of the block solver (csrc/block_core.h, host emulation of the wavefront) against the exact host solver (csrc/milp.cpp) on
// random worker-class blocks: same canonical optimum, column for column.
//   g++ -O2 -std=c++17 -o /tmp/block_fuzz tools/block_fuzz.cpp hyperqueue_amd/csrc/milp.cpp && /tmp/block_fuzz [n_cases] [seed] [family]
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../hyperqueue_amd/csrc/block_core.h"
#include "../hyperqueue_amd/csrc/milp.h"

static uint64_t sm(uint64_t &s) { s += 0x9E3779B97F4A7C15ull; uint64_t z = s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

struct Case {
    uint32_t n_cols, R;
    std::vector<uint32_t> ent_off, ent_res, weight; std::vector<uint8_t> ent_kind; std::vector<uint64_t> ent_amount; std::vector<double> pool;
    std::vector<uint64_t> free_, total; uint64_t elig;
};

// family 0: the c3 classes on a partly used 128 c / 8 g / 512 m worker; family 1: random requests on random resources
static Case make_case(uint64_t &s, int family) {
    Case c;
    if (family == 0) {
        static const double cls[8][3] = {{1, 0, 0}, {4, 0, 0}, {2, 1, 0}, {1, .5, 0}, {1, .25, 0}, {8, 0, 64}, {16, 2, 128}, {1, 0, 1}};
        c.R = 3; c.n_cols = 8; c.ent_off.push_back(0);
        for (int j = 0; j < 8; j++) {
            for (int r = 0; r < 3; r++) if (cls[j][r] > 0) { c.ent_res.push_back(r); c.ent_kind.push_back(0); c.ent_amount.push_back((uint64_t)std::llround(cls[j][r] * 10000)); }
            c.ent_off.push_back((uint32_t)c.ent_res.size());
            c.weight.push_back(sm(s) % 4 == 0 ? 5000 + (uint32_t)(sm(s) % 20000) : 10000);
        }
        c.total = {1280000, 80000, 5120000};
        const double fill = (double)(sm(s) % 1000) / 1000.0;
        c.free_ = {(uint64_t)((sm(s) % 129) * (1.0 - 0.7 * fill)) * 10000, (uint64_t)(sm(s) % 33) * 2500, (uint64_t)(sm(s) % 513) * 10000};
        c.pool = {1024.0 * 128 * (0.05 + 0.9 * ((sm(s) % 1000) / 1000.0)), 1024.0 * 8 * (0.05 + 0.9 * ((sm(s) % 1000) / 1000.0)), 1024.0 * 512 * (0.05 + 0.9 * ((sm(s) % 1000) / 1000.0))};
        c.elig = sm(s) % 8 == 0 ? (sm(s) & 0xFF) : 0xFF;
    } else {
        c.R = 1 + (uint32_t)(sm(s) % 4); c.n_cols = 1 + (uint32_t)(sm(s) % (family == 2 ? 24 : 10)); c.ent_off.push_back(0);
        for (uint32_t r = 0; r < c.R; r++) {
            uint64_t tot = (1 + sm(s) % 64) * 10000;
            c.total.push_back(tot); c.free_.push_back(sm(s) % 5 == 0 ? tot : (sm(s) % (tot / 100 + 1)) * 100);
            c.pool.push_back((double)(1 + sm(s) % 50000) / 7.0);
        }
        for (uint32_t j = 0; j < c.n_cols; j++) {
            bool any = false;
            for (uint32_t r = 0; r < c.R; r++) {
                if (sm(s) % 2 && !(r == c.R - 1 && !any)) continue;
                any = true;
                c.ent_res.push_back(r);
                const bool all = sm(s) % 16 == 0;
                c.ent_kind.push_back(all ? 1 : 0);
                static const uint64_t grid[] = {10000, 20000, 40000, 5000, 2500, 80000, 30000, 15000, 100, 70000};
                c.ent_amount.push_back(grid[sm(s) % 10]);
            }
            c.ent_off.push_back((uint32_t)c.ent_res.size());
            c.weight.push_back(sm(s) % 3 == 0 ? 1000 + (uint32_t)(sm(s) % 30000) : 10000);
        }
        c.elig = c.n_cols >= 64 ? ~0ull : ((1ull << c.n_cols) - 1);
        if (sm(s) % 4 == 0) c.elig &= sm(s);
    }
    return c;
}

// the same block as host_model.cpp builds it for the exact solver
static bool solve_milp(const Case &c, std::vector<uint32_t> &x, bool *canonical) {
    hqmilp::Model m;
    std::vector<std::vector<std::pair<int, double>>> rt(c.R);
    std::vector<int> colmap(c.n_cols, -1);
    for (uint32_t g = 0; g < c.n_cols; g++) {
        if (!((c.elig >> g) & 1)) continue;
        double sc = 0.0;
        for (uint32_t e = c.ent_off[g]; e < c.ent_off[g + 1]; e++) {
            double pool = c.pool[c.ent_res[e]];
            uint64_t amt = c.ent_kind[e] ? c.total[c.ent_res[e]] : c.ent_amount[e];
            sc += pool < 0.000001 ? 0.0 : ((double)amt / 10000.0) / pool;
        }
        int col = m.add_col(sc * ((double)c.weight[g] / 10000.0), hqmilp::COL_NAT);
        colmap[g] = col;
        for (uint32_t e = c.ent_off[g]; e < c.ent_off[g + 1]; e++) {
            uint64_t amt = c.ent_kind[e] ? c.total[c.ent_res[e]] : c.ent_amount[e];
            rt[c.ent_res[e]].push_back({col, (double)amt / 10000.0});
        }
    }
    for (uint32_t r = 0; r < c.R; r++) if (!rt[r].empty()) { m.begin_row(hqmilp::ROW_MAX, (double)c.free_[r] / 10000.0); for (auto &t : rt[r]) m.term(t.first, t.second); m.end_row(); }
    hqmilp::Result res = hqmilp::solve(m, 20.0, true);
    if (!res.feasible || !res.optimal) return false;
    *canonical = res.canonical;
    x.assign(c.n_cols, 0);
    for (uint32_t g = 0; g < c.n_cols; g++) if (colmap[g] >= 0) x[g] = (uint32_t)std::llround(res.x[colmap[g]]);
    return true;
}

int main(int argc, char **argv) {
    long n_cases = argc > 1 ? atol(argv[1]) : 2000;
    uint64_t seed = argc > 2 ? strtoull(argv[2], nullptr, 10) : 1;
    int family = argc > 3 ? atoi(argv[3]) : -1;
    uint32_t budget = argc > 4 ? (uint32_t)atol(argv[4]) : 20000;
    static hqblock::Shared S;
    long bad = 0, budget_out = 0, unsup = 0, noncanon = 0; double t_blk = 0, t_milp = 0; unsigned long steps_sum = 0, steps_max = 0;
    for (long i = 0; i < n_cases; i++) {
        uint64_t s = seed * 1000003ull + (uint64_t)i;
        int fam = family >= 0 ? family : (int)(i % 3);
        Case c = make_case(s, fam);
        if (const char *bo = getenv("BF_ORDER")) {  // experiment: permute the columns (phase-1 step counts only; the canonical answer changes with the order)
            int mode = atoi(bo);
            std::vector<double> key(c.n_cols);
            for (uint32_t g = 0; g < c.n_cols; g++) {
                double size = 0, cost = 0;
                for (uint32_t e = c.ent_off[g]; e < c.ent_off[g + 1]; e++) { uint32_t r = c.ent_res[e]; double amt = c.ent_kind[e] ? c.total[r] : c.ent_amount[e]; size += amt / (double)(c.free_[r] + 1); cost += amt / 10000.0 / c.pool[r]; }
                cost *= c.weight[g] / 10000.0;
                key[g] = mode == 1 ? size : mode == 2 ? -size : mode == 3 ? cost / size : mode == 4 ? -cost / size : mode == 5 ? cost : -cost;
            }
            std::vector<uint32_t> idx(c.n_cols); for (uint32_t g = 0; g < c.n_cols; g++) idx[g] = g;
            std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return key[a] < key[b]; });
            Case d = c; d.ent_off.assign(1, 0); d.ent_res.clear(); d.ent_kind.clear(); d.ent_amount.clear(); d.weight.clear(); d.elig = 0;
            for (uint32_t q = 0; q < c.n_cols; q++) { uint32_t g = idx[q]; for (uint32_t e = c.ent_off[g]; e < c.ent_off[g + 1]; e++) { d.ent_res.push_back(c.ent_res[e]); d.ent_kind.push_back(c.ent_kind[e]); d.ent_amount.push_back(c.ent_amount[e]); } d.ent_off.push_back((uint32_t)d.ent_res.size()); d.weight.push_back(c.weight[g]); if ((c.elig >> g) & 1) d.elig |= 1ull << q; }
            c = d;
        }
        hqblock::ColTable ct{c.n_cols, c.R, c.ent_off.data(), c.ent_res.data(), c.ent_kind.data(), c.ent_amount.data(), c.weight.data(), c.pool.data(), nullptr, 0};
        hqblock::ClassTable cl{1, c.free_.data(), c.total.data(), &c.elig};
        std::vector<uint32_t> x(c.n_cols, 7); uint32_t status = 9, steps = 0;
        hqblock::Output out{x.data(), &status, &steps, nullptr};
        hqblock::HostWave wv;
        auto t0 = std::chrono::steady_clock::now();
        hqblock::solve_block(wv, S, ct, cl, 0, out, budget);
        auto t1 = std::chrono::steady_clock::now();
        std::vector<uint32_t> want; bool canon = true;
        bool ok = solve_milp(c, want, &canon);
        auto t2 = std::chrono::steady_clock::now();
        t_blk += std::chrono::duration<double>(t1 - t0).count(); t_milp += std::chrono::duration<double>(t2 - t1).count();
        if (status == hqblock::ST_UNSUPPORTED) { unsup++; continue; }
        if (status == hqblock::ST_BUDGET) { budget_out++; continue; }
        steps_sum += steps; if (steps > steps_max) steps_max = steps;
        if (getenv("BF_VERBOSE")) printf("steps %u n %d p1 %u\n", steps, S.n, S.steps_p1);
        if (!ok) { printf("case %ld: host solver failed\n", i); continue; }
        if (!canon) { noncanon++; continue; }
        if (x != want) {
            bad++;
            if (bad <= 10) {
                printf("MISMATCH case %ld (family %d, seed %llu): n_cols %u R %u elig %llx\n  got ", i, fam, (unsigned long long)seed, c.n_cols, c.R, (unsigned long long)c.elig);
                for (auto v : x) printf("%u ", v);
                printf("\n  want ");
                for (auto v : want) printf("%u ", v);
                printf("\n");
            }
        }
    }
    printf("%ld cases: %ld mismatches, %ld over budget, %ld unsupported, %ld host-non-canonical; steps avg %.1f max %lu; emulation %.3f s, host solver %.3f s\n", n_cases, bad, budget_out, unsup, noncanon,
           (double)steps_sum / (double)(n_cases - unsup - budget_out > 0 ? n_cases - unsup - budget_out : 1), steps_max, t_blk, t_milp);
#ifdef HQB_TRACE
    printf("max level list %u, max pool %u, probes %lu\n", hqblock::g_trace_maxlist, hqblock::g_trace_maxpool, hqblock::g_trace_probes);
#endif
    return bad ? 1 : 0;
}
