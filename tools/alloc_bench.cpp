// Times libhqalloc.so from C (no ctypes in the way): three allocation mixes on one worker.  Built and run by tools/alloc_bench.py --native.
#include <chrono>
#include <cstdio>
#include <vector>

#include "../include/hqalloc.h"

struct Desc {
    std::vector<uint8_t> kind;
    std::vector<uint64_t> size;
    std::vector<uint32_t> group_off{0}, index_off{0}, index, c1, g1, c2, g2;
    std::vector<uint16_t> w;
    void groups(uint32_t n, uint32_t per) {
        kind.push_back(n == 1 ? HQALLOC_POOL_INDICES : HQALLOC_POOL_GROUPS);
        size.push_back(0);
        uint32_t base = 0;
        for (uint32_t g = 0; g < n; g++) {
            for (uint32_t i = 0; i < per; i++) index.push_back(base++);
            index_off.push_back((uint32_t)index.size());
        }
        group_off.push_back((uint32_t)index_off.size() - 1);
    }
    hqalloc_descriptor c() {
        hqalloc_descriptor d{};
        d.abi_version = HQALLOC_ABI_VERSION;
        d.n_resources = (uint32_t)kind.size();
        d.pool_kind = kind.data(); d.sum_size = size.data(); d.group_off = group_off.data(); d.index_off = index_off.data(); d.index = index.data();
        d.n_couplings = (uint32_t)w.size();
        d.coupling_resource1 = c1.data(); d.coupling_group1 = g1.data(); d.coupling_resource2 = c2.data(); d.coupling_group2 = g2.data(); d.coupling_weight = w.data();
        return d;
    }
};

static double run(Desc &d, std::vector<uint32_t> rid, std::vector<uint8_t> kind, std::vector<uint64_t> amt, int fill) {
    hqalloc_descriptor dc = d.c();
    hqalloc_ctx *ctx = nullptr;
    if (hqalloc_create(&dc, &ctx) != 0) return -1;
    hqalloc_request rq{(uint32_t)rid.size(), rid.data(), kind.data(), amt.data()};
    std::vector<uint32_t> r(8), off(9), idx(256), grp(256), fr(256);
    std::vector<uint64_t> am(8), ids;
    hqalloc_allocation out{};
    out.cap_resources = 8; out.cap_indices = 256;
    out.resource_id = r.data(); out.amount = am.data(); out.idx_off = off.data(); out.index = idx.data(); out.group_idx = grp.data(); out.fractions = fr.data();
    long n = 0;
    auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 1.0) {
        ids.clear();
        for (int i = 0; i < fill; i++) {
            if (hqalloc_try_allocate(ctx, &rq, &out) != 1) break;
            ids.push_back(out.allocation_id);
        }
        for (uint64_t id : ids) hqalloc_release(ctx, id);
        n += (long)ids.size();
    }
    double us = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / (n ? n : 1) * 1e6;
    hqalloc_destroy(ctx);
    return us;
}

int main() {
    Desc flat;
    flat.groups(1, 128);
    printf("1 cpu, flat 128-core worker                      %8.2f us per allocate+release\n", run(flat, {0}, {HQALLOC_COMPACT}, {10000}, 128));
    Desc numa;
    numa.groups(8, 16);
    numa.groups(8, 1);
    for (uint32_t g = 0; g < 8; g++) { numa.c1.push_back(0); numa.g1.push_back(g); numa.c2.push_back(1); numa.g2.push_back(g); numa.w.push_back(256); }
    printf("4 cpus + 0.5 gpu, 8 NUMA groups, coupled         %8.2f us per allocate+release\n", run(numa, {0, 1}, {HQALLOC_COMPACT, HQALLOC_COMPACT}, {40000, 5000}, 16));
    printf("4 cpus compact! + 1 gpu compact!, coupled        %8.2f us per allocate+release\n", run(numa, {0, 1}, {HQALLOC_FORCE_COMPACT, HQALLOC_FORCE_COMPACT}, {40000, 10000}, 8));
    return 0;
}
