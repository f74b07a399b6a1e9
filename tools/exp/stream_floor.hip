// What is the floor of a launch shaped like K1 at the BASELINE size?   (MI355X, ROCm 7.2)
// K1 (k_level_hist, csrc/kernels.hip) at 1 M ready tasks: 977 workgroups of 256 threads, each wavefront reads 256 tasks (a dwordx4 of priorities and a dwordx2 of
// request ids per lane and 128-task tile, two tiles in flight), writes 2 B per task and 8 counters: 12 MB read, 2.25 MB written, 4.4-5.1 us inside the tick.
// Same grid, same loads, no classification: (a) nothing, (b) loads only (summed into one word per wavefront so they stay), (c) loads + the 2 B/task store.
// Durations from the dispatch's own start / stop events (hipExtLaunchKernelGGL), 200 launches each, back to back and with the host idle 50 us between launches.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/stream_floor tools/exp/stream_floor.hip && /tmp/stream_floor [n_tasks]
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

__global__ void __launch_bounds__(256) k_nothing(const uint64_t *, const uint32_t *, uint64_t, uint16_t *, uint32_t *) {}

template <bool STORE>
__global__ void __launch_bounds__(256) k_stream(const uint64_t *__restrict__ prio, const uint32_t *__restrict__ rq, uint64_t n, uint16_t *__restrict__ key, uint32_t *__restrict__ sink) {
    const uint32_t lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint64_t begin = (uint64_t)wave * 256;
    if (begin >= n) return;
    ulonglong2 pv[2]; uint2 qv[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const uint64_t i = begin + (uint64_t)u * 128 + 2 * lane;
        if (i + 1 < n) { pv[u] = *reinterpret_cast<const ulonglong2 *>(prio + i); qv[u] = *reinterpret_cast<const uint2 *>(rq + i); }
        else { pv[u] = make_ulonglong2(0, 0); qv[u] = make_uint2(0, 0); }
    }
    uint32_t acc = 0;
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const uint64_t i = begin + (uint64_t)u * 128 + 2 * lane;
        const uint32_t k0 = (uint32_t)pv[u].x ^ qv[u].x, k1 = (uint32_t)pv[u].y ^ qv[u].y;
        acc += k0 + k1;
        if (STORE && i + 1 < n) *reinterpret_cast<uint32_t *>(key + i) = (k0 & 0xFFFFu) | (k1 << 16);
    }
    if (acc == 0x12345678u) sink[wave] = acc;  // keeps the loads alive; practically never taken
}

// (d) .. (f): what K1 does beyond streaming — per-wavefront LDS counters fed by atomics, and the 8 counters published to the slice table, either as K1 lays it
// out ([group][slice]: 8 scattered dwords per wavefront, 16 bytes per workgroup and row) or slice-major ([slice][group]: 32 contiguous bytes per wavefront)
template <int TABLE, bool LDSATOM>
__global__ void __launch_bounds__(256) k_like_k1(const uint64_t *__restrict__ prio, const uint32_t *__restrict__ rq, uint64_t n, uint16_t *__restrict__ key, uint32_t *__restrict__ tab) {
    __shared__ uint32_t s_all[4 * 8];
    uint32_t *s_cnt = s_all + (threadIdx.x >> 6) * 8;
    const uint32_t lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t n_waves = (uint32_t)((n + 255) / 256), stride = (n_waves + 15u) & ~15u;
    if (lane < 8) s_cnt[lane] = 0;
    const uint64_t begin = (uint64_t)wave * 256;
    if (begin >= n) return;
    ulonglong2 pv[2]; uint2 qv[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const uint64_t i = begin + (uint64_t)u * 128 + 2 * lane;
        if (i + 1 < n) { pv[u] = *reinterpret_cast<const ulonglong2 *>(prio + i); qv[u] = *reinterpret_cast<const uint2 *>(rq + i); }
        else { pv[u] = make_ulonglong2(0, 0); qv[u] = make_uint2(0, 0); }
    }
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const uint64_t i = begin + (uint64_t)u * 128 + 2 * lane;
        const uint32_t k0 = (pv[u].x == 0x8000000000000000ull ? 0u : 8u) + qv[u].x, k1 = (pv[u].y == 0x8000000000000000ull ? 0u : 8u) + qv[u].y;
        if (LDSATOM) { atomicAdd(&s_cnt[k0 & 7u], 1u); atomicAdd(&s_cnt[k1 & 7u], 1u); }
        if (i + 1 < n) *reinterpret_cast<uint32_t *>(key + i) = (k0 & 0xFFFFu) | (k1 << 16);
    }
    if (TABLE == 1) { if (lane < 8) tab[(size_t)lane * stride + wave] = s_cnt[lane]; }
    if (TABLE == 2) { if (lane < 8) tab[(size_t)wave * 8 + lane] = s_cnt[lane]; }
}

static double med(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

template <class K> static void run(const char *name, K kern, uint32_t grid, hipStream_t s, const uint64_t *p, const uint32_t *q, uint64_t n, uint16_t *key, uint32_t *sink, double mb) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; mode++) {
        std::vector<double> us;
        for (int it = 0; it < 220; it++) {
            hipExtLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, s, e0, e1, 0, p, q, n, key, sink);
            if (mode == 1) { hipStreamSynchronize(s); std::this_thread::sleep_for(std::chrono::microseconds(50)); }
            else if (it % 20 == 19) hipStreamSynchronize(s);
            hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            if (it >= 20) us.push_back(ms * 1e3);
        }
        const double m = med(us);
        printf("%-28s %-14s %6.2f us", name, mode ? "(idle between)" : "(back to back)", m);
        if (mb > 0) printf("   %5.2f TB/s of read bytes = %.2f of 8 TB/s", mb / m, mb / m / 8.0);
        printf("\n");
    }
}

// 100 launches captured into one graph: no host launch cost, no per-launch events — (graph duration) / 100 is what the GPU spends per launch, boundary included
template <class K> static void run_graph(const char *name, K kern, uint32_t grid, hipStream_t s, const uint64_t *p, const uint32_t *q, uint64_t n, uint16_t *key, uint32_t *sink, double mb) {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < 100; i++) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, s, p, q, n, key, sink);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<double> us;
    for (int it = 0; it < 25; it++) {
        hipEventRecord(e0, s); hipGraphLaunch(ge, s); hipEventRecord(e1, s); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        if (it >= 5) us.push_back(ms * 1e3 / 100.0);
    }
    const double m = med(us);
    printf("%-28s %-14s %6.2f us", name, "(graph of 100)", m);
    if (mb > 0) printf("   %5.2f TB/s of read bytes = %.2f of 8 TB/s", mb / m, mb / m / 8.0);
    printf("\n");
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
}

int main(int argc, char **argv) {
    const uint64_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1000000ull;
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    uint64_t *p; uint32_t *q; uint16_t *key; uint32_t *sink;
    hipMalloc(&p, n * 8 + 64); hipMalloc(&q, n * 4 + 64); hipMalloc(&key, n * 2 + 64); hipMalloc(&sink, (n / 256 + 8) * 4);
    std::vector<uint64_t> hp(n); std::vector<uint32_t> hq(n);
    for (uint64_t i = 0; i < n; i++) { hp[i] = 0x8000000000000000ull + (i * 2654435761ull) % 3; hq[i] = (uint32_t)(i * 40503u) % 8; }
    hipMemcpy(p, hp.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(q, hq.data(), n * 4, hipMemcpyHostToDevice);
    const uint32_t grid = (uint32_t)((n + 1023) / 1024);
    printf("n = %llu tasks, %u workgroups of 256 threads, %.1f MB read per launch\n", (unsigned long long)n, grid, n * 12 / 1e6);
    run("empty kernel, same grid", k_nothing, grid, s, p, q, n, key, sink, 0.0);
    run("loads only", k_stream<false>, grid, s, p, q, n, key, sink, n * 12 / 1e6);
    run("loads + 2 B/task store", k_stream<true>, grid, s, p, q, n, key, sink, n * 12 / 1e6);
    uint32_t *tab; hipMalloc(&tab, ((n / 256 + 32) * 8) * 4);
    run_graph("(d) + LDS counters", k_like_k1<0, true>, grid, s, p, q, n, key, tab, n * 12 / 1e6);
    run_graph("(e) + table [group][slice]", k_like_k1<1, true>, grid, s, p, q, n, key, tab, n * 12 / 1e6);
    run_graph("(f) + table [slice][group]", k_like_k1<2, true>, grid, s, p, q, n, key, tab, n * 12 / 1e6);
    run_graph("(g) table [g][s], no atomics", k_like_k1<1, false>, grid, s, p, q, n, key, tab, n * 12 / 1e6);
    run_graph("empty kernel, same grid", k_nothing, grid, s, p, q, n, key, sink, 0.0);
    run_graph("loads only", k_stream<false>, grid, s, p, q, n, key, sink, n * 12 / 1e6);
    run_graph("loads + 2 B/task store", k_stream<true>, grid, s, p, q, n, key, sink, n * 12 / 1e6);
    return 0;
}
