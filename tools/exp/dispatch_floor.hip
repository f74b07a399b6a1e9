// What does the SHAPE of K1's launch cost at the BASELINE size, and what can a persistent grid stream?   (MI355X, ROCm 7.2; round 3)
// K1 (k_level_hist) at 1 M ready tasks was 977 workgroups x 256 threads and sat on its dispatch floor: an EMPTY kernel of that grid records 4.1 us under the
// per-dispatch events the bench uses.  Measured here, same events, 200 launches each, back to back and with the host idle in between:
//   (1) an empty kernel over a range of (grid, block) shapes;
//   (2) K1's work — 12 B/task read, 2 B/task key column, per-slice LDS counters, the [group][slice] table — as a PERSISTENT grid: every wavefront takes the
//       256-task slices  wave, wave + total_waves, ...  with the next slice's loads issued before the current one is classified.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/dispatch_floor tools/exp/dispatch_floor.hip && /tmp/dispatch_floor [n_tasks]
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

__global__ void k_nothing(const uint64_t *, const uint32_t *, uint64_t, uint16_t *, uint32_t *) {}

struct Tile { ulonglong2 p[2]; uint2 q[2]; };

__device__ __forceinline__ void load_slice(Tile &t, const uint64_t *__restrict__ prio, const uint32_t *__restrict__ rq, uint64_t begin, uint64_t n, uint32_t lane) {
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const uint64_t i = begin + (uint64_t)u * 128 + 2 * lane;
        if (i + 1 < n) { t.p[u] = *reinterpret_cast<const ulonglong2 *>(prio + i); t.q[u] = *reinterpret_cast<const uint2 *>(rq + i); }
        else { t.p[u] = make_ulonglong2(0, 0); t.q[u] = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu); }
    }
}

template <int DEPTH>
__global__ void k_persistent(const uint64_t *__restrict__ prio, const uint32_t *__restrict__ rq, uint64_t n, uint16_t *__restrict__ key, uint32_t *__restrict__ tab) {
    extern __shared__ uint32_t s_all[];
    const uint32_t wpb = blockDim.x >> 6, lane = threadIdx.x & 63;
    uint32_t *s_cnt = s_all + (threadIdx.x >> 6) * 8;
    const uint32_t total = gridDim.x * wpb, first = blockIdx.x * wpb + (threadIdx.x >> 6);
    const uint32_t n_waves = (uint32_t)((n + 255) / 256), stride = (n_waves + 15u) & ~15u;
    Tile t[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; d++) { const uint32_t w = first + d * total; if (w < n_waves) load_slice(t[d], prio, rq, (uint64_t)w * 256, n, lane); }
    for (uint32_t w0 = first; w0 < n_waves; w0 += DEPTH * total) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            const uint32_t w = w0 + d * total;
            if (w >= n_waves) break;
            if (lane < 8) s_cnt[lane] = 0;
            Tile cur = t[d];
            const uint32_t wn = w + DEPTH * total;
            if (wn < n_waves) load_slice(t[d], prio, rq, (uint64_t)wn * 256, n, lane);  // the slice after next goes out before this one is classified
            const uint64_t begin = (uint64_t)w * 256;
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const uint64_t i = begin + (uint64_t)u * 128 + 2 * lane;
                const uint32_t k0 = (cur.p[u].x == 0x8000000000000000ull ? 0u : 8u) + cur.q[u].x, k1 = (cur.p[u].y == 0x8000000000000000ull ? 0u : 8u) + cur.q[u].y;
                if (i < n) atomicAdd(&s_cnt[k0 & 7u], 1u);
                if (i + 1 < n) { atomicAdd(&s_cnt[k1 & 7u], 1u); *reinterpret_cast<uint32_t *>(key + i) = (k0 & 0xFFFFu) | (k1 << 16); }
            }
            if (lane < 8) tab[(size_t)lane * stride + w] = s_cnt[lane];
        }
    }
}

static double med(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

template <class K> static void run(const char *name, K kern, uint32_t grid, uint32_t block, size_t lds, hipStream_t s, const uint64_t *p, const uint32_t *q, uint64_t n, uint16_t *key, uint32_t *tab, double mb) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    double m[2];
    for (int mode = 0; mode < 2; mode++) {
        std::vector<double> us;
        for (int it = 0; it < 220; it++) {
            hipExtLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, s, e0, e1, 0, p, q, n, key, tab);
            if (mode == 1) { hipStreamSynchronize(s); std::this_thread::sleep_for(std::chrono::microseconds(50)); }
            else if (it % 20 == 19) hipStreamSynchronize(s);
            hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            if (it >= 20) us.push_back(ms * 1e3);
        }
        m[mode] = med(us);
    }
    printf("%-34s grid %5u x %4u   %6.2f us back to back   %6.2f us idle between", name, grid, block, m[0], m[1]);
    if (mb > 0) printf("   %5.2f TB/s = %.2f of 8 TB/s", mb / m[0], mb / m[0] / 8.0);
    printf("\n");
}

int main(int argc, char **argv) {
    const uint64_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1000000ull;
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    uint64_t *p; uint32_t *q; uint16_t *key; uint32_t *tab;
    hipMalloc(&p, n * 8 + 64); hipMalloc(&q, n * 4 + 64); hipMalloc(&key, n * 2 + 64); hipMalloc(&tab, ((n / 256 + 32) * 8) * 4);
    std::vector<uint64_t> hp(n); std::vector<uint32_t> hq(n);
    for (uint64_t i = 0; i < n; i++) { hp[i] = 0x8000000000000000ull + (i * 2654435761ull) % 3; hq[i] = (uint32_t)(i * 40503u) % 8; }
    hipMemcpy(p, hp.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(q, hq.data(), n * 4, hipMemcpyHostToDevice);
    const uint32_t slices = (uint32_t)((n + 255) / 256);
    printf("n = %llu tasks = %u slices of 256, %.1f MB read per launch\n", (unsigned long long)n, slices, n * 12 / 1e6);
    const uint32_t shapes[][2] = {{(slices + 3) / 4, 256}, {(slices + 7) / 8, 512}, {(slices + 15) / 16, 1024}, {1024, 256}, {512, 256}, {512, 512}, {256, 256}, {256, 512}, {256, 1024}, {128, 1024}, {64, 1024}};
    for (auto &sh : shapes) run("empty kernel", k_nothing, sh[0], sh[1], 0, s, p, q, n, key, tab, 0.0);
    const double mb = n * 12 / 1e6;
    for (auto &sh : shapes) {
        const size_t lds = (sh[1] / 64) * 8 * 4;
        run("K1-like, persistent, depth 1", k_persistent<1>, sh[0], sh[1], lds, s, p, q, n, key, tab, mb);
        run("K1-like, persistent, depth 2", k_persistent<2>, sh[0], sh[1], lds, s, p, q, n, key, tab, mb);
        if (sh[0] <= 512) run("K1-like, persistent, depth 4", k_persistent<4>, sh[0], sh[1], lds, s, p, q, n, key, tab, mb);
    }
    return 0;
}
