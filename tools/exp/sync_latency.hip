// How long after a kernel's last store does the host know?  hipStreamSynchronize vs spinning on a flag the kernel's last workgroup writes into pinned memory.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/sync_latency tools/exp/sync_latency.hip && /tmp/sync_latency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <immintrin.h>

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// every workgroup writes 64 B of "results" into pinned memory, burns `spin` clocks, then arrives; the last one publishes seq
__global__ void k_work(uint32_t *out, uint32_t *ctr, volatile uint32_t *flag, uint32_t seq, uint32_t spin, int publish) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
    if (threadIdx.x < 16) out[blockIdx.x * 16 + threadIdx.x] = seq + threadIdx.x;
    if (!publish) return;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t old = atomicAdd(ctr, 1u);
        if (old == gridDim.x - 1) { *ctr = 0; __threadfence_system(); *flag = seq; }
    }
}

int main() {
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    uint32_t *h_out, *h_flag, *d_ctr; hipHostMalloc(&h_out, 4096 * 64, hipHostMallocDefault); hipHostMalloc(&h_flag, 64, hipHostMallocDefault);
    hipMalloc(&d_ctr, 64); hipMemset(d_ctr, 0, 64);
    uint32_t *d_out, *d_flag; hipHostGetDevicePointer((void **)&d_out, h_out, 0); hipHostGetDevicePointer((void **)&d_flag, h_flag, 0);
    const int grids[] = {34, 1024};
    for (int grid : grids) for (uint32_t spin : {500u, 2000u}) {  // wall_clock64 ticks at 100 MHz: 5 us / 20 us kernels
        std::vector<double> a, b;
        uint32_t seq = 1;
        for (int it = 0; it < 300; it++) {
            seq += 100;
            double t0 = now_us();
            hipLaunchKernelGGL(k_work, dim3(grid), dim3(256), 0, s, d_out, d_ctr, d_flag, seq, spin, 0);
            hipStreamSynchronize(s);
            double t1 = now_us();
            if (it >= 50) a.push_back(t1 - t0);
            seq += 100;
            *(volatile uint32_t *)h_flag = 0;
            t0 = now_us();
            hipLaunchKernelGGL(k_work, dim3(grid), dim3(256), 0, s, d_out, d_ctr, d_flag, seq, spin, 1);
            while (__atomic_load_n(h_flag, __ATOMIC_ACQUIRE) != seq) _mm_pause();
            t1 = now_us();
            bool ok = true;
            for (int blk = 0; blk < grid; blk++) for (int k = 0; k < 16; k++) if (h_out[blk * 16 + k] != seq + k) ok = false;
            if (!ok) { printf("STALE RESULTS at it %d\n", it); return 1; }
            if (it >= 50) b.push_back(t1 - t0);
            hipStreamSynchronize(s);
        }
        std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
        printf("grid %4d kernel ~%2u us: launch + hipStreamSynchronize p50 %.1f us   launch + flag spin p50 %.1f us (results checked every iteration)\n", grid, spin / 100, a[a.size() / 2], b[b.size() / 2]);
    }
    return 0;
}
