// How many single-wavefront workgroups does a CU of the MI355X really hold at once, as a function of their LDS allocation?   (round 4)
// The block solvers (k_block_solve, k_price_sweep) are one 64-thread workgroup per block with 40.8 KB of static LDS: 3 per CU by the arithmetic of 160 KB.  A variant
// with 32 KB (5 per CU by that arithmetic) changed no duration.  This measures residency directly: every workgroup records on which CU / XCC / SE it ran
// (s_getreg HW_ID) and the wall-clock interval [start, end] (s_memrealtime, 100 MHz) around a fixed busy loop of ~20 us; from the intervals the host counts, per CU,
// the largest number of workgroups whose intervals overlap.
// hipcc --offload-arch=gfx950 -O3 -o tools/exp/bin/resident_wg tools/exp/resident_wg.hip && tools/exp/bin/resident_wg
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <map>
#include <vector>

struct Rec { uint64_t t0, t1; uint32_t hw_id, xcc_id; };

template <int LDS_BYTES>
__global__ __launch_bounds__(64) void k_hold(Rec *out, uint32_t spin, uint32_t *sink) {
    __shared__ uint32_t lds[LDS_BYTES / 4];
    const uint64_t t0 = __builtin_readcyclecounter() * 0 + wall_clock64();
    uint32_t acc = threadIdx.x;
    for (uint32_t i = 0; i < spin; i++) { lds[(acc + i) % (LDS_BYTES / 4)] = acc; acc = acc * 1664525u + lds[(acc >> 3) % (LDS_BYTES / 4)] + 1013904223u; }
    const uint64_t t1 = wall_clock64();
    if (threadIdx.x == 0) {
        uint32_t hw = 0, xcc = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[blockIdx.x] = Rec{t0, t1, hw, xcc};
        if (acc == 0xDEADBEEFu) *sink = acc;
    }
}

template <int LDS_BYTES>
void run(uint32_t n_wg, uint32_t spin) {
    Rec *d; uint32_t *sink;
    hipMalloc(&d, sizeof(Rec) * n_wg); hipMalloc(&sink, 4);
    for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k_hold<LDS_BYTES>, dim3(n_wg), dim3(64), 0, 0, d, spin, sink); hipDeviceSynchronize(); }
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(k_hold<LDS_BYTES>, dim3(n_wg), dim3(64), 0, 0, d, spin, sink);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    std::vector<Rec> h(n_wg);
    hipMemcpy(h.data(), d, sizeof(Rec) * n_wg, hipMemcpyDeviceToHost);
    // CU identity: XCC id + (SE, SH, CU) fields of HW_ID (bits: cu_id 8-11, sh_id 12, se_id 13-15 on gfx9)
    std::map<uint32_t, std::vector<std::pair<uint64_t, int>>> ev;
    double dur = 0;
    for (const Rec &r : h) {
        const uint32_t cu = ((r.xcc_id & 0xF) << 16) | ((r.hw_id >> 8) & 0xFF);
        ev[cu].push_back({r.t0, +1}); ev[cu].push_back({r.t1, -1});
        dur += (double)(r.t1 - r.t0);
    }
    int worst = 0; double mean_peak = 0;
    for (auto &kv : ev) {
        std::sort(kv.second.begin(), kv.second.end(), [](auto &x, auto &y) { return x.first < y.first || (x.first == y.first && x.second < y.second); });
        int cur = 0, peak = 0;
        for (auto &e : kv.second) { cur += e.second; peak = std::max(peak, cur); }
        worst = std::max(worst, peak); mean_peak += peak;
    }
    printf("LDS %6d B/workgroup, %5u workgroups of 1 wave: kernel %8.1f us, mean workgroup %6.1f us, distinct CUs seen %zu, resident per CU: max %d, mean of the CUs' peaks %.2f\n",
           LDS_BYTES, n_wg, ms * 1e3, dur / n_wg * 0.01, ev.size(), worst, mean_peak / ev.size());
    hipFree(d); hipFree(sink);
}

int main() {
    const uint32_t spin = 6000;
    for (uint32_t n : {4096u}) {
        run<8192>(n, spin); run<16384>(n, spin); run<20480>(n, spin); run<22528>(n, spin); run<23296>(n, spin); run<24576>(n, spin); run<26624>(n, spin); run<27136>(n, spin); run<28672>(n, spin); run<30720>(n, spin); run<31744>(n, spin); run<32256>(n, spin); run<40776>(n, spin); run<49152>(n, spin); run<65536>(n, spin);
    }
    return 0;
}
