// Where do the ~12 us between "host calls launch" and "host knows the kernel is done" go?   (MI355X, ROCm 7.2)
// hipcc --offload-arch=gfx950 -O3 -o /tmp/launch_latency tools/exp/launch_latency.hip && /tmp/launch_latency [spin]
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include <immintrin.h>

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_empty() {}
struct Big { uint64_t v[96]; };   // 768 bytes of kernel arguments (K4: a 512-byte plan + 20 scalars; K5b: a 27-field table struct)
__global__ void k_bigargs(Big b, uint32_t *out) { if (out && threadIdx.x == 9999) out[0] = (uint32_t)b.v[3]; }
// one workgroup: tells the host it has STARTED (flag0), burns `spin` x 10 ns, tells the host it is done (flag1)
__global__ void k_marks(volatile uint32_t *flag, uint32_t seq, uint32_t spin) {
    if (threadIdx.x == 0) flag[0] = seq;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
    if (threadIdx.x == 0) { __threadfence_system(); flag[16] = seq; }
}
static double med(std::vector<double> &v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main(int argc, char **argv) {
    if (argc > 1 && !strcmp(argv[1], "spin")) { hipSetDeviceFlags(hipDeviceScheduleSpin); printf("hipDeviceScheduleSpin\n"); }
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    uint32_t *h_flag, *d_flag; hipHostMalloc(&h_flag, 256, hipHostMallocDefault); hipHostGetDevicePointer((void **)&d_flag, h_flag, 0);
    memset(h_flag, 0, 256);
    std::vector<double> call, total, started, done_flag, done_sync, two;
    uint32_t seq = 0;
    for (int it = 0; it < 400; it++) {
        double t0 = now_us();
        hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s);
        double t1 = now_us();
        hipStreamSynchronize(s);
        double t2 = now_us();
        if (it >= 100) { call.push_back(t1 - t0); total.push_back(t2 - t0); }
    }
    printf("empty kernel: launch call %.1f us, launch + hipStreamSynchronize %.1f us\n", med(call), med(total));
    for (int it = 0; it < 400; it++) {
        seq++;
        double t0 = now_us();
        hipLaunchKernelGGL(k_marks, dim3(1), dim3(64), 0, s, d_flag, seq, 500u);
        while (__atomic_load_n(h_flag, __ATOMIC_ACQUIRE) != seq) _mm_pause();
        double t1 = now_us();
        while (__atomic_load_n(h_flag + 16, __ATOMIC_ACQUIRE) != seq) _mm_pause();
        double t2 = now_us();
        hipStreamSynchronize(s);
        double t3 = now_us();
        if (it >= 100) { started.push_back(t1 - t0); done_flag.push_back(t2 - t0); done_sync.push_back(t3 - t0); }
    }
    printf("5 us kernel: host sees START flag at %.1f us, DONE flag at %.1f us, hipStreamSynchronize returns at %.1f us (all since the launch call)\n", med(started), med(done_flag), med(done_sync));
    for (int it = 0; it < 400; it++) {
        double t0 = now_us();
        hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s);
        hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s);
        hipStreamSynchronize(s);
        double t2 = now_us();
        if (it >= 100) two.push_back(t2 - t0);
    }
    printf("two empty kernels + sync: %.1f us\n", med(two));
    {   // ways to learn that a ~5 us kernel is done, all since the launch call
        hipEvent_t ev; hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        std::vector<double> a, b, c, d, e2;
        for (int it = 0; it < 400; it++) {
            seq++; double t0 = now_us();
            hipLaunchKernelGGL(k_marks, dim3(1), dim3(64), 0, s, d_flag, seq, 500u);
            hipStreamSynchronize(s);
            if (it >= 100) a.push_back(now_us() - t0);
            seq++; t0 = now_us();
            hipLaunchKernelGGL(k_marks, dim3(1), dim3(64), 0, s, d_flag, seq, 500u);
            while (hipStreamQuery(s) == hipErrorNotReady) {}
            if (it >= 100) b.push_back(now_us() - t0);
            seq++; t0 = now_us();
            hipLaunchKernelGGL(k_marks, dim3(1), dim3(64), 0, s, d_flag, seq, 500u);
            hipEventRecord(ev, s);
            while (hipEventQuery(ev) == hipErrorNotReady) {}
            if (it >= 100) c.push_back(now_us() - t0);
            seq++; t0 = now_us();
            hipLaunchKernelGGL(k_marks, dim3(1), dim3(64), 0, s, d_flag, seq, 500u);
            hipEventRecord(ev, s);
            hipEventSynchronize(ev);
            if (it >= 100) d.push_back(now_us() - t0);
            seq++; t0 = now_us();
            hipLaunchKernelGGL(k_marks, dim3(1), dim3(64), 0, s, d_flag, seq, 500u);
            while (__atomic_load_n(h_flag + 16, __ATOMIC_ACQUIRE) != seq) _mm_pause();
            double t1 = now_us();
            hipStreamSynchronize(s);  // the kernel is done: what does the bookkeeping alone cost?
            if (it >= 100) { e2.push_back(now_us() - t1); }
        }
        printf("5 us kernel, done known after: hipStreamSynchronize %.1f | hipStreamQuery spin %.1f | event record + hipEventQuery spin %.1f | event record + hipEventSynchronize %.1f us;  hipStreamSynchronize on a finished stream: %.1f us\n",
               med(a), med(b), med(c), med(d), med(e2));
    }
    {   // the kernel's OWN completion signal: hipExtLaunchKernelGGL with a stop event, then wait on that event
        hipEvent_t st, st2, sp; hipEventCreate(&st); hipEventCreate(&sp); hipEventCreateWithFlags(&st2, hipEventDisableTiming);
        std::vector<double> a, b, c;
        for (int it = 0; it < 400; it++) {
            seq++; double t0 = now_us();
            hipExtLaunchKernelGGL(k_marks, dim3(1), dim3(64), 0, s, nullptr, sp, 0, d_flag, seq, 500u);
            hipEventSynchronize(sp);
            if (it >= 100) a.push_back(now_us() - t0);
            seq++; t0 = now_us();
            hipExtLaunchKernelGGL(k_marks, dim3(1), dim3(64), 0, s, nullptr, sp, 0, d_flag, seq, 500u);
            while (hipEventQuery(sp) == hipErrorNotReady) {}
            if (it >= 100) b.push_back(now_us() - t0);
            seq++; t0 = now_us();
            hipExtLaunchKernelGGL(k_marks, dim3(1), dim3(64), 0, s, nullptr, st2, 0, d_flag, seq, 500u);
            while (hipEventQuery(st2) == hipErrorNotReady) {}
            if (it >= 100) c.push_back(now_us() - t0);
            hipStreamSynchronize(s);
        }
        printf("5 us kernel launched with a stop event (hipExtLaunchKernelGGL): hipEventSynchronize(stop) %.1f | hipEventQuery(stop) spin %.1f | same, event without timing %.1f us\n", med(a), med(b), med(c));
    }
    {   // host cost of the launch call itself, by flavour (the stream is drained between the calls)
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        std::vector<double> a, b, c, d;
        for (int it = 0; it < 400; it++) {
            double t0 = now_us(); hipLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 0, s); double t1 = now_us(); hipStreamSynchronize(s);
            if (it >= 100) a.push_back(t1 - t0);
            t0 = now_us(); hipExtLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 0, s, nullptr, nullptr, 0); t1 = now_us(); hipStreamSynchronize(s);
            if (it >= 100) b.push_back(t1 - t0);
            t0 = now_us(); hipExtLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 0, s, nullptr, e1, 0); t1 = now_us(); hipStreamSynchronize(s);
            if (it >= 100) c.push_back(t1 - t0);
            t0 = now_us(); hipExtLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 0, s, e0, e1, 0); t1 = now_us(); hipStreamSynchronize(s);
            if (it >= 100) d.push_back(t1 - t0);
        }
        printf("launch CALL on the host: hipLaunchKernelGGL %.2f | hipExtLaunchKernelGGL without events %.2f | with a stop event %.2f | with start + stop %.2f us\n", med(a), med(b), med(c), med(d));
        {
            Big big{}; std::vector<double> f, g2;
            for (int it = 0; it < 400; it++) {
                double t0 = now_us(); hipLaunchKernelGGL(k_bigargs, dim3(1024), dim3(256), 0, s, big, (uint32_t *)nullptr); double t1 = now_us(); hipStreamSynchronize(s);
                if (it >= 100) f.push_back(t1 - t0);
                t0 = now_us(); hipExtLaunchKernelGGL(k_bigargs, dim3(1024), dim3(256), 12000, s, nullptr, e1, 0, big, (uint32_t *)nullptr); t1 = now_us(); hipStreamSynchronize(s);
                if (it >= 100) g2.push_back(t1 - t0);
            }
            printf("launch CALL with 776 bytes of kernel arguments: plain %.2f | with a stop event and 12 KB of dynamic LDS %.2f us\n", med(f), med(g2));
        }
        // two launches back to back, the second with a stop event (what phase C does)
        std::vector<double> e;
        for (int it = 0; it < 400; it++) {
            double t0 = now_us(); hipLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 0, s); hipExtLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 0, s, nullptr, e1, 0); double t1 = now_us(); hipStreamSynchronize(s);
            if (it >= 100) e.push_back(t1 - t0);
        }
        printf("two launch calls back to back (plain, then with a stop event): %.2f us\n", med(e));
    }
    // a graph of one kernel
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s);
    hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    std::vector<double> gr;
    for (int it = 0; it < 400; it++) { double t0 = now_us(); hipGraphLaunch(ge, s); hipStreamSynchronize(s); double t2 = now_us(); if (it >= 100) gr.push_back(t2 - t0); }
    printf("graph of one empty kernel + sync: %.1f us\n", med(gr));
    return 0;
}
