// k_block_solve: the per-worker-class blocks of the separable placement model (run_scheduling_solver's hot loop,
// /root/reference/crates/tako/src/internal/scheduler/solver.rs:95-192, then the solve of solver/highs.rs:65-88) — one wavefront per class.
//
// Launch shape: grid = number of worker classes (about one per worker on a steady-state cluster: 1024-4096), block = 64 threads = ONE wave64,
// 39 KB of LDS per block (dual vertices, level stack, the 64 greedy vectors) -> 4 blocks per CU, 1024 blocks resident on the 256 CUs, spread
// round-robin over the 8 XCDs by the dispatcher; blocks share nothing but the read-only column table (a few hundred bytes, L2-resident), so no
// XCD-aware mapping is needed.  Integer / f64 scalar work on data that lives in LDS: not an HBM-bound kernel, not MFMA work either — its
// figure of merit is classes solved per second.  The algorithm is in block_core.h (shared with the CPU emulation the tests run).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "block_core.h"
#include "kernels.h"
#include "block_solve.h"

namespace hqblock {

namespace {

struct DevWave {
    __device__ uint64_t now() const { return wall_clock64(); }  // 100 MHz
    __device__ bool first() const { return threadIdx.x == 0; }
    __device__ void sync() { __syncthreads(); }
    __device__ uint32_t atomic_inc(uint32_t *p) { return atomicAdd(p, 1u); }
    __device__ void atomic_or64(uint64_t *p, uint64_t v) { atomicOr((unsigned long long *)p, (unsigned long long)v); }
    __device__ static int ctz(uint64_t m) { return __ffsll((long long)m) - 1; }
    template <class F> __device__ void each(F f) { f((int)threadIdx.x); }
    template <class F> __device__ uint64_t ballot(F f) { return __ballot(f((int)threadIdx.x) ? 1 : 0); }
    template <class I, class Ch> __device__ uint64_t ballot_chunked(int nchunks, I init, Ch chunk) {
        Probe st;
        bool alive = init((int)threadIdx.x, st);
        for (int c = 0; c < nchunks; c++) {
            if (!__ballot(alive ? 1 : 0)) break;  // every child is pruned: the remaining duals cannot bring one back
            if (alive) alive = chunk((int)threadIdx.x, st, c);
        }
        return __ballot(alive ? 1 : 0);
    }
    template <class F> __device__ double argmax(F f, int *lane) {
        // butterfly over the wavefront: every lane ends with (largest value, lowest lane holding it)
        double v = f((int)threadIdx.x);
        int l = (int)threadIdx.x;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const double ov = __shfl_xor(v, off, 64);
            const int ol = __shfl_xor(l, off, 64);
            if (ov > v || (ov == v && ol < l)) { v = ov; l = ol; }
        }
        *lane = v < 0.0 ? -1 : l;
        return v < 0.0 ? -1.0 : v;
    }
};

__global__ __launch_bounds__(WAVE) void k_block_solve(ColTable ct, ClassTable cl, Output out, uint32_t budget) {
    __shared__ Shared S;
    DevWave wv;
    solve_block(wv, S, ct, cl, blockIdx.x, out, budget);
}

}  // namespace

hipError_t block_solve(const ColTable &ct, const ClassTable &cl, const Output &out, uint32_t budget, hipStream_t s) {
    if (cl.n_classes == 0) return hipSuccess;
    hqk::LaunchTimer t = hqk::take_launch_timer();
    hipExtLaunchKernelGGL(k_block_solve, dim3(cl.n_classes), dim3(WAVE), 0, s, t.start, t.stop, 0, ct, cl, out, budget);
    return hipGetLastError();
}

}  // namespace hqblock
