// k_block_solve: the per-worker-class blocks of the separable placement model (run_scheduling_solver's hot loop,
// /root/reference/crates/tako/src/internal/scheduler/solver.rs:95-192, then the solve of solver/highs.rs:65-88) — one wavefront per class.
//
// Launch shape: grid = number of worker classes (about one per worker on a steady-state cluster: 1024-4096), block = 64 threads = ONE wave64,
// 25.9 KB of LDS per block (dual vertices, level stack; the 64 greedy vectors share the level lists' storage) -> 6 blocks per CU, 1536 resident on the 256 CUs, spread
// round-robin over the 8 XCDs by the dispatcher; blocks share nothing but the read-only column table (a few hundred bytes, L2-resident), so no
// XCD-aware mapping is needed.  Integer / f64 scalar work on data that lives in LDS: not an HBM-bound kernel, not MFMA work either — its
// figure of merit is classes solved per second.  The algorithm is in block_core.h (shared with the CPU emulation the tests run).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "block_core.h"
#include "dev_wave.h"
#include "kernels.h"
#include "block_solve.h"

namespace hqblock {

namespace {

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
