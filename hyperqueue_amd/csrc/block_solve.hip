// k_block_solve: the per-worker-class blocks of the separable placement model (run_scheduling_solver's hot loop,
// /root/reference/crates/tako/src/internal/scheduler/solver.rs:95-192, then the solve of solver/highs.rs:65-88) — one workgroup per class.
//
// Launch shape: grid = number of worker classes (about one per worker on a steady-state cluster: 1024-4096), block = 256 threads = FOUR wave64 (one runs the class's chain,
// the others share its dual pool and run its greedy fills, then leave: block_core.h, pool_sections), 36.9 KB of LDS per block (dual vertices, level stack, greedy
// vectors) -> 4 blocks per CU, 1024 resident on the 256 CUs, spread round-robin over the 8 XCDs by the dispatcher; blocks share nothing but the read-only column table (a few hundred bytes, L2-resident), so no
// XCD-aware mapping is needed.  Integer / f64 scalar work on data that lives in LDS: not an HBM-bound kernel, not MFMA work either — its
// figure of merit is classes solved per second.  The algorithm is in block_core.h (shared with the CPU emulation the tests run).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdlib>

#include "block_core.h"
#include "dev_wave.h"
#include "kernels.h"
#include "block_solve.h"

namespace hqblock {

namespace {

// NW wavefronts per class block: wavefront 0 runs the block's chain, the others take part in its dual pool / greedy section and leave (block_core.h: pool_sections).
// A steady-state tick has a few hundred classes on 1024 SIMDs: the helpers sit on SIMDs that would idle.
template <int NW>
__global__ __launch_bounds__(WAVE * NW) void k_block_solve(ColTable ct, ClassTable cl, Output out, uint32_t budget) {
    __shared__ Shared S;
    DevGroup<NW> wv;
    if (threadIdx.x >= WAVE) { pool_helper(wv, S); return; }
    solve_block(wv, S, ct, cl, blockIdx.x, out, budget);
}
template <>
__global__ __launch_bounds__(WAVE) void k_block_solve<1>(ColTable ct, ClassTable cl, Output out, uint32_t budget) {
    __shared__ Shared S;
    DevWave wv;
    solve_block(wv, S, ct, cl, blockIdx.x, out, budget);
}

}  // namespace

hipError_t block_solve(const ColTable &ct, const ClassTable &cl, const Output &out, uint32_t budget, hipStream_t s) {
    if (cl.n_classes == 0) return hipSuccess;
    hqk::LaunchTimer t = hqk::take_launch_timer();
    static const int force_waves = getenv("HQTICK_BLOCK_WAVES") ? atoi(getenv("HQTICK_BLOCK_WAVES")) : 0;   // A/B switch (1 / 2 / 4)
    const int nw = force_waves == 1 || force_waves == 2 ? force_waves : 4;
    if (nw == 4) hipExtLaunchKernelGGL(k_block_solve<4>, dim3(cl.n_classes), dim3(WAVE * 4), 0, s, t.start, t.stop, 0, ct, cl, out, budget);
    else if (nw == 2) hipExtLaunchKernelGGL(k_block_solve<2>, dim3(cl.n_classes), dim3(WAVE * 2), 0, s, t.start, t.stop, 0, ct, cl, out, budget);
    else hipExtLaunchKernelGGL(k_block_solve<1>, dim3(cl.n_classes), dim3(WAVE), 0, s, t.start, t.stop, 0, ct, cl, out, budget);
    return hipGetLastError();
}

}  // namespace hqblock
