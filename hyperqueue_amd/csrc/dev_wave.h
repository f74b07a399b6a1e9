// The Wave policies of block_core.h / price_core.h on the device.
//   DevWave       one wave64 per block (blockDim.x == 64), lane = threadIdx.x.
//   DevGroup<NW>  a workgroup of NW wave64 per block (blockDim.x == 64 NW, one wavefront on each SIMD of the CU for NW = 4): wavefront 0 is the block's MAIN wavefront and
//                 runs the chain; the others enter through hqblock::pool_helper and take part in the pool section only.  sync() is the main wavefront's own (a
//                 wavefront executes in lockstep and its LDS operations complete in order: a compiler fence, no s_barrier — the helpers have left by then, and a
//                 hardware barrier per search step would cost more than the helpers save); group_sync() is the workgroup's barrier.
// (The host emulation the CPU tests run is hqblock::HostWave in block_core.h.)
#pragma once
#include <hip/hip_runtime.h>

#include "block_core.h"

namespace hqblock {

// The children of a level of the walk against the level's duals (block_core.h: walk): every lane is one child; chunks of 8 duals, tightest first, and once a chunk has
// pruned every child the rest is not looked at — which is the common case after the FIRST chunk.  (Round 6 also tried the duals sliced over the idle lanes when a level
// has fewer children than lanes, with a butterfly minimum over the slices: every step then pays the full minimum and five shuffles, and the walks were 15-20 % slower.)
template <class I, class P, class D>
__device__ __forceinline__ uint64_t dev_ballot_bound(int lane, int cnt, int nchild, I init, P part, D decide) {
    (void)nchild;
    const int nch = (cnt + 7) >> 3;
    Probe st;
    bool alive = init(lane, st);
    double b = 1e300;
    for (int c = 0; c < nch; c++) {
        if (!__ballot(alive ? 1 : 0)) break;
        if (alive) { const double v = part(st, c * 8, (c + 1) * 8 < cnt ? (c + 1) * 8 : cnt); b = v < b ? v : b; alive = decide(st, b); }
    }
    return __ballot(alive ? 1 : 0);
}

// The wavefront's largest value of a double per lane, in every lane: a prefix maximum inside the rows of 16 lanes by DPP row shifts, the rows joined by the two row
// broadcasts, lane 63 read back — a dozen cycles per step where a butterfly of ds_bpermute shuffles paid an LDS-crossbar round trip per step (the leaves of a walk take
// one arg-max per step: ~0.4 us of a ~1 us step).  max() is exact, so the value is the butterfly's.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_max_step(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int tlo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xF, false), thi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xF, false);
    const double t = __hiloint2double(thi, tlo);
    return t > v ? t : v;
}
__device__ __forceinline__ double wave_max_f64(double v) {
    v = dpp_max_step<0x111, 0xF>(v);  // row_shr:1
    v = dpp_max_step<0x112, 0xF>(v);  // row_shr:2
    v = dpp_max_step<0x114, 0xF>(v);  // row_shr:4
    v = dpp_max_step<0x118, 0xF>(v);  // row_shr:8   (lane 15 of every row: the row's maximum)
    v = dpp_max_step<0x142, 0xA>(v);  // row_bcast:15 into rows 1 and 3
    v = dpp_max_step<0x143, 0xC>(v);  // row_bcast:31 into rows 2 and 3  (lane 63: the wavefront's)
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
// largest value, lowest lane among equals (the value returned is that lane's own); *lane = -1 when every value is negative
__device__ __forceinline__ double dev_argmax(double v, int *lane_out) {
    const double m = wave_max_f64(v);
    if (m < 0.0) { *lane_out = -1; return -1.0; }
    const int l = __ffsll((long long)__ballot(v == m ? 1 : 0)) - 1;
    *lane_out = l;
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

struct DevWave {
    static constexpr int WAVES = 1;
    __device__ int wave_index() const { return 0; }
    __device__ void group_sync() { __syncthreads(); }
    __device__ void pool_barrier(uint32_t *, uint32_t) { __syncthreads(); }
    __device__ void raise(uint32_t *) {}
    __device__ void await(uint32_t *) {}
    // a result other workgroups of the launch read: written through to where the whole device sees it (no L2 write-back fence needed behind it, only the acknowledgement)
    __device__ void put(double *p, double v) { __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __device__ void put(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __device__ uint32_t grab(uint32_t *p, uint32_t n) { uint32_t v = 0; if (threadIdx.x == 0) v = atomicAdd(p, n); return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
    template <class F> __device__ void each_of_pool(F f) { f((int)threadIdx.x, WAVE); }
    __device__ uint64_t now() const { return wall_clock64(); }  // 100 MHz
    __device__ bool first() const { return threadIdx.x == 0; }
    __device__ void sync() { __syncthreads(); }
    __device__ uint32_t atomic_inc(uint32_t *p) { return atomicAdd(p, 1u); }
    __device__ void atomic_or64(uint64_t *p, uint64_t v) { atomicOr((unsigned long long *)p, (unsigned long long)v); }
    __device__ void atomic_add_i64(long long *p, long long v) { atomicAdd((unsigned long long *)p, (unsigned long long)v); }  // two's complement: the sum of signed terms
    __device__ void lds_add_i64(long long *p, long long v) { atomicAdd((unsigned long long *)p, (unsigned long long)v); }  // (an LDS address: ds_add_u64)
    __device__ static int ctz(uint64_t m) { return __ffsll((long long)m) - 1; }
    template <class F> __device__ void each(F f) { f((int)threadIdx.x); }
    template <class F> __device__ uint64_t ballot(F f) { return __ballot(f((int)threadIdx.x) ? 1 : 0); }
    template <class I, class P, class D> __device__ uint64_t ballot_bound(int cnt, int nchild, I init, P part, D decide) { return dev_ballot_bound((int)threadIdx.x, cnt, nchild, init, part, decide); }
    template <class F> __device__ double argmax(F f, int *lane) { return dev_argmax(f((int)threadIdx.x), lane); }
};

template <int NW>
struct DevGroup {
    static constexpr int WAVES = NW;
    __device__ static int lane() { return (int)(threadIdx.x & 63u); }
    __device__ int wave_index() const { return (int)(threadIdx.x >> 6); }
    __device__ void group_sync() { __syncthreads(); }
    // barrier number `phase` of the NW - 1 pool wavefronts (all but the last): a counter in LDS, lane 0 of each adds one and waits for (NW - 1) * phase
    __device__ void pool_barrier(uint32_t *ctr, uint32_t phase) {
        if (NW <= 2) { sync(); return; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane() == 0) {
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < (uint32_t)(NW - 1) * phase) __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    __device__ void raise(uint32_t *flag) {   // the calling wavefront's LDS writes, then the flag
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        if (lane() == 0) __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __device__ void await(uint32_t *flag) {
        if (lane() == 0) while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    // a result other workgroups of the launch read: written through to where the whole device sees it (no L2 write-back fence needed behind it, only the acknowledgement)
    __device__ void put(double *p, double v) { __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __device__ void put(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __device__ uint32_t grab(uint32_t *p, uint32_t n) { uint32_t v = 0; if (lane() == 0) v = atomicAdd(p, n); return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
    template <class F> __device__ void each_of_pool(F f) { f((int)threadIdx.x, WAVE * (NW - 1)); }
    __device__ uint64_t now() const { return wall_clock64(); }
    __device__ bool first() const { return threadIdx.x == 0; }  // (lane 0 of the main wavefront: the helpers never ask)
    __device__ void sync() {  // within the calling wavefront (see above)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    __device__ uint32_t atomic_inc(uint32_t *p) { return atomicAdd(p, 1u); }
    __device__ void atomic_or64(uint64_t *p, uint64_t v) { atomicOr((unsigned long long *)p, (unsigned long long)v); }
    __device__ void atomic_add_i64(long long *p, long long v) { atomicAdd((unsigned long long *)p, (unsigned long long)v); }
    __device__ void lds_add_i64(long long *p, long long v) { atomicAdd((unsigned long long *)p, (unsigned long long)v); }
    __device__ static int ctz(uint64_t m) { return __ffsll((long long)m) - 1; }
    template <class F> __device__ void each(F f) { f(lane()); }
    template <class F> __device__ uint64_t ballot(F f) { return __ballot(f(lane()) ? 1 : 0); }
    template <class I, class P, class D> __device__ uint64_t ballot_bound(int cnt, int nchild, I init, P part, D decide) { return dev_ballot_bound((int)lane(), cnt, nchild, init, part, decide); }
    template <class F> __device__ double argmax(F f, int *lane_out) { return dev_argmax(f(lane()), lane_out); }
};

}  // namespace hqblock
