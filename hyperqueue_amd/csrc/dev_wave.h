// The Wave policy of block_core.h / price_core.h on the device: one wave64 per block (blockDim.x == 64), lane = threadIdx.x.
// (The host emulation the CPU tests run is hqblock::HostWave in block_core.h.)
#pragma once
#include <hip/hip_runtime.h>

#include "block_core.h"

namespace hqblock {

struct DevWave {
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

}  // namespace hqblock
