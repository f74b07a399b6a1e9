// Device / pinned-host buffers that only ever grow (no allocation in the steady state of a tick).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

namespace hqbuf {

struct DevBuf {
    void *p = nullptr; size_t cap = 0;
    bool ensure(size_t bytes) {
        if (bytes <= cap) return true;
        if (p) hipFree(p);
        size_t want = bytes + bytes / 4 + 256;
        if (hipMalloc(&p, want) != hipSuccess) { p = nullptr; cap = 0; return false; }
        cap = want; return true;
    }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
    void release() { if (p) hipFree(p); p = nullptr; cap = 0; }
};

struct PinBuf {  // page-locked, device-mapped host memory: kernels read small inputs from it and write small outputs into it
    void *p = nullptr; void *dp = nullptr; size_t cap = 0;  // dp = the same memory as the device sees it
    bool ensure(size_t bytes) {
        if (bytes <= cap) return true;
        if (p) hipHostFree(p);
        size_t want = bytes + bytes / 4 + 4096;
        if (hipHostMalloc(&p, want, hipHostMallocMapped) != hipSuccess) { p = nullptr; dp = nullptr; cap = 0; return false; }
        if (hipHostGetDevicePointer(&dp, p, 0) != hipSuccess) { hipHostFree(p); p = nullptr; dp = nullptr; cap = 0; return false; }
        cap = want; return true;
    }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
    template <typename T> T *dev() const { return reinterpret_cast<T *>(dp); }
    void release() { if (p) hipHostFree(p); p = nullptr; dp = nullptr; cap = 0; }
};

}  // namespace hqbuf
