// hqwire_encode_device (include/hqwire.h): the three kernels of the worker-message wire encoding, and the CPU debug hook that runs the same
// phases on host memory (include/hqtick_debug.h).  The phases themselves live in wire_core.h.
//
// Launch shape: one 256-thread workgroup (4 wavefronts) per message slot -- a tick at the BASELINE sizes has 1024-4096 worker slots, i.e.
// 4-16 workgroups per CU over the 256 CUs / 8 XCDs; slots are independent, so the default round-robin of workgroups over XCDs is the right
// mapping (no slot shares data with its neighbour except the read-only tables, which sit in L2 / Infinity Cache).  HBM-bound byte work:
// per record ~20 B of attribute reads + ~43 B written, per distinct configuration its body read once per message that uses it.
#include <hip/hip_runtime.h>

#include <cstring>
#include <new>

#include "../../include/hqtick.h"
#ifdef HQTICK_TEST_HOOKS
#include "../../include/hqtick_debug.h"
#endif
#include "wire_core.h"

using namespace hqwire;

namespace {

__global__ __launch_bounds__(BLOCK) void k_wire_plan(Args a) {
    __shared__ PlanLds lds;
    const uint32_t s = blockIdx.x;
    const int tid = (int)threadIdx.x;
    plan_p0(a, lds, s, tid);
    __syncthreads();
    plan_p1(a, lds, s, tid);
    __syncthreads();
    plan_p2(a, lds, s, tid);
    __syncthreads();
    plan_p3a(a, lds, s, tid);
    __syncthreads();
    plan_p3b(a, lds, s, tid);
    __syncthreads();
    plan_p3c(a, lds, s, tid);
    __syncthreads();
    plan_p4(a, lds, s, tid);
    __syncthreads();
    plan_p5(a, lds, s, tid);
    __syncthreads();
    plan_p6a(a, lds, s, tid);
    __syncthreads();
    plan_p6(a, lds, s, tid);
    __syncthreads();
    plan_p7(a, lds, s, tid);
}

__global__ __launch_bounds__(BLOCK) void k_wire_scan(Args a) {
    __shared__ ScanLds lds;
    const int tid = (int)threadIdx.x;
    scan_p1(a, lds, tid);
    __syncthreads();
    scan_p2a(a, lds, tid);
    __syncthreads();
    scan_p2(a, lds, tid);
    __syncthreads();
    scan_p2c(a, lds, tid);
    __syncthreads();
    scan_p3(a, lds, tid);
}

__global__ __launch_bounds__(BLOCK) void k_wire_emit(Args a) {
    __shared__ EmitLds lds;
    const uint32_t s = blockIdx.x;
    const int tid = (int)threadIdx.x;
    const uint32_t nf = nfrag_of(a, s) ? nfrag_of(a, s) : 1;  // uniform per workgroup; fragment 0 also carries the slot's RetractTasks message
    for (uint32_t f = 0; f < nf; f++) {
        emit_p1(a, lds, s, f, tid);
        __syncthreads();
        emit_p2a(a, lds, s, f, tid);
        __syncthreads();
        emit_p2(a, lds, s, f, tid);
        __syncthreads();
        emit_p2c(a, lds, s, f, tid);
        __syncthreads();
        emit_p3(a, lds, s, f, tid);
        __syncthreads();
        emit_p4(a, lds, s, f, tid);
        __syncthreads();
    }
}

bool make_args(const hqwire_tables *t, const hqwire_records *r, const hqwire_output *o, Args &a) {
    if (!t || !r || !o) return false;
    if (r->n_workers && (!r->worker_id || !r->rec_off)) return false;
    if (r->n_records && (!r->rec_task || !r->rec_variant || !r->rec_kind)) return false;
    if (r->n_mn && (!r->mn_task || !r->mn_worker_off || !r->mn_worker)) return false;
    if (t->n_tasks && (!t->task_id || !t->task_rq || !t->task_instance || !t->task_priority || !t->task_config || !t->entry_some || !t->entry_off)) return false;
    if (t->n_configs && (!t->config_time_some || !t->config_time_secs || !t->config_time_nanos || !t->body_off)) return false;
    if (!o->slot_off || !o->slot_status || !o->header || !o->scratch || (o->capacity && !o->bytes)) return false;
    a.t = *t;
    a.r = *r;
    a.o = *o;
    a.n_slots = r->n_workers + r->n_mn;
    if (o->scratch_bytes < scratch_bytes((uint64_t)r->n_records + r->n_mn, a.n_slots)) return false;
    if ((reinterpret_cast<uintptr_t>(o->scratch) & 7) || (reinterpret_cast<uintptr_t>(o->slot_off) & 7)) return false;  // u64 stores in bind_scratch / scan_p3
    if ((o->slot_nfrag == nullptr) != (o->frag_end == nullptr)) return false;
    if (o->frag_end && (reinterpret_cast<uintptr_t>(o->frag_end) & 7)) return false;
    bind_scratch(a);
    return true;
}

}  // namespace

extern "C" {

uint32_t hqwire_abi_version(void) { return HQWIRE_ABI_VERSION; }

uint64_t hqwire_scratch_bytes(uint64_t n_records_incl_mn, uint64_t n_slots) { return scratch_bytes(n_records_incl_mn, n_slots); }

int hqwire_encode_device(const hqwire_tables *tables, const hqwire_records *records, const hqwire_output *out, void *hip_stream) {
    Args a;
    if (!make_args(tables, records, out, a)) return HQTICK_E_INVALID;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return HQTICK_E_NO_DEVICE;  // no CPU path behind this entry point
    hipStream_t st = (hipStream_t)hip_stream;
    if (a.n_slots) hipLaunchKernelGGL(k_wire_plan, dim3(a.n_slots), dim3(BLOCK), 0, st, a);
    hipLaunchKernelGGL(k_wire_scan, dim3(1), dim3(BLOCK), 0, st, a);
    if (a.n_slots) hipLaunchKernelGGL(k_wire_emit, dim3(a.n_slots), dim3(BLOCK), 0, st, a);
    return hipGetLastError() == hipSuccess ? 0 : HQTICK_E_DEVICE;
}

#ifdef HQTICK_TEST_HOOKS  // libhqtick_test.so only
// CPU debug hook: the same phases on HOST memory, one emulated thread after the other (a barrier = the end of a loop over tid).
// Test infrastructure for machines without a GPU; the product entry point is hqwire_encode_device.
// `order` picks the sequence in which the 256 emulated threads of a phase run (0 ascending, 1 descending, 2 a fixed permutation): a phase
// whose result depended on that sequence would be a race on the GPU.
int hqwire_debug_encode_host_order(const hqwire_tables *tables, const hqwire_records *records, const hqwire_output *out, int order) {
    Args a;
    if (!make_args(tables, records, out, a)) return HQTICK_E_INVALID;
    return run_on_host(a, order) ? 0 : HQTICK_E_DEVICE;
}

int hqwire_debug_encode_host(const hqwire_tables *tables, const hqwire_records *records, const hqwire_output *out) {
    return hqwire_debug_encode_host_order(tables, records, out, 0);
}
#endif  // HQTICK_TEST_HOOKS

}  // extern "C"
