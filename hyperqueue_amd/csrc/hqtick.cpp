// libhqtick.so — the C ABI of include/hqtick.h: one scheduling tick of tako on an MI355X.
//
// Data flow of hqtick_run():
//   H2D   ready-set columns (unless resident), worker/request tables
//   GPU   K0 distinct priorities -> K0b sorted level table -> K1 (level, rq) histogram per wave slice -> K1b scan
//         K2 per-(worker, variant) capability flags + task_max_count
//   D2H   level table, histogram, flags                                   (small)
//   HOST  batches (batches.rs) -> MILP (solver.rs) via the exact solver -> counts; selection plan
//   H2D   take/base per group, round-robin tables                         (small)
//   GPU   K4 select+scatter the taken tasks in queue order -> K5 expand to per-worker records
//   D2H   assignment records
// There is no CPU implementation of the scans/mapping: without a HIP device every entry point fails.
#include "../../include/hqtick.h"
#ifdef HQTICK_TEST_HOOKS
#include "../../include/hqtick_debug.h"
#endif

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "block_solve.h"
#include "price_dev.h"
#ifdef HQTICK_TEST_HOOKS
#include "price_emul.h"
#endif
#include "devbuf.h"
#include "graph.h"
#include "hb_order.h"
#include "host_model.h"
#include "kernels.h"

using hqbuf::DevBuf;
using hqbuf::PinBuf;

namespace {

double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

namespace {
// Reusable host scratch of the mapping plan (no per-tick allocation in the steady state).
struct PlanScratch {
    std::vector<uint32_t> q_total, pf_n, pf_start, seq_taken, pf_drained, zq_taken, new_pf_total, pfl_size, rq_sel_base;
    std::vector<uint32_t> key_seg, key_sum, key_rq, key_var_w, key_ord_off, ord_cnt, key_t_off, key_bits_off;
    std::vector<uint32_t> key_tr;   // per key: c > 0 = stored worker-major by K4 (kernels.h: MapKeys::key_tr)
    std::vector<uint32_t> list_pos, pf_flag, pf_flag_prev, sn_ok, items, n_assign, asg_qw, has_pf, wm_order, pfq_src, pfq_size, out_off, take_base, mn_first;
    std::vector<uint8_t> now_mn;
    // the three per-(key | request, worker) tables of the plan are built IN the pinned buffer K4's ride-along workgroups copy from (a memcpy of ~100 KB per
    // tick otherwise): [wpos nkeys * W][wcnt nkeys * W][pfl_j Q * W] at its head, the small tables behind them (phase_c)
    uint32_t *wpos = nullptr, *wcnt = nullptr, *pfl_j = nullptr; size_t pfl_rows = 0, plan_head_words = 0;
    std::vector<uint32_t> sink_hdr;
    std::vector<uint64_t> holes;                                 // (rq << 32 | logical position) of Retracting tasks taken this tick
    std::vector<std::pair<uint32_t, uint32_t>> freed;            // (worker, variant slot) given back by re-targeted redirects
    std::vector<std::pair<uint32_t, uint64_t>> retract_pairs;  // (old worker, task)
    // cached worker_map iteration order (emulated) for the last worker-id set
    std::vector<uint32_t> cached_ids, cached_order;
};

}  // namespace

struct hqtick_ctx {
    hqtick_config cfg;
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;  // K2 (worker evaluation: PCIe-latency-bound reads of the worker tables) runs here, next to K1 + K1b on `stream`
    bool k2_on_hist = false;       // HQTICK_K2_RIDE_ALONG=1: K2 rides along K1's launch (round 1 / first half of round 2); default: along K1b's
    bool k2_own_stream = false;    // HQTICK_K2_RIDE_ALONG=0: K2 as its own launch on stream2.  Measured (profiles/r02/README.md): K1 alone then takes 4.8 us instead of
                                   // 8.2 us (0.31 vs 0.18 of the HBM roofline), but the second stream's launch + synchronisation make phase A 36 us instead of 28 us:
                                   // the ride-along layout stays the default because the TICK is what counts
    hipEvent_t ev[12] = {};
    std::string err = "";
    // ready set
    DevBuf d_tid, d_tprio, d_trq; uint64_t n_ready = 0; bool resident = false;  // n_ready = physical length (tombstones included)
    DevBuf d_tid2, d_tprio2, d_trq2, d_slice, d_add, d_pre8;   // alternate columns + scratch of the resident deltas (hqtick_ready_*)
    uint64_t n_live = 0; uint32_t last_n_sel = 0; bool last_consumed = true;
    uint64_t max_id = 0; bool max_id_valid = false;  // an upper bound of every resident id (the last one after an upload / a rebuild): a batch that starts above it is appended
    uint32_t n_appends = 0;
    bool consumed_unconfirmed = false;  // HQTICK_FLAG_CONSUME_IN_TICK: the running tick has written its tombstones; cleared when it returns without an error
    hqhost::BlockMemo block_memo;  // host class blocks of earlier ticks (host_model.h)
    uint64_t add_staged_n = 0;  // tasks hqtick_ready_add_stage made room for (0: nothing staged)
    // scans
    DevBuf d_set, d_flags, d_levels, d_nlevels, d_wave_tab, d_hist, d_gkey;
    uint32_t lv_seq = 0; bool set_clean = false; bool no_spec_scan = getenv("HQTICK_NO_SPEC_SCAN") != nullptr; bool levels_valid = false; uint32_t cached_L = 0; std::vector<uint64_t> h_levels; bool timing = true, timing_k1 = false;  // (timing_k1: events around K1 alone, hqtick_set_kernel_timing(ctx, 2))  // level table of the previous tick (re-validated by K1 every tick)
    PinBuf h_up, h_up2, h_q, h_a, h_plan, h_rec, h_sinkhdr, h_add, h_addp, h_retr, h_blk, h_k5a, h_lv;   // (h_lv: the level table as k_sort_levels writes it)
    PinBuf h_blkprof; uint32_t n_blkprof = 0; bool block_profile = false;
    hqprice::DeviceSweeper *pricer = nullptr;  // k_price_sweep: the block sweeps of the coupled placement (csrc/price.hip); HQTICK_PRICE=0 keeps coupled ticks on the host search
    uint32_t block_budget = 4096, block_min_classes = 12;  // k_block_solve: search steps per class before the host solver takes it; classes below which the host solves alone
    // workers / requests
    DevBuf d_up, d_vflags, d_vtmc, d_blk;
    // cluster tables resident in HBM (hqtick_cluster_*): worker rows + request tables in the layout of upload_tables; the host sends rows that changed
    DevBuf d_cluster; PinBuf h_cl, h_cld; bool cluster_valid = false, cluster_check = false, cl_pending = false; uint32_t cl_W = 0, cl_R = 0;
    std::vector<unsigned char> cl_rt;  // host copy of the request-table part as uploaded (compared per tick: a few hundred bytes)
    hipEvent_t cl_ev = nullptr;
    // host mirror of the resident worker set (ABI 7): what a snapshot with worker_id == NULL is completed from, kept current by the hqtick_cluster_* deltas
    struct ClusterMirror {
        bool valid = false; uint32_t n_groups = 1;
        std::vector<uint32_t> id, group; std::vector<uint64_t> total, free_; std::vector<int64_t> rem; std::vector<float> min_util; std::vector<uint8_t> flags;
        std::map<uint32_t, std::vector<std::pair<uint32_t, uint8_t>>> blocked;   // worker id -> (rq, variant)
        std::vector<uint32_t> blk_worker, blk_rq; std::vector<uint8_t> blk_variant; bool blk_dirty = true;  // the same as (worker index, rq, variant) triples
    } mirror;
    DevBuf d_cluster2;  // the re-packed tables of a membership change (swapped with d_cluster)
    // resident table of Retracting tasks (ABI 7): task -> (worker id it is retracting from, still in its queue?, redirect target id / variant)
    struct RetrEntry { uint32_t old_id; bool in_queue; bool has_redirect; uint32_t target_id; uint8_t variant; };
    std::map<uint64_t, RetrEntry> retr;
    std::vector<uint64_t> retr_task, resp_task; std::vector<uint32_t> retr_worker, retr_red_worker, resp_worker; std::vector<uint8_t> retr_red_variant, resp_variant;
    bool sweep_inflight = false;  // an early K5a launch not yet covered by a stream synchronisation
    bool wait_on_kernel = true;   // HQ_HIP_LAST; HQTICK_WAIT_ON_KERNEL=0: hipStreamSynchronize at every wait (A/B)
    // selection + mapping
    DevBuf d_sel_task, d_sel_level, d_map, d_rec, d_tsweep, d_bits, d_pre;
    hqhost::Problem pb;
    PlanScratch plan;
    // results (host)
    std::vector<uint32_t> b_rq, b_size, b_limit, b_cut_off, c_size, c_bl_off, bl_rq, bl_size; std::vector<uint8_t> b_lr, b_blk;
    std::vector<uint32_t> cnt_rq, cnt_worker, cnt_value; std::vector<uint8_t> cnt_variant;
    std::vector<uint32_t> rec_off, retract_off, red_worker, mn_off, mn_worker; std::vector<uint64_t> rec_task, retract_task, red_task, mn_task, new_free;
    std::vector<uint8_t> rec_variant, rec_kind, red_variant, red_kind, q_loaded;
    hqtick_kernel_stats stats{};
    uint32_t tpw_hint = 0;                            // HQTICK_TPW (tuning knob): tasks per wavefront slice
    uint32_t shard_index = 0, shard_count = 1;       // hqtick_set_shard
    void *sink = nullptr; size_t sink_bytes = 0;      // hqtick_set_record_sink (device memory)
    // launch state of the last tick (hqtick_time_kernel re-launches K1 / K4 on it)
    hqk::WaveGeom last_geom{}; uint32_t last_L = 0, last_Q = 0, last_G = 0; size_t last_tb = 0, last_plan_bytes = 0, last_hist_off = 0; bool last_valid = false;
    hqgraph::Graph graph;  // hqtick_graph_*: dependency counters + consumer lists in HBM
    hqtick_ctx *qctx = nullptr;  // hqtick_query's private sub-context
    ncclComm_t comm = nullptr; uint32_t comm_rank = 0, comm_world = 0;  // hqtick_comm_init (RCCL, loaded on first use)
    // sharded placement solve: the ranks split the coupled solve's price sweeps and the separable solve's class blocks, one small all-gather of host
    // buffers per sweep / per launch — through the host's callback (hqtick_set_exchange) or the library's own RCCL communicator
    hqtick_exchange_fn xfn = nullptr; void *xuser = nullptr;
    DevBuf d_xsend, d_xrecv; PinBuf h_xsend, h_xrecv;
    uint32_t shard_min_blocks = 1025, shard_min_classes = 1025; bool shard_solve = true;
    uint32_t block_verify = 2, tick_seq = 0;  // HQTICK_BLOCK_VERIFY: classes of a k_block_solve launch the host re-solves while the kernel runs (host_model.h: Problem::block_verify)
    uint64_t x_calls = 0, x_bytes = 0; double x_us = 0;
    double tl[32] = {}; int ntl = 0;  // debug timeline (us since tick start), hqtick_timeline()
};

namespace {

#define HQ_HIP(call)                                                                                          \
    do {                                                                                                      \
        hipError_t e_ = (call);                                                                               \
        if (e_ != hipSuccess) { ctx->err = std::string(#call) + ": " + hipGetErrorString(e_); return HQTICK_E_DEVICE; } \
    } while (0)

// a launch wrapper that was handed a timer (hqk::time_next_launch) but returned without launching must not leave it to the next launch
#define HQ_HIP_TIMED(call)                                                                                    \
    do {                                                                                                      \
        hipError_t e_ = (call);                                                                               \
        hqk::take_launch_timer();                                                                             \
        if (e_ != hipSuccess) { ctx->err = std::string(#call) + ": " + hipGetErrorString(e_); return HQTICK_E_DEVICE; } \
    } while (0)

// The last launch of a phase carries a stop event (hipExtLaunchKernel: the dispatch's own completion signal), and the host waits on THAT instead of
// hipStreamSynchronize, which submits a marker packet of its own and returns 5.6 us after the kernel is done where the kernel's signal is seen after 2.8 us
// (tools/exp/launch_latency.hip, MI355X / ROCm 7.2).  `launched` = the wrapper really launched (it consumed the pending timer); otherwise the
// caller falls back to the stream.
#define HQ_HIP_LAST(call, launched)                                                                           \
    do {                                                                                                      \
        hipError_t e_ = (call);                                                                               \
        (launched) = hqk::take_launch_timer().stop == nullptr && ctx->wait_on_kernel;                        \
        if (e_ != hipSuccess) { ctx->err = std::string(#call) + ": " + hipGetErrorString(e_); return HQTICK_E_DEVICE; } \
    } while (0)

int fail(hqtick_ctx *ctx, int code, const std::string &msg) { ctx->err = msg; return code; }

// duration between two dispatch events in us, or -1 when the pair was not recorded by a launch of this tick (the runtime's error state is
// cleared: an unrecorded event must not surface as the "last error" of the next launch)
double elapsed_us(hipEvent_t a, hipEvent_t b) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, a, b) == hipSuccess) return (double)ms * 1000.0;
    (void)hipGetLastError();
    return -1.0;
}

int validate(hqtick_ctx *ctx, const hqtick_snapshot *s, bool need_tasks) {
    if (!s) return fail(ctx, HQTICK_E_INVALID, "null snapshot");
    if (s->n_workers && (!s->worker_id || !s->worker_total || !s->worker_free)) return fail(ctx, HQTICK_E_INVALID, "worker arrays missing");
    for (uint32_t w = 1; w < s->n_workers; w++) if (s->worker_id[w - 1] >= s->worker_id[w]) return fail(ctx, HQTICK_E_INVALID, "worker ids not ascending");
    if (s->n_requests && (!s->rq_variant_off || !s->variant_entry_off)) return fail(ctx, HQTICK_E_INVALID, "request arrays missing");
    for (uint32_t q = 0; q < s->n_requests; q++) {
        if (s->rq_variant_off[q + 1] <= s->rq_variant_off[q]) return fail(ctx, HQTICK_E_INVALID, "request without variants");
        if (s->rq_variant_off[q + 1] - s->rq_variant_off[q] > 32) return fail(ctx, HQTICK_E_INVALID, "more than 32 variants");
    }
    uint32_t nv = s->n_requests ? s->rq_variant_off[s->n_requests] : 0;
    for (uint32_t e = 0; e < (nv ? s->variant_entry_off[nv] : 0); e++) {
        if (s->entry_resource[e] >= s->n_resources) return fail(ctx, HQTICK_E_INVALID, "entry resource id out of range");
        if (s->entry_kind[e] == HQ_ENTRY_AMOUNT && s->entry_amount[e] == 0) return fail(ctx, HQTICK_E_INVALID, "zero amount request");
    }
    if (need_tasks && s->n_ready) {
        if (!s->task_id || !s->task_priority || !s->task_rq) return fail(ctx, HQTICK_E_INVALID, "ready-set columns missing");
        for (uint64_t i = 1; i < s->n_ready; i++) if (s->task_id[i - 1] >= s->task_id[i]) return fail(ctx, HQTICK_E_INVALID, "ready set not sorted by task id");
    }
    if (s->n_blocked && (!s->blocked_worker || !s->blocked_rq || !s->blocked_variant)) return fail(ctx, HQTICK_E_INVALID, "blocked arrays missing");
    for (uint32_t k = 0; k < s->n_blocked; k++) {
        if (s->blocked_worker[k] >= s->n_workers) return fail(ctx, HQTICK_E_INVALID, "blocked worker index");
        if (s->blocked_rq[k] >= s->n_requests) return fail(ctx, HQTICK_E_INVALID, "blocked request id out of range");
        if (s->blocked_variant[k] >= s->rq_variant_off[s->blocked_rq[k] + 1] - s->rq_variant_off[s->blocked_rq[k]]) return fail(ctx, HQTICK_E_INVALID, "blocked variant out of range");
    }
    if (s->worker_group) for (uint32_t w = 0; w < s->n_workers; w++) if (s->worker_group[w] >= (s->n_groups ? s->n_groups : 1)) return fail(ctx, HQTICK_E_INVALID, "worker_group >= n_groups");
    // CSR offsets: monotone, and their arrays present.  (The ENTRIES of assigned_* are checked where they are dereferenced — the gap rows of a
    // tick with priority cuts, host_model.cpp — so that the common tick does not pay a pass over every running task.)
    if (s->assigned_off) {
        for (uint32_t w = 0; w < s->n_workers; w++) if (s->assigned_off[w] > s->assigned_off[w + 1]) return fail(ctx, HQTICK_E_INVALID, "assigned_off not monotone");
        if (s->n_workers && s->assigned_off[s->n_workers] && (!s->assigned_rq || !s->assigned_variant)) return fail(ctx, HQTICK_E_INVALID, "assigned arrays missing");
    }
    if (s->prefilled_off) {
        for (uint32_t w = 0; w < s->n_workers; w++) if (s->prefilled_off[w] > s->prefilled_off[w + 1]) return fail(ctx, HQTICK_E_INVALID, "prefilled_off not monotone");
        if (s->n_workers && s->prefilled_off[s->n_workers] && !s->prefilled_rq) return fail(ctx, HQTICK_E_INVALID, "prefilled_rq missing");
    }
    if (s->prefill_off) {
        for (uint32_t q = 0; q < s->n_requests; q++) if (s->prefill_off[q] > s->prefill_off[q + 1]) return fail(ctx, HQTICK_E_INVALID, "prefill_off not monotone");
        const uint32_t np = s->n_requests ? s->prefill_off[s->n_requests] : 0;
        if (np && (!s->prefill_priority || !s->prefill_task || !s->prefill_worker)) return fail(ctx, HQTICK_E_INVALID, "prefill arrays missing");
        for (uint32_t i = 0; i < np; i++) if (s->prefill_worker[i] >= s->n_workers) return fail(ctx, HQTICK_E_INVALID, "prefill worker index");
    }
    if (s->n_retracting && (!s->retracting_task || !s->retracting_worker)) return fail(ctx, HQTICK_E_INVALID, "retracting arrays missing");
    for (uint32_t k = 0; k < s->n_retracting; k++) {
        if (s->retracting_worker[k] >= s->n_workers) return fail(ctx, HQTICK_E_INVALID, "retracting worker index");
        if (k && s->retracting_task[k - 1] >= s->retracting_task[k]) return fail(ctx, HQTICK_E_INVALID, "retracting tasks not ascending");
        if (s->retracting_redirect_worker && s->retracting_redirect_worker[k] != HQ_NO_WORKER && s->retracting_redirect_worker[k] >= s->n_workers) return fail(ctx, HQTICK_E_INVALID, "retracting redirect worker index");
    }
    return 0;
}

struct WorkerEval { const uint8_t *flags = nullptr; const uint32_t *tmc = nullptr; };

// Packs worker tables + request tables into ONE pinned, device-mapped staging buffer that K2 reads in place (every byte
// crosses PCIe once per workgroup; no H2D copy command).
struct UpView { const uint64_t *total, *free_; const int64_t *rem; hqk::RequestTable rt; uint32_t n_entries; };
struct TabLayout { size_t o_tot, o_free, o_rem, o_amt, o_time, o_off, o_res, o_kind, bytes; uint32_t nv, ne; };
TabLayout table_layout(const hqtick_snapshot *s, uint32_t W) {
    TabLayout L{};
    const uint32_t R = s->n_resources;
    L.nv = s->n_requests ? s->rq_variant_off[s->n_requests] : 0;
    L.ne = L.nv ? s->variant_entry_off[L.nv] : 0;
    L.o_tot = 0; L.o_free = L.o_tot + (size_t)W * R * 8; L.o_rem = L.o_free + (size_t)W * R * 8; L.o_amt = L.o_rem + (size_t)W * 8; L.o_time = L.o_amt + (size_t)L.ne * 8;
    L.o_off = L.o_time + (size_t)L.nv * 8; L.o_res = L.o_off + (size_t)(L.nv + 1) * 4; L.o_kind = L.o_res + (size_t)L.ne * 4; L.bytes = L.o_kind + L.ne + 64;
    return L;
}
void pack_worker_rows(unsigned char *h, const TabLayout &L, uint32_t W, uint32_t R, const uint64_t *total, const uint64_t *free_, const int64_t *rem) {
    if (W && R) { memcpy(h + L.o_tot, total, (size_t)W * R * 8); memcpy(h + L.o_free, free_, (size_t)W * R * 8); }
    int64_t *hr = reinterpret_cast<int64_t *>(h + L.o_rem);
    if (rem) memcpy(hr, rem, (size_t)W * 8); else for (uint32_t w = 0; w < W; w++) hr[w] = HQ_NO_TIME_LIMIT;
}
void pack_request_tables(unsigned char *h, const TabLayout &L, const hqtick_snapshot *s) {
    if (!L.nv) return;
    memcpy(h + L.o_amt, s->entry_amount, (size_t)L.ne * 8);
    if (s->variant_min_time_ns) memcpy(h + L.o_time, s->variant_min_time_ns, (size_t)L.nv * 8); else memset(h + L.o_time, 0, (size_t)L.nv * 8);
    memcpy(h + L.o_off, s->variant_entry_off, (size_t)(L.nv + 1) * 4);
    memcpy(h + L.o_res, s->entry_resource, (size_t)L.ne * 4);
    memcpy(h + L.o_kind, s->entry_kind, L.ne);
}
void view_tables(unsigned char *d, const TabLayout &L, UpView *uv) {
    uv->total = (const uint64_t *)(d + L.o_tot); uv->free_ = (const uint64_t *)(d + L.o_free); uv->rem = (const int64_t *)(d + L.o_rem);
    uv->rt.entry_amount = (const uint64_t *)(d + L.o_amt); uv->rt.variant_min_time_ns = (const uint64_t *)(d + L.o_time);
    uv->rt.variant_entry_off = (const uint32_t *)(d + L.o_off); uv->rt.entry_resource = (const uint32_t *)(d + L.o_res);
    uv->rt.entry_kind = (const uint8_t *)(d + L.o_kind); uv->rt.n_variants = L.nv; uv->n_entries = L.ne;
}
int upload_tables(hqtick_ctx *ctx, const hqtick_snapshot *s, uint32_t W, const uint64_t *total, const uint64_t *free_, const int64_t *rem, PinBuf &buf, UpView *uv) {
    const uint32_t R = s->n_resources;
    const TabLayout L = table_layout(s, W);
    if (hqk::worker_eval_lds(R, L.nv, L.ne) > 150 * 1024) return fail(ctx, HQTICK_E_CAPACITY, "request table + 32 worker rows exceed the 150 KiB the worker-evaluation kernel stages in LDS");
    if (!buf.ensure(L.bytes)) return fail(ctx, HQTICK_E_DEVICE, "allocating upload staging");
    unsigned char *h = buf.as<unsigned char>();
    pack_worker_rows(h, L, W, R, total, free_, rem);
    pack_request_tables(h, L, s);
    view_tables(buf.dev<unsigned char>(), L, uv);
    return 0;
}

// The same tables from HBM (hqtick_cluster_upload): worker rows are the caller's responsibility (hqtick_cluster_update_workers), the request tables are
// compared with what was uploaded and re-sent when the snapshot brings new request classes.
int resident_tables(hqtick_ctx *ctx, const hqtick_snapshot *s, uint32_t W, UpView *uv) {
    const uint32_t R = s->n_resources;
    if (W != ctx->cl_W || R != ctx->cl_R) return fail(ctx, HQTICK_E_INVALID, "cluster tables in HBM were uploaded for another worker set (hqtick_cluster_upload after workers join or leave)");
    const TabLayout L = table_layout(s, W);
    if (hqk::worker_eval_lds(R, L.nv, L.ne) > 150 * 1024) return fail(ctx, HQTICK_E_CAPACITY, "request table + 32 worker rows exceed the 150 KiB the worker-evaluation kernel stages in LDS");
    const size_t rt_bytes = L.bytes - L.o_amt;
    if (ctx->cl_pending) { HQ_HIP(hipEventSynchronize(ctx->cl_ev)); ctx->cl_pending = false; }  // h_cl is the staging of the previous request-table upload
    if (L.bytes > ctx->d_cluster.cap) {  // the request tables outgrew the allocation: move the worker rows over
        DevBuf nb;
        if (!nb.ensure(L.bytes * 2)) return fail(ctx, HQTICK_E_DEVICE, "allocating cluster tables");
        HQ_HIP(hipStreamSynchronize(ctx->stream));
        HQ_HIP(hipMemcpy(nb.p, ctx->d_cluster.p, L.o_amt, hipMemcpyDeviceToDevice));
        ctx->d_cluster.release(); ctx->d_cluster = nb;
    }
    if (!ctx->h_cl.ensure(L.bytes)) return fail(ctx, HQTICK_E_DEVICE, "allocating cluster tables");
    unsigned char *h = ctx->h_cl.as<unsigned char>();
    memset(h + L.o_amt, 0, rt_bytes);
    pack_request_tables(h, L, s);
    if (ctx->cl_rt.size() != rt_bytes || memcmp(ctx->cl_rt.data(), h + L.o_amt, rt_bytes) != 0) {  // new request classes: a few hundred bytes, stream-ordered before K2
        HQ_HIP(hipMemcpyAsync(ctx->d_cluster.as<unsigned char>() + L.o_amt, h + L.o_amt, rt_bytes, hipMemcpyHostToDevice, ctx->stream));
        HQ_HIP(hipEventRecord(ctx->cl_ev, ctx->stream)); ctx->cl_pending = true;
        ctx->cl_rt.assign(h + L.o_amt, h + L.o_amt + rt_bytes);
    }
    if (ctx->cluster_check) {  // HQTICK_CHECK_CLUSTER=1 (tests): the rows in HBM must be the rows of the snapshot
        std::vector<unsigned char> dev(L.o_amt);
        HQ_HIP(hipMemcpyAsync(dev.data(), ctx->d_cluster.p, L.o_amt, hipMemcpyDeviceToHost, ctx->stream));
        HQ_HIP(hipStreamSynchronize(ctx->stream));
        std::vector<unsigned char> want(L.o_amt);
        pack_worker_rows(want.data(), L, W, R, s->worker_total, s->worker_free, s->worker_remaining_ns);
        if (memcmp(dev.data(), want.data(), L.o_amt) != 0) return fail(ctx, HQTICK_E_INVALID, "cluster tables in HBM differ from the snapshot's worker rows (a missed hqtick_cluster_update_workers)");
    }
    view_tables(ctx->d_cluster.as<unsigned char>(), L, uv);
    return 0;
}

// K2 on a worker set, synchronous (used for the fake workers of hqtick_query); results land in pinned memory
int eval_workers_sync(hqtick_ctx *ctx, const hqtick_snapshot *s, uint32_t W, const uint64_t *total, const uint64_t *free_, const int64_t *rem, WorkerEval *out) {
    UpView uv; int rc;
    if ((rc = upload_tables(ctx, s, W, total, free_, rem, ctx->h_up2, &uv))) return rc;
    size_t n = (size_t)W * uv.rt.n_variants;
    if (!ctx->h_q.ensure(n * 5 + 64)) return fail(ctx, HQTICK_E_DEVICE, "hipHostMalloc query eval");
    out->tmc = ctx->h_q.as<uint32_t>(); out->flags = ctx->h_q.as<uint8_t>() + n * 4;
    if (n == 0) return 0;
    HQ_HIP(hqk::worker_eval(uv.total, uv.free_, uv.rem, W, s->n_resources, uv.rt, uv.n_entries, ctx->h_q.dev<uint8_t>() + n * 4, ctx->h_q.dev<uint32_t>(), ctx->stream));
    HQ_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

void fill_problem(hqhost::Problem &pb, const hqtick_snapshot *s, const hqtick_config &cfg, const WorkerEval &ev) {
    pb.R = s->n_resources; pb.n_groups = s->n_groups ? s->n_groups : 1; pb.time_limit_s = cfg.mip_time_limit_s; pb.certificate_only = (cfg.flags & HQTICK_FLAG_CERTIFICATE_ONLY) != 0;
    uint32_t nv = s->n_requests ? s->rq_variant_off[s->n_requests] : 0;
    pb.rqs.resize(s->n_requests); pb.variants.resize(nv);
    for (uint32_t q = 0; q < s->n_requests; q++) pb.rqs[q] = {s->rq_variant_off[q], s->rq_variant_off[q + 1] - s->rq_variant_off[q]};
    for (uint32_t v = 0; v < nv; v++) {
        uint32_t e0 = s->variant_entry_off[v];
        pb.variants[v] = {s->entry_resource + e0, s->entry_kind + e0, s->entry_amount + e0, s->variant_entry_off[v + 1] - e0,
                          s->variant_n_nodes ? s->variant_n_nodes[v] : 0, s->variant_weight ? s->variant_weight[v] : 10000,
                          s->variant_min_time_ns ? s->variant_min_time_ns[v] : 0};
    }
    hqhost::WorkerSet &ws = pb.real;
    ws.n = s->n_workers; ws.R = s->n_resources; ws.id = s->worker_id; ws.total = s->worker_total; ws.free_ = s->worker_free;
    ws.remaining_ns = s->worker_remaining_ns; ws.min_util = s->worker_min_utilization; ws.flags = s->worker_flags; ws.group = s->worker_group;
    ws.vflags = ev.flags; ws.vtmc = ev.tmc; ws.n_variant_slots = nv;
    if (ws.blocked.size() != ws.n || ws.n_blocked_lists) { ws.blocked.assign(ws.n, {}); ws.n_blocked_lists = 0; }  // (4096 empty lists re-made per tick cost more than the batches stage)
    ws.assigned_off = s->assigned_off; ws.assigned_rq = s->assigned_rq; ws.assigned_variant = s->assigned_variant;
    for (uint32_t k = 0; k < s->n_blocked; k++) ws.blocked[s->blocked_worker[k]].push_back({s->blocked_rq[k], s->blocked_variant[k]});
    ws.n_blocked_lists = s->n_blocked;
    hqhost::group_equal_rows(ws, false);  // needs nothing of K2's output: in a tick this runs while the GPU works on phase A
}

void export_batches(hqtick_ctx *ctx, const std::vector<hqhost::TaskBatch> &batches, hqtick_result *out) {
    ctx->b_rq.clear(); ctx->b_size.clear(); ctx->b_limit.clear(); ctx->b_lr.clear(); ctx->b_blk.clear();
    ctx->b_cut_off.assign(1, 0); ctx->c_size.clear(); ctx->c_bl_off.assign(1, 0); ctx->bl_rq.clear(); ctx->bl_size.clear();
    for (auto &b : batches) {
        ctx->b_rq.push_back(b.rq); ctx->b_size.push_back(b.size); ctx->b_limit.push_back(b.limit); ctx->b_lr.push_back(b.limit_reached); ctx->b_blk.push_back(b.is_blocker);
        for (auto &c : b.cuts) {
            ctx->c_size.push_back(c.size);
            for (auto &bl : c.blockers) { ctx->bl_rq.push_back(bl.first); ctx->bl_size.push_back(bl.second); }
            ctx->c_bl_off.push_back((uint32_t)ctx->bl_rq.size());
        }
        ctx->b_cut_off.push_back((uint32_t)ctx->c_size.size());
    }
    out->n_batches = (uint32_t)ctx->b_rq.size(); out->batch_rq = ctx->b_rq.data(); out->batch_size = ctx->b_size.data(); out->batch_limit = ctx->b_limit.data();
    out->batch_limit_reached = ctx->b_lr.data(); out->batch_is_blocker = ctx->b_blk.data(); out->batch_cut_off = ctx->b_cut_off.data();
    out->cut_size = ctx->c_size.data(); out->cut_blocker_off = ctx->c_bl_off.data(); out->blocker_rq = ctx->bl_rq.data(); out->blocker_size = ctx->bl_size.data();
}

struct Scan {  // result of GPU phase A
    uint32_t L = 0, Q = 0, G = 0;
    std::vector<uint64_t> levels;
    std::vector<uint32_t> hist;  // [G], g = level*Q + rq
    hqk::WaveGeom geom{};
};

// GPU phase A: K2 on the real workers, level table (cached across ticks, re-validated by K1), K1 + K1b on the ready set in
// ctx->d_t*.  One stream synchronisation in the steady state.
// `while_gpu_runs` (may be null) is host work that needs only the ADDRESSES of K2's output, run between the launches and the sync.
int phase_a(hqtick_ctx *ctx, const hqtick_snapshot *s, WorkerEval *ev, Scan *sc, const std::function<void()> *while_gpu_runs = nullptr) {
    const uint32_t W = s->n_workers, R = s->n_resources, Q = s->n_requests;
    const uint64_t N = ctx->n_ready;
    sc->Q = Q; sc->L = 0; sc->G = 0; sc->levels.clear(); sc->hist.clear();
    if (!ctx->d_set.ensure((size_t)(hqk::PRIO_SET_CAP + hqk::MAX_LEVELS) * 8) || !ctx->h_lv.ensure((size_t)(hqk::MAX_LEVELS + 4) * 8) || !ctx->d_flags.ensure(64) || !ctx->d_levels.ensure((size_t)(hqk::MAX_LEVELS + 2) * 8) || !ctx->d_nlevels.ensure(64))   // (a 40-byte head: count + the first four levels, kernels.hip: k_sort_levels)
        return fail(ctx, HQTICK_E_DEVICE, "hipMalloc phase A");
    const bool scan = N != 0 && Q != 0;
    const uint32_t nvs = Q ? s->rq_variant_off[Q] : 0;
    const size_t nwv = (size_t)W * nvs;
    for (int attempt = 0; attempt < 2; attempt++) {
        uint32_t L = 0;
        bool levels_fresh = false;  // the level table was (re)built by this attempt
        bool spec = false;          // ... and the scan was launched behind the discovery without waiting for it (sized for four levels)
        if (scan) {
            if (!ctx->levels_valid) {
                // (the set and the flag words are EMPTY / zero between ticks: k_sort_levels clears what a discovery used, the scan clears its own flag — the two fills are for
                // the first discovery of a context and for the one after a tick that did not come back)
                if (!ctx->set_clean) {
                    HQ_HIP(hipMemsetAsync(ctx->d_set.p, 0xFF, (size_t)hqk::PRIO_SET_CAP * 8, ctx->stream));
                    HQ_HIP(hipMemsetAsync(ctx->d_flags.p, 0, 64, ctx->stream));
                }
                ctx->set_clean = false;   // until this tick has seen the discovery's result
                const bool time_k0 = ctx->timing;   // (two marker packets around K0: only when every kernel is timed)
                if (time_k0) HQ_HIP(hipEventRecord(ctx->ev[0], ctx->stream));
                HQ_HIP(hqk::distinct_priorities(ctx->d_tprio.as<uint64_t>(), ctx->d_trq.as<uint32_t>(), N, ctx->d_set.as<uint64_t>(), ctx->d_flags.as<uint32_t>(), ctx->stream));
                levels_fresh = true;
                if (time_k0) HQ_HIP(hipEventRecord(ctx->ev[1], ctx->stream));
                // (the sort kernel writes count, flags and the table straight into pinned memory and clears the flag words for the scan: one synchronisation, no copy)
                volatile uint32_t *hl = ctx->h_lv.as<uint32_t>();
                const uint32_t lv_seq = ++ctx->lv_seq ? ctx->lv_seq : ++ctx->lv_seq;   // (never 0)
                hl[0] = 0; hl[1] = 0; hl[2] = 0; hl[3] = 0;
                HQ_HIP(hqk::sort_levels(ctx->d_set.as<uint64_t>(), ctx->d_flags.as<uint32_t>(), ctx->d_levels.as<uint64_t>(), ctx->d_nlevels.as<uint32_t>(), ctx->h_lv.dev<uint64_t>(), lv_seq, ctx->stream));
                // SPECULATIVE SCAN (round 6): with at most 4 levels x Q <= 64 groups — the variant of K1 the BASELINE ticks run — the scan is launched right behind the
                // discovery, sized for four levels, and reads the table and its length from HBM (kernels.h: level_hist, n_levels_dev): no host round trip, no idle GPU
                // between the two (K1 behind a 15-33 us gap took 6.4-7.2 us against 5.5, profiles/r06).  More than four levels: K1 refuses, the table is read below
                // and this loop's second pass launches the general variant.
                spec = attempt == 0 && (uint64_t)4 * Q <= 64 && !ctx->no_spec_scan;
                if (spec) L = 4;   // (levels_valid stays false until the table has been read, after the scan)
                else
                {   // wait on the kernel's own completion word (the stream synchronisation is the fallback after 2 s)
                    const double w0 = now_us();
                    for (uint64_t spins = 0;; spins++) {
                        if (__atomic_load_n(&ctx->h_lv.as<uint32_t>()[3], __ATOMIC_ACQUIRE) == lv_seq) break;
                        if ((spins & 0xFFFF) == 0xFFFF && now_us() - w0 > 2.0e6) { HQ_HIP(hipStreamSynchronize(ctx->stream)); if (__atomic_load_n(&ctx->h_lv.as<uint32_t>()[3], __ATOMIC_ACQUIRE) != lv_seq) return fail(ctx, HQTICK_E_DEVICE, "level discovery did not complete"); break; }
                    }
                }
                if (!spec) {
                L = hl[0];
                const uint32_t flags[2] = {hl[1], hl[2]};
                if (flags[1] || L == 0xFFFFFFFFu || L > hqk::MAX_LEVELS) return fail(ctx, HQTICK_E_CAPACITY, "more than 4096 distinct priority levels in the ready set");
                if (L == 0) return fail(ctx, HQTICK_E_DEVICE, "level discovery returned no level");
                ctx->h_levels.assign(ctx->h_lv.as<uint64_t>() + 2, ctx->h_lv.as<uint64_t>() + 2 + L);
                ctx->set_clean = true;
                }
                float ms = 0;
                if (!spec && time_k0)
                if (hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]) == hipSuccess || (hipEventSynchronize(ctx->ev[1]) == hipSuccess && hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]) == hipSuccess)) ctx->stats.distinct_us = ms * 1000.0;  // (the kernel behind ev[1] has finished: the wait, if the runtime has not noticed yet, is short)
                if (!spec) { ctx->levels_valid = true; ctx->cached_L = L; }
            }
            L = spec ? 4u : ctx->cached_L;
            uint64_t G64 = (uint64_t)L * Q;
            if (G64 > hqk::MAX_GROUPS) return fail(ctx, HQTICK_E_CAPACITY, "levels x requests exceeds 16384 groups");
            sc->L = L; sc->G = (uint32_t)G64;
        }
        // host-visible outputs of phase A, written in place by the kernels: [flags 16][hist G*4][vtmc nwv*4][vflags nwv]
        size_t o_hist = 16, o_tmc = o_hist + (size_t)sc->G * 4, o_fl = o_tmc + nwv * 4, bytes = o_fl + nwv + 16;
        if (!ctx->h_a.ensure(bytes)) return fail(ctx, HQTICK_E_DEVICE, "hipHostMalloc phase A");
        unsigned char *h = ctx->h_a.as<unsigned char>(), *hd = ctx->h_a.dev<unsigned char>();
        memset(h, 0, 16);
        // K2 reads the packed tables from pinned memory; with a ready set to scan it rides along the K1 launch
        bool scan_is_last = false;  // K1b was launched and nothing follows it on the stream
        UpView uv; int rc;
        if (ctx->cluster_valid) { if ((rc = resident_tables(ctx, s, W, &uv))) return rc; }
        else if ((rc = upload_tables(ctx, s, W, s->worker_total, s->worker_free, s->worker_remaining_ns, ctx->h_up, &uv))) return rc;
        hqk::WorkerEvalArgs wea{uv.total, uv.free_, uv.rem, W, R, uv.rt, uv.n_entries, hd + o_fl, reinterpret_cast<uint32_t *>(hd + o_tmc)};
        if (scan) {
            hqk::WaveGeom &g = sc->geom;
            g.waves_per_block = sc->G <= hqk::MAX_GROUPS_4W ? 4 : 1;
            uint64_t tpw = ctx->tpw_hint ? ctx->tpw_hint : 256;
            while (((N + tpw - 1) / tpw) * sc->G > (1ull << 24)) tpw *= 2;  // keep the per-slice table under 64 MiB
            if (!ctx->tpw_hint) while ((N + tpw - 1) / tpw > 32768) tpw *= 2;  // and the rows K1b scans short: one wavefront scans a row of n_waves entries, 4096 per
                                                                                 // step (64 M tasks at 256 per slice: 250 k entries = 400 us for K1b against 160 us for K1;
                                                                                 // 32 k slices of 2048 tasks keep K1 / K4 at 8 k workgroups and K1b at 8 steps)
            g.tasks_per_wave = (uint32_t)tpw; g.n_waves = (uint32_t)((N + tpw - 1) / tpw); g.tab_stride = (g.n_waves + 15u) & ~15u;
            if (!ctx->d_wave_tab.ensure((size_t)g.tab_stride * sc->G * 4) || !ctx->d_gkey.ensure(N * 2 + 16)) return fail(ctx, HQTICK_E_DEVICE, "hipMalloc histogram");
            if (ctx->timing || ctx->timing_k1) hqk::time_next_launch(ctx->ev[2], ctx->ev[3]);
            if (ctx->k2_own_stream) HQ_HIP(hqk::worker_eval(uv.total, uv.free_, uv.rem, W, R, uv.rt, uv.n_entries, hd + o_fl, reinterpret_cast<uint32_t *>(hd + o_tmc), ctx->stream2));
            HQ_HIP_TIMED(hqk::level_hist(ctx->d_tprio.as<uint64_t>(), ctx->d_trq.as<uint32_t>(), N, ctx->d_levels.as<uint64_t>(), ctx->h_levels.data(), L, Q, g, ctx->d_wave_tab.as<uint32_t>(),
                                   ctx->d_gkey.as<uint16_t>(), ctx->d_flags.as<uint32_t>() + 2, ctx->k2_on_hist ? &wea : nullptr, ctx->stream, spec ? ctx->d_nlevels.as<uint32_t>() : nullptr));
            hqk::time_next_launch((ctx->timing && !spec) ? ctx->ev[0] : nullptr, ctx->ev[8]);  // ev[8]: K1b's completion — what the host waits on below
            HQ_HIP_LAST(hqk::scan_waves(ctx->d_wave_tab.as<uint32_t>(), g, sc->G, reinterpret_cast<uint32_t *>(hd + o_hist), ctx->d_flags.as<uint32_t>() + 2,
                                   reinterpret_cast<uint32_t *>(hd) + 2, ctx->stream, (ctx->k2_own_stream || ctx->k2_on_hist) ? nullptr : &wea), scan_is_last);
            if (s->n_retracting) scan_is_last = false;  // k_rank_of follows
            if (s->n_retracting) {  // where do the Retracting tasks sit in their queues?  (mapping.rs:66-80 treats them apart)
                const uint32_t nr = s->n_retracting;
                if (!ctx->h_retr.ensure((size_t)nr * 16 + 64)) return fail(ctx, HQTICK_E_DEVICE, "hipHostMalloc retracting");
                memcpy(ctx->h_retr.p, s->retracting_task, (size_t)nr * 8);
                uint8_t *rd = ctx->h_retr.dev<uint8_t>();
                HQ_HIP(hqk::rank_of(ctx->d_tid.as<uint64_t>(), ctx->d_gkey.as<uint16_t>(), N, ctx->d_wave_tab.as<uint32_t>(), g, reinterpret_cast<const uint64_t *>(rd), nr,
                                    reinterpret_cast<uint32_t *>(rd + (size_t)nr * 8), reinterpret_cast<uint32_t *>(rd + (size_t)nr * 12), ctx->stream));
            }
        } else {
            HQ_HIP(hqk::worker_eval(uv.total, uv.free_, uv.rem, W, R, uv.rt, uv.n_entries, hd + o_fl, reinterpret_cast<uint32_t *>(hd + o_tmc), ctx->stream));
        }
        ev->flags = h + o_fl; ev->tmc = reinterpret_cast<const uint32_t *>(h + o_tmc);
        if (while_gpu_runs) (*while_gpu_runs)();
        if (scan && scan_is_last) HQ_HIP(hipEventSynchronize(ctx->ev[8])); else HQ_HIP(hipStreamSynchronize(ctx->stream));
        if (scan && ctx->k2_own_stream) HQ_HIP(hipStreamSynchronize(ctx->stream2));
        const uint32_t *flags = reinterpret_cast<const uint32_t *>(h);
        if (spec) {   // the discovery's own results, now that everything behind it has finished too
            const uint32_t *hl = ctx->h_lv.as<uint32_t>();
            const uint32_t La = hl[0];
            if (hl[2] || La == 0xFFFFFFFFu || La > hqk::MAX_LEVELS) { ctx->levels_valid = false; return fail(ctx, HQTICK_E_CAPACITY, "more than 4096 distinct priority levels in the ready set"); }
            if (La == 0) { ctx->levels_valid = false; return fail(ctx, HQTICK_E_DEVICE, "level discovery returned no level"); }
            ctx->h_levels.assign(ctx->h_lv.as<uint64_t>() + 2, ctx->h_lv.as<uint64_t>() + 2 + La);
            ctx->levels_valid = true; ctx->cached_L = La; ctx->set_clean = true;
            float ms = 0;
            if (ctx->timing && hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]) == hipSuccess) ctx->stats.distinct_us = ms * 1000.0;
            if ((flags[2] & 4u) || La > 4) continue;   // more levels than the speculative launch was sized for: the table is known now, scan again with the right variant
            sc->L = La; sc->G = La * Q;                // (group keys and the rows of the per-slice table do not depend on how many levels the launch was sized for)
        }
        if (scan && (flags[2] & 2u)) return fail(ctx, HQTICK_E_INVALID, "ready set holds a request id >= n_requests");
        if (scan && (flags[2] & 1u)) {  // a priority the cached level table does not know: rebuild the table once
            ctx->levels_valid = false;
            if (attempt == 0) continue;
            return fail(ctx, HQTICK_E_DEVICE, "level table inconsistent with the ready set");
        }
        ev->flags = h + o_fl; ev->tmc = reinterpret_cast<const uint32_t *>(h + o_tmc);
        if (scan) {
            sc->levels = ctx->h_levels;
            sc->hist.assign(reinterpret_cast<const uint32_t *>(h + o_hist), reinterpret_cast<const uint32_t *>(h + o_hist) + sc->G);
            // A level without a single task (its tasks were handed out or cancelled since the table was built — or the table was built from a column that
            // was still being written) only costs: more groups, and beyond 4 levels / 2048 groups the slower kernel variants.  This tick is right either
            // way (empty levels take part in nothing); the next one rediscovers the levels.
            for (uint32_t l = 0; l < sc->L && ctx->levels_valid && !levels_fresh; l++) {  // (a table built in this very tick is not questioned: no rediscovery loop)
                uint64_t tot = 0;
                for (uint32_t q = 0; q < Q; q++) tot += sc->hist[(size_t)l * Q + q];
                if (tot == 0) ctx->levels_valid = false;
            }
            if (ctx->timing || ctx->timing_k1) { const double us_ = elapsed_us(ctx->ev[2], ctx->ev[3]); if (us_ >= 0) ctx->stats.level_hist_us = us_; }
            if (ctx->timing && !spec) { const double us_ = elapsed_us(ctx->ev[0], ctx->ev[8]); if (us_ >= 0) ctx->stats.scan_us = us_; }  // (a speculative scan shares ev[0] with the discovery: K1b untimed on that tick)
        }
        return 0;
    }
    return 0;
}

// TaskQueue::iter_priority_sizes (taskqueue.rs:273-302) for every request, from the histogram and the prefill sets
std::vector<hqhost::QueueLevels> queue_levels(const Scan &sc, const hqtick_snapshot *s) {
    std::vector<hqhost::QueueLevels> qs(s->n_requests);
    for (uint32_t q = 0; q < s->n_requests; q++) {
        auto &lv = qs[q].levels;
        for (uint32_t l = 0; l < sc.L; l++) { uint32_t c = sc.hist[(size_t)l * sc.Q + q]; if (c) lv.push_back({sc.levels[l], c}); }
        uint32_t pfn = s->prefill_off ? s->prefill_off[q + 1] - s->prefill_off[q] : 0;
        if (pfn) {
            uint64_t pp = s->prefill_priority[q];
            if (!lv.empty() && lv[0].first == pp) lv[0].second += pfn; else lv.insert(lv.begin(), {pp, pfn});
        }
    }
    return qs;
}

// The per-class blocks of the separable placement on the device (csrc/block_solve.hip): tables staged in ONE pinned, device-mapped buffer
// that the kernel reads in place (a few dozen bytes per class) and writes its answers into; one launch, one synchronisation.
struct DeviceBlocks : hqhost::BlockSolver {
    hqtick_ctx *ctx;
    explicit DeviceBlocks(hqtick_ctx *c) : ctx(c) {}
    bool solve(const hqblock::ColTable &ct, const hqblock::ClassTable &cl, const hqblock::Output &out) override { return begin(ct, cl, out) && finish(); }
    bool overlaps() const override { return true; }
    // state between begin() and finish()
    hqblock::Output p_out{}; size_t p_ox = 0, p_ost = 0, p_osteps = 0; uint32_t p_nd = 0, p_nc = 0; bool p_launched = false, p_pending = false;
    bool begin(const hqblock::ColTable &ct, const hqblock::ClassTable &cl, const hqblock::Output &out) override {
        p_pending = false;
        const uint32_t NC = ct.n_cols, R = ct.R, nd = cl.n_classes, ne = ct.ent_off[NC];
        auto al8 = [](size_t v) { return (v + 7) & ~(size_t)7; };
        const size_t o_off = 0, o_res = al8(o_off + (size_t)(NC + 1) * 4), o_w = al8(o_res + (size_t)ne * 4), o_kind = al8(o_w + (size_t)NC * 4), o_amt = al8(o_kind + ne),
                     o_pool = o_amt + (size_t)ne * 8, o_free = (o_pool + (size_t)R * 8 + 15) & ~(size_t)15, o_tot = o_free + (size_t)nd * R * 8, o_elig = o_tot + (size_t)nd * R * 8,
                     o_x = o_elig + (size_t)nd * 8, o_st = o_x + al8((size_t)nd * NC * 4), o_steps = o_st + al8((size_t)nd * 4), bytes = o_steps + al8((size_t)nd * 4) + 64;
        if (!ctx->h_blk.ensure(bytes)) return false;
        if (!ctx->d_blk.ensure(o_x + 64)) return false;
        unsigned char *h = ctx->h_blk.as<unsigned char>(), *d = ctx->d_blk.as<unsigned char>(), *dpin = ctx->h_blk.dev<unsigned char>();
        memcpy(h + o_off, ct.ent_off, (size_t)(NC + 1) * 4); memcpy(h + o_res, ct.ent_res, (size_t)ne * 4); memcpy(h + o_w, ct.weight, (size_t)NC * 4);
        memcpy(h + o_kind, ct.ent_kind, ne); memcpy(h + o_amt, ct.ent_amount, (size_t)ne * 8); memcpy(h + o_pool, ct.pool, (size_t)R * 8);
        memcpy(h + o_free, cl.free_, (size_t)nd * R * 8); memcpy(h + o_tot, cl.total, (size_t)nd * R * 8); memcpy(h + o_elig, cl.elig, (size_t)nd * 8);
        // inputs: one copy into HBM (929 wavefronts reading the same table through PCIe reads would queue behind each other); outputs: written by the
        // kernel straight into pinned memory
        if (hqk::copy_pinned_to_hbm(dpin, d, o_x, ctx->stream) != hipSuccess) return false;  // (a copy kernel, not the copy engine: ~10 us less per launch)
        hqblock::ColTable dct{NC, R, (const uint32_t *)(d + o_off), (const uint32_t *)(d + o_res), (const uint8_t *)(d + o_kind), (const uint64_t *)(d + o_amt), (const uint32_t *)(d + o_w), (const double *)(d + o_pool), d, (uint32_t)o_free};  // [0, o_free) = the column table: staged into LDS by the kernel
        hqblock::ClassTable dcl{nd, (const uint64_t *)(d + o_free), (const uint64_t *)(d + o_tot), (const uint64_t *)(d + o_elig)};
        uint64_t *dprof = nullptr;
        if (ctx->block_profile) {  // HQTICK_BLOCK_PROFILE=1: per-class stage timestamps (tools/block_profile.py)
            if (!ctx->h_blkprof.ensure((size_t)nd * 64 + 64)) return false;
            memset(ctx->h_blkprof.p, 0, (size_t)nd * 64);
            dprof = ctx->h_blkprof.dev<uint64_t>(); ctx->n_blkprof = nd;
        }
        hqblock::Output dout{(uint32_t *)(dpin + o_x), (uint32_t *)(dpin + o_st), (uint32_t *)(dpin + o_steps), dprof};
        hqk::time_next_launch(ctx->timing ? ctx->ev[9] : nullptr, ctx->ev[10]);
        const hipError_t be = hqblock::block_solve(dct, dcl, dout, ctx->block_budget, ctx->stream);
        const bool launched = hqk::take_launch_timer().stop == nullptr && ctx->wait_on_kernel;
        if (be != hipSuccess) return false;
        p_out = out; p_ox = o_x; p_ost = o_st; p_osteps = o_steps; p_nd = nd; p_nc = NC; p_launched = launched; p_pending = true;
        return true;
    }
    bool finish() override {
        if (!p_pending) return false;
        p_pending = false;
        if ((p_launched ? hipEventSynchronize(ctx->ev[10]) : hipStreamSynchronize(ctx->stream)) != hipSuccess) return false;  // the kernel's own completion signal (HQ_HIP_LAST)
        const unsigned char *h = ctx->h_blk.as<unsigned char>();
        memcpy(p_out.x, h + p_ox, (size_t)p_nd * p_nc * 4); memcpy(p_out.status, h + p_ost, (size_t)p_nd * 4); memcpy(p_out.steps, h + p_osteps, (size_t)p_nd * 4);
        if (ctx->timing) { const double us_ = elapsed_us(ctx->ev[9], ctx->ev[10]); if (us_ >= 0) ctx->stats.block_solve_us = us_; }
        ctx->stats.n_classes_device = p_nd;
        return true;
    }
};

// The ranks' exchange of small host buffers (csrc/price.h: Exchange): the host's callback if one is set, else the library's RCCL communicator.
bool rccl_allgather_host(hqtick_ctx *ctx, const void *send, void *recv, size_t bytes);  // (bottom of this file, next to the other RCCL calls)
struct CtxExchange : hqprice::Exchange {
    hqtick_ctx *ctx;
    explicit CtxExchange(hqtick_ctx *c) : ctx(c) { rank = c->shard_index; world = c->shard_count; }
    static bool available(const hqtick_ctx *c) { return c->shard_solve && c->shard_count > 1 && (c->xfn || (c->comm && c->comm_world == c->shard_count)); }
    bool allgather(const void *send, void *recv, size_t bytes) override {
        if (ctx->xfn) return ctx->xfn(ctx->xuser, send, recv, bytes) == 0;
        return rccl_allgather_host(ctx, send, recv, bytes);
    }
    ~CtxExchange() override { ctx->x_calls += n_calls; ctx->x_bytes += n_bytes; ctx->x_us += us; }
};

// The class blocks of a separable tick over the ranks: class i of the launch goes to rank i % world (the launch is ordered longest first: a strided deal spreads the
// hard classes), every rank runs its share through the inner solver, ONE all-gather of (x, status, steps) completes the answer on every rank.  A rank whose inner
// solver fails reports its classes as unsolved — every rank then sends the same classes to its host solver, and the replicas stay in step.
struct ShardedBlocks : hqhost::BlockSolver {
    hqhost::BlockSolver &inner; hqprice::Exchange &ex; uint32_t min_classes;
    hqblock::Output p_out{}; uint32_t p_nd = 0, p_nc = 0, p_nm = 0; bool p_pass = false, p_inner_ok = false, p_pending = false;
    std::vector<uint64_t> cfree, ctot, celig; std::vector<uint32_t> dx, dst, dsteps; std::vector<unsigned char> send, recv;
    ShardedBlocks(hqhost::BlockSolver &in, hqprice::Exchange &e, uint32_t mc) : inner(in), ex(e), min_classes(mc) {}
    bool overlaps() const override { return inner.overlaps(); }
    bool solve(const hqblock::ColTable &ct, const hqblock::ClassTable &cl, const hqblock::Output &out) override { return begin(ct, cl, out) && finish(); }
    bool begin(const hqblock::ColTable &ct, const hqblock::ClassTable &cl, const hqblock::Output &out) override {
        p_pass = ex.world <= 1 || cl.n_classes < min_classes;
        if (p_pass) return inner.begin(ct, cl, out);
        const uint32_t nd = cl.n_classes, R = ct.R, NC = ct.n_cols, W = ex.world, me = ex.rank;
        const uint32_t nm = nd > me ? (nd - me + W - 1) / W : 0;
        cfree.resize((size_t)nm * R); ctot.resize((size_t)nm * R); celig.resize(nm);
        for (uint32_t j = 0; j < nm; j++) {
            const size_t i = (size_t)me + (size_t)j * W;
            memcpy(cfree.data() + (size_t)j * R, cl.free_ + i * R, (size_t)R * 8); memcpy(ctot.data() + (size_t)j * R, cl.total + i * R, (size_t)R * 8); celig[j] = cl.elig[i];
        }
        dx.assign((size_t)nm * NC, 0); dst.assign(nm, hqblock::ST_UNSUPPORTED); dsteps.assign(nm, 0);
        p_out = out; p_nd = nd; p_nc = NC; p_nm = nm; p_pending = true;
        p_inner_ok = nm == 0 || inner.begin(ct, hqblock::ClassTable{nm, cfree.data(), ctot.data(), celig.data()}, hqblock::Output{dx.data(), dst.data(), dsteps.data(), nullptr});
        return true;  // (whatever the inner solver said: this rank still owes the others its part of the exchange)
    }
    bool finish() override {
        if (p_pass) return inner.finish();
        if (!p_pending) return false;
        p_pending = false;
        if (p_nm && !(p_inner_ok && inner.finish())) { std::fill(dst.begin(), dst.end(), (uint32_t)hqblock::ST_UNSUPPORTED); std::fill(dx.begin(), dx.end(), 0u); }
        const uint32_t W = ex.world, NC = p_nc, mx = (p_nd + W - 1) / W;
        const size_t o_st = (size_t)mx * NC * 4, o_steps = o_st + (size_t)mx * 4, bytes = (o_steps + (size_t)mx * 4 + 7) & ~(size_t)7;
        send.assign(bytes, 0); recv.resize(bytes * W);
        if (p_nm) { memcpy(send.data(), dx.data(), (size_t)p_nm * NC * 4); memcpy(send.data() + o_st, dst.data(), (size_t)p_nm * 4); memcpy(send.data() + o_steps, dsteps.data(), (size_t)p_nm * 4); }
        const double t0 = now_us();
        if (!ex.allgather(send.data(), recv.data(), bytes)) return false;
        ex.us += now_us() - t0; ex.n_calls++; ex.n_bytes += bytes * W;
        for (uint32_t r = 0; r < W; r++) {
            const unsigned char *b = recv.data() + (size_t)r * bytes;
            const uint32_t nr = p_nd > r ? (p_nd - r + W - 1) / W : 0;
            for (uint32_t j = 0; j < nr; j++) {
                const size_t i = (size_t)r + (size_t)j * W;
                memcpy(p_out.x + i * NC, b + (size_t)j * NC * 4, (size_t)NC * 4); memcpy(p_out.status + i, b + o_st + (size_t)j * 4, 4); memcpy(p_out.steps + i, b + o_steps + (size_t)j * 4, 4);
            }
        }
        return true;
    }
};

// One tick, stage by stage.  The stages share the plan scratch of the ctx (no per-tick allocation in the steady state) and a handful of
// sizes; run() is the only entry point and mirrors run_scheduling_inner (scheduler/main.rs:50-72).
struct TickRun {
    hqtick_ctx *ctx; const hqtick_snapshot *s; hqtick_result *out; const bool use_resident;
    PlanScratch &ps; hqhost::Problem &pb;
    const uint32_t W, R, Q;
    static constexpr uint32_t NONE = 0xFFFFFFFFu;
    WorkerEval ev; Scan sc; hqhost::Counts cnt;
    uint64_t N = 0; uint32_t L = 0; int status = HQTICK_DONE;
    uint32_t nkeys = 0, max_count = 0, max_nk = 0, n_bit_words = 0, n_sel = 0, n_pfq = 0, n_rec = 0, max_items = 0;
    std::vector<uint32_t> pfq_rq;                             // requests that prefill in this tick, ascending
    std::vector<std::pair<uint32_t, uint32_t>> retr_pos;      // (rq, queue position) of every Retracting task
    std::vector<std::vector<uint32_t>> key_T;                 // lazily built T_k(s) tables of worker_of()
    size_t o_rv = 0, o_rk = 0, o_mn = 0, o_fl = 0;            // layout of the pinned record buffer
    bool sweep_launched = false;  // K5a went out right after the key tables (launch_sweep_early)
    uint32_t n_tr = 0;            // keys K4 stores worker-major (PlanScratch::key_tr)
    const bool transpose_on = !(getenv("HQTICK_TRANSPOSE") && atoi(getenv("HQTICK_TRANSPOSE")) == 0);
    bool may_reorder = false;  // the mapping kernel needs its stable sort: several priority levels, Retracting holes or prefilled tasks inside the queues
    bool compact = false, delta16 = false; size_t o_rs = 0, o_rf = 0; uint32_t max_out = 0;  // compact emission (HQTICK_FLAG_COMPACT_RECORDS / _DELTA16)
    uint64_t *h_rec_task = nullptr, *mn_ids = nullptr; uint8_t *h_rec_var = nullptr, *h_rec_kind = nullptr;
    bool assembled = false;
    double t0 = 0;

    TickRun(hqtick_ctx *c, const hqtick_snapshot *snap, hqtick_result *o, bool resident)
        : ctx(c), s(snap), out(o), use_resident(resident), ps(c->plan), pb(c->pb), W(snap->n_workers), R(snap->n_resources), Q(snap->n_requests) {}

    void mark() { if (ctx->ntl < 32) ctx->tl[ctx->ntl++] = now_us() - t0; }
    uint32_t hist(uint32_t l, uint32_t q) const { return sc.hist[(size_t)l * Q + q]; }

    // host mirror of the K5 arithmetic: the worker that receives the task at index idx of key k's take_tasks() vector
    uint32_t worker_of(uint32_t k, uint32_t idx) {
        cnt.pairs();  // (rare path: redirects)
        const auto &pk = cnt.per_key[k];
        std::vector<uint32_t> &T = key_T[k];
        if (T.empty()) {
            uint32_t maxc = 0; for (auto &wc : pk) maxc = std::max(maxc, wc.second);
            std::vector<uint32_t> ge(maxc + 2, 0);
            for (auto &wc : pk) ge[wc.second]++;
            uint32_t more = (uint32_t)pk.size(), acc = 0;
            for (uint32_t sw = 0; sw <= maxc; sw++) { T.push_back(acc); more -= ge[sw]; acc += more; }
        }
        uint32_t sw = (uint32_t)(std::upper_bound(T.begin(), T.end(), idx) - T.begin()) - 1, nth = idx - T[sw];
        for (auto &wc : pk) if (wc.second > sw) { if (nth == 0) return wc.first; nth--; }
        return HQ_NO_WORKER;
    }

    // the ready set: uploaded with the snapshot, or already resident
    int load_ready_set() {
        if (!use_resident) {
            uint64_t N = s->n_ready;
            if (!ctx->d_tid.ensure(N * 8 + 8) || !ctx->d_tprio.ensure(N * 8 + 8) || !ctx->d_trq.ensure(N * 4 + 8)) return fail(ctx, HQTICK_E_DEVICE, "hipMalloc ready set");
            if (N) {
                HQ_HIP(hipMemcpyAsync(ctx->d_tid.p, s->task_id, N * 8, hipMemcpyHostToDevice, ctx->stream));
                HQ_HIP(hipMemcpyAsync(ctx->d_tprio.p, s->task_priority, N * 8, hipMemcpyHostToDevice, ctx->stream));
                HQ_HIP(hipMemcpyAsync(ctx->d_trq.p, s->task_rq, N * 4, hipMemcpyHostToDevice, ctx->stream));
            }
            ctx->n_ready = N; ctx->resident = false; ctx->levels_valid = false;
        } else if (!ctx->resident) return fail(ctx, HQTICK_E_INVALID, "hqtick_run_resident without hqtick_upload_ready");
        N = ctx->n_ready;
        return 0;
    }

    // create_task_mapping as index arithmetic, part 1: the take sequence of every queue and the per-key tables  (mapping.rs:36-157)
    int plan_keys() {
        L = sc.L;
        // per request: logical take sequence = [first level][prefilled][rest] when the prefill priority equals the top
        // priority of the queue, else [prefilled][all levels]   (taskqueue.rs:320-355)
        ps.q_total.assign(Q, 0); ps.pf_n.assign(Q, 0); ps.pf_start.assign(Q, 0); ps.seq_taken.assign(Q, 0);
        for (uint32_t q = 0; q < Q; q++) {
            for (uint32_t l = 0; l < L; l++) ps.q_total[q] += hist(l, q);
            ps.pf_n[q] = s->prefill_off ? s->prefill_off[q + 1] - s->prefill_off[q] : 0;
            if (ps.pf_n[q]) {
                uint32_t first = L; for (uint32_t l = 0; l < L; l++) if (hist(l, q)) { first = l; break; }
                ps.pf_start[q] = (first < L && sc.levels[first] == s->prefill_priority[q]) ? hist(first, q) : 0;
            }
        }
        nkeys = (uint32_t)cnt.keys.size();
        ps.key_seg.assign(nkeys, 0); ps.key_sum.assign(nkeys, 0); ps.key_rq.assign(nkeys, 0); ps.key_var_w.assign((nkeys + 3) / 4 + 1, 0);
        ps.key_ord_off.assign(nkeys + 1, 0); ps.key_t_off.assign(nkeys + 1, 0); ps.key_bits_off.assign(nkeys + 1, 0);
        const bool by_class = cnt.by_class && cnt.wclass.size() == W && cnt.key_col.size() == nkeys;
        ctx->last_valid = false;  // the previous tick's plan is overwritten from here on: hqtick_ready_consume_last refers to THIS tick or to none (if it fails)
        {   // the big tables live at the head of the pinned plan buffer; everything else of the plan (phase_c) is a few KB behind them
            ps.plan_head_words = (size_t)nkeys * W * 2 + (size_t)Q * W;
            size_t n_cnt0 = 0; for (uint32_t k = 0; k < nkeys; k++) n_cnt0 += cnt.key_size(k);
            const size_t small = (size_t)9 * (nkeys + 4) + n_cnt0 + (size_t)6 * (Q + 2) + (W + 2) + (size_t)4 * sc.G + (size_t)2 * s->n_retracting + 64;
            if (!ctx->h_plan.ensure((ps.plan_head_words + small) * 4 + 64)) return fail(ctx, HQTICK_E_DEVICE, "hipHostMalloc plan");
            ps.wpos = ctx->h_plan.as<uint32_t>(); ps.wcnt = ps.wpos + (size_t)nkeys * W; ps.pfl_j = ps.wcnt + (size_t)nkeys * W; ps.pfl_rows = 0;
        }
        if (!by_class) { std::fill(ps.wpos, ps.wpos + (size_t)nkeys * W, NONE); std::fill(ps.wcnt, ps.wcnt + (size_t)nkeys * W, 0u); }  // (by class: every row is written in full below)
        ps.items.assign(W, 0); ps.n_assign.assign(W, 0); ps.asg_qw.assign((size_t)Q * W, 0);
        size_t n_cnt = 0; for (uint32_t k = 0; k < nkeys; k++) n_cnt += cnt.key_size(k);
        ctx->cnt_rq.resize(n_cnt); ctx->cnt_variant.resize(n_cnt); ctx->cnt_worker.resize(n_cnt); ctx->cnt_value.resize(n_cnt); ps.ord_cnt.resize(n_cnt);
        uint32_t *c_rq = ctx->cnt_rq.data(), *c_w = ctx->cnt_worker.data(), *c_v = ctx->cnt_value.data(), *c_ord = ps.ord_cnt.data(); uint8_t *c_var = ctx->cnt_variant.data();
        size_t ci = 0;
        max_count = 0; max_nk = 0;
        if (by_class) {  // position of every worker in each distinct worker list (Map order), built once per list
            ps.list_pos.assign(cnt.lists.size() * (size_t)W, NONE);
            for (size_t l = 0; l < cnt.lists.size(); l++) {
                uint32_t *lp = ps.list_pos.data() + l * W; const std::vector<uint32_t> &wi = cnt.lists[l].widx;
                for (uint32_t i = 0; i < wi.size(); i++) lp[wi[i]] = i;
            }
        }
        for (uint32_t k = 0; k < nkeys; k++) {
            const uint32_t q = cnt.keys[k].first; const uint8_t v = cnt.keys[k].second;
            ps.key_rq[k] = q; reinterpret_cast<uint8_t *>(ps.key_var_w.data())[k] = v;
            uint32_t sum = 0, maxc = 0, pos = 0;
            uint32_t *wp = ps.wpos + (size_t)k * W, *wcn = ps.wcnt + (size_t)k * W, *aq = ps.asg_qw.data() + (size_t)q * W, *items = ps.items.data(), *nas = ps.n_assign.data();
            if (by_class) {  // separable tick: the count of a worker is its class's; sequential passes instead of scattering (worker, count) pairs
                const uint32_t *xg = cnt.class_x.data() + cnt.key_col[k], *wcl = cnt.wclass.data(); const uint32_t NCc = cnt.n_cols;
                const std::vector<uint32_t> &wi = cnt.lists[cnt.key_list[k]].widx;
                if (cnt.one_class) { const uint32_t c = xg[0]; for (uint32_t w = 0; w < W; w++) { wcn[w] = c; items[w] += c; aq[w] += c; } }  // a cold tick on identical workers (vectorises)
                else for (uint32_t w = 0; w < W; w++) { const uint32_t c = xg[(size_t)wcl[w] * NCc]; wcn[w] = c; items[w] += c; aq[w] += c; }
                memcpy(wp, ps.list_pos.data() + (size_t)cnt.key_list[k] * W, (size_t)W * 4);
                pos = (uint32_t)wi.size();
                std::fill(c_rq + ci, c_rq + ci + pos, q); memset(c_var + ci, v, pos); memcpy(c_w + ci, wi.data(), (size_t)pos * 4);
                if (cnt.one_class) { const uint32_t c = xg[0]; std::fill(c_v + ci, c_v + ci + pos, c); std::fill(c_ord + ci, c_ord + ci + pos, c); sum = c * pos; maxc = c; }
                else for (uint32_t i = 0; i < pos; i++) { const uint32_t c = wcn[wi[i]]; c_v[ci + i] = c; c_ord[ci + i] = c; sum += c; maxc = std::max(maxc, c); }
                ci += pos;
            } else for (auto &wc : (cnt.pairs(), cnt.per_key[k])) {  // (worker, count) in the Map's iteration order
                const uint32_t w = wc.first, c = wc.second;
                sum += c; maxc = std::max(maxc, c);
                c_rq[ci] = q; c_var[ci] = v; c_w[ci] = w; c_v[ci] = c; c_ord[ci] = c; ci++;
                wp[w] = pos++; wcn[w] = c; items[w] += c; nas[w] += c; aq[w] += c;
            }
            ps.key_ord_off[k + 1] = (uint32_t)ci;
            ps.key_t_off[k + 1] = ps.key_t_off[k] + maxc + 1;  // sweeps 0..maxc
            ps.key_bits_off[k + 1] = ps.key_bits_off[k] + (maxc + 1) * ((pos + 63) / 64);
            max_count = std::max(max_count, maxc); max_nk = std::max(max_nk, pos);
            ps.key_seg[k] = ps.seq_taken[q]; ps.key_sum[k] = sum; ps.seq_taken[q] += sum;
            if (ps.seq_taken[q] > ps.q_total[q] + ps.pf_n[q]) return fail(ctx, HQTICK_E_QUEUE_UNDERFLOW, "solver placed more tasks than the queue holds (reference panics, taskqueue.rs:327)");
        }
        if (by_class) memcpy(ps.n_assign.data(), ps.items.data(), (size_t)W * 4);  // equal until the redirects are taken off
        n_bit_words = ps.key_bits_off[nkeys];
        // Worker-major selection (kernels.hip: SelPlanN): a request served by ONE key that starts at the head of its queue and gives every one of its workers the same
        // count — the cold tick on identical workers, and every saturated class of a homogeneous cluster — is written by K4 so that each worker's ids are contiguous.
        // Only plain queues: no prefill set in front of the request, no Retracting task anywhere in this tick (holes are positions of the queue order).
        ps.key_tr.assign(nkeys, 0);
        n_tr = 0;
        if (transpose_on && s->n_retracting == 0) {
            std::vector<uint32_t> &nk = ps.pf_drained;  // (scratch: keys per request; plan_redirects re-assigns it)
            nk.assign(Q, 0);
            for (uint32_t k = 0; k < nkeys; k++) nk[ps.key_rq[k]]++;
            for (uint32_t k = 0; k < nkeys; k++) {
                const uint32_t q = ps.key_rq[k], n = ps.key_ord_off[k + 1] - ps.key_ord_off[k], c = ps.key_t_off[k + 1] - ps.key_t_off[k] - 1;
                if (nk[q] != 1 || ps.key_seg[k] != 0 || ps.pf_n[q] != 0 || n == 0 || c == 0 || c > 0xFFFFu || n > 0xFFFFu || (uint64_t)n * c != ps.key_sum[k]) continue;
                ps.key_tr[k] = c; n_tr++;
            }
        }
        // multi-node placements take one task each from the head of their queue (mapping.rs:133-154)
        ps.mn_first.assign(cnt.mn_rq.size(), 0);
        for (size_t i = 0; i < cnt.mn_rq.size(); i++) {
            uint32_t q = cnt.mn_rq[i];
            ps.mn_first[i] = ps.seq_taken[q]; ps.seq_taken[q] += (uint32_t)cnt.mn_sets[i].size();
            if (ps.pf_n[q]) return fail(ctx, HQTICK_E_UNSUPPORTED, "multi-node queue with a prefill set");
            if (ps.seq_taken[q] > ps.q_total[q]) return fail(ctx, HQTICK_E_QUEUE_UNDERFLOW, "multi-node placement exceeds its queue");
        }
        return 0;
    }

    // part 2: tasks that leave a prefill set or are Retracting in their queue — redirects instead of records  (mapping.rs:66-101)
    int plan_redirects() {
        // already-prefilled tasks that this tick hands out: Prefilled{old} -> retract + redirect  (mapping.rs:81-101)
        ps.retract_pairs.clear();
        ctx->red_task.clear(); ctx->red_worker.clear(); ctx->red_variant.clear();
        ps.pf_drained.assign(Q, 0);
        ps.has_pf.assign((size_t)Q * W, 0);  // SingleNodeTaskAssignment::prefilled_tasks as per-(rq, worker) counts
        if (s->prefilled_off) for (uint32_t w = 0; w < W; w++) for (uint32_t i = s->prefilled_off[w]; i < s->prefilled_off[w + 1]; i++) if (s->prefilled_rq[i] < Q) ps.has_pf[(size_t)s->prefilled_rq[i] * W + w]++;
        key_T.assign(nkeys, {});
        ctx->red_kind.clear();
        for (uint32_t k = 0; k < nkeys; k++) {
            const uint32_t q = cnt.keys[k].first;
            if (!ps.pf_n[q]) continue;
            uint32_t a = std::max(ps.key_seg[k], ps.pf_start[q]), b = std::min(ps.key_seg[k] + ps.key_sum[k], ps.pf_start[q] + ps.pf_n[q]);
            for (uint32_t p = a; p < b; p++) {
                uint32_t slot = s->prefill_off[q] + (p - ps.pf_start[q]);
                uint64_t task = s->prefill_task[slot]; uint32_t oldw = s->prefill_worker[slot];
                uint32_t neww = worker_of(k, p - ps.key_seg[k]);
                ps.retract_pairs.push_back({oldw, task});
                if (oldw < W && ps.has_pf[(size_t)q * W + oldw]) ps.has_pf[(size_t)q * W + oldw]--;
                ctx->red_task.push_back(task); ctx->red_worker.push_back(neww); ctx->red_variant.push_back(cnt.keys[k].second); ctx->red_kind.push_back(HQ_REDIRECT_FROM_PREFILL);
                if (neww < W) { ps.asg_qw[(size_t)q * W + neww]--; ps.n_assign[neww]--; }  // a redirect is not a new `assigned` record
                ps.pf_drained[q]++;
            }
        }
        // ready tasks in state Retracting{old} that this tick takes (mapping.rs:66-80): no `assigned` record, a redirect instead
        ps.holes.clear(); ps.freed.clear();
        retr_pos.clear();  // (rq, queue position) of every Retracting task, for the prefill check
        if (s->n_retracting) {
            if (N == 0) return fail(ctx, HQTICK_E_INVALID, "retracting tasks without a ready set");
            const uint32_t nr = s->n_retracting;
            const uint8_t *rh = ctx->h_retr.as<uint8_t>();
            const uint32_t *rkey = reinterpret_cast<const uint32_t *>(rh + (size_t)nr * 8), *rrank = reinterpret_cast<const uint32_t *>(rh + (size_t)nr * 12);
            for (uint32_t i = 0; i < nr; i++) {
                if (rkey[i] == 0xFFFFFFFFu || Q == 0) return fail(ctx, HQTICK_E_INVALID, "a retracting task is not in the ready set");
                const uint32_t l = rkey[i] / Q, q = rkey[i] % Q;
                uint32_t z = rrank[i]; for (uint32_t l2 = 0; l2 < l; l2++) z += hist(l2, q);         // position in the queue (levels, then id)
                const uint32_t p = z < ps.pf_start[q] ? z : z + ps.pf_n[q];                             // position in the logical take sequence
                retr_pos.push_back({q, z});
                if (p >= ps.seq_taken[q]) continue;                                                   // stays in its queue
                uint32_t k = nkeys;
                for (uint32_t kk = 0; kk < nkeys; kk++) if (ps.key_rq[kk] == q && p >= ps.key_seg[kk] && p < ps.key_seg[kk] + ps.key_sum[kk]) { k = kk; break; }
                if (k == nkeys) return fail(ctx, HQTICK_E_UNSUPPORTED, "a Retracting task was taken by a multi-node placement");
                const uint32_t neww = worker_of(k, p - ps.key_seg[k]), oldw = s->retracting_worker[i];
                const uint8_t v = cnt.keys[k].second;
                ps.holes.push_back(((uint64_t)q << 32) | p);
                if (neww < W) { ps.asg_qw[(size_t)q * W + neww]--; ps.n_assign[neww]--; }
                ctx->red_task.push_back(s->retracting_task[i]); ctx->red_worker.push_back(neww); ctx->red_variant.push_back(v);
                if (oldw != neww) {
                    ctx->red_kind.push_back(HQ_REDIRECT_RETARGET);
                    const uint32_t tw = s->retracting_redirect_worker ? s->retracting_redirect_worker[i] : HQ_NO_WORKER;
                    if (tw != HQ_NO_WORKER) ps.freed.push_back({tw, pb.rqs[q].first_variant + (s->retracting_redirect_variant ? s->retracting_redirect_variant[i] : 0)});  // remove_sn_task(previous target)
                } else {
                    ctx->red_kind.push_back(HQ_REDIRECT_SAME_WORKER);
                }
            }
            std::sort(ps.holes.begin(), ps.holes.end());
        }
        // queue tasks taken per request (excluding the prefilled block)
        ps.zq_taken.assign(Q, 0);
        for (uint32_t q = 0; q < Q; q++) ps.zq_taken[q] = ps.seq_taken[q] - ps.pf_drained[q];
        // workers that received a multi-node task are no longer SN (set_mn_task)
        ps.now_mn.assign(W, 0);
        for (auto &sets : cnt.mn_sets) for (auto &set : sets) for (uint32_t w : set) ps.now_mn[w] = 1;
        return 0;
    }

    // part 3: process_proactive_filling  (mapping.rs:159-234)
    int plan_prefill() {
        if (s->worker_map_rank) { ps.wm_order.assign(W, 0); for (uint32_t w = 0; w < W; w++) ps.wm_order[s->worker_map_rank[w]] = w; }
        else {
            if (ps.cached_ids.size() != W || (W && memcmp(ps.cached_ids.data(), s->worker_id, (size_t)W * 4) != 0)) {
                ps.cached_ids.assign(s->worker_id, s->worker_id + W);
                hqhb::insertion_order_u32(s->worker_id, W, ps.cached_order);
            }
        }
        const std::vector<uint32_t> &wm_order = s->worker_map_rank ? ps.wm_order : ps.cached_order;  // (no copy of the cached order)
        ps.sn_ok.resize(W);  // per worker: still a single-node worker after this tick's multi-node placements
        for (uint32_t w = 0; w < W; w++) ps.sn_ok[w] = ((s->worker_flags ? (s->worker_flags[w] & HQ_WORKER_SN) != 0 : true) && !ps.now_mn[w]) ? 1u : 0u;
        ps.new_pf_total.assign(Q, 0); ps.pfl_size.assign(Q, 0);
        ps.pfq_src.clear(); ps.pfq_size.clear(); ps.pfl_rows = 0;
        pfq_rq.clear();
        {
            // state of every queue after the takes: first level that still has tasks
            std::vector<int> top_level(Q, -1); std::vector<uint32_t> top_left(Q, 0);
            uint64_t global_top = 0;
            for (uint32_t q = 0; q < Q; q++) {
                uint32_t left = ps.zq_taken[q];
                for (uint32_t l = 0; l < L; l++) { uint32_t h = hist(l, q); if (h > left) { top_level[q] = (int)l; top_left[q] = h - left; break; } left -= h; }
                if (top_level[q] >= 0) global_top = std::max(global_top, sc.levels[top_level[q]]);  // TaskQueues::top_priority  taskqueue.rs:62-68
            }
            size_t prev_row = SIZE_MAX;  // the pfl_j row that belongs to pf_flag_prev
            for (uint32_t q = 0; q < Q; q++) {
                if (top_level[q] < 0 || sc.levels[top_level[q]] != global_top) continue;
                bool pf_left = ps.pf_n[q] > ps.pf_drained[q];
                uint32_t tsz = (pf_left && s->prefill_priority[q] != global_top) ? 0 : top_left[q];  // top_size_no_prefill  taskqueue.rs:241-253
                uint32_t size = tsz > ctx->cfg.proactive_filling_reserve ? tsz - ctx->cfg.proactive_filling_reserve : 0;
                if (!size) continue;
                // eligible workers: flags in index order (a branch-free pass), their ranks in worker_map order only when the set differs from the
                // previous request's (on a saturated tick every worker serves every class: one walk of the order for all requests)
                const uint32_t *aq = ps.asg_qw.data() + (size_t)q * W, *hp = ps.has_pf.data() + (size_t)q * W;
                ps.pf_flag.resize(W);
                uint32_t n_elig = 0;
                {
                    uint32_t *fl = ps.pf_flag.data(); const uint32_t *sn = ps.sn_ok.data();
                    for (uint32_t w = 0; w < W; w++) { const uint32_t f = sn[w] & (aq[w] != 0 ? 1u : 0u) & (hp[w] == 0 ? 1u : 0u); fl[w] = f; n_elig += f; }  // (vectorises)
                }
                if (!n_elig) continue;
                uint32_t psz = std::min(size / n_elig, ctx->cfg.proactive_filling_max);
                if (!psz) continue;
                ps.pfl_size[q] = psz;
                pfq_rq.push_back(q);
                const size_t o = ps.pfl_rows++ * W;  // (at most Q rows: one per request)
                if (prev_row != SIZE_MAX && memcmp(ps.pf_flag.data(), ps.pf_flag_prev.data(), (size_t)W * 4) == 0) memcpy(ps.pfl_j + o, ps.pfl_j + prev_row, (size_t)W * 4);
                else {
                    uint32_t *row = ps.pfl_j + o; const uint32_t *fl = ps.pf_flag.data();
                    uint32_t j = 0;
                    for (uint32_t w : wm_order) { const uint32_t f = fl[w]; row[w] = f ? j : NONE; j += f; }
                    ps.pf_flag_prev.swap(ps.pf_flag);
                }
                prev_row = o;
                ps.new_pf_total[q] = psz * n_elig;
            }
        }
        for (auto &rp : retr_pos) {  // take_tasks_for_prefill on a Retracting task: the reference asserts task.is_waiting()  (mapping.rs:221)
            const uint32_t q = rp.first, z = rp.second;
            if (ps.new_pf_total[q] && z >= ps.zq_taken[q] && z < ps.zq_taken[q] + ps.new_pf_total[q])
                return fail(ctx, HQTICK_E_UNSUPPORTED, "a Retracting task reached take_tasks_for_prefill: the reference asserts task.is_waiting() (mapping.rs:221)");
        }
        return 0;
    }

    // part 4: what K4 selects per (level, rq) group, where every worker's records go, capacity checks
    int plan_outputs() {
        // ---- selection plan per (level, rq) group ----
        ps.rq_sel_base.assign(Q + 1, 0);
        for (uint32_t q = 0; q < Q; q++) ps.rq_sel_base[q + 1] = ps.rq_sel_base[q] + ps.zq_taken[q] + ps.new_pf_total[q];
        n_sel = ps.rq_sel_base[Q];
        ps.take_base.assign((size_t)4 * sc.G, 0);  // [take][base][tnc][tsb] per (level, rq) group (kernels.hip: SelPlanN)
        for (uint32_t q = 0; q < Q; q++) {
            uint32_t want = ps.zq_taken[q] + ps.new_pf_total[q], cum = 0;
            for (uint32_t l = 0; l < L; l++) {
                uint32_t h = hist(l, q), t = want > cum ? std::min(h, want - cum) : 0;
                ps.take_base[(size_t)l * Q + q] = t; ps.take_base[(size_t)sc.G + (size_t)l * Q + q] = ps.rq_sel_base[q] + cum;
                cum += h;
            }
        }
        if (n_tr) {
            if (!ps.holes.empty()) return fail(ctx, HQTICK_E_UNSUPPORTED, "internal: worker-major selection on a tick with holes");  // (cannot happen: no Retracting task, no hole)
            for (uint32_t k = 0; k < nkeys; k++) {
                if (!ps.key_tr[k]) continue;
                const uint32_t q = ps.key_rq[k], n = ps.key_ord_off[k + 1] - ps.key_ord_off[k];
                for (uint32_t l = 0; l < L; l++) { ps.take_base[(size_t)2 * sc.G + (size_t)l * Q + q] = (n << 16) | ps.key_tr[k]; ps.take_base[(size_t)3 * sc.G + (size_t)l * Q + q] = ps.rq_sel_base[q]; }
            }
        }
        for (uint32_t q : pfq_rq) { ps.pfq_src.push_back(ps.rq_sel_base[q] + ps.zq_taken[q]); ps.pfq_size.push_back(ps.pfl_size[q]); }
        n_pfq = (uint32_t)pfq_rq.size();
        // ---- output offsets ----
        ps.out_off.assign(W + 1, 0);
        max_items = 0; max_out = 0;
        compact = (ctx->cfg.flags & (HQTICK_FLAG_COMPACT_RECORDS | HQTICK_FLAG_COMPACT_DELTA16)) != 0 && !ctx->sink;
        for (uint32_t w = 0; w < W; w++) {
            if (ctx->shard_count > 1 && hqhb::hash_worker_id(s->worker_id[w]) % ctx->shard_count != ctx->shard_index) { ps.out_off[w + 1] = ps.out_off[w]; continue; }  // another rank's worker
            uint32_t npf = 0;
            for (uint32_t pi = 0; pi < n_pfq; pi++) if (ps.pfl_j[(size_t)pi * W + w] != NONE) npf += ps.pfq_size[pi];
            ps.out_off[w + 1] = ps.out_off[w] + npf + ps.n_assign[w];
            max_items = std::max(max_items, ps.items[w]);
            max_out = std::max(max_out, npf + ps.n_assign[w]);
        }
        n_rec = ps.out_off[W];
        may_reorder = sc.L > 1 || !ps.holes.empty() || (s->prefill_off && s->prefill_off[Q] > 0);
        if (hqk::expand_mapping_lds(max_items, nkeys, compact ? max_out : 0, may_reorder) > 150 * 1024) return fail(ctx, HQTICK_E_CAPACITY, "a worker receives more tasks in one tick than the mapping kernel stages in LDS");
        if (max_nk > hqk::SWEEP_MAX_WORKERS) return fail(ctx, HQTICK_E_CAPACITY, "more than 24576 workers share one (request, variant) placement: beyond the round-robin kernel's LDS staging");
        return 0;
    }

    // everything of the result that needs no GPU output: runs while the GPU works on phase C
    void assemble_host_part() {
        assembled = true;
        ctx->rec_off = ps.out_off;
        ctx->retract_off.assign(W + 1, 0); ctx->retract_task.assign(ps.retract_pairs.size(), 0);
        {
            for (auto &rp : ps.retract_pairs) if (rp.first < W) ctx->retract_off[rp.first + 1]++;
            for (uint32_t w = 0; w < W; w++) ctx->retract_off[w + 1] += ctx->retract_off[w];
            std::vector<uint32_t> cur(ctx->retract_off.begin(), ctx->retract_off.end() - 1);
            for (auto &rp : ps.retract_pairs) if (rp.first < W) ctx->retract_task[cur[rp.first]++] = rp.second;  // stable: per worker in emission order
        }
        // Worker::insert_sn_task for every placed task (server/worker.rs:188-196 -> workerload.rs:156-165)
        ctx->new_free.assign(s->worker_free, s->worker_free + (size_t)W * R);
        for (uint32_t k = 0; k < nkeys; k++) {
            const hqhost::VariantView &vv = pb.variants[pb.rqs[cnt.keys[k].first].first_variant + cnt.keys[k].second];
            const uint32_t *wcn = ps.wcnt + (size_t)k * W;  // this key's count per worker (0 = none): the plan's own row
            for (uint32_t w = 0; w < W; w++) {
                const uint32_t c = wcn[w];
                if (!c) continue;
                for (uint32_t e = 0; e < vv.n_entries; e++) {
                    uint64_t &f = ctx->new_free[(size_t)w * R + vv.res[e]];
                    if (vv.kind[e] == HQ_ENTRY_ALL) f = 0; else { uint64_t d = vv.amount[e] * (uint64_t)c; f = f > d ? f - d : 0; }
                }
            }
        }
        for (auto &fr : ps.freed) {  // remove_sn_task on the previous target of a re-targeted redirect  (worker.rs:223-234 -> workerload.rs:194-202)
            const hqhost::VariantView &vv = pb.variants[fr.second];
            for (uint32_t e = 0; e < vv.n_entries; e++) {
                uint64_t &f = ctx->new_free[(size_t)fr.first * R + vv.res[e]];
                f = vv.kind[e] == HQ_ENTRY_ALL ? s->worker_total[(size_t)fr.first * R + vv.res[e]] : f + vv.amount[e];
            }
        }
    }

    // K5a needs the key tables only (counts in Map order per key): it goes out as soon as plan_keys() has them, reading its four small tables in place
    // from pinned memory, and runs while the host plans redirects, prefills and outputs — 5.4 us and a launch boundary off the critical path.
    int launch_sweep_early() {
        sweep_launched = false;
        if (nkeys == 0 || n_bit_words == 0 || max_nk > hqk::SWEEP_MAX_WORKERS) return 0;  // (the capacity error is reported by plan_outputs)
        if (n_tr == nkeys) { sweep_launched = true; return 0; }  // every key is stored worker-major: no bit row is ever read — no launch at all
        if (ctx->sweep_inflight) { HQ_HIP(hipStreamSynchronize(ctx->stream)); ctx->sweep_inflight = false; }  // a tick that failed after its early launch: its kernel still reads h_k5a
        const size_t n_ord = ps.ord_cnt.size(), words = 4 * (size_t)(nkeys + 1) + n_ord;
        if (!ctx->h_k5a.ensure(words * 4 + 64) || !ctx->d_tsweep.ensure((size_t)ps.key_t_off[nkeys] * 4 + 16) || !ctx->d_bits.ensure((size_t)n_bit_words * 8 + 16) ||
            !ctx->d_pre.ensure((size_t)n_bit_words * 4 + 16))
            return fail(ctx, HQTICK_E_DEVICE, "hipMalloc mapping");
        uint32_t *h = ctx->h_k5a.as<uint32_t>();
        const uint32_t *d = ctx->h_k5a.dev<uint32_t>();
        const size_t o_toff = 0, o_ordoff = nkeys + 1, o_boff = 2 * (size_t)(nkeys + 1), o_tr = 3 * (size_t)(nkeys + 1), o_ord = 4 * (size_t)(nkeys + 1);
        memcpy(h + o_toff, ps.key_t_off.data(), (size_t)(nkeys + 1) * 4); memcpy(h + o_ordoff, ps.key_ord_off.data(), (size_t)(nkeys + 1) * 4);
        memcpy(h + o_boff, ps.key_bits_off.data(), (size_t)(nkeys + 1) * 4); memcpy(h + o_tr, ps.key_tr.data(), (size_t)nkeys * 4); if (n_ord) memcpy(h + o_ord, ps.ord_cnt.data(), n_ord * 4);
        hqk::MapKeys mk{};
        mk.n_keys = nkeys; mk.key_t_off = d + o_toff; mk.key_ord_off = d + o_ordoff; mk.key_bits_off = d + o_boff; mk.ord_cnt = d + o_ord; mk.key_tr = d + o_tr;
        mk.t_sweep = ctx->d_tsweep.as<uint32_t>(); mk.bits = ctx->d_bits.as<uint64_t>(); mk.pre = ctx->d_pre.as<uint32_t>();
        if (ctx->timing) hqk::time_next_launch(ctx->ev[1], ctx->ev[6]);
        HQ_HIP_TIMED(hqk::sweep_bits(mk, max_count, max_nk, ctx->stream));
        sweep_launched = true; ctx->sweep_inflight = true;
        return 0;
    }

    // GPU phase C: selection, round-robin bit rows, per-worker expansion; records land in pinned memory (or the HBM sink)
    int phase_c() {
        size_t n_mn_ids = 0; for (auto &sets : cnt.mn_sets) n_mn_ids += sets.size();
        delta16 = compact && (ctx->cfg.flags & HQTICK_FLAG_COMPACT_DELTA16) != 0;
        if (compact) {  // [rec_lo u32 x n_rec | units u16 x 4 n_rec][run_span (start, count) x W][runs 12 | 16 B x n_rec] (at most one run per record)
            o_rs = delta16 ? (size_t)n_rec * 8 : (((size_t)n_rec * 4 + 7) & ~(size_t)7); o_rf = o_rs + (size_t)W * 8;
            o_rv = o_rk = 0; o_mn = (o_rf + (size_t)n_rec * (delta16 ? 16 : 12) + 7) & ~(size_t)7;
        } else { o_rv = (size_t)n_rec * 8; o_rk = o_rv + n_rec; o_mn = (o_rk + n_rec + 7) & ~(size_t)7; }
        o_fl = o_mn + n_mn_ids * 8;
        const size_t rec_bytes = o_fl + 64;
        if (!ctx->h_rec.ensure(rec_bytes)) return fail(ctx, HQTICK_E_DEVICE, "hipHostMalloc records");
        h_rec_task = ctx->h_rec.as<uint64_t>(); h_rec_var = ctx->h_rec.as<uint8_t>() + o_rv; h_rec_kind = ctx->h_rec.as<uint8_t>() + o_rk;
        mn_ids = reinterpret_cast<uint64_t *>(ctx->h_rec.as<uint8_t>() + o_mn);
        if (n_sel) {
            if (!ctx->d_sel_task.ensure((size_t)n_sel * 8) || !ctx->d_sel_level.ensure((size_t)n_sel * 2 + 2))
                return fail(ctx, HQTICK_E_DEVICE, "hipMalloc selection");
            // the plan in one upload: the big tables are already at the head of the pinned buffer (plan_keys / plan_prefill), the small ones go behind them
            uint32_t *hp = ctx->h_plan.as<uint32_t>();
            size_t cur = ps.plan_head_words;
            auto put = [&](const std::vector<uint32_t> &v) { const size_t o = cur; if (!v.empty()) memcpy(hp + cur, v.data(), v.size() * 4); cur += v.size(); if (v.empty()) hp[cur++] = 0; return o; };
            const size_t o_wpos = 0, o_wcnt = (size_t)nkeys * W, o_pflj = (size_t)2 * nkeys * W;
            size_t o_rq = put(ps.key_rq), o_var = put(ps.key_var_w), o_seg = put(ps.key_seg), o_ordoff = put(ps.key_ord_off), o_ord = put(ps.ord_cnt), o_toff = put(ps.key_t_off),
                   o_boff = put(ps.key_bits_off), o_tr = put(ps.key_tr), o_base = put(ps.rq_sel_base), o_pfs = put(ps.pf_start), o_pfn = put(ps.pf_n),
                   o_pqs = put(ps.pfq_src), o_pqz = put(ps.pfq_size), o_out = put(ps.out_off);
            size_t o_tb = put(ps.take_base);
            if (cur & 1) hp[cur++] = 0;  // 8-byte alignment for the u64 hole list
            const size_t o_holes = cur;
            for (uint64_t hk : ps.holes) { hp[cur++] = (uint32_t)(hk & 0xFFFFFFFFu); hp[cur++] = (uint32_t)(hk >> 32); }
            if (ps.holes.empty()) { hp[cur++] = 0; hp[cur++] = 0; }
            const size_t plan_words = cur;
            if (plan_words * 4 + 16 > ctx->h_plan.cap) return fail(ctx, HQTICK_E_DEVICE, "mapping plan larger than its buffer");  // (sized in plan_keys)
            if (!ctx->d_map.ensure(plan_words * 4 + 16) ||
                !ctx->d_tsweep.ensure((size_t)ps.key_t_off[nkeys] * 4 + 16) || !ctx->d_bits.ensure((size_t)n_bit_words * 8 + 16) || !ctx->d_pre.ensure((size_t)n_bit_words * 4 + 16))
                return fail(ctx, HQTICK_E_DEVICE, "hipMalloc mapping");
            mark();  // 6: pack
            const uint32_t *d = ctx->d_map.as<uint32_t>();
            hqk::MapKeys mk{};
            mk.n_keys = nkeys; mk.key_rq = d + o_rq; mk.key_variant = reinterpret_cast<const uint8_t *>(d + o_var); mk.key_seg_start = d + o_seg;
            mk.key_ord_off = d + o_ordoff; mk.ord_cnt = d + o_ord; mk.key_t_off = d + o_toff; mk.key_bits_off = d + o_boff; mk.key_tr = d + o_tr;
            mk.t_sweep = ctx->d_tsweep.as<uint32_t>(); mk.bits = ctx->d_bits.as<uint64_t>(); mk.pre = ctx->d_pre.as<uint32_t>();
            mk.wpos = d + o_wpos; mk.wcnt = d + o_wcnt; mk.rq_sel_base = d + o_base; mk.rq_pf_start = d + o_pfs; mk.rq_pf_n = d + o_pfn;
            mk.n_holes = (uint32_t)ps.holes.size(); mk.holes = reinterpret_cast<const uint64_t *>(d + o_holes);
            mk.n_pfq = n_pfq; mk.pfq_src = d + o_pqs; mk.pfq_size = d + o_pqz; mk.pfl_j = d + o_pflj; mk.out_off = d + o_out;
            uint32_t *flags = reinterpret_cast<uint32_t *>(ctx->h_rec.as<uint8_t>() + o_fl);
            flags[0] = 0;  // K5b reports a capacity overflow straight into this pinned word
            ctx->last_n_sel = n_sel; ctx->last_consumed = false;
            const bool consume_in_tick = use_resident && (ctx->cfg.flags & HQTICK_FLAG_CONSUME_IN_TICK) != 0;  // K4 writes the tombstones of what it selects
            ctx->last_geom = sc.geom; ctx->last_L = L; ctx->last_Q = Q; ctx->last_G = sc.G; ctx->last_tb = o_tb; ctx->last_plan_bytes = plan_words * 4; ctx->last_valid = true;
            if (ctx->timing) hqk::time_next_launch(ctx->ev[4], ctx->ev[5]);
            HQ_HIP_TIMED(hqk::select_scatter(ctx->d_tid.as<uint64_t>(), ctx->d_gkey.as<uint16_t>(), N, Q, sc.G, sc.geom, ctx->d_wave_tab.as<uint32_t>(), ctx->h_plan.as<uint32_t>() + o_tb,
                                d + o_tb, ctx->d_sel_task.as<uint64_t>(), ctx->d_sel_level.as<uint16_t>(), ctx->h_plan.dev<void>(), ctx->d_map.p, plan_words * 4,
                                consume_in_tick ? ctx->d_trq.as<uint32_t>() : nullptr, ctx->stream, consume_in_tick ? 1u : 0u));
            if (consume_in_tick) { ctx->n_live -= n_sel; ctx->last_consumed = true; ctx->consumed_unconfirmed = true; }  // (confirmed when the tick returns without an error: run_tick)
            if (!sweep_launched && n_tr < nkeys) {
                if (ctx->timing) hqk::time_next_launch(ctx->ev[1], ctx->ev[6]);
                HQ_HIP_TIMED(hqk::sweep_bits(mk, max_count, max_nk, ctx->stream));
            }
            uint8_t *drec = ctx->h_rec.dev<uint8_t>();  // K5b writes the records straight into the caller-visible pinned buffer (PCIe-bound, no copy command)
            uint64_t *k_task = reinterpret_cast<uint64_t *>(drec); uint8_t *k_var = drec + o_rv, *k_kind = drec + o_rk;
            if (ctx->sink) {  // multi-GPU: records stay in HBM, laid out for the all-gather (include/hqtick.h)
                const uint32_t cap = hqtick_sink_capacity_records(W, ctx->sink_bytes);
                if (n_rec > cap || hqtick_sink_bytes(W, cap) > ctx->sink_bytes) return fail(ctx, HQTICK_E_CAPACITY, "record sink too small for this tick");
                uint8_t *sk = reinterpret_cast<uint8_t *>(ctx->sink);
                const size_t so_off = 16, so_task = (so_off + (size_t)(W + 1) * 4 + 7) & ~(size_t)7, so_var = so_task + (size_t)cap * 8, so_kind = so_var + cap;
                // header + rec_off ride in the plan buffer's tail: stage them in pinned memory and copy with the stream
                std::vector<uint32_t> &hdr = ps.sink_hdr; hdr.assign(4 + W + 1, 0);
                hdr[0] = n_rec; hdr[1] = cnt.checksum(); hdr[2] = HQTICK_SINK_MAGIC; hdr[3] = cap;
                memcpy(hdr.data() + 4, ps.out_off.data(), (size_t)(W + 1) * 4);
                if (!ctx->h_sinkhdr.ensure(hdr.size() * 4)) return fail(ctx, HQTICK_E_DEVICE, "hipHostMalloc sink header");
                memcpy(ctx->h_sinkhdr.p, hdr.data(), hdr.size() * 4);
                HQ_HIP(hipMemcpyAsync(sk, ctx->h_sinkhdr.p, hdr.size() * 4, hipMemcpyHostToDevice, ctx->stream));
                k_task = reinterpret_cast<uint64_t *>(sk + so_task); k_var = sk + so_var; k_kind = sk + so_kind;
            }
            hqk::time_next_launch(ctx->timing ? ctx->ev[7] : nullptr, ctx->ev[11]);  // ev[11]: K5b's completion — what the host waits on below
            hqk::CompactOut co{};
            if (compact) co = hqk::CompactOut{reinterpret_cast<uint32_t *>(drec), reinterpret_cast<uint2 *>(drec + o_rs), reinterpret_cast<uint32_t *>(drec + o_rf),
                                              delta16 ? reinterpret_cast<uint16_t *>(drec) : nullptr};
            bool expand_is_last = false;
            HQ_HIP_LAST(hqk::expand_mapping(mk, W, ctx->d_sel_task.as<uint64_t>(), ctx->d_sel_level.as<uint16_t>(), Q, max_items, k_task, k_var, k_kind,
                                reinterpret_cast<uint32_t *>(drec + o_fl), co, max_out, may_reorder, ctx->stream), expand_is_last);
            if (!cnt.mn_rq.empty()) expand_is_last = false;  // copies of the multi-node task ids follow
            // multi-node tasks: the heads of their queues
            {
                size_t pos = 0;
                for (size_t i = 0; i < cnt.mn_rq.size(); i++) {
                    size_t n = cnt.mn_sets[i].size();
                    HQ_HIP(hipMemcpyAsync(mn_ids + pos, ctx->d_sel_task.as<uint64_t>() + ps.rq_sel_base[cnt.mn_rq[i]] + ps.mn_first[i], n * 8, hipMemcpyDeviceToHost, ctx->stream));
                    pos += n;
                }
            }
            mark();  // 7: phase C enqueued
            assemble_host_part();
            if (expand_is_last) HQ_HIP(hipEventSynchronize(ctx->ev[11])); else HQ_HIP(hipStreamSynchronize(ctx->stream));
            ctx->sweep_inflight = false;
            if (flags[0]) return fail(ctx, HQTICK_E_CAPACITY, "mapping kernel capacity exceeded");
            if (ctx->timing) { const double us_ = elapsed_us(ctx->ev[4], ctx->ev[5]); if (us_ >= 0) ctx->stats.select_us = us_; }
            if (ctx->timing) { const double us_ = elapsed_us(ctx->ev[1], ctx->ev[6]); if (us_ >= 0) ctx->stats.sweep_us = us_; }
            if (ctx->timing) { const double us_ = elapsed_us(ctx->ev[7], ctx->ev[11]); if (us_ >= 0) ctx->stats.other_us = us_; }
        }
        if (!n_sel) { ctx->last_n_sel = 0; ctx->last_consumed = true; }
        if (!n_sel && ctx->sink) {  // nothing placed: still publish an empty, well-formed sink
            if (hqtick_sink_bytes(W, 0) > ctx->sink_bytes) return fail(ctx, HQTICK_E_CAPACITY, "record sink too small for this tick");
            std::vector<uint32_t> &hdr = ps.sink_hdr; hdr.assign(4 + W + 1, 0);
            hdr[1] = cnt.checksum(); hdr[2] = HQTICK_SINK_MAGIC; hdr[3] = hqtick_sink_capacity_records(W, ctx->sink_bytes);
            HQ_HIP(hipMemcpy(ctx->sink, hdr.data(), hdr.size() * 4, hipMemcpyHostToDevice));
        }
        return 0;
    }

    void finish(double t1, double t2, double t3) {
        if (!assembled) assemble_host_part();
        ctx->mn_task.clear(); ctx->mn_off.assign(1, 0); ctx->mn_worker.clear();
        {
            size_t pos = 0;
            for (size_t i = 0; i < cnt.mn_rq.size(); i++) for (auto &set : cnt.mn_sets[i]) {
                ctx->mn_task.push_back(mn_ids[pos++]);
                for (uint32_t w : set) ctx->mn_worker.push_back(w);
                ctx->mn_off.push_back((uint32_t)ctx->mn_worker.size());
            }
        }
        out->status = status;
        out->n_counts = (uint32_t)ctx->cnt_rq.size(); out->count_rq = ctx->cnt_rq.data(); out->count_variant = ctx->cnt_variant.data();
        out->count_worker = ctx->cnt_worker.data(); out->count_value = ctx->cnt_value.data();
        out->rec_off = ctx->rec_off.data();
        if (compact && n_sel) {
            const uint8_t *hb = ctx->h_rec.as<uint8_t>();
            out->run_span = reinterpret_cast<const hqtick_run_span *>(hb + o_rs);
            if (delta16) { out->rec_delta16 = reinterpret_cast<const uint16_t *>(hb); out->runs16 = reinterpret_cast<const hqtick_rec_run16 *>(hb + o_rf); }
            else { out->rec_task_lo = reinterpret_cast<const uint32_t *>(hb); out->runs = reinterpret_cast<const hqtick_rec_run *>(hb + o_rf); }
        } else if (!ctx->sink) { out->rec_task = h_rec_task; out->rec_variant = h_rec_var; out->rec_kind = h_rec_kind; }
        out->retract_off = ctx->retract_off.data(); out->retract_task = ctx->retract_task.data();
        out->n_redirects = (uint32_t)ctx->red_task.size(); out->redirect_task = ctx->red_task.data(); out->redirect_worker = ctx->red_worker.data(); out->redirect_variant = ctx->red_variant.data(); out->redirect_kind = ctx->red_kind.data();
        out->n_mn = (uint32_t)ctx->mn_task.size(); out->mn_task = ctx->mn_task.data(); out->mn_worker_off = ctx->mn_off.data(); out->mn_worker = ctx->mn_worker.data();
        out->new_free = ctx->new_free.data();
        mark();  // 9: result assembled
        double t5 = now_us();
        out->t_total_us = t5 - t0; out->t_scan_us = t1 - t0; out->t_batches_us = t2 - t1; out->t_solve_us = t3 - t2; out->t_mapping_us = t5 - t3;
        uint64_t n_pref = 0; for (uint32_t q = 0; q < Q; q++) n_pref += ps.new_pf_total[q];
        uint64_t n_asg = 0; for (uint32_t w = 0; w < W; w++) n_asg += ps.n_assign[w];
        uint32_t nv = Q ? s->rq_variant_off[Q] : 0;
        ctx->stats.n_assigned = n_asg; ctx->stats.n_prefilled = n_pref;
        ctx->stats.algorithmic_bytes = N * 20 + (uint64_t)W * R * 16 + (uint64_t)nv * R * 9 + n_asg * 13 + n_pref * 12;
        ctx->stats.tick_gpu_us = ctx->stats.distinct_us + ctx->stats.level_hist_us + ctx->stats.scan_us + ctx->stats.block_solve_us + ctx->stats.select_us + ctx->stats.sweep_us + ctx->stats.other_us;
    }

    int run() {
        t0 = now_us();
        ctx->ntl = 0;
        if (ctx->cfg.flags & HQTICK_FLAG_NO_TICK_CACHES) { ctx->levels_valid = false; hqhost::flush_tick_caches(); }  // nothing derived survives from the last tick
        memset(out, 0, sizeof(*out));
        ctx->stats = hqtick_kernel_stats{};
        int rc = validate(ctx, s, !use_resident);
        if (rc) return rc;
        HQ_HIP(hipSetDevice(ctx->device));
        if ((rc = load_ready_set())) return rc;
        // ---------------- GPU phase A ----------------
        const std::function<void()> prep = [&]() {   // request/worker views: no GPU output needed yet
            fill_problem(pb, s, ctx->cfg, ev);
            // HQTICK_FLAG_CERTIFICATE_ONLY is for ONE scheduler: replicas (a shard of several, a record sink that is merged with others', an exchange) must return the
            // canonical point — what the merge compares byte for byte — so the flag is ignored there (include/hqtick.h).
            if (pb.certificate_only && (ctx->shard_count > 1 || ctx->sink || ctx->xfn || ctx->comm)) pb.certificate_only = false;
        };
        if ((rc = phase_a(ctx, s, &ev, &sc, &prep))) return rc;
        mark();  // 0: phase A done
        const double t1 = now_us();
        // ---------------- host: batches + placement ----------------
        std::vector<hqhost::QueueLevels> qlv = queue_levels(sc, s);
        std::vector<hqhost::TaskBatch> batches = hqhost::create_task_batches(pb, qlv);
        export_batches(ctx, batches, out);
        mark();  // 1: batches
        const double t2 = now_us();
        DeviceBlocks dev_blocks(ctx);
        pb.blocks = &dev_blocks; pb.block_min_classes = ctx->block_min_classes; pb.pricer = ctx->pricer;
        pb.block_verify = ctx->block_verify; pb.tick_seq = ctx->tick_seq++;
        pb.memo = (ctx->cfg.flags & (HQTICK_FLAG_NO_BLOCK_MEMO | HQTICK_FLAG_NO_TICK_CACHES)) ? nullptr : &ctx->block_memo;
        double shard_sweep_us = -1.0;
        ctx->x_calls = 0; ctx->x_bytes = 0; ctx->x_us = 0;
        if (CtxExchange::available(ctx)) {  // the ranks of a sharded scheduler split the sweeps and the class blocks (no-ops below their thresholds)
            CtxExchange xch(ctx);
            ShardedBlocks sh_blocks(dev_blocks, xch, ctx->shard_min_classes);
            pb.blocks = &sh_blocks;
            if (ctx->pricer) {
                hqprice::ShardedSweeper sh_sweeps(*ctx->pricer, xch, ctx->shard_min_blocks);
                pb.pricer = &sh_sweeps;
                cnt = hqhost::run_scheduling_solver(pb, batches);
                shard_sweep_us = sh_sweeps.stat_sweep_us;
            } else cnt = hqhost::run_scheduling_solver(pb, batches);
        } else cnt = hqhost::run_scheduling_solver(pb, batches);
        pb.blocks = nullptr; pb.pricer = nullptr;
        ctx->stats.price_sweeps = (uint32_t)cnt.price_sweeps; ctx->stats.price_rounds = (uint32_t)cnt.price_rounds; ctx->stats.price_us = cnt.price_us; ctx->stats.milp_us = cnt.milp_us; ctx->stats.model_us = cnt.model_us; ctx->stats.solve_pre_us = cnt.pre_us;
        ctx->stats.price_sweep_us = shard_sweep_us >= 0.0 ? shard_sweep_us : (ctx->pricer ? ctx->pricer->stat_sweep_us : 0.0); ctx->stats.milp_cols = (uint32_t)cnt.milp_cols; ctx->stats.milp_rows = (uint32_t)cnt.milp_rows;
        ctx->stats.n_classes_verified = cnt.blocks_verified; ctx->stats.n_classes_mismatch = cnt.blocks_mismatch; ctx->stats.n_classes_rejected = cnt.blocks_rejected; ctx->stats.n_classes_memo = cnt.blocks_memo;
        ctx->stats.exchange_calls = (uint32_t)ctx->x_calls; ctx->stats.exchange_bytes = ctx->x_bytes; ctx->stats.exchange_us = ctx->x_us;
        if (cnt.error) return fail(ctx, cnt.error, cnt.errmsg);
        ctx->stats.n_classes_device = cnt.blocks_device; ctx->stats.n_classes_host = cnt.blocks_host; ctx->stats.block_steps_max = cnt.block_steps_max; ctx->stats.n_classes = cnt.n_classes;
        ctx->stats.solve_classify_us = cnt.t_classify_us; ctx->stats.solve_blocks_us = cnt.t_blocks_us; ctx->stats.solve_decode_us = cnt.t_decode_us;
        mark();  // 2: solve
        const double t3 = now_us();
        out->is_optimal = cnt.is_optimal;
        out->is_canonical = cnt.is_canonical && cnt.is_optimal;
        status = HQTICK_DONE;  // scheduler/main.rs:57-68
        if (!cnt.is_optimal) status = cnt.empty() ? HQTICK_NO_PROGRESS : HQTICK_NEED_MORE_COMPUTE;
        // ---------------- host: mapping plan ----------------
        if ((rc = plan_keys())) return rc;
        if ((rc = launch_sweep_early())) return rc;
        if ((rc = plan_redirects())) return rc;
        mark();  // 3: key tables + retracts
        if ((rc = plan_prefill())) return rc;
        mark();  // 4: prefill plan
        if ((rc = plan_outputs())) return rc;
        mark();  // 5: K5 tables
        // ---------------- GPU phase C ----------------
        if ((rc = phase_c())) return rc;
        mark();  // 8: phase C synced
        finish(t1, t2, t3);
        return status;
    }
};

int run_tick(hqtick_ctx *ctx, const hqtick_snapshot *s, hqtick_result *out, bool use_resident) {
    hqtick_snapshot full;
    bool copied = false;
    if (s && s->n_workers == HQ_WORKERS_RESIDENT && !(s->worker_id == nullptr && ctx->cluster_valid && ctx->mirror.valid))
        return fail(ctx, HQTICK_E_INVALID, s->worker_id ? "n_workers == HQ_WORKERS_RESIDENT with worker arrays in the snapshot"
                                                         : "n_workers == HQ_WORKERS_RESIDENT without a resident worker set (hqtick_cluster_upload; dropped by hqtick_cluster_drop)");
    if (s && s->worker_id == nullptr && ctx->cluster_valid && ctx->mirror.valid) {  // the worker side lives in the library (hqtick_cluster_*, ABI 7)
        hqtick_ctx::ClusterMirror &m = ctx->mirror;
        const uint32_t W = (uint32_t)m.id.size();
        if (s->n_workers != 0 && s->n_workers != HQ_WORKERS_RESIDENT && s->n_workers != W) return fail(ctx, HQTICK_E_INVALID, "snapshot without worker arrays: n_workers must be HQ_WORKERS_RESIDENT, 0 or the resident worker count");
        if (s->n_resources != ctx->cl_R) return fail(ctx, HQTICK_E_INVALID, "snapshot without worker arrays: n_resources differs from the resident tables");
        if (m.blk_dirty) {
            m.blk_worker.clear(); m.blk_rq.clear(); m.blk_variant.clear();
            for (uint32_t w = 0; w < W; w++) { auto it = m.blocked.find(m.id[w]); if (it == m.blocked.end()) continue; for (auto &p : it->second) { m.blk_worker.push_back(w); m.blk_rq.push_back(p.first); m.blk_variant.push_back(p.second); } }
            m.blk_dirty = false;
        }
        full = *s; copied = true;
        full.n_workers = W; full.worker_id = m.id.data(); full.worker_total = m.total.data(); full.worker_free = m.free_.data(); full.worker_remaining_ns = m.rem.data();
        full.worker_min_utilization = m.min_util.data(); full.worker_flags = m.flags.data(); full.worker_group = m.group.data(); full.n_groups = m.n_groups; full.worker_map_rank = nullptr;
        full.n_blocked = (uint32_t)m.blk_worker.size(); full.blocked_worker = m.blk_worker.data(); full.blocked_rq = m.blk_rq.data(); full.blocked_variant = m.blk_variant.data();
        s = &full;
    }
    const bool retr_resident = s && s->n_retracting == HQ_RETRACTING_RESIDENT;
    if (retr_resident) {  // the Retracting tasks of the queues come from the library's table (hqtick_retracting_*, ABI 7)
        if (!s->worker_id) return fail(ctx, HQTICK_E_INVALID, "resident retracting table without worker ids (hqtick_cluster_upload, or worker arrays in the snapshot)");
        const uint32_t W = s->n_workers;
        auto index_of = [&](uint32_t id) -> uint32_t { const uint32_t *b = s->worker_id, *e = b + W, *it = std::lower_bound(b, e, id); return (it != e && *it == id) ? (uint32_t)(it - b) : HQ_NO_WORKER; };
        ctx->retr_task.clear(); ctx->retr_worker.clear(); ctx->retr_red_worker.clear(); ctx->retr_red_variant.clear();
        for (auto &kv : ctx->retr) {  // (std::map: ascending task id, as the snapshot wants it)
            if (!kv.second.in_queue) continue;
            const uint32_t oi = index_of(kv.second.old_id);
            if (oi == HQ_NO_WORKER) return fail(ctx, HQTICK_E_INVALID, "a Retracting task's worker is not in the worker set");
            ctx->retr_task.push_back(kv.first); ctx->retr_worker.push_back(oi);
            ctx->retr_red_worker.push_back(kv.second.has_redirect ? index_of(kv.second.target_id) : HQ_NO_WORKER); ctx->retr_red_variant.push_back(kv.second.variant);
        }
        if (!copied) { full = *s; copied = true; s = &full; }
        full.n_retracting = (uint32_t)ctx->retr_task.size();
        full.retracting_task = ctx->retr_task.data(); full.retracting_worker = ctx->retr_worker.data();
        full.retracting_redirect_worker = ctx->retr_red_worker.data(); full.retracting_redirect_variant = ctx->retr_red_variant.data();
    }
    int rc;
    {
        TickRun run(ctx, s, out, use_resident);
        rc = run.run();
    }
    if (ctx->consumed_unconfirmed) {  // HQTICK_FLAG_CONSUME_IN_TICK: the selection has written its tombstones
        ctx->consumed_unconfirmed = false;
        if (rc < 0) {
            // ... and the tick did not come back (a record sink too small, a mapping capacity exceeded, a failed exchange: errors a host can recover from).  What it took
            // goes back: this tick's K1 left a valid group key on every task that was live, so the tombstones K4 wrote are recognisable and their request ids
            // recoverable (k_restore_consumed: one pass over the key and request-id columns).  Only if THAT fails is the set dropped and the host uploads it again.
            const std::string why = ctx->err;
            bool restored = false;
            if (hipStreamSynchronize(ctx->stream) == hipSuccess && ctx->last_valid && ctx->last_Q && ctx->h_q.ensure(64)) {
                uint32_t *cntp = ctx->h_q.as<uint32_t>();
                cntp[0] = 0;
                if (hqk::ready_restore_consumed(ctx->d_gkey.as<uint16_t>(), ctx->d_trq.as<uint32_t>(), ctx->n_ready, ctx->last_Q, ctx->h_q.dev<uint32_t>(), ctx->stream) == hipSuccess &&
                    hipStreamSynchronize(ctx->stream) == hipSuccess && cntp[0] <= ctx->last_n_sel) {
                    ctx->n_live += ctx->last_n_sel;   // (what the tick had subtracted when it launched the selection; the kernel may have run for any part of it)
                    restored = true;
                }
            }
            ctx->last_valid = false; ctx->last_consumed = true; ctx->last_n_sel = 0;   // nothing of this tick is left to consume
            ctx->err = why;
            if (restored) ctx->err += " (HQTICK_FLAG_CONSUME_IN_TICK: what the tick had taken is back in the resident ready set)";
            else { ctx->resident = false; ctx->err += " (HQTICK_FLAG_CONSUME_IN_TICK: the tick had already taken its tasks and they could not be put back; the resident ready set is dropped, upload it again)"; }
        }
    }
    if (rc >= 0 && retr_resident) {  // what create_task_mapping did to task states and redirects (mapping.rs:66-101), applied to the table
        const uint32_t W = s->n_workers;
        for (uint32_t w = 0; w < W && w + 1 < ctx->retract_off.size(); w++)   // Prefilled{old} -> Retracting{old}: out of a prefill set, not in a queue
            for (uint32_t i = ctx->retract_off[w]; i < ctx->retract_off[w + 1]; i++) ctx->retr[ctx->retract_task[i]] = hqtick_ctx::RetrEntry{s->worker_id[w], false, false, 0, 0};
        for (size_t i = 0; i < ctx->red_task.size(); i++) {
            auto it = ctx->retr.find(ctx->red_task[i]);
            if (it == ctx->retr.end()) continue;
            hqtick_ctx::RetrEntry &e = it->second;
            const uint8_t kind = i < ctx->red_kind.size() ? ctx->red_kind[i] : (uint8_t)HQ_REDIRECT_FROM_PREFILL;
            e.in_queue = false;  // take_tasks removed it from its queue (or it came out of a prefill set)
            if (kind == HQ_REDIRECT_SAME_WORKER) continue;  // back on the worker it is retracting from: insert_sn_task(old) only, the redirect table is untouched (mapping.rs:66-80)
            if (ctx->red_worker[i] < W) { e.has_redirect = true; e.target_id = s->worker_id[ctx->red_worker[i]]; e.variant = ctx->red_variant[i]; }
        }
    }
    return rc;
}

}  // namespace

// ================================================================================================ C ABI
extern "C" {

uint32_t hqtick_abi_version(void) { return HQTICK_ABI_VERSION; }
const char *hqtick_build_arch(void) { return "gfx950"; }

int hqtick_create(const hqtick_config *config, hqtick_ctx **out_ctx) {
    if (!config || !out_ctx || config->abi_version != HQTICK_ABI_VERSION) return HQTICK_E_INVALID;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || config->device_index < 0 || config->device_index >= n) return HQTICK_E_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, config->device_index) != hipSuccess) return HQTICK_E_NO_DEVICE;
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) return HQTICK_E_NO_DEVICE;  // kernels are built for gfx950 only
    if (hipSetDevice(config->device_index) != hipSuccess) return HQTICK_E_NO_DEVICE;
    hqtick_ctx *ctx = new hqtick_ctx();
    ctx->cfg = *config; ctx->device = config->device_index;
    ctx->timing = (config->flags & HQTICK_FLAG_NO_KERNEL_TIMING) == 0;
    if (const char *e = getenv("HQTICK_TPW")) { long v = atol(e); if (v >= 256 && v <= (1 << 20) && v % 256 == 0) ctx->tpw_hint = (uint32_t)v; }
    if (const char *e = getenv("HQTICK_BLOCK_PROFILE")) ctx->block_profile = atoi(e) != 0;
    if (const char *e = getenv("HQTICK_BLOCK_BUDGET")) { long v = atol(e); if (v >= 1 && v <= (1 << 24)) ctx->block_budget = (uint32_t)v; }
    if (const char *e = getenv("HQTICK_BLOCK_MIN_CLASSES")) { long v = atol(e); if (v >= 0) ctx->block_min_classes = (uint32_t)v; }
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return HQTICK_E_DEVICE; }
    if (hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking) != hipSuccess) { hipStreamDestroy(ctx->stream); delete ctx; return HQTICK_E_DEVICE; }
    if (const char *e = getenv("HQTICK_K2_RIDE_ALONG")) { ctx->k2_own_stream = atoi(e) == 0; ctx->k2_on_hist = atoi(e) == 1; }
    if (const char *e = getenv("HQTICK_CHECK_CLUSTER")) ctx->cluster_check = atoi(e) != 0;
    if (const char *e = getenv("HQTICK_WAIT_ON_KERNEL")) ctx->wait_on_kernel = atoi(e) != 0;
    {
        bool price = true; uint32_t min_cols = 0;
        if (const char *e = getenv("HQTICK_PRICE")) price = atoi(e) != 0;
        if (const char *e = getenv("HQTICK_PRICE_MIN_COLS")) { long v = atol(e); if (v > 0) min_cols = (uint32_t)v; }
        if (price) { ctx->pricer = new hqprice::DeviceSweeper(ctx->stream); ctx->pricer->budget = ctx->block_budget; if (min_cols) ctx->pricer->min_cols = min_cols; }
    }
    if (const char *e = getenv("HQTICK_BLOCK_VERIFY")) ctx->block_verify = strcmp(e, "all") == 0 ? 0xFFFFFFFFu : (uint32_t)std::max(0L, atol(e));
    if (const char *e = getenv("HQTICK_SHARD_SOLVE")) ctx->shard_solve = atoi(e) != 0;
    if (const char *e = getenv("HQTICK_SHARD_MIN_BLOCKS")) { long v = atol(e); if (v >= 0) ctx->shard_min_blocks = (uint32_t)v; }
    if (const char *e = getenv("HQTICK_SHARD_MIN_CLASSES")) { long v = atol(e); if (v >= 0) ctx->shard_min_classes = (uint32_t)v; }
    if (hipEventCreate(&ctx->cl_ev) != hipSuccess) { delete ctx; return HQTICK_E_DEVICE; }
    for (auto &e : ctx->ev) if (hipEventCreate(&e) != hipSuccess) { delete ctx; return HQTICK_E_DEVICE; }
    if (!ctx->d_flags.ensure(64) || hipMemset(ctx->d_flags.p, 0, 64) != hipSuccess) { delete ctx; return HQTICK_E_DEVICE; }
    *out_ctx = ctx;
    return 0;
}

static void rccl_destroy_comm(hqtick_ctx *ctx);
void hqtick_destroy(hqtick_ctx *ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    if (ctx->stream2) hipStreamSynchronize(ctx->stream2);
    delete ctx->pricer; ctx->pricer = nullptr;
    DevBuf *bufs[] = {&ctx->d_tid, &ctx->d_tprio, &ctx->d_trq, &ctx->d_set, &ctx->d_flags, &ctx->d_levels, &ctx->d_nlevels, &ctx->d_wave_tab, &ctx->d_hist,
                      &ctx->d_up, &ctx->d_vflags, &ctx->d_vtmc, &ctx->d_sel_task, &ctx->d_gkey,
                      &ctx->d_sel_level, &ctx->d_map, &ctx->d_rec, &ctx->d_tsweep, &ctx->d_bits, &ctx->d_pre, &ctx->d_tid2, &ctx->d_tprio2, &ctx->d_trq2, &ctx->d_slice, &ctx->d_add, &ctx->d_pre8, &ctx->d_blk, &ctx->d_cluster};
    for (DevBuf *b : bufs) b->release();
    ctx->h_cl.release(); ctx->h_cld.release(); ctx->d_cluster2.release();
    if (ctx->cl_ev) hipEventDestroy(ctx->cl_ev);
    if (ctx->qctx) { hqtick_destroy(ctx->qctx); ctx->qctx = nullptr; }
    hipSetDevice(ctx->device);
    if (ctx->comm) { rccl_destroy_comm(ctx); }
    ctx->graph.release();
    ctx->h_up.release(); ctx->h_up2.release(); ctx->h_q.release(); ctx->h_a.release(); ctx->h_plan.release(); ctx->h_rec.release(); ctx->h_sinkhdr.release(); ctx->h_add.release(); ctx->h_addp.release(); ctx->h_retr.release(); ctx->h_blk.release(); ctx->h_blkprof.release(); ctx->h_k5a.release(); ctx->h_lv.release();
    for (auto &e : ctx->ev) if (e) hipEventDestroy(e);
    if (ctx->stream) hipStreamDestroy(ctx->stream);
    if (ctx->stream2) hipStreamDestroy(ctx->stream2);
    delete ctx;
}

const char *hqtick_last_error(const hqtick_ctx *ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

int hqtick_run(hqtick_ctx *ctx, const hqtick_snapshot *snapshot, hqtick_result *out) {
    if (!ctx || !out) return HQTICK_E_INVALID;
    return run_tick(ctx, snapshot, out, false);
}

int hqtick_upload_ready(hqtick_ctx *ctx, uint64_t n, const uint64_t *task_id, const uint64_t *task_priority, const uint32_t *task_rq, int sorted) {
    if (!ctx) return HQTICK_E_INVALID;
    if (n && (!task_id || !task_priority || !task_rq)) return fail(ctx, HQTICK_E_INVALID, "null ready-set column");
    if (sorted) for (uint64_t i = 1; i < n; i++) if (task_id[i - 1] >= task_id[i]) return fail(ctx, HQTICK_E_INVALID, "ready set not sorted by task id");
    for (uint64_t i = 0; i < n; i++) if (task_rq[i] == 0xFFFFFFFFu) return fail(ctx, HQTICK_E_INVALID, "request id 0xFFFFFFFF is reserved");
    HQ_HIP(hipSetDevice(ctx->device));
    ctx->resident = false;  // whatever was resident is being overwritten: only a complete upload makes the set usable again
    uint64_t n_pow2 = 1; while (n_pow2 < n) n_pow2 <<= 1;
    const uint64_t cap = sorted ? n : n_pow2;
    if (!ctx->d_tid.ensure(cap * 8 + 8) || !ctx->d_tprio.ensure(cap * 8 + 8) || !ctx->d_trq.ensure(cap * 4 + 8)) return fail(ctx, HQTICK_E_DEVICE, "hipMalloc ready set");
    if (n) {
        // blocking copies: the caller's columns are pageable memory, and this is the one call whose result every later tick trusts (a level table built from
        // a column that was still arriving would list whatever the buffer held before; seen once under rocprofv3 with the asynchronous form)
        HQ_HIP(hipMemcpy(ctx->d_tid.p, task_id, n * 8, hipMemcpyHostToDevice));
        HQ_HIP(hipMemcpy(ctx->d_tprio.p, task_priority, n * 8, hipMemcpyHostToDevice));
        HQ_HIP(hipMemcpy(ctx->d_trq.p, task_rq, n * 4, hipMemcpyHostToDevice));
        if (!sorted) {  // the host handed over its queues in whatever order it walked them: sort by id on the device, once
            if (!ctx->h_q.ensure(64)) return fail(ctx, HQTICK_E_DEVICE, "hipHostMalloc");
            uint32_t *flag = ctx->h_q.as<uint32_t>(); flag[0] = 0;
            HQ_HIP(hqk::sort_ready(ctx->d_tid.as<uint64_t>(), ctx->d_tprio.as<uint64_t>(), ctx->d_trq.as<uint32_t>(), n, n_pow2, ctx->h_q.dev<uint32_t>(), ctx->stream));
            HQ_HIP(hipStreamSynchronize(ctx->stream));
            if (flag[0]) return fail(ctx, HQTICK_E_INVALID, "ready set holds a task id twice (or the id 2^64 - 1)");
        } else {
            HQ_HIP(hipStreamSynchronize(ctx->stream));
        }
    }
    ctx->n_ready = n; ctx->n_live = n; ctx->resident = true; ctx->levels_valid = false; ctx->last_valid = false; ctx->last_consumed = true;
    ctx->max_id = 0; ctx->max_id_valid = true;
    if (sorted) { if (n) ctx->max_id = task_id[n - 1]; } else for (uint64_t i = 0; i < n; i++) ctx->max_id = std::max(ctx->max_id, task_id[i]);
    return 0;
}

int hqtick_run_resident(hqtick_ctx *ctx, const hqtick_snapshot *snapshot, hqtick_result *out) {
    if (!ctx || !out) return HQTICK_E_INVALID;
    return run_tick(ctx, snapshot, out, true);
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------- resident ready-set deltas (f1)
namespace {
// Drops tombstones and merges `n_add` new tasks (device arrays, ids ascending) into fresh columns; swaps them in.
bool append_enabled() { static const bool on = !(getenv("HQTICK_APPEND") && atoi(getenv("HQTICK_APPEND")) == 0); return on; }
// `first_id` / `last_id`: the batch's smallest and largest id when the caller knows them on the host (0 / 0: it does not — the device graph's releases).
int rebuild_ready(hqtick_ctx *ctx, const uint64_t *aid, const uint64_t *aprio, const uint32_t *arq, uint32_t n_add, uint64_t first_id = 0, uint64_t last_id = 0) {
    const uint64_t N = ctx->n_ready, new_n = ctx->n_live + n_add;
    if (append_enabled() && N > 0 && n_add > 0 && last_id >= first_id && first_id > 0 && ctx->max_id_valid && first_id > ctx->max_id &&
        ctx->d_tid.cap >= (N + n_add) * 8 + 8 && ctx->d_tprio.cap >= (N + n_add) * 8 + 8 && ctx->d_trq.cap >= (N + n_add) * 4 + 8) {
        // Fresh ids behind everything resident, and room at the tail: ONE kernel instead of four (the columns are not re-written; what the ticks consumed stays as
        // tombstones until hqtick_ready_consume_last / _compact find more tombstones than tasks).  The steady loop's add: 136 -> ~50 us.
        if (!ctx->h_q.ensure(64)) return fail(ctx, HQTICK_E_DEVICE, "hipHostMalloc");
        uint32_t *flag = ctx->h_q.as<uint32_t>(); flag[0] = 0;
        bool waits_on_kernel = false;
        hqk::time_next_launch(nullptr, ctx->ev[11]);
        HQ_HIP_LAST(hqk::ready_append(aid, aprio, arq, n_add, ctx->max_id, ctx->d_tid.as<uint64_t>() + N, ctx->d_tprio.as<uint64_t>() + N, ctx->d_trq.as<uint32_t>() + N, ctx->h_q.dev<uint32_t>(), ctx->stream), waits_on_kernel);
        if (waits_on_kernel) HQ_HIP(hipEventSynchronize(ctx->ev[11])); else HQ_HIP(hipStreamSynchronize(ctx->stream));  // (the dispatch's own completion signal: HQ_HIP_LAST)
        if (flag[0] & 4u) return fail(ctx, HQTICK_E_INVALID, "hqtick_ready_add: a task id is already in the ready set");
        if (flag[0] & 8u) return fail(ctx, HQTICK_E_INVALID, "hqtick_ready_add: ids not strictly ascending");
        if (flag[0] & 16u) return fail(ctx, HQTICK_E_INVALID, "hqtick_ready_add: request id 0xFFFFFFFF is reserved");
        ctx->n_ready = N + n_add; ctx->n_live += n_add; ctx->last_valid = false; ctx->last_consumed = true; ctx->max_id = last_id; ctx->n_appends++;
        if (ctx->n_ready > 4096 && ctx->n_live * 2 < ctx->n_ready) return rebuild_ready(ctx, nullptr, nullptr, nullptr, 0);  // (see hqtick_ready_add_packed)
        return 0;
    }
    // (a rebuild leaves room behind the columns: the next fresh batches are appended)
    const uint64_t room = new_n + std::max<uint64_t>(new_n, 4096);
    if (!ctx->d_tid2.ensure(room * 8 + 8) || !ctx->d_tprio2.ensure(room * 8 + 8) || !ctx->d_trq2.ensure(room * 4 + 8)) return fail(ctx, HQTICK_E_DEVICE, "hipMalloc ready set (rebuild)");
    uint64_t first_batch_last_id = 0;
    if (N == 0) {  // nothing resident: the batch becomes the ready set — through the same validation as an appended one (ascending ids, no reserved request id:
                   // a refused batch leaves the set as it was, here: empty), written by the kernel that checks it
        if (n_add) {
            if (!ctx->h_q.ensure(64)) return fail(ctx, HQTICK_E_DEVICE, "hipHostMalloc");
            uint32_t *flag = ctx->h_q.as<uint32_t>(); flag[0] = 0;
            HQ_HIP(hqk::ready_append(aid, aprio, arq, n_add, 0xFFFFFFFFFFFFFFFFull, ctx->d_tid2.as<uint64_t>(), ctx->d_tprio2.as<uint64_t>(), ctx->d_trq2.as<uint32_t>(), ctx->h_q.dev<uint32_t>(), ctx->stream));
            uint64_t *last_back = reinterpret_cast<uint64_t *>(ctx->h_q.as<unsigned char>() + 16);
            HQ_HIP(hipMemcpyAsync(last_back, ctx->d_tid2.as<uint64_t>() + (n_add - 1), 8, hipMemcpyDeviceToHost, ctx->stream));
            // (the batch came through the pinned staging buffer: the caller may stage the NEXT batch into it as soon as this call returns, so the kernel must be over by then)
            HQ_HIP(hipStreamSynchronize(ctx->stream));
            if (flag[0] & 8u) return fail(ctx, HQTICK_E_INVALID, "hqtick_ready_add: ids not strictly ascending");
            if (flag[0] & 16u) return fail(ctx, HQTICK_E_INVALID, "hqtick_ready_add: request id 0xFFFFFFFF is reserved");
            first_batch_last_id = *last_back;
        }
    } else {
        const uint32_t n_slices = (uint32_t)((N + 255) / 256), stride = (n_slices + 15u) & ~15u;
        if (!ctx->d_slice.ensure((size_t)stride * 4 + 64) || !ctx->h_q.ensure(64) || !ctx->d_pre8.ensure(N + 16)) return fail(ctx, HQTICK_E_DEVICE, "hipMalloc slice table");
        uint32_t *flag = ctx->h_q.as<uint32_t>(); flag[0] = 0; flag[1] = 0;
        hqk::WaveGeom g{256, n_slices, 4, stride};
        HQ_HIP(hqk::ready_live_count(ctx->d_trq.as<uint32_t>(), N, ctx->d_slice.as<uint32_t>(), ctx->stream));
        HQ_HIP(hqk::scan_waves(ctx->d_slice.as<uint32_t>(), g, 1, ctx->h_q.dev<uint32_t>() + 1, nullptr, nullptr, ctx->stream));
        HQ_HIP(hqk::ready_rebuild(ctx->d_tid.as<uint64_t>(), ctx->d_tprio.as<uint64_t>(), ctx->d_trq.as<uint32_t>(), N, (uint32_t)ctx->n_live, ctx->d_slice.as<uint32_t>(), aid, aprio, arq, n_add,
                                  ctx->d_tid2.as<uint64_t>(), ctx->d_tprio2.as<uint64_t>(), ctx->d_trq2.as<uint32_t>(), ctx->d_pre8.as<uint8_t>(), ctx->h_q.dev<uint32_t>(), ctx->stream));
        uint64_t *last_back = reinterpret_cast<uint64_t *>(ctx->h_q.as<unsigned char>() + 16);  // the new last id comes back with the flags: the bound appends are decided by
        if (new_n) HQ_HIP(hipMemcpyAsync(last_back, ctx->d_tid2.as<uint64_t>() + (new_n - 1), 8, hipMemcpyDeviceToHost, ctx->stream));
        HQ_HIP(hipStreamSynchronize(ctx->stream));
        if (flag[1] != ctx->n_live) return fail(ctx, HQTICK_E_DEVICE, "resident ready set: live-task count out of sync");
        if (flag[0] & 4u) return fail(ctx, HQTICK_E_INVALID, "hqtick_ready_add: a task id is already in the ready set");
        if (flag[0] & 8u) return fail(ctx, HQTICK_E_INVALID, "hqtick_ready_add: ids not strictly ascending");
        if (flag[0] & 16u) return fail(ctx, HQTICK_E_INVALID, "hqtick_ready_add: request id 0xFFFFFFFF is reserved");
        if (new_n) { ctx->max_id = *last_back; ctx->max_id_valid = true; }  // (only of columns that are swapped in: a refused batch leaves the bound as it was)
    }
    std::swap(ctx->d_tid, ctx->d_tid2); std::swap(ctx->d_tprio, ctx->d_tprio2); std::swap(ctx->d_trq, ctx->d_trq2);
    ctx->n_ready = new_n; ctx->n_live = new_n; ctx->last_valid = false; ctx->last_consumed = true;
    if (N == 0 && n_add) { ctx->max_id = first_batch_last_id; ctx->max_id_valid = true; }  // (the validated batch's own last id, read back from the device: the bound later appends are decided by)
    return 0;
}
}  // namespace

extern "C" {

uint64_t hqtick_ready_count(const hqtick_ctx *ctx) { return ctx && ctx->resident ? ctx->n_live : 0; }

int hqtick_ready_consume_last(hqtick_ctx *ctx) {
    if (!ctx) return HQTICK_E_INVALID;
    if (!ctx->resident) return fail(ctx, HQTICK_E_INVALID, "no resident ready set");
    if (ctx->last_consumed) {  // nothing handed out since the last consume (or the tick consumed it itself: HQTICK_FLAG_CONSUME_IN_TICK)
        if (ctx->n_ready > 4096 && ctx->n_live * 2 < ctx->n_ready) { HQ_HIP(hipSetDevice(ctx->device)); return rebuild_ready(ctx, nullptr, nullptr, nullptr, 0); }
        return 0;
    }
    if (!ctx->last_valid) return fail(ctx, HQTICK_E_INVALID, "hqtick_ready_consume_last needs a preceding hqtick_run_resident");
    HQ_HIP(hipSetDevice(ctx->device));
    const uint32_t *d = ctx->d_map.as<uint32_t>();
    // the selection of the last tick once more, writing tombstones instead of the selected ids (same offsets table, same plan)
    HQ_HIP(hqk::select_scatter(ctx->d_tid.as<uint64_t>(), ctx->d_gkey.as<uint16_t>(), ctx->n_ready, ctx->last_Q, ctx->last_G, ctx->last_geom, ctx->d_wave_tab.as<uint32_t>(),
                               ctx->h_plan.as<uint32_t>() + ctx->last_tb, d + ctx->last_tb, ctx->d_sel_task.as<uint64_t>(), ctx->d_sel_level.as<uint16_t>(), nullptr, nullptr, 0,
                               ctx->d_trq.as<uint32_t>(), ctx->stream));
    ctx->n_live -= ctx->last_n_sel; ctx->last_consumed = true; ctx->last_valid = false;
    if (ctx->n_ready > 4096 && ctx->n_live * 2 < ctx->n_ready) return rebuild_ready(ctx, nullptr, nullptr, nullptr, 0);  // more tombstones than tasks: compact
    return 0;
}

int hqtick_ready_remove(hqtick_ctx *ctx, uint64_t n, const uint64_t *task_id) {
    if (!ctx || (n && !task_id)) return HQTICK_E_INVALID;
    if (!ctx->resident) return fail(ctx, HQTICK_E_INVALID, "no resident ready set");
    if (n == 0 || ctx->n_ready == 0) return 0;
    if (n > 0x7FFFFFFFull) return fail(ctx, HQTICK_E_CAPACITY, "more than 2^31 - 1 ids in one delta (the count of removed tasks is the int return value)");
    HQ_HIP(hipSetDevice(ctx->device));
    if (!ctx->d_add.ensure(n * 8 + 8) || !ctx->h_q.ensure(64)) return fail(ctx, HQTICK_E_DEVICE, "hipMalloc delta staging");
    uint32_t *cnt = ctx->h_q.as<uint32_t>(); cnt[0] = 0;
    if (!ctx->h_add.ensure(n * 8 + 8)) return fail(ctx, HQTICK_E_DEVICE, "hipHostMalloc delta staging");
    memcpy(ctx->h_add.p, task_id, n * 8);
    HQ_HIP(hipMemcpyAsync(ctx->d_add.p, ctx->h_add.p, n * 8, hipMemcpyHostToDevice, ctx->stream));
    HQ_HIP(hqk::ready_mark_removed(ctx->d_tid.as<uint64_t>(), ctx->d_trq.as<uint32_t>(), ctx->n_ready, ctx->d_add.as<uint64_t>(), (uint32_t)n, ctx->h_q.dev<uint32_t>(), ctx->stream));
    HQ_HIP(hipStreamSynchronize(ctx->stream));
    ctx->n_live -= cnt[0]; ctx->last_valid = false; ctx->last_consumed = true;
    return (int)cnt[0];  // number of tasks that were in the set
}

// The batch of new ready tasks is staged in pinned memory (a pageable hipMemcpy of a few MB costs ~1 ms in page pinning).  The caller can write it there
// directly: hqtick_ready_add_stage hands out the three column pointers for n tasks, hqtick_ready_add_staged merges what was written (no copy on the host);
// hqtick_ready_add = stage + memcpy + merge for callers that have the columns elsewhere.
namespace {
struct AddLayout { size_t o_p, o_q, bytes; };
AddLayout add_layout(uint64_t n) { AddLayout L; L.o_p = (n * 8 + 15) & ~(size_t)15; L.o_q = L.o_p * 2; L.bytes = L.o_q + n * 4 + 16; return L; }
}  // namespace

int hqtick_ready_add_stage(hqtick_ctx *ctx, uint64_t n, uint64_t **task_id, uint64_t **task_priority, uint32_t **task_rq) {
    if (!ctx || !task_id || !task_priority || !task_rq) return HQTICK_E_INVALID;
    if (!ctx->resident) return fail(ctx, HQTICK_E_INVALID, "no resident ready set (hqtick_upload_ready with n = 0 creates an empty one)");
    if (n > 0xFFFFFFFFull) return fail(ctx, HQTICK_E_CAPACITY, "more than 2^32 ids in one delta");
    HQ_HIP(hipSetDevice(ctx->device));
    const AddLayout L = add_layout(n);
    if (!ctx->h_add.ensure(L.bytes)) return fail(ctx, HQTICK_E_DEVICE, "hipHostMalloc delta staging");
    unsigned char *h = ctx->h_add.as<unsigned char>();
    *task_id = reinterpret_cast<uint64_t *>(h); *task_priority = reinterpret_cast<uint64_t *>(h + L.o_p); *task_rq = reinterpret_cast<uint32_t *>(h + L.o_q);
    ctx->add_staged_n = n;
    return 0;
}

int hqtick_ready_add_staged(hqtick_ctx *ctx, uint64_t n) {
    if (!ctx) return HQTICK_E_INVALID;
    if (!ctx->resident) return fail(ctx, HQTICK_E_INVALID, "no resident ready set (hqtick_upload_ready with n = 0 creates an empty one)");
    if (n > ctx->add_staged_n) return fail(ctx, HQTICK_E_INVALID, "hqtick_ready_add_staged: more tasks than hqtick_ready_add_stage made room for");
    const AddLayout L = add_layout(ctx->add_staged_n);  // the layout the pointers were handed out for
    ctx->add_staged_n = 0;
    if (n == 0) return 0;
    const unsigned char *h = ctx->h_add.as<unsigned char>();
    const uint64_t *task_id = reinterpret_cast<const uint64_t *>(h); const uint32_t *task_rq = reinterpret_cast<const uint32_t *>(h + L.o_q);
    if (ctx->n_ready == 0) {  // nothing resident: the batch is copied as it is, so it is checked here; otherwise the merge kernel validates it
        for (uint64_t i = 1; i < n; i++) if (task_id[i - 1] >= task_id[i]) return fail(ctx, HQTICK_E_INVALID, "hqtick_ready_add: ids not strictly ascending");
        for (uint64_t i = 0; i < n; i++) if (task_rq[i] == 0xFFFFFFFFu) return fail(ctx, HQTICK_E_INVALID, "hqtick_ready_add: request id 0xFFFFFFFF is reserved");
    }
    HQ_HIP(hipSetDevice(ctx->device));
    if (!ctx->d_add.ensure(L.bytes)) return fail(ctx, HQTICK_E_DEVICE, "hipMalloc delta staging");
    unsigned char *d = ctx->d_add.as<unsigned char>();
    HQ_HIP(hipMemcpyAsync(d, h, L.o_q + n * 4, hipMemcpyHostToDevice, ctx->stream));
    return rebuild_ready(ctx, reinterpret_cast<const uint64_t *>(d), reinterpret_cast<const uint64_t *>(d + L.o_p), reinterpret_cast<const uint32_t *>(d + L.o_q), (uint32_t)n, task_id[0], task_id[n - 1]);
}

int hqtick_ready_add_packed(hqtick_ctx *ctx, uint64_t n, uint32_t n_id_runs, const uint64_t *id_run_start, const uint32_t *id_run_len, const uint32_t *id_off,
                            uint32_t n_prio_runs, const uint64_t *prio_run_value, const uint32_t *prio_run_len, const uint16_t *task_rq) {
    if (!ctx) return HQTICK_E_INVALID;
    if (!ctx->resident) return fail(ctx, HQTICK_E_INVALID, "no resident ready set (hqtick_upload_ready with n = 0 creates an empty one)");
    if (n == 0) return 0;
    if (n > 0xFFFFFFFFull) return fail(ctx, HQTICK_E_CAPACITY, "more than 2^32 ids in one delta");
    if (!n_id_runs || !n_prio_runs || !id_run_start || !id_run_len || !prio_run_value || !prio_run_len || !task_rq) return fail(ctx, HQTICK_E_INVALID, "hqtick_ready_add_packed: null array");
    // the small run tables are checked here (a handful of entries); ids ascending across the whole batch, duplicates against the resident set and the reserved
    // request id are checked by the merge kernel, as for hqtick_ready_add
    uint64_t tot = 0; for (uint32_t r = 0; r < n_id_runs; r++) { if (!id_run_len[r]) return fail(ctx, HQTICK_E_INVALID, "hqtick_ready_add_packed: empty id run"); tot += id_run_len[r]; }
    if (tot != n) return fail(ctx, HQTICK_E_INVALID, "hqtick_ready_add_packed: the id runs do not add up to n");
    tot = 0; for (uint32_t r = 0; r < n_prio_runs; r++) { if (!prio_run_len[r]) return fail(ctx, HQTICK_E_INVALID, "hqtick_ready_add_packed: empty priority run"); tot += prio_run_len[r]; }
    if (tot != n) return fail(ctx, HQTICK_E_INVALID, "hqtick_ready_add_packed: the priority runs do not add up to n");
    HQ_HIP(hipSetDevice(ctx->device));
    auto al = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const size_t o_is = 0, o_if = al(o_is + (size_t)n_id_runs * 8), o_pv = al(o_if + (size_t)(n_id_runs + 1) * 4), o_pf = al(o_pv + (size_t)n_prio_runs * 8), o_off = al(o_pf + (size_t)(n_prio_runs + 1) * 4),
                 o_rq = al(o_off + (id_off ? (size_t)n * 4 : 0)), bytes = al(o_rq + (size_t)n * 2);
    static const bool trace_add = getenv("HQTICK_TRACE_ADD") != nullptr;  // (experiments: where the microseconds of an add go, one line per call on stderr)
    const double ta0 = trace_add ? now_us() : 0.0;
    if (!ctx->h_addp.ensure(bytes)) return fail(ctx, HQTICK_E_DEVICE, "hipHostMalloc packed delta staging");
    const AddLayout L = add_layout(n);
    if (!ctx->d_add.ensure(L.bytes)) return fail(ctx, HQTICK_E_DEVICE, "hipMalloc delta staging");
    unsigned char *h = ctx->h_addp.as<unsigned char>(), *hd = ctx->h_addp.dev<unsigned char>(), *d = ctx->d_add.as<unsigned char>();
    memcpy(h + o_is, id_run_start, (size_t)n_id_runs * 8); memcpy(h + o_pv, prio_run_value, (size_t)n_prio_runs * 8);
    uint32_t *idf = reinterpret_cast<uint32_t *>(h + o_if), *pf = reinterpret_cast<uint32_t *>(h + o_pf);
    idf[0] = 0; for (uint32_t r = 0; r < n_id_runs; r++) idf[r + 1] = idf[r] + id_run_len[r];
    pf[0] = 0; for (uint32_t r = 0; r < n_prio_runs; r++) pf[r + 1] = pf[r] + prio_run_len[r];
    if (id_off) memcpy(h + o_off, id_off, (size_t)n * 4);
    memcpy(h + o_rq, task_rq, (size_t)n * 2);
    // the batch's id range, from the run tables: what decides between appending and merging
    const uint64_t first_id = id_run_start[0] + (id_off ? id_off[0] : 0u);
    const uint64_t last_id = id_run_start[n_id_runs - 1] + (id_off ? id_off[n - 1] : id_run_len[n_id_runs - 1] - 1u);
    const uint64_t N = ctx->n_ready;
    if (append_enabled() && N > 0 && ctx->max_id_valid && first_id > ctx->max_id && last_id >= first_id &&
        ctx->d_tid.cap >= (N + n) * 8 + 8 && ctx->d_tprio.cap >= (N + n) * 8 + 8 && ctx->d_trq.cap >= (N + n) * 4 + 8) {
        // fresh ids, room at the tail: the expansion kernel writes the batch where it belongs and validates it — the whole add is this one launch
        const double ta1 = trace_add ? now_us() : 0.0;
        if (!ctx->h_q.ensure(64)) return fail(ctx, HQTICK_E_DEVICE, "hipHostMalloc");
        uint32_t *flag = ctx->h_q.as<uint32_t>(); flag[0] = 0;
        bool waits_on_kernel = false;
        hqk::time_next_launch(nullptr, ctx->ev[11]);
        HQ_HIP_LAST(hqk::ready_unpack_adds((uint32_t)n, n_id_runs, reinterpret_cast<const uint64_t *>(hd + o_is), reinterpret_cast<const uint32_t *>(hd + o_if), id_off ? reinterpret_cast<const uint32_t *>(hd + o_off) : nullptr,
                                      n_prio_runs, reinterpret_cast<const uint64_t *>(hd + o_pv), reinterpret_cast<const uint32_t *>(hd + o_pf), reinterpret_cast<const uint16_t *>(hd + o_rq),
                                      ctx->d_tid.as<uint64_t>() + N, ctx->d_tprio.as<uint64_t>() + N, ctx->d_trq.as<uint32_t>() + N, ctx->max_id, ctx->h_q.dev<uint32_t>(), ctx->stream), waits_on_kernel);
        const double ta2 = trace_add ? now_us() : 0.0;
        if (waits_on_kernel) HQ_HIP(hipEventSynchronize(ctx->ev[11])); else HQ_HIP(hipStreamSynchronize(ctx->stream));  // (the dispatch's own completion signal: HQ_HIP_LAST)
        if (trace_add) fprintf(stderr, "hqtick add (append, packed): n %llu prepare %.1f us, launch %.1f us, wait %.1f us\n", (unsigned long long)n, ta1 - ta0, ta2 - ta1, now_us() - ta2);
        if (flag[0] & 4u) return fail(ctx, HQTICK_E_INVALID, "hqtick_ready_add: a task id is already in the ready set");
        if (flag[0] & 8u) return fail(ctx, HQTICK_E_INVALID, "hqtick_ready_add: ids not strictly ascending");
        if (flag[0] & 16u) return fail(ctx, HQTICK_E_INVALID, "hqtick_ready_add: request id 0xFFFFFFFF is reserved");
        ctx->n_ready = N + n; ctx->n_live += n; ctx->last_valid = false; ctx->last_consumed = true; ctx->max_id = last_id; ctx->n_appends++;
        if (ctx->n_ready > 4096 && ctx->n_live * 2 < ctx->n_ready) return rebuild_ready(ctx, nullptr, nullptr, nullptr, 0);  // (a host that never calls consume_last — HQTICK_FLAG_CONSUME_IN_TICK — compacts here)
        return 0;
    }
    // the expansion kernel reads the packed batch in place (pinned, device-mapped): what crosses PCIe is the packed form
    HQ_HIP(hqk::ready_unpack_adds((uint32_t)n, n_id_runs, reinterpret_cast<const uint64_t *>(hd + o_is), reinterpret_cast<const uint32_t *>(hd + o_if), id_off ? reinterpret_cast<const uint32_t *>(hd + o_off) : nullptr,
                                  n_prio_runs, reinterpret_cast<const uint64_t *>(hd + o_pv), reinterpret_cast<const uint32_t *>(hd + o_pf), reinterpret_cast<const uint16_t *>(hd + o_rq),
                                  reinterpret_cast<uint64_t *>(d), reinterpret_cast<uint64_t *>(d + L.o_p), reinterpret_cast<uint32_t *>(d + L.o_q), 0, nullptr, ctx->stream));
    return rebuild_ready(ctx, reinterpret_cast<const uint64_t *>(d), reinterpret_cast<const uint64_t *>(d + L.o_p), reinterpret_cast<const uint32_t *>(d + L.o_q), (uint32_t)n, first_id, last_id);
}

int hqtick_ready_add(hqtick_ctx *ctx, uint64_t n, const uint64_t *task_id, const uint64_t *task_priority, const uint32_t *task_rq) {
    if (!ctx || (n && (!task_id || !task_priority || !task_rq))) return HQTICK_E_INVALID;
    if (!ctx->resident) return fail(ctx, HQTICK_E_INVALID, "no resident ready set (hqtick_upload_ready with n = 0 creates an empty one)");
    if (n == 0) return 0;
    uint64_t *hi = nullptr, *hp = nullptr; uint32_t *hq = nullptr;
    if (int rc = hqtick_ready_add_stage(ctx, n, &hi, &hp, &hq)) return rc;
    memcpy(hi, task_id, n * 8); memcpy(hp, task_priority, n * 8); memcpy(hq, task_rq, n * 4);
    return hqtick_ready_add_staged(ctx, n);
}

// ---------------------------------------------------------------------------------------------- dependency graph (f1)
namespace {
int graph_fail(hqtick_ctx *ctx, int rc) { ctx->err = ctx->graph.err; return rc; }
int graph_pre(hqtick_ctx *ctx) {
    if (!ctx->resident) return fail(ctx, HQTICK_E_INVALID, "no resident ready set (hqtick_upload_ready with n = 0 creates an empty one)");
    if (hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, HQTICK_E_DEVICE, "hipSetDevice");
    return 0;
}
}  // namespace

int hqtick_graph_add_tasks(hqtick_ctx *ctx, uint64_t n, const uint64_t *task_id, const uint64_t *task_priority, const uint32_t *task_rq, const uint32_t *dep_off,
                           const uint64_t *dep_task_id) {
    if (!ctx || (n && (!task_id || !task_priority || !task_rq))) return HQTICK_E_INVALID;
    if (n && dep_off && dep_off[n] && !dep_task_id) return HQTICK_E_INVALID;
    if (int rc = graph_pre(ctx)) return rc;
    int r = ctx->graph.add(n, task_id, task_priority, task_rq, dep_off, dep_task_id, ctx->stream);
    if (r < 0) return graph_fail(ctx, r);
    if (r > 0) { if (int rc = rebuild_ready(ctx, ctx->graph.out_id(), ctx->graph.out_prio(), ctx->graph.out_rq(), (uint32_t)r)) return rc; }
    return r;
}

int hqtick_graph_finish(hqtick_ctx *ctx, uint64_t n, const uint64_t *task_id) {
    if (!ctx || (n && !task_id)) return HQTICK_E_INVALID;
    if (int rc = graph_pre(ctx)) return rc;
    int r = ctx->graph.finish(n, task_id, ctx->stream);
    // released consumers join the ready set even when the call reports ERR_NOT_READY: the graph itself is consistent
    const uint32_t rel = ctx->graph.n_out();
    if (rel) { if (int rc = rebuild_ready(ctx, ctx->graph.out_id(), ctx->graph.out_prio(), ctx->graph.out_rq(), rel)) return rc; }
    if (r < 0) return graph_fail(ctx, r);
    return r;
}

int hqtick_graph_remove(hqtick_ctx *ctx, uint64_t n, const uint64_t *task_id, int recursive) {
    if (!ctx || (n && !task_id)) return HQTICK_E_INVALID;
    if (int rc = graph_pre(ctx)) return rc;
    int r = ctx->graph.remove(n, task_id, recursive != 0, ctx->stream);
    if (r < 0) return graph_fail(ctx, r);
    if (r > 0 && ctx->n_ready) {  // TaskQueue::remove for the ones that were ready  core.rs:227-231
        if (!ctx->h_q.ensure(64)) return fail(ctx, HQTICK_E_DEVICE, "hipHostMalloc");
        uint32_t *cnt = ctx->h_q.as<uint32_t>(); cnt[0] = 0;
        HQ_HIP(hqk::ready_mark_removed(ctx->d_tid.as<uint64_t>(), ctx->d_trq.as<uint32_t>(), ctx->n_ready, ctx->graph.out_id(), (uint32_t)r, ctx->h_q.dev<uint32_t>(), ctx->stream));
        HQ_HIP(hipStreamSynchronize(ctx->stream));
        ctx->n_live -= cnt[0]; ctx->last_valid = false; ctx->last_consumed = true;
    }
    return r;
}

const uint64_t *hqtick_graph_last_ids(const hqtick_ctx *ctx, uint64_t *n) {
    if (!ctx) { if (n) *n = 0; return nullptr; }
    if (n) *n = ctx->graph.n_out();
    return ctx->graph.n_out() ? ctx->graph.out_id_host() : nullptr;
}

uint64_t hqtick_graph_last_unknown(const hqtick_ctx *ctx) { return ctx ? ctx->graph.n_unknown() : 0; }

int hqtick_graph_unfinished(hqtick_ctx *ctx, uint64_t n, const uint64_t *task_id, uint32_t *out) {
    if (!ctx || (n && (!task_id || !out))) return HQTICK_E_INVALID;
    if (hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, HQTICK_E_DEVICE, "hipSetDevice");
    int r = ctx->graph.unfinished(n, task_id, out, ctx->stream);
    return r < 0 ? graph_fail(ctx, r) : 0;
}

int hqtick_graph_blevel(hqtick_ctx *ctx, uint32_t flags, uint32_t *max_level, uint32_t *n_ready_updated) {
    if (!ctx) return HQTICK_E_INVALID;
    if (flags & ~(uint32_t)HQTICK_BLEVEL_UPDATE_READY) return fail(ctx, HQTICK_E_INVALID, "hqtick_graph_blevel: unknown flag");
    if (hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, HQTICK_E_DEVICE, "hipSetDevice");
    const bool upd = (flags & HQTICK_BLEVEL_UPDATE_READY) && ctx->resident && ctx->n_ready;
    int r = ctx->graph.blevel(max_level, upd ? ctx->d_tid.as<uint64_t>() : nullptr, upd ? ctx->d_trq.as<uint32_t>() : nullptr, upd ? ctx->d_tprio.as<uint64_t>() : nullptr, upd ? ctx->n_ready : 0,
                              n_ready_updated, ctx->stream);
    if (r < 0) return graph_fail(ctx, r);
    // the ready set's priorities changed: the next tick rediscovers its levels.  A pending hqtick_ready_consume_last is left alone — its replay reads the group keys, the
    // per-slice counters and the plan of the last tick, none of which the priority rewrite touches.
    if (upd) ctx->levels_valid = false;
    return r;
}

int hqtick_graph_priorities(hqtick_ctx *ctx, uint64_t n, const uint64_t *task_id, uint64_t *out) {
    if (!ctx || (n && (!task_id || !out))) return HQTICK_E_INVALID;
    if (hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, HQTICK_E_DEVICE, "hipSetDevice");
    int r = ctx->graph.priorities(n, task_id, out, ctx->stream);
    return r < 0 ? graph_fail(ctx, r) : 0;
}

int hqtick_graph_get_stats(const hqtick_ctx *ctx, hqtick_graph_stats *out) {
    if (!ctx || !out) return HQTICK_E_INVALID;
    hqgraph::Stats st = ctx->graph.stats();
    out->n_tasks = st.n_tasks; out->n_slots = st.n_slots; out->n_edges_live = st.n_edges_live; out->n_edges_pool = st.n_edges_pool; out->n_runs = st.n_runs;
    out->hash_capacity = st.hash_capacity; out->hash_tombstones = st.hash_tombstones; out->bytes_hbm = st.bytes_hbm; out->last_kernel_us = ctx->graph.last_kernel_us();
    return 0;
}

// ---- cluster tables resident in HBM (row f1: the reactor's worker bookkeeping as deltas) ----
int hqtick_cluster_upload(hqtick_ctx *ctx, const hqtick_snapshot *s) {
    if (!ctx || !s) return HQTICK_E_INVALID;
    if (hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, HQTICK_E_NO_DEVICE, "hipSetDevice failed");
    int rc = validate(ctx, s, false);
    if (rc) return rc;
    const uint32_t W = s->n_workers, R = s->n_resources;
    const TabLayout L = table_layout(s, W);
    if (ctx->cl_pending) { HQ_HIP(hipEventSynchronize(ctx->cl_ev)); ctx->cl_pending = false; }
    HQ_HIP(hipStreamSynchronize(ctx->stream));
    if (!ctx->h_cl.ensure(L.bytes) || !ctx->d_cluster.ensure(L.bytes + 65536)) return fail(ctx, HQTICK_E_DEVICE, "allocating cluster tables");
    unsigned char *h = ctx->h_cl.as<unsigned char>();
    memset(h, 0, L.bytes);
    pack_worker_rows(h, L, W, R, s->worker_total, s->worker_free, s->worker_remaining_ns);
    pack_request_tables(h, L, s);
    HQ_HIP(hipMemcpyAsync(ctx->d_cluster.p, h, L.bytes, hipMemcpyHostToDevice, ctx->stream));
    HQ_HIP(hipStreamSynchronize(ctx->stream));
    ctx->cl_rt.assign(h + L.o_amt, h + L.bytes);
    ctx->cl_W = W; ctx->cl_R = R; ctx->cluster_valid = true;
    {   // the host mirror (ABI 7)
        hqtick_ctx::ClusterMirror &m = ctx->mirror;
        m.id.assign(s->worker_id, s->worker_id + W);
        m.total.assign(s->worker_total, s->worker_total + (size_t)W * R); m.free_.assign(s->worker_free, s->worker_free + (size_t)W * R);
        m.rem.assign(W, HQ_NO_TIME_LIMIT); if (s->worker_remaining_ns) m.rem.assign(s->worker_remaining_ns, s->worker_remaining_ns + W);
        m.min_util.assign(W, 0.0f); if (s->worker_min_utilization) m.min_util.assign(s->worker_min_utilization, s->worker_min_utilization + W);
        m.flags.assign(W, HQ_WORKER_SN); if (s->worker_flags) m.flags.assign(s->worker_flags, s->worker_flags + W);
        m.group.assign(W, 0); if (s->worker_group) m.group.assign(s->worker_group, s->worker_group + W);
        m.n_groups = s->n_groups ? s->n_groups : 1;
        m.blocked.clear();
        for (uint32_t i = 0; i < s->n_blocked; i++) m.blocked[s->worker_id[s->blocked_worker[i]]].push_back({s->blocked_rq[i], s->blocked_variant[i]});
        m.blk_dirty = true; m.valid = true;
    }
    return 0;
}

int hqtick_cluster_update_workers(hqtick_ctx *ctx, uint32_t n, const uint32_t *worker_index, const uint64_t *free_rows, const int64_t *remaining_ns) {
    if (!ctx) return HQTICK_E_INVALID;
    if (!ctx->cluster_valid) return fail(ctx, HQTICK_E_INVALID, "hqtick_cluster_update_workers without hqtick_cluster_upload");
    if (n == 0) return 0;
    if (!worker_index || !free_rows) return fail(ctx, HQTICK_E_INVALID, "hqtick_cluster_update_workers: null array");
    if (hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, HQTICK_E_NO_DEVICE, "hipSetDevice failed");
    const uint32_t W = ctx->cl_W, R = ctx->cl_R;
    for (uint32_t i = 0; i < n; i++) if (worker_index[i] >= W) return fail(ctx, HQTICK_E_INVALID, "hqtick_cluster_update_workers: worker index out of range");
    // staging: [free n*R u64][rem n i64][index n u32]; the scatter kernel reads it in place (pinned, device-mapped) — wait for the previous one first
    if (ctx->cl_pending) { HQ_HIP(hipEventSynchronize(ctx->cl_ev)); ctx->cl_pending = false; }
    const size_t o_rem = (size_t)n * R * 8, o_idx = o_rem + (size_t)n * 8, bytes = o_idx + (size_t)n * 4 + 16;
    if (!ctx->h_cld.ensure(bytes)) return fail(ctx, HQTICK_E_DEVICE, "allocating delta staging");
    unsigned char *h = ctx->h_cld.as<unsigned char>(), *d = ctx->h_cld.dev<unsigned char>();
    memcpy(h, free_rows, o_rem);
    if (remaining_ns) memcpy(h + o_rem, remaining_ns, (size_t)n * 8);
    memcpy(h + o_idx, worker_index, (size_t)n * 4);
    if (ctx->mirror.valid) for (uint32_t i = 0; i < n; i++) {  // the host mirror follows
        memcpy(ctx->mirror.free_.data() + (size_t)worker_index[i] * R, free_rows + (size_t)i * R, (size_t)R * 8);
        if (remaining_ns) ctx->mirror.rem[worker_index[i]] = remaining_ns[i];
    }
    unsigned char *base = ctx->d_cluster.as<unsigned char>();
    const size_t WR8 = (size_t)W * R * 8;
    HQ_HIP(hqk::scatter_worker_rows(reinterpret_cast<uint64_t *>(base + WR8), reinterpret_cast<int64_t *>(base + 2 * WR8), R, n, reinterpret_cast<const uint32_t *>(d + o_idx),
                                    reinterpret_cast<const uint64_t *>(d), remaining_ns ? reinterpret_cast<const int64_t *>(d + o_rem) : nullptr, ctx->stream));
    HQ_HIP(hipEventRecord(ctx->cl_ev, ctx->stream)); ctx->cl_pending = true;
    return 0;
}

int hqtick_cluster_drop(hqtick_ctx *ctx) {
    if (!ctx) return HQTICK_E_INVALID;
    ctx->cluster_valid = false; ctx->mirror.valid = false;
    return 0;
}

// Membership change: the rows `src` (old row index, or W_old + k for the k-th staged new worker) become the new table; one re-pack kernel, request tables copied
// device to device.  The staging of `add_*` rows sits behind the index list in the pinned delta buffer.
static int cluster_repack(hqtick_ctx *ctx, const std::vector<uint32_t> &src, uint32_t n_add, const uint64_t *add_total, const uint64_t *add_free, const int64_t *add_rem) {
    const uint32_t W_old = ctx->cl_W, R = ctx->cl_R, W_new = (uint32_t)src.size();
    HQ_HIP(hipSetDevice(ctx->device));
    if (ctx->cl_pending) { HQ_HIP(hipEventSynchronize(ctx->cl_ev)); ctx->cl_pending = false; }
    const size_t rt_bytes = ctx->cl_rt.size();
    const size_t o_amt_old = ((size_t)2 * W_old * R + W_old) * 8, o_amt_new = ((size_t)2 * W_new * R + W_new) * 8;
    if (!ctx->d_cluster2.ensure(o_amt_new + rt_bytes + 65536)) return fail(ctx, HQTICK_E_DEVICE, "allocating cluster tables");
    const size_t o_tot = ((size_t)W_new * 4 + 15) & ~(size_t)15, o_fr = o_tot + (size_t)n_add * R * 8, o_rem = o_fr + (size_t)n_add * R * 8, bytes = o_rem + (size_t)n_add * 8 + 64;
    if (!ctx->h_cld.ensure(bytes)) return fail(ctx, HQTICK_E_DEVICE, "allocating delta staging");
    unsigned char *h = ctx->h_cld.as<unsigned char>(), *d = ctx->h_cld.dev<unsigned char>();
    memcpy(h, src.data(), (size_t)W_new * 4);
    if (n_add) { memcpy(h + o_tot, add_total, (size_t)n_add * R * 8); memcpy(h + o_fr, add_free, (size_t)n_add * R * 8); memcpy(h + o_rem, add_rem, (size_t)n_add * 8); }
    unsigned char *ob = ctx->d_cluster.as<unsigned char>(), *nb = ctx->d_cluster2.as<unsigned char>();
    const size_t WR8o = (size_t)W_old * R * 8, WR8n = (size_t)W_new * R * 8;
    HQ_HIP(hqk::repack_worker_rows(reinterpret_cast<const uint64_t *>(ob), reinterpret_cast<const uint64_t *>(ob + WR8o), reinterpret_cast<const int64_t *>(ob + 2 * WR8o), W_old, R, W_new,
                                   reinterpret_cast<const uint32_t *>(d), reinterpret_cast<const uint64_t *>(d + o_tot), reinterpret_cast<const uint64_t *>(d + o_fr), reinterpret_cast<const int64_t *>(d + o_rem),
                                   reinterpret_cast<uint64_t *>(nb), reinterpret_cast<uint64_t *>(nb + WR8n), reinterpret_cast<int64_t *>(nb + 2 * WR8n), ctx->stream));
    if (rt_bytes) HQ_HIP(hipMemcpyAsync(nb + o_amt_new, ob + o_amt_old, rt_bytes, hipMemcpyDeviceToDevice, ctx->stream));
    HQ_HIP(hipEventRecord(ctx->cl_ev, ctx->stream)); ctx->cl_pending = true;
    std::swap(ctx->d_cluster, ctx->d_cluster2);
    ctx->cl_W = W_new;
    return 0;
}

int hqtick_cluster_add_workers(hqtick_ctx *ctx, uint32_t n, const uint32_t *worker_id, const uint64_t *total_rows, const uint64_t *free_rows, const int64_t *remaining_ns,
                               const float *min_utilization, const uint8_t *flags, const uint32_t *group) {
    if (!ctx) return HQTICK_E_INVALID;
    if (!ctx->cluster_valid || !ctx->mirror.valid) return fail(ctx, HQTICK_E_INVALID, "hqtick_cluster_add_workers without hqtick_cluster_upload");
    if (n == 0) return 0;
    if (!worker_id || !total_rows || !free_rows) return fail(ctx, HQTICK_E_INVALID, "hqtick_cluster_add_workers: null array");
    hqtick_ctx::ClusterMirror &m = ctx->mirror;
    const uint32_t W = ctx->cl_W, R = ctx->cl_R;
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t prev = i ? worker_id[i - 1] : (W ? m.id[W - 1] : 0u);
        if ((i || W) && worker_id[i] <= prev) return fail(ctx, HQTICK_E_INVALID, "hqtick_cluster_add_workers: ids must ascend above every id present");
        if (group && group[i] >= 65536) return fail(ctx, HQTICK_E_INVALID, "hqtick_cluster_add_workers: group index");
    }
    std::vector<uint32_t> src(W + n);
    for (uint32_t i = 0; i < W + n; i++) src[i] = i;
    std::vector<int64_t> rem(n, HQ_NO_TIME_LIMIT);
    if (remaining_ns) rem.assign(remaining_ns, remaining_ns + n);
    if (int rc = cluster_repack(ctx, src, n, total_rows, free_rows, rem.data())) return rc;
    m.id.insert(m.id.end(), worker_id, worker_id + n);
    m.total.insert(m.total.end(), total_rows, total_rows + (size_t)n * R); m.free_.insert(m.free_.end(), free_rows, free_rows + (size_t)n * R);
    m.rem.insert(m.rem.end(), rem.begin(), rem.end());
    for (uint32_t i = 0; i < n; i++) {
        m.min_util.push_back(min_utilization ? min_utilization[i] : 0.0f); m.flags.push_back(flags ? flags[i] : (uint8_t)HQ_WORKER_SN);
        m.group.push_back(group ? group[i] : 0u); if (group && group[i] + 1 > m.n_groups) m.n_groups = group[i] + 1;
    }
    m.blk_dirty = true;
    return 0;
}

int hqtick_cluster_remove_workers(hqtick_ctx *ctx, uint32_t n, const uint32_t *worker_id) {
    if (!ctx) return HQTICK_E_INVALID;
    if (!ctx->cluster_valid || !ctx->mirror.valid) return fail(ctx, HQTICK_E_INVALID, "hqtick_cluster_remove_workers without hqtick_cluster_upload");
    if (n == 0) return 0;
    if (!worker_id) return fail(ctx, HQTICK_E_INVALID, "hqtick_cluster_remove_workers: null array");
    hqtick_ctx::ClusterMirror &m = ctx->mirror;
    const uint32_t W = ctx->cl_W, R = ctx->cl_R;
    std::vector<uint8_t> gone(W, 0);
    for (uint32_t i = 0; i < n; i++) {
        auto it = std::lower_bound(m.id.begin(), m.id.end(), worker_id[i]);
        if (it == m.id.end() || *it != worker_id[i] || gone[it - m.id.begin()]) return fail(ctx, HQTICK_E_INVALID, "hqtick_cluster_remove_workers: unknown (or repeated) worker id");
        gone[it - m.id.begin()] = 1;
    }
    std::vector<uint32_t> src; src.reserve(W - n);
    for (uint32_t w = 0; w < W; w++) if (!gone[w]) src.push_back(w);
    if (int rc = cluster_repack(ctx, src, 0, nullptr, nullptr, nullptr)) return rc;
    uint32_t k = 0;
    for (uint32_t w = 0; w < W; w++) {
        if (gone[w]) { m.blocked.erase(m.id[w]); continue; }
        if (k != w) {
            m.id[k] = m.id[w]; m.rem[k] = m.rem[w]; m.min_util[k] = m.min_util[w]; m.flags[k] = m.flags[w]; m.group[k] = m.group[w];
            memmove(m.total.data() + (size_t)k * R, m.total.data() + (size_t)w * R, (size_t)R * 8); memmove(m.free_.data() + (size_t)k * R, m.free_.data() + (size_t)w * R, (size_t)R * 8);
        }
        k++;
    }
    ctx->resp_task.clear(); ctx->resp_worker.clear(); ctx->resp_variant.clear();
    if (!ctx->retr.empty()) {  // on_remove_worker's two passes over the Retracting tasks (server/reactor.rs:86-147), one pass over the table whatever n is
        std::vector<uint32_t> lost(worker_id, worker_id + n);
        std::sort(lost.begin(), lost.end());
        auto is_lost = [&](uint32_t id) { return std::binary_search(lost.begin(), lost.end(), id); };
        for (auto it = ctx->retr.begin(); it != ctx->retr.end();) {
            hqtick_ctx::RetrEntry &e = it->second;
            if (is_lost(e.old_id)) {
                // the worker it was retracting from is gone: with a redirect (to a worker that stays) the task is Assigned{target} now and the host sends its
                // ComputeTasks message (reactor.rs:131-141) — reported through hqtick_cluster_last_reassigned; without one it is a Waiting task of its queue
                if (e.has_redirect && !is_lost(e.target_id)) { ctx->resp_task.push_back(it->first); ctx->resp_worker.push_back(e.target_id); ctx->resp_variant.push_back(e.variant); }
                it = ctx->retr.erase(it);
                continue;
            }
            // the redirect TARGET is gone: the redirect is dropped and the task goes back into its queue, still Retracting{old} (reactor.rs:89-94: redirects.remove +
            // add_ready_task) — the host re-adds it to the resident ready set with the other tasks of the lost worker; the next tick sees it as Retracting again
            if (e.has_redirect && is_lost(e.target_id)) { e.has_redirect = false; e.in_queue = true; }
            ++it;
        }
    }
    m.id.resize(k); m.rem.resize(k); m.min_util.resize(k); m.flags.resize(k); m.group.resize(k); m.total.resize((size_t)k * R); m.free_.resize((size_t)k * R);
    m.blk_dirty = true;
    return 0;
}

int hqtick_cluster_last_reassigned(const hqtick_ctx *ctx, uint32_t *n, const uint64_t **task_id, const uint32_t **worker_id, const uint8_t **variant) {
    if (!ctx) return HQTICK_E_INVALID;
    if (n) *n = (uint32_t)ctx->resp_task.size();
    if (task_id) *task_id = ctx->resp_task.data();
    if (worker_id) *worker_id = ctx->resp_worker.data();
    if (variant) *variant = ctx->resp_variant.data();
    return 0;
}

int hqtick_cluster_set_blocked(hqtick_ctx *ctx, uint32_t worker_id, uint32_t n, const uint32_t *rq, const uint8_t *variant) {
    if (!ctx) return HQTICK_E_INVALID;
    if (!ctx->cluster_valid || !ctx->mirror.valid) return fail(ctx, HQTICK_E_INVALID, "hqtick_cluster_set_blocked without hqtick_cluster_upload");
    hqtick_ctx::ClusterMirror &m = ctx->mirror;
    if (!std::binary_search(m.id.begin(), m.id.end(), worker_id)) return fail(ctx, HQTICK_E_INVALID, "hqtick_cluster_set_blocked: unknown worker id");
    if (n && (!rq || !variant)) return fail(ctx, HQTICK_E_INVALID, "hqtick_cluster_set_blocked: null array");
    if (n == 0) m.blocked.erase(worker_id);
    else { auto &v = m.blocked[worker_id]; v.clear(); for (uint32_t i = 0; i < n; i++) v.push_back({rq[i], variant[i]}); }
    m.blk_dirty = true;
    return 0;
}

int hqtick_cluster_workers(const hqtick_ctx *ctx, uint32_t *n_workers, const uint32_t **worker_id) {
    if (!ctx || !ctx->mirror.valid) return HQTICK_E_INVALID;
    if (n_workers) *n_workers = (uint32_t)ctx->mirror.id.size();
    if (worker_id) *worker_id = ctx->mirror.id.data();
    return 0;
}

int hqtick_retracting_add(hqtick_ctx *ctx, uint32_t n, const uint64_t *task_id, const uint32_t *worker_id) {
    if (!ctx) return HQTICK_E_INVALID;
    if (n && (!task_id || !worker_id)) return fail(ctx, HQTICK_E_INVALID, "hqtick_retracting_add: null array");
    for (uint32_t i = 0; i < n; i++) ctx->retr[task_id[i]] = hqtick_ctx::RetrEntry{worker_id[i], true, false, 0, 0};
    return 0;
}

int hqtick_retract_response(hqtick_ctx *ctx, uint32_t worker_id, uint32_t n, const uint64_t *task_id, uint32_t *n_assigned, const uint64_t **assigned_task,
                            const uint32_t **assigned_worker_id, const uint8_t **assigned_variant) {
    if (!ctx) return HQTICK_E_INVALID;
    if (n && !task_id) return fail(ctx, HQTICK_E_INVALID, "hqtick_retract_response: null array");
    ctx->resp_task.clear(); ctx->resp_worker.clear(); ctx->resp_variant.clear();
    int left = 0;
    for (uint32_t i = 0; i < n; i++) {
        auto it = ctx->retr.find(task_id[i]);
        if (it == ctx->retr.end() || it->second.old_id != worker_id) continue;  // "retracted task is in invalid state"  reactor.rs:476-481
        if (it->second.has_redirect) { ctx->resp_task.push_back(task_id[i]); ctx->resp_worker.push_back(it->second.target_id); ctx->resp_variant.push_back(it->second.variant); }
        ctx->retr.erase(it); left++;
    }
    if (n_assigned) *n_assigned = (uint32_t)ctx->resp_task.size();
    if (assigned_task) *assigned_task = ctx->resp_task.data();
    if (assigned_worker_id) *assigned_worker_id = ctx->resp_worker.data();
    if (assigned_variant) *assigned_variant = ctx->resp_variant.data();
    return left;
}

uint32_t hqtick_retracting_count(const hqtick_ctx *ctx) { return ctx ? (uint32_t)ctx->retr.size() : 0u; }

int hqtick_ready_compact(hqtick_ctx *ctx) {
    if (!ctx) return HQTICK_E_INVALID;
    if (!ctx->resident) return fail(ctx, HQTICK_E_INVALID, "no resident ready set");
    if (ctx->n_live == ctx->n_ready) return 0;
    HQ_HIP(hipSetDevice(ctx->device));
    return rebuild_ready(ctx, nullptr, nullptr, nullptr, 0);
}

}  // extern "C"

extern "C" {

// The what-if query runs on the live Core between two ticks (crates/tako/src/control.rs:125-155): it must not disturb what the ctx keeps
// resident (ready-set columns, level table, the last tick's selection that hqtick_ready_consume_last replays, the dependency graph).  It
// therefore runs on a private sub-context — same device, own stream and buffers — created on first use.
static int query_on(hqtick_ctx *ctx, const hqtick_snapshot *s, const hqtick_query_workers *fake, hqtick_query_result *out);
int hqtick_query(hqtick_ctx *ctx, const hqtick_snapshot *s, const hqtick_query_workers *fake, hqtick_query_result *out) {
    if (!ctx || !out || !fake) return HQTICK_E_INVALID;
    if (!ctx->qctx) {
        hqtick_config cfg = ctx->cfg; cfg.flags |= HQTICK_FLAG_NO_KERNEL_TIMING;
        int rc = hqtick_create(&cfg, &ctx->qctx);
        if (rc) { ctx->qctx = nullptr; return fail(ctx, rc, "creating the query sub-context failed"); }
    }
    int rc = query_on(ctx->qctx, s, fake, out);
    if (rc < 0) ctx->err = ctx->qctx->err;
    return rc;
}
static int query_on(hqtick_ctx *ctx, const hqtick_snapshot *s, const hqtick_query_workers *fake, hqtick_query_result *out) {
    int rc = validate(ctx, s, true);
    if (rc) return rc;
    HQ_HIP(hipSetDevice(ctx->device));
    const uint32_t R = s->n_resources, Q = s->n_requests;
    uint64_t N = s->n_ready;
    if (!ctx->d_tid.ensure(N * 8 + 8) || !ctx->d_tprio.ensure(N * 8 + 8) || !ctx->d_trq.ensure(N * 4 + 8)) return fail(ctx, HQTICK_E_DEVICE, "hipMalloc ready set");
    if (N) {
        HQ_HIP(hipMemcpyAsync(ctx->d_tprio.p, s->task_priority, N * 8, hipMemcpyHostToDevice, ctx->stream));
        HQ_HIP(hipMemcpyAsync(ctx->d_trq.p, s->task_rq, N * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    ctx->n_ready = N; ctx->resident = false;
    ctx->levels_valid = false;
    WorkerEval ev_real, ev_fake;
    if ((rc = eval_workers_sync(ctx, s, fake->n_workers, fake->worker_total, fake->worker_total, fake->worker_remaining_ns, &ev_fake))) return rc;  // fresh fake workers: free == total
    Scan sc;
    if ((rc = phase_a(ctx, s, &ev_real, &sc))) return rc;
    hqhost::Problem pb;
    fill_problem(pb, s, ctx->cfg, ev_real);
    hqhost::WorkerSet fw;
    fw.n = fake->n_workers; fw.R = R; fw.id = fake->worker_id; fw.total = fake->worker_total; fw.free_ = fake->worker_total;
    fw.remaining_ns = fake->worker_remaining_ns; fw.min_util = fake->worker_min_utilization; fw.flags = nullptr; fw.group = nullptr;
    fw.vflags = ev_fake.flags; fw.vtmc = ev_fake.tmc; fw.n_variant_slots = Q ? s->rq_variant_off[Q] : 0;
    fw.blocked.assign(fw.n, {});
    hqhost::group_equal_rows(fw, true);
    pb.custom = &fw;
    std::vector<hqhost::QueueLevels> qlv = queue_levels(sc, s);
    std::vector<hqhost::TaskBatch> batches = hqhost::create_task_batches(pb, qlv);
    DeviceBlocks dev_blocks(ctx);
    pb.blocks = &dev_blocks; pb.block_min_classes = ctx->block_min_classes; pb.pricer = ctx->pricer;
    hqhost::Counts cnt = hqhost::run_scheduling_solver(pb, batches);
    if (cnt.error) return fail(ctx, cnt.error, cnt.errmsg);
    ctx->q_loaded.assign(fake->n_workers, 0);
    for (auto &k : cnt.per_key) for (auto &wc : k) if (wc.second > 0) ctx->q_loaded[wc.first] = 1;  // query.rs:73-81
    out->n_workers = fake->n_workers; out->is_loaded = ctx->q_loaded.data(); out->is_optimal = cnt.is_optimal;
    return 0;
}

#ifdef HQTICK_TEST_HOOKS
// ---- libhqtick_test.so only (include/hqtick_debug.h): CPU hooks for the test suite.  Not compiled into libhqtick.so. ----
namespace {
// k_block_solve's algorithm (csrc/block_core.h) with the wavefront emulated by a loop over its 64 lanes
struct EmulatedBlocks : hqhost::BlockSolver {
    uint32_t budget;
    explicit EmulatedBlocks(uint32_t b) : budget(b) {}
    bool solve(const hqblock::ColTable &ct, const hqblock::ClassTable &cl, const hqblock::Output &out) override {
        static thread_local hqblock::Shared *S = new hqblock::Shared();
        hqblock::HostWave wv;
        // the column table as ONE 16-byte-aligned allocation, as the tick stages it for the kernel: the emulation then takes the LDS-staging path too
        const uint32_t NC = ct.n_cols, R = ct.R, ne = ct.ent_off[NC];
        auto al8 = [](size_t v) { return (v + 7) & ~(size_t)7; };
        const size_t o_res = al8((size_t)(NC + 1) * 4), o_w = al8(o_res + (size_t)ne * 4), o_kind = al8(o_w + (size_t)NC * 4), o_amt = al8(o_kind + ne), o_pool = o_amt + (size_t)ne * 8,
                     bytes = (o_pool + (size_t)R * 8 + 15) & ~(size_t)15;
        std::vector<hqblock::V16> store(bytes / 16 + 1);
        unsigned char *b = reinterpret_cast<unsigned char *>(store.data());
        memcpy(b, ct.ent_off, (size_t)(NC + 1) * 4); memcpy(b + o_res, ct.ent_res, (size_t)ne * 4); memcpy(b + o_w, ct.weight, (size_t)NC * 4);
        memcpy(b + o_kind, ct.ent_kind, ne); memcpy(b + o_amt, ct.ent_amount, (size_t)ne * 8); memcpy(b + o_pool, ct.pool, (size_t)R * 8);
        hqblock::ColTable bt{NC, R, (const uint32_t *)b, (const uint32_t *)(b + o_res), b + o_kind, (const uint64_t *)(b + o_amt), (const uint32_t *)(b + o_w), (const double *)(b + o_pool), b, (uint32_t)bytes};
        for (uint32_t c = 0; c < cl.n_classes; c++) hqblock::solve_block(wv, *S, (c & 1) ? ct : bt, cl, c, out, budget);  // odd classes: tables read in place
        if (corrupt_mode && corrupt_class < cl.n_classes && out.status[corrupt_class] == hqblock::ST_OK) {  // fault injection (tests/test_block_solve.py): what a wrong kernel would hand back
            uint32_t *x = out.x + (size_t)corrupt_class * NC;
            if (corrupt_mode == 1) { for (uint32_t g = 0; g < NC; g++) if (x[g]) { x[g]--; break; } }               // one task short: feasible, not maximal
            else if (corrupt_mode == 2) { for (uint32_t g = NC; g-- > 0;) if (x[g]) { x[g] += 1000; break; } }       // does not fit the rows
            else if (corrupt_mode == 3) {                                                                             // everything on ONE column: corrupt_fill = column << 16 | count
                const uint32_t col = corrupt_fill >> 16, n = corrupt_fill & 0xFFFFu;
                if (col < NC) { for (uint32_t g = 0; g < NC; g++) x[g] = 0; x[col] = n; }
            }
        }
        return true;
    }
    int corrupt_mode = 0; uint32_t corrupt_class = 0, corrupt_fill = 0;
};
thread_local int g_block_emulation = 0, g_price_emulation = 0; thread_local uint32_t g_price_min_cols = 0, g_last_price_sweeps = 0, g_last_price_rounds = 0;
thread_local uint32_t g_block_budget = 4096, g_last_blocks_device = 0, g_last_blocks_host = 0;
thread_local uint32_t g_block_verify = 2, g_tick_seq = 0, g_last_guard[3] = {0, 0, 0}; thread_local int g_corrupt_mode = 0; thread_local uint32_t g_corrupt_class = 0, g_corrupt_fill = 0;
thread_local double g_last_stage_us[3] = {0, 0, 0};
}  // namespace

void hqtick_debug_set_block_guard(uint32_t verify, uint32_t tick_seq, int corrupt_mode, uint32_t corrupt_class, uint32_t corrupt_fill) {
    g_block_verify = verify; g_tick_seq = tick_seq; g_corrupt_mode = corrupt_mode; g_corrupt_class = corrupt_class; g_corrupt_fill = corrupt_fill;
}
void hqtick_debug_last_block_guard(uint32_t *verified, uint32_t *mismatch, uint32_t *rejected) {
    if (verified) *verified = g_last_guard[0]; if (mismatch) *mismatch = g_last_guard[1]; if (rejected) *rejected = g_last_guard[2];
}
thread_local hqhost::BlockMemo *g_memo = nullptr; thread_local uint32_t g_last_memo = 0;
void hqtick_debug_set_block_memo(int on) { if (on && !g_memo) g_memo = new hqhost::BlockMemo(); if (!on) { delete g_memo; g_memo = nullptr; } }
uint32_t hqtick_debug_last_block_memo(void) { return g_last_memo; }
void hqtick_debug_set_price_emulation(int on, uint32_t min_cols) { g_price_emulation = on; g_price_min_cols = min_cols; }
void hqtick_debug_set_fast_path(int on) { hqmilp::set_fast_path(on); }
void hqtick_debug_check_model_hints(int on) { hqprice::set_check_hints(on != 0); }
int hqtick_debug_model_hint_mismatches(void) { return hqprice::hint_mismatches(); }
thread_local int g_price_fault = -1;
void hqtick_debug_set_price_fault(int fail_at) { g_price_fault = fail_at; }
// the host stages as ONE RANK of a sharded scheduler: emulated sweeps / class blocks over this rank's share, completed through `fn` (tests/test_sharded.py: gloo)
thread_local hqtick_exchange_fn g_xfn = nullptr; thread_local void *g_xuser = nullptr; thread_local uint32_t g_xrank = 0, g_xworld = 1, g_xmin_blocks = 1, g_xmin_classes = 1, g_xcalls = 0;
void hqtick_debug_set_exchange(hqtick_exchange_fn fn, void *user, uint32_t rank, uint32_t world, uint32_t min_blocks, uint32_t min_classes) {
    g_xfn = fn; g_xuser = user; g_xrank = rank; g_xworld = world ? world : 1; g_xmin_blocks = min_blocks; g_xmin_classes = min_classes;
}
uint32_t hqtick_debug_last_exchange_calls(void) { return g_xcalls; }
void hqtick_debug_last_stage_us(double *out3) { if (out3) for (int i = 0; i < 3; i++) out3[i] = g_last_stage_us[i]; }
void hqtick_debug_last_price(uint32_t *sweeps, uint32_t *rounds) { if (sweeps) *sweeps = g_last_price_sweeps; if (rounds) *rounds = g_last_price_rounds; }
void hqtick_debug_set_block_emulation(int on, uint32_t budget) { g_block_emulation = on; if (budget) g_block_budget = budget; }
void hqtick_debug_last_blocks(uint32_t *n_emulated, uint32_t *n_host) { if (n_emulated) *n_emulated = g_last_blocks_device; if (n_host) *n_host = g_last_blocks_host; }

int hqtick_debug_block_solve_host(uint32_t n_cols, uint32_t n_resources, const uint32_t *ent_off, const uint32_t *ent_res, const uint8_t *ent_kind, const uint64_t *ent_amount,
                                  const uint32_t *weight, const double *pool, uint32_t n_classes, const uint64_t *free_, const uint64_t *total, const uint64_t *elig,
                                  uint32_t budget, uint32_t *x, uint32_t *status, uint32_t *steps) {
    hqblock::ColTable ct{n_cols, n_resources, ent_off, ent_res, ent_kind, ent_amount, weight, pool, nullptr, 0};
    hqblock::ClassTable cl{n_classes, free_, total, elig};
    hqblock::Output out{x, status, steps, nullptr};
    EmulatedBlocks emu(budget ? budget : 4096);
    return emu.solve(ct, cl, out) ? 0 : HQTICK_E_DEVICE;
}

// Test hook (include/hqtick_debug.h): the host stages on caller-supplied scan outputs.  No device is touched.
int hqtick_debug_host_stages(const hqtick_config *config, const hqtick_snapshot *s, const uint8_t *vflags, const uint32_t *vtmc, uint32_t n_levels,
                             const uint64_t *levels, const uint32_t *hist, hqtick_result *out) {
    if (!config || !s || !out) return HQTICK_E_INVALID;
    static thread_local hqtick_ctx *holder = nullptr;  // owns the result arrays; never creates a stream or a buffer
    if (!holder) holder = new hqtick_ctx();
    hqtick_ctx *ctx = holder;
    ctx->cfg = *config;
    if (int rc = validate(ctx, s, false)) return rc;
    WorkerEval ev; ev.flags = vflags; ev.tmc = vtmc;
    hqhost::Problem pb;
    fill_problem(pb, s, ctx->cfg, ev);
    Scan sc; sc.Q = s->n_requests; sc.L = n_levels; sc.G = n_levels * s->n_requests;
    sc.levels.assign(levels, levels + n_levels); sc.hist.assign(hist, hist + (size_t)n_levels * s->n_requests);
    std::vector<hqhost::QueueLevels> qlv = queue_levels(sc, s);
    std::vector<hqhost::TaskBatch> batches = hqhost::create_task_batches(pb, qlv);
    memset(out, 0, sizeof(*out));
    export_batches(ctx, batches, out);
    EmulatedBlocks emu(g_block_budget);
    emu.corrupt_mode = g_corrupt_mode; emu.corrupt_class = g_corrupt_class; emu.corrupt_fill = g_corrupt_fill;
    if (g_block_emulation) { pb.blocks = &emu; pb.block_min_classes = 1; }
    pb.block_verify = g_block_verify; pb.tick_seq = g_tick_seq; pb.memo = g_memo;
    hqprice::EmulatedSweeper pemu;
    if (g_price_emulation) { pemu.fail_at = g_price_fault; pemu.budget = g_block_budget; if (g_price_min_cols) pemu.min_cols = g_price_min_cols; pb.pricer = &pemu; }
    hqhost::Counts cnt;
    if (g_xfn && g_xworld > 1) {  // this call is one rank of a sharded scheduler
        struct FnExchange : hqprice::Exchange { bool allgather(const void *snd, void *rcv, size_t bytes) override { return g_xfn(g_xuser, snd, rcv, bytes) == 0; } } xch;
        xch.rank = g_xrank; xch.world = g_xworld;
        ShardedBlocks sh_blocks(emu, xch, g_xmin_classes);
        hqprice::ShardedSweeper sh_sweeps(pemu, xch, g_xmin_blocks);
        if (pb.blocks) pb.blocks = &sh_blocks;
        if (pb.pricer) pb.pricer = &sh_sweeps;
        cnt = hqhost::run_scheduling_solver(pb, batches);
        g_xcalls = (uint32_t)xch.n_calls;
    } else cnt = hqhost::run_scheduling_solver(pb, batches);
    if (cnt.error) return fail(ctx, cnt.error, cnt.errmsg);
    g_last_guard[0] = cnt.blocks_verified; g_last_guard[1] = cnt.blocks_mismatch; g_last_guard[2] = cnt.blocks_rejected;
    g_last_memo = cnt.blocks_memo;
    g_last_blocks_device = cnt.blocks_device; g_last_blocks_host = cnt.blocks_host; g_last_price_sweeps = (uint32_t)cnt.price_sweeps; g_last_price_rounds = (uint32_t)cnt.price_rounds;
    g_last_stage_us[0] = cnt.t_classify_us; g_last_stage_us[1] = cnt.t_blocks_us; g_last_stage_us[2] = cnt.t_decode_us;
    ctx->cnt_rq.clear(); ctx->cnt_variant.clear(); ctx->cnt_worker.clear(); ctx->cnt_value.clear();
    cnt.pairs();
    for (size_t k = 0; k < cnt.keys.size(); k++)
        for (auto &wc : cnt.per_key[k]) { ctx->cnt_rq.push_back(cnt.keys[k].first); ctx->cnt_variant.push_back(cnt.keys[k].second); ctx->cnt_worker.push_back(wc.first); ctx->cnt_value.push_back(wc.second); }
    out->n_counts = (uint32_t)ctx->cnt_rq.size(); out->count_rq = ctx->cnt_rq.data(); out->count_variant = ctx->cnt_variant.data();
    out->count_worker = ctx->cnt_worker.data(); out->count_value = ctx->cnt_value.data();
    out->is_optimal = cnt.is_optimal; out->is_canonical = cnt.is_canonical && cnt.is_optimal;
    out->status = cnt.is_optimal ? HQTICK_DONE : (cnt.empty() ? HQTICK_NO_PROGRESS : HQTICK_NEED_MORE_COMPUTE);
    return out->status;
}

int hqtick_debug_host_query(const hqtick_config *config, const hqtick_snapshot *s, const hqtick_query_workers *fake, const uint8_t *vflags, const uint32_t *vtmc,
                            const uint8_t *fake_vflags, const uint32_t *fake_vtmc, uint32_t n_levels, const uint64_t *levels, const uint32_t *hist,
                            hqtick_query_result *out) {
    if (!config || !s || !fake || !out) return HQTICK_E_INVALID;
    static thread_local hqtick_ctx *holder = nullptr;
    if (!holder) holder = new hqtick_ctx();
    hqtick_ctx *ctx = holder;
    ctx->cfg = *config;
    if (int rc = validate(ctx, s, false)) return rc;
    const uint32_t R = s->n_resources, Q = s->n_requests;
    WorkerEval ev; ev.flags = vflags; ev.tmc = vtmc;
    hqhost::Problem pb;
    fill_problem(pb, s, ctx->cfg, ev);
    hqhost::WorkerSet fw;  // as in hqtick_query
    fw.n = fake->n_workers; fw.R = R; fw.id = fake->worker_id; fw.total = fake->worker_total; fw.free_ = fake->worker_total;
    fw.remaining_ns = fake->worker_remaining_ns; fw.min_util = fake->worker_min_utilization; fw.flags = nullptr; fw.group = nullptr;
    fw.vflags = fake_vflags; fw.vtmc = fake_vtmc; fw.n_variant_slots = Q ? s->rq_variant_off[Q] : 0;
    fw.blocked.assign(fw.n, {});
    hqhost::group_equal_rows(fw, true);
    pb.custom = &fw;
    Scan sc; sc.Q = Q; sc.L = n_levels; sc.G = n_levels * Q;
    sc.levels.assign(levels, levels + n_levels); sc.hist.assign(hist, hist + (size_t)n_levels * Q);
    std::vector<hqhost::QueueLevels> qlv = queue_levels(sc, s);
    std::vector<hqhost::TaskBatch> batches = hqhost::create_task_batches(pb, qlv);
    EmulatedBlocks emu(g_block_budget);
    if (g_block_emulation) { pb.blocks = &emu; pb.block_min_classes = 1; }
    hqhost::Counts cnt = hqhost::run_scheduling_solver(pb, batches);
    if (cnt.error) return fail(ctx, cnt.error, cnt.errmsg);
    ctx->q_loaded.assign(fake->n_workers, 0);
    for (auto &k : cnt.per_key) for (auto &wc : k) if (wc.second > 0) ctx->q_loaded[wc.first] = 1;  // query.rs:73-81
    out->n_workers = fake->n_workers; out->is_loaded = ctx->q_loaded.data(); out->is_optimal = cnt.is_optimal;
    return 0;
}

#endif  // HQTICK_TEST_HOOKS

int hqtick_set_shard(hqtick_ctx *ctx, uint32_t shard_index, uint32_t shard_count) {
    if (!ctx) return HQTICK_E_INVALID;
    if (shard_count > 1 && shard_index >= shard_count) return fail(ctx, HQTICK_E_INVALID, "shard_index >= shard_count");
    ctx->shard_index = shard_count > 1 ? shard_index : 0; ctx->shard_count = shard_count > 1 ? shard_count : 1;
    return 0;
}

int hqtick_set_exchange(hqtick_ctx *ctx, hqtick_exchange_fn fn, void *user) {
    if (!ctx) return HQTICK_E_INVALID;
    ctx->xfn = fn; ctx->xuser = fn ? user : nullptr;
    return 0;
}

size_t hqtick_sink_bytes(uint32_t n_workers, uint32_t capacity_records) {
    size_t o_task = (16 + (size_t)(n_workers + 1) * 4 + 7) & ~(size_t)7;
    return (o_task + (size_t)capacity_records * 10 + 15) & ~(size_t)15;
}

uint32_t hqtick_sink_capacity_records(uint32_t n_workers, size_t capacity_bytes) {
    size_t fixed = hqtick_sink_bytes(n_workers, 0);
    if (capacity_bytes < fixed) return 0;
    size_t cap = (capacity_bytes - fixed) / 10;
    while (cap && hqtick_sink_bytes(n_workers, (uint32_t)cap) > capacity_bytes) cap--;
    return cap > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)cap;
}

int hqtick_set_record_sink(hqtick_ctx *ctx, void *device_ptr, size_t capacity_bytes) {
    if (!ctx) return HQTICK_E_INVALID;
    if (device_ptr && (reinterpret_cast<uintptr_t>(device_ptr) & 15)) return fail(ctx, HQTICK_E_INVALID, "record sink must be 16-byte aligned");
    ctx->sink = device_ptr; ctx->sink_bytes = device_ptr ? capacity_bytes : 0;
    return 0;
}

#ifdef HQTICK_TEST_HOOKS  // measurement hooks of the tools (include/hqtick_debug.h): libhqtick_test.so only
int hqtick_time_kernel(hqtick_ctx *ctx, int which, int iters, double *avg_us) {
    if (!ctx || !avg_us || iters <= 0) return HQTICK_E_INVALID;
    if (!ctx->last_valid || !ctx->resident) return fail(ctx, HQTICK_E_INVALID, "hqtick_time_kernel needs a preceding hqtick_run_resident tick that placed tasks");
    HQ_HIP(hipSetDevice(ctx->device));
    const uint64_t N = ctx->n_ready; const hqk::WaveGeom g = ctx->last_geom;
    const uint32_t *d = ctx->d_map.as<uint32_t>();
    uint32_t *hist_dev = reinterpret_cast<uint32_t *>(ctx->h_a.dev<unsigned char>() + 16);
    if (which == 2) {  // calibration: an EMPTY kernel of K1's grid, each launch bracketed by its own start / stop events like the stats pass of a tick
        double sum = 0.0; int cnt = 0;
        for (int i = 0; i < iters; i++) {
            hqk::time_next_launch(ctx->ev[2], ctx->ev[3]);
            HQ_HIP(hqk::empty_like_level_hist(g, ctx->stream));
            HQ_HIP(hipStreamSynchronize(ctx->stream));
            const double us_ = elapsed_us(ctx->ev[2], ctx->ev[3]);
            if (us_ >= 0) { sum += us_; cnt++; }
        }
        *avg_us = cnt ? sum / cnt : 0.0;
        return 0;
    }
    // HQTICK_KTIME_GRAPH: the launches captured into one graph and replayed — no host launch cost between them (a launch call with K1's argument block takes
    // longer than the kernel runs at 1 M tasks), so (graph duration) / iters is what the GPU spends per launch, kernel boundary included
    const bool as_graph = getenv("HQTICK_KTIME_GRAPH") != nullptr;
    if (as_graph) HQ_HIP(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
    else HQ_HIP(hipEventRecord(ctx->ev[10], ctx->stream));
    for (int i = 0; i < iters; i++) {
        if (which == 0) HQ_HIP(hqk::level_hist(ctx->d_tprio.as<uint64_t>(), ctx->d_trq.as<uint32_t>(), N, ctx->d_levels.as<uint64_t>(), ctx->h_levels.data(), ctx->last_L, ctx->last_Q, g, ctx->d_wave_tab.as<uint32_t>(),
                                              ctx->d_gkey.as<uint16_t>(), ctx->d_flags.as<uint32_t>() + 2, nullptr, ctx->stream));
        else if (which == 1) HQ_HIP(hqk::select_scatter(ctx->d_tid.as<uint64_t>(), ctx->d_gkey.as<uint16_t>(), N, ctx->last_Q, ctx->last_G, g, ctx->d_wave_tab.as<uint32_t>(), ctx->h_plan.as<uint32_t>() + ctx->last_tb,
                                                        d + ctx->last_tb, ctx->d_sel_task.as<uint64_t>(), ctx->d_sel_level.as<uint16_t>(), ctx->h_plan.dev<void>(), ctx->d_map.p, ctx->last_plan_bytes, nullptr, ctx->stream));
        else return fail(ctx, HQTICK_E_INVALID, "which: 0 = level_hist, 1 = select_scatter, 2 = empty kernel of level_hist's grid");
    }
    if (as_graph) {
        hipGraph_t gr = nullptr; hipGraphExec_t ge = nullptr;
        HQ_HIP(hipStreamEndCapture(ctx->stream, &gr));
        HQ_HIP(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
        HQ_HIP(hipGraphLaunch(ge, ctx->stream));  // once untimed
        HQ_HIP(hipEventRecord(ctx->ev[10], ctx->stream));
        HQ_HIP(hipGraphLaunch(ge, ctx->stream));
        HQ_HIP(hipEventRecord(ctx->ev[11], ctx->stream));
        HQ_HIP(hipStreamSynchronize(ctx->stream));
        hipGraphExecDestroy(ge); hipGraphDestroy(gr);
    } else HQ_HIP(hipEventRecord(ctx->ev[11], ctx->stream));
    if (which == 0)  // K1 left raw counts in the slice table: turn them back into offsets
        HQ_HIP(hqk::scan_waves(ctx->d_wave_tab.as<uint32_t>(), g, ctx->last_G, hist_dev, ctx->d_flags.as<uint32_t>() + 2, reinterpret_cast<uint32_t *>(ctx->h_a.dev<unsigned char>()) + 2, ctx->stream));
    HQ_HIP(hipStreamSynchronize(ctx->stream));
    float ms = 0;
    HQ_HIP(hipEventElapsedTime(&ms, ctx->ev[10], ctx->ev[11]));
    *avg_us = (double)ms * 1000.0 / iters;
    return 0;
}

int hqtick_timeline(const hqtick_ctx *ctx, double *out, int cap) {
    if (!ctx || !out) return 0;
    int n = ctx->ntl < cap ? ctx->ntl : cap;
    for (int i = 0; i < n; i++) out[i] = ctx->tl[i];
    return n;
}

const uint64_t *hqtick_block_profile_last(const hqtick_ctx *ctx, uint32_t *n_classes) {
    if (!ctx || !ctx->block_profile || !ctx->n_blkprof) { if (n_classes) *n_classes = 0; return nullptr; }
    if (n_classes) *n_classes = ctx->n_blkprof;
    return ctx->h_blkprof.as<uint64_t>();
}
#endif

int hqtick_set_kernel_timing(hqtick_ctx *ctx, int on) {
    if (!ctx) return HQTICK_E_INVALID;
    ctx->timing = on != 0 && on != 2; ctx->timing_k1 = on == 2;  // (any non-zero value but 2 = every measured kernel, as before ABI 9)
    return 0;
}

int hqtick_kernel_stats_last(const hqtick_ctx *ctx, hqtick_kernel_stats *out) {
    if (!ctx || !out) return HQTICK_E_INVALID;
    *out = ctx->stats;
    out->ready_appends = ctx->n_appends;
    return 0;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------- RCCL all-gather of the shards' record sinks
// north_star: "workers hash-partitioned across the GPUs of one node with a single RCCL allgather over xGMI to merge the assignment vector".
// librccl is loaded on first use (dlopen), so that a single-GPU host needs no RCCL at all and a process that already carries one (e.g.
// PyTorch's) shares it.
namespace {
struct Rccl {
    void *h = nullptr;
    decltype(&ncclGetUniqueId) get_id = nullptr;
    decltype(&ncclCommInitRank) init_rank = nullptr;
    decltype(&ncclAllGather) all_gather = nullptr;
    decltype(&ncclCommDestroy) destroy = nullptr;
    decltype(&ncclGetErrorString) err = nullptr;
    bool ok() const { return h && get_id && init_rank && all_gather && destroy && err; }
};
Rccl &rccl() {
    static Rccl r;
    if (!r.h) {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { r.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL); if (r.h) break; }
        if (r.h) {
            r.get_id = (decltype(r.get_id))dlsym(r.h, "ncclGetUniqueId"); r.init_rank = (decltype(r.init_rank))dlsym(r.h, "ncclCommInitRank");
            r.all_gather = (decltype(r.all_gather))dlsym(r.h, "ncclAllGather"); r.destroy = (decltype(r.destroy))dlsym(r.h, "ncclCommDestroy");
            r.err = (decltype(r.err))dlsym(r.h, "ncclGetErrorString");
        }
    }
    return r;
}
}  // namespace

extern "C" {

int hqtick_comm_unique_id(void *id_out) {
    if (!id_out) return HQTICK_E_INVALID;
    Rccl &r = rccl();
    if (!r.ok()) return HQTICK_E_NO_DEVICE;
    ncclUniqueId id;
    if (r.get_id(&id) != ncclSuccess) return HQTICK_E_DEVICE;
    static_assert(sizeof(id) == HQTICK_COMM_ID_BYTES, "ncclUniqueId size");
    memcpy(id_out, &id, sizeof(id));
    return 0;
}

int hqtick_comm_init(hqtick_ctx *ctx, const void *id, uint32_t rank, uint32_t world) {
    if (!ctx || !id || world == 0 || rank >= world) return HQTICK_E_INVALID;
    Rccl &r = rccl();
    if (!r.ok()) return fail(ctx, HQTICK_E_NO_DEVICE, "librccl.so could not be loaded");
    HQ_HIP(hipSetDevice(ctx->device));
    if (ctx->comm) { r.destroy(ctx->comm); ctx->comm = nullptr; }
    ncclUniqueId uid; memcpy(&uid, id, sizeof(uid));
    ncclResult_t rc = r.init_rank(&ctx->comm, (int)world, uid, (int)rank);
    if (rc != ncclSuccess) { ctx->comm = nullptr; return fail(ctx, HQTICK_E_DEVICE, std::string("ncclCommInitRank: ") + r.err(rc)); }
    ctx->comm_rank = rank; ctx->comm_world = world;
    return hqtick_set_shard(ctx, rank, world);
}

int hqtick_shard_allgather(hqtick_ctx *ctx, void *recv_device, size_t recv_bytes) {
    if (!ctx || !recv_device) return HQTICK_E_INVALID;
    if (!ctx->comm) return fail(ctx, HQTICK_E_INVALID, "hqtick_shard_allgather without hqtick_comm_init");
    if (!ctx->sink || !ctx->sink_bytes) return fail(ctx, HQTICK_E_INVALID, "hqtick_shard_allgather without a record sink (hqtick_set_record_sink)");
    if (recv_bytes < (size_t)ctx->comm_world * ctx->sink_bytes) return fail(ctx, HQTICK_E_CAPACITY, "receive buffer smaller than world x sink bytes");
    HQ_HIP(hipSetDevice(ctx->device));
    ncclResult_t rc = rccl().all_gather(ctx->sink, recv_device, ctx->sink_bytes, ncclUint8, ctx->comm, ctx->stream);
    if (rc != ncclSuccess) return fail(ctx, HQTICK_E_DEVICE, std::string("ncclAllGather: ") + rccl().err(rc));
    HQ_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

}  // extern "C"

namespace {
// all-gather of a small HOST buffer through the RCCL communicator: staged through HBM (ncclAllGather wants device memory on both sides)
bool rccl_allgather_host(hqtick_ctx *ctx, const void *send, void *recv, size_t bytes) {
    if (!ctx->comm || bytes == 0) return false;
    const size_t total = bytes * ctx->comm_world;
    if (!ctx->h_xsend.ensure(bytes) || !ctx->h_xrecv.ensure(total) || !ctx->d_xsend.ensure(bytes) || !ctx->d_xrecv.ensure(total)) return false;
    if (hipSetDevice(ctx->device) != hipSuccess) return false;
    memcpy(ctx->h_xsend.p, send, bytes);
    if (hipMemcpyAsync(ctx->d_xsend.p, ctx->h_xsend.p, bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return false;
    if (rccl().all_gather(ctx->d_xsend.p, ctx->d_xrecv.p, bytes, ncclUint8, ctx->comm, ctx->stream) != ncclSuccess) return false;
    if (hipMemcpyAsync(ctx->h_xrecv.p, ctx->d_xrecv.p, total, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) return false;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return false;
    memcpy(recv, ctx->h_xrecv.p, total);
    return true;
}
}  // namespace

extern "C" {

#ifdef HQTICK_TEST_HOOKS
// libhqtick_test.so: one all-gather of a host buffer through the library's RCCL communicator, as the sharded solve issues it per sweep — for the single-rank
// round trip of the GPU suite (recv == send) and for timing the exchange's fixed cost (H2D + ncclAllGather + D2H + one stream synchronisation)
int hqtick_debug_exchange(hqtick_ctx *ctx, const void *send, void *recv, size_t bytes_per_rank) {
    if (!ctx || !send || !recv) return HQTICK_E_INVALID;
    if (!ctx->comm) return fail(ctx, HQTICK_E_INVALID, "hqtick_debug_exchange without hqtick_comm_init");
    return rccl_allgather_host(ctx, send, recv, bytes_per_rank) ? 0 : fail(ctx, HQTICK_E_DEVICE, "the RCCL exchange failed");
}
#endif

int hqtick_comm_destroy(hqtick_ctx *ctx) {
    if (!ctx) return HQTICK_E_INVALID;
    if (ctx->comm) { hipSetDevice(ctx->device); rccl().destroy(ctx->comm); ctx->comm = nullptr; }
    ctx->comm_world = 0; ctx->comm_rank = 0;
    return 0;
}

}  // extern "C"

static void rccl_destroy_comm(hqtick_ctx *ctx) { hqtick_comm_destroy(ctx); }
