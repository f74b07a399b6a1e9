// k_price_sweep: one PRICE SWEEP of the coupled placement model on the MI355X — every worker's block solved exactly under the current prices of the
// model's wide rows (run_scheduling_solver's model, /root/reference/crates/tako/src/internal/scheduler/solver.rs:95-430, which the reference hands to
// HiGHS, solver/highs.rs:65-88; the method is in price.h, the per-block algorithm in price_core.h / block_core.h).
//
// Launch shape: grid = number of blocks (one per worker: 1024-4096), block = 64 threads = ONE wave64, 25.9 KB of LDS per block (dual vertices, level
// stack; the 64 greedy vectors share the level lists' storage) -> 6 blocks per CU (measured: tools/exp/resident_wg.hip), 1536 resident on the 256 CUs, dealt round-robin over the 8 XCDs by the dispatcher; the blocks
// share nothing but the prices (<= 1 KB, in the kernel arguments) — no XCD-aware mapping is needed.  Integer / f64 scalar work on LDS-resident data:
// not an HBM kernel (a block reads ~0.5-2 KB of tables) and not MFMA work; its figure of merit is block solves per second.
//
// A sweep is one link of a serial chain (master LP on the host -> prices -> sweep -> cut -> master ...), so its LATENCY is what counts:
//   * prices travel in the kernel arguments: no copy, no PCIe read by 1024 wavefronts;
//   * the wide rows' activities are integer, accumulated with 64-bit atomics in HBM: exact, order-free, the same on every replica;
//   * the last workgroup to finish (a ticket in HBM) adds up the per-block values in a fixed order and writes the sweep's totals + a sequence
//     number straight into pinned host memory; the host waits on that word instead of a stream synchronisation (the marker packet behind
//     hipStreamSynchronize costs ~6 us per call);
//   * the patterns stay in HBM (a ring of sweeps) and cross PCIe once, when the master has converged and the primal side needs them.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "dev_wave.h"
#include "price_core.h"
#include "price_dev.h"

namespace hqprice {

static_assert(PARTS == ASLOTS, "the master's parts are the kernel's activity slots");
// Activity vectors per part on the device (SweepOut::asub).  With one, the 64 blocks of a part (1024-block model; 256 at 4096 blocks) add into the same K addresses.
// Four vectors per part were measured in round 5 (a quarter of the queue per address; the last workgroup adds them up): the sweep took as long and the slowest block's
// last stage stayed at ~16 us — the atomics are not what it waits for.  One vector it stays; the plumbing is kept for the next attempt (a power of two).
constexpr int ASUB = 1;

namespace {

struct SweepResult {   // pinned host memory, written by the last workgroup of a sweep
    double cx, rc, bnd;
    uint32_t n_budget, max_steps;
    long long act[KMAX];
    double part_cx[ASLOTS];
    long long part_act[ASLOTS * KMAX];   // [part * K + k]: only the first ASLOTS * K entries are written
    uint32_t seq;      // written last (release, system scope)
};

struct SweepArgs {
    Tables t;
    SweepOut out;
    uint32_t budget, seq;
    uint32_t *ticket;
    SweepResult *res;
    // worker-range shards (price.h: ShardedSweeper): the grid covers the blocks [first, first + gridDim.x) only, and instead of the sweep's totals the last
    // workgroup leaves the range's per-block values in pinned memory (lv_*: arrays indexed by absolute block) — the totals are added up after the ranks' all-gather
    uint32_t first, local;
    double *lv_cx, *lv_rc, *lv_bnd; uint32_t *lv_steps;
    double pi[KMAX];
};

// SH = hqblock::SharedN<N>, N = 8 / 16 / 32: the smallest working set the model's widest block fits (block_core.h) — 13.5 / 17.6 / 25.9 KB of LDS per workgroup, i.e.
// eleven / nine / six blocks resident per CU: a 4096-block sweep of 16-column blocks (BASELINE configs[3]) runs in two generations of resident workgroups instead of
// three.  The answers do not depend on N.
template <class SH>
__global__ __launch_bounds__(WAVE) void k_price_sweep(const SweepArgs a) {
    __shared__ SH S;
    __shared__ uint32_t s_last;
    hqblock::DevWave wv;
    solve_priced_block(wv, S, a.t, a.pi, a.first + blockIdx.x, a.out, a.budget);
    __threadfence();  // the block's results before its ticket
    if (threadIdx.x == 0) s_last = atomicAdd(a.ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    if (a.local) {  // this rank's share of a sharded sweep: per-block values and the partial activity vectors as they are, no totals
        for (uint32_t b0 = a.first + threadIdx.x; b0 < a.first + gridDim.x; b0 += WAVE * 8) {   // (eight blocks' loads in flight, then their stores: see below)
            double vcx[8], vrc[8], vbd[8]; uint32_t vst[8];
            const uint32_t end = a.first + gridDim.x;
#pragma unroll
            for (int u = 0; u < 8; u++) { const uint32_t b = b0 + (uint32_t)u * WAVE; const bool in = b < end; vcx[u] = in ? a.out.blk_cx[b] : 0.0; vrc[u] = in ? a.out.blk_rc[b] : 0.0; vbd[u] = in ? a.out.blk_bnd[b] : 0.0; vst[u] = in ? a.out.blk_steps[b] : 0u; }
#pragma unroll
            for (int u = 0; u < 8; u++) { const uint32_t b = b0 + (uint32_t)u * WAVE; if (b < end) { a.lv_cx[b] = vcx[u]; a.lv_rc[b] = vrc[u]; a.lv_bnd[b] = vbd[u]; a.lv_steps[b] = vst[u]; } }
        }
        for (uint32_t i = threadIdx.x; i < (uint32_t)ASLOTS * a.t.K; i += WAVE) {
            const uint32_t sl = i / a.t.K, k = i - sl * a.t.K;
            long long vs[ASUB];
#pragma unroll
            for (int sub = 0; sub < ASUB; sub++) vs[sub] = a.out.act[((size_t)sl * ASUB + sub) * a.t.K + k];
#pragma unroll
            for (int sub = 0; sub < ASUB; sub++) a.out.act[((size_t)sl * ASUB + sub) * a.t.K + k] = 0;
            long long v = 0;
#pragma unroll
            for (int sub = 0; sub < ASUB; sub++) v += vs[sub];
            a.res->part_act[i] = v;
        }
        if (threadIdx.x == 0) *a.ticket = 0;
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(&a.res->seq, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    // the last workgroup: totals in a fixed order (lane l takes blocks l, l + 64, ...; lane 0 adds the 64 partial sums in lane order)
    const uint32_t nb = a.t.n_blocks;
    double cx = 0.0, rc = 0.0, bnd = 0.0; uint32_t nbud = 0, mx = 0;
    // (eight of a lane's blocks per round: the loads of a round are issued together and the sums then run in the same order as before — block l, l + 64, ... —
    // so the totals are bit for bit the old ones; one round trip to L2 per eight blocks instead of one per block: this workgroup is the tail of every sweep)
    for (uint32_t b0 = threadIdx.x; b0 < nb; b0 += WAVE * 8) {
        double vcx[8], vrc[8], vbd[8]; uint32_t vst[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const uint32_t b = b0 + (uint32_t)u * WAVE; const bool in = b < nb; vcx[u] = in ? a.out.blk_cx[b] : 0.0; vrc[u] = in ? a.out.blk_rc[b] : 0.0; vbd[u] = in ? a.out.blk_bnd[b] : 0.0; vst[u] = in ? a.out.blk_steps[b] : 0u; }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (b0 + (uint32_t)u * WAVE >= nb) break;
            cx += vcx[u]; rc += vrc[u]; bnd += vbd[u];
            nbud += vst[u] >> 31; const uint32_t s = vst[u] & 0x7FFFFFFFu; mx = s > mx ? s : mx;
        }
    }
    double *red = &S.py[0][0];  // the pool's storage again: 3 x 64 doubles + 2 x 64 words
    uint32_t *redu = reinterpret_cast<uint32_t *>(red + 3 * WAVE);
    {   // c.x per part, in a fixed order: four lanes per part take every fourth block of its range, the first of them adds the four partial sums
        const uint32_t per = part_size(nb), g = threadIdx.x >> 2, p = threadIdx.x & 3u;
        const uint32_t b0 = g * per, b1 = b0 + per < nb ? b0 + per : nb;
        double s = 0.0;
        for (uint32_t bb = b0 + p; bb < b1; bb += 32) {   // (eight loads in flight, summed in the order b0 + p, + 4, + 8, ... as before)
            double v8[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { const uint32_t b = bb + 4u * (uint32_t)u; v8[u] = b < b1 ? a.out.blk_cx[b] : 0.0; }
#pragma unroll
            for (int u = 0; u < 8; u++) { if (bb + 4u * (uint32_t)u >= b1) break; s += v8[u]; }
        }
        red[threadIdx.x] = s;
        __syncthreads();
        if (p == 0) a.res->part_cx[g] = ((red[threadIdx.x] + red[threadIdx.x + 1]) + red[threadIdx.x + 2]) + red[threadIdx.x + 3];
        __syncthreads();
    }
    red[threadIdx.x] = cx; red[WAVE + threadIdx.x] = rc; red[2 * WAVE + threadIdx.x] = bnd; redu[threadIdx.x] = nbud; redu[WAVE + threadIdx.x] = mx;
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < a.t.K; k += WAVE) {  // the ASLOTS partial vectors -> the sweep's activities (and the slots ready for the next sweep)
        // (all loads first, then the stores: with a store to the host's pinned result between two loads the compiler keeps their order — it cannot know the
        // two do not alias — and the sixteen loads become sixteen round trips to memory the other workgroups have just written)
        long long vs[ASLOTS * ASUB];
#pragma unroll
        for (int i = 0; i < ASLOTS * ASUB; i++) vs[i] = a.out.act[(size_t)i * a.t.K + k];
#pragma unroll
        for (int i = 0; i < ASLOTS * ASUB; i++) a.out.act[(size_t)i * a.t.K + k] = 0;
        long long sum = 0;
#pragma unroll
        for (int sl = 0; sl < ASLOTS; sl++) {
            long long v = 0;
#pragma unroll
            for (int sub = 0; sub < ASUB; sub++) v += vs[sl * ASUB + sub];
            sum += v; a.res->part_act[(size_t)sl * a.t.K + k] = v;
        }
        a.res->act[k] = sum;
    }
    if (threadIdx.x == 0) {
        double tcx = 0.0, trc = 0.0, tb = 0.0; uint32_t tn = 0, tm = 0;
        for (int l = 0; l < WAVE; l++) { tcx += red[l]; trc += red[WAVE + l]; tb += red[2 * WAVE + l]; tn += redu[l]; tm = redu[WAVE + l] > tm ? redu[WAVE + l] : tm; }
        a.res->cx = tcx; a.res->rc = trc; a.res->bnd = tb; a.res->n_budget = tn; a.res->max_steps = tm;
        *a.ticket = 0;
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&a.res->seq, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

size_t al16(size_t v) { return (v + 15) & ~(size_t)15; }
double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

DeviceSweeper::~DeviceSweeper() { h_stage.release(); h_res.release(); h_pats.release(); h_blkv.release(); d_tab.release(); d_pats.release(); d_blk.release(); d_sync.release(); }

bool DeviceSweeper::begin(const HostTables &t, uint32_t max_sweeps) {
    if (t.K > (uint32_t)KMAX || t.n_blocks == 0) return false;
    if (profile && !h_prof.ensure((size_t)t.n_blocks * 64 + 64)) return false;
    T = &t; n_sweeps = 0; cap_sweeps = max_sweeps;
    max_block_cols = 0;
    for (uint32_t b = 0; b < t.n_blocks; b++) max_block_cols = std::max(max_block_cols, t.blk_off[b + 1] - t.blk_off[b]);
    if (force_nmax) max_block_cols = (uint32_t)hqblock::NMAX;   // (HQTICK_PRICE_NMAX=1: the full-size working set whatever the model — A/B switch)
    const size_t nw = t.w_row.size();
    o_off = 0; o_m = al16(o_off + (size_t)(t.n_blocks + 1) * 4); o_cap = al16(o_m + t.n_blocks); o_cost = al16(o_cap + (size_t)t.n_blocks * MMAX * 8);
    o_a = al16(o_cost + (size_t)t.n_cols * 8); o_ccap = al16(o_a + (size_t)t.n_cols * MMAX * 8); o_woff = al16(o_ccap + (size_t)t.n_cols * 4);
    o_wrow = al16(o_woff + (size_t)(t.n_cols + 1) * 4); o_wcoef = al16(o_wrow + nw * 2); tab_bytes = al16(o_wcoef + nw * 4);
    if (!h_stage.ensure(tab_bytes) || !d_tab.ensure(tab_bytes) || !h_res.ensure(sizeof(SweepResult) + 64)) return false;
    if (!d_pats.ensure((size_t)max_sweeps * t.n_cols * 2) || !d_blk.ensure((size_t)t.n_blocks * 28 + 64) || !d_sync.ensure(64 + (size_t)ASLOTS * ASUB * KMAX * 8)) return false;
    unsigned char *h = h_stage.as<unsigned char>();
    memcpy(h + o_off, t.blk_off.data(), (size_t)(t.n_blocks + 1) * 4); memcpy(h + o_m, t.blk_m.data(), t.n_blocks); memcpy(h + o_cap, t.blk_cap.data(), (size_t)t.n_blocks * MMAX * 8);
    memcpy(h + o_cost, t.col_cost.data(), (size_t)t.n_cols * 8); memcpy(h + o_a, t.col_a.data(), (size_t)t.n_cols * MMAX * 8); memcpy(h + o_ccap, t.col_cap.data(), (size_t)t.n_cols * 4);
    memcpy(h + o_woff, t.col_woff.data(), (size_t)(t.n_cols + 1) * 4);
    if (nw) { memcpy(h + o_wrow, t.w_row.data(), nw * 2); memcpy(h + o_wcoef, t.w_coef.data(), nw * 4); }
    if (hipMemcpyAsync(d_tab.p, h, tab_bytes, hipMemcpyHostToDevice, stream) != hipSuccess) return false;
    if (hipMemsetAsync(d_sync.p, 0, 64 + (size_t)ASLOTS * ASUB * KMAX * 8, stream) != hipSuccess) return false;  // ticket + the wide rows' accumulators
    SweepResult *r = h_res.as<SweepResult>();
    seq = r->seq;  // (whatever the last solve left: the next sweep writes seq + 1)
    return true;
}

bool DeviceSweeper::set_caps(const int32_t *col_cap) {
    if (!T) return false;
    memcpy(h_stage.as<unsigned char>() + o_ccap, col_cap, (size_t)T->n_cols * 4);
    return hipMemcpyAsync(d_tab.as<unsigned char>() + o_ccap, h_stage.as<unsigned char>() + o_ccap, (size_t)T->n_cols * 4, hipMemcpyHostToDevice, stream) == hipSuccess;
}

bool DeviceSweeper::set_block_caps(const double *blk_cap) {
    if (!T) return false;
    const size_t bytes = (size_t)T->n_blocks * MMAX * 8;
    memcpy(h_stage.as<unsigned char>() + o_cap, blk_cap, bytes);
    return hipMemcpyAsync(d_tab.as<unsigned char>() + o_cap, h_stage.as<unsigned char>() + o_cap, bytes, hipMemcpyHostToDevice, stream) == hipSuccess;
}

bool DeviceSweeper::sweep(const double *pi, SweepTotals &out) { return launch(pi, 0, T ? T->n_blocks : 0, false, &out); }

bool DeviceSweeper::sweep_range(const double *pi, uint32_t b0, uint32_t b1, RangeValues &rv) {
    if (!T || b1 > T->n_blocks || b0 > b1) return false;
    const uint32_t nb = T->n_blocks;
    if (!h_blkv.ensure((size_t)nb * 28 + 64)) return false;
    if (b1 > b0) { if (!launch(pi, b0, b1, true, nullptr)) return false; }
    else {  // a rank without blocks (more ranks than parts): nothing to launch, the sweep still counts (the ring of patterns stays aligned across the ranks)
        if (n_sweeps >= cap_sweeps) return false;
        SweepResult *r = h_res.as<SweepResult>();
        memset(r->part_act, 0, sizeof(r->part_act));
        n_sweeps++;
    }
    unsigned char *h = h_blkv.as<unsigned char>();
    rv = RangeValues{(const double *)h, (const double *)(h + (size_t)nb * 8), (const double *)(h + (size_t)nb * 16), (const uint32_t *)(h + (size_t)nb * 24), h_res.as<SweepResult>()->part_act};
    return true;
}

bool DeviceSweeper::launch(const double *pi, uint32_t b0, uint32_t b1, bool local, SweepTotals *outp) {
    if (!T || n_sweeps >= cap_sweeps || b1 <= b0) return false;
    const HostTables &t = *T;
    unsigned char *d = d_tab.as<unsigned char>();
    SweepArgs a;
    a.t = Tables{t.n_blocks, t.n_cols, t.K, (const uint32_t *)(d + o_off), (const uint8_t *)(d + o_m), (const double *)(d + o_cap), (const double *)(d + o_cost), (const double *)(d + o_a),
                 (const int32_t *)(d + o_ccap), (const uint32_t *)(d + o_woff), (const uint16_t *)(d + o_wrow), (const int32_t *)(d + o_wcoef)};
    unsigned char *blk = d_blk.as<unsigned char>();
    a.out = SweepOut{d_pats.as<uint16_t>() + (size_t)n_sweeps * t.n_cols, (double *)blk, (double *)(blk + (size_t)t.n_blocks * 8), (double *)(blk + (size_t)t.n_blocks * 16),
                     (long long *)(d_sync.as<unsigned char>() + 64), (uint32_t *)(blk + (size_t)t.n_blocks * 24), profile ? h_prof.dev<uint64_t>() : nullptr, (uint32_t)ASUB};
    a.budget = budget; a.seq = ++seq; a.ticket = d_sync.as<uint32_t>(); a.res = h_res.dev<SweepResult>();
    a.first = b0; a.local = local ? 1u : 0u;
    if (local) { unsigned char *lv = h_blkv.dev<unsigned char>(); a.lv_cx = (double *)lv; a.lv_rc = (double *)(lv + (size_t)t.n_blocks * 8); a.lv_bnd = (double *)(lv + (size_t)t.n_blocks * 16); a.lv_steps = (uint32_t *)(lv + (size_t)t.n_blocks * 24); }
    else { a.lv_cx = a.lv_rc = a.lv_bnd = nullptr; a.lv_steps = nullptr; }
    memset(a.pi, 0, sizeof(a.pi));
    memcpy(a.pi, pi, (size_t)t.K * 8);
    const double t0 = now_us();
    if (max_block_cols <= 8) hipLaunchKernelGGL(k_price_sweep<hqblock::SharedN<8>>, dim3(b1 - b0), dim3(WAVE), 0, stream, a);
    else if (max_block_cols <= 16) hipLaunchKernelGGL(k_price_sweep<hqblock::SharedN<16>>, dim3(b1 - b0), dim3(WAVE), 0, stream, a);
    else hipLaunchKernelGGL(k_price_sweep<hqblock::SharedN<hqblock::NMAX>>, dim3(b1 - b0), dim3(WAVE), 0, stream, a);
    if (hipGetLastError() != hipSuccess) return false;
    // wait for the sweep's own completion word (pinned memory); the stream synchronisation is the fallback after 2 s
    volatile SweepResult *r = h_res.as<SweepResult>();
    for (uint64_t spins = 0;; spins++) {
        if (__atomic_load_n(&r->seq, __ATOMIC_ACQUIRE) == a.seq) break;
        if ((spins & 0xFFFF) == 0xFFFF && now_us() - t0 > 2.0e6) {
            if (hipStreamSynchronize(stream) != hipSuccess) return false;
            if (__atomic_load_n(&r->seq, __ATOMIC_ACQUIRE) != a.seq) return false;
            break;
        }
    }
    last_kernel_us = now_us() - t0;
    if (profile) {  // HQTICK_PRICE_PROFILE=1: per-stage medians over the blocks of this sweep (100 MHz wavefront clock), accumulated for end()
        const uint64_t *pr = h_prof.as<uint64_t>();
        for (int st = 0; st < 6; st++) { std::vector<double> v; v.reserve(t.n_blocks); for (uint32_t b = 0; b < t.n_blocks; b++) if (pr[(size_t)b * 8 + st + 1] >= pr[(size_t)b * 8 + st] && pr[(size_t)b * 8 + st + 1]) v.push_back((double)(pr[(size_t)b * 8 + st + 1] - pr[(size_t)b * 8 + st]) / 100.0); if (v.empty()) continue; std::sort(v.begin(), v.end()); prof_med[st] += v[v.size() / 2]; prof_max[st] += v.back(); }
        double smax = 0; for (uint32_t b = 0; b < t.n_blocks; b++) smax = std::max(smax, (double)pr[(size_t)b * 8 + 7]);
        prof_steps += smax; prof_n++;
    }
    total_sweeps++; total_block_solves += b1 - b0; total_us += last_kernel_us;
    n_sweeps++;
    if (!outp) return true;
    SweepTotals &out = *outp;
    out.cx = r->cx; out.rc = r->rc; out.bnd = r->bnd; out.n_budget = r->n_budget; out.max_steps = r->max_steps;
    out.act.resize(t.K);
    for (uint32_t k = 0; k < t.K; k++) out.act[k] = r->act[k];
    out.part_cx.assign(r->part_cx, r->part_cx + ASLOTS);
    out.part_act.resize((size_t)ASLOTS * t.K);
    for (size_t i = 0; i < (size_t)ASLOTS * t.K; i++) out.part_act[i] = r->part_act[i];
    return true;
}

const uint16_t *DeviceSweeper::patterns(uint32_t first, uint32_t count) {
    if (!T || first + count > n_sweeps) return nullptr;
    const size_t bytes = (size_t)count * T->n_cols * 2;
    if (!h_pats.ensure(bytes + 64)) return nullptr;
    if (count == 0) return h_pats.as<uint16_t>();
    if (hipMemcpyAsync(h_pats.p, d_pats.as<uint16_t>() + (size_t)first * T->n_cols, bytes, hipMemcpyDeviceToHost, stream) != hipSuccess) return nullptr;
    if (hipStreamSynchronize(stream) != hipSuccess) return nullptr;
    return h_pats.as<uint16_t>();
}

void DeviceSweeper::end() {
    if (profile && prof_n) {
        static const char *names[6] = {"reduced costs + compaction", "dual pool + order", "greedy fills", "level lists + root bound", "walk", "results + activities"};
        fprintf(stderr, "[price profile] %d sweeps, per block and sweep (us; median over blocks / slowest block, averaged over the sweeps):", prof_n);
        for (int st = 0; st < 6; st++) fprintf(stderr, "  %s %.1f / %.1f;", names[st], prof_med[st] / prof_n, prof_max[st] / prof_n);
        fprintf(stderr, "  most search steps of a block %.0f; sweep as the host saw it %.1f us\n", prof_steps / prof_n, total_us / (double)std::max<uint64_t>(1, total_sweeps));
        for (int st = 0; st < 6; st++) prof_med[st] = prof_max[st] = 0; prof_steps = 0; prof_n = 0;
    }
    T = nullptr;
}

}  // namespace hqprice
