// Worker-range shards of the coupled solve's sweeps: ShardedSweeper runs this rank's blocks only and
// completes every sweep with ONE small all-gather; the master (csrc/price.cpp) stays replicated and sees the same totals on every rank, bit for bit.
// The model being swept is run_scheduling_solver's (/root/reference/crates/tako/src/internal/scheduler/solver.rs:95-430): one block per worker, wide rows
// across them — the per-worker part is what shards (north_star: "workers hash-partitioned across the GPUs"; here by contiguous worker ranges, which is what
// the master's 16 parts are).
#include <algorithm>
#include <chrono>
#include <cstring>
#include <vector>

#include "price.h"
#include "price_core.h"

namespace hqprice {

namespace {
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
size_t al8(size_t v) { return (v + 7) & ~(size_t)7; }
}  // namespace

void totals_from_blocks(uint32_t nb, uint32_t K, const double *cx, const double *rc, const double *bnd, const uint32_t *steps, const long long *part_act, SweepTotals &out) {
    out.act.assign(K, 0);
    for (int sl = 0; sl < ASLOTS; sl++) for (uint32_t k = 0; k < K; k++) out.act[k] += part_act[(size_t)sl * K + k];
    out.part_act.assign(part_act, part_act + (size_t)ASLOTS * K);
    out.part_cx.assign(ASLOTS, 0.0);
    const uint32_t per = part_size(nb);
    for (uint32_t g = 0; g < (uint32_t)ASLOTS; g++) {  // four lanes per part, every fourth block each, then the four partial sums
        const uint32_t b0 = g * per, b1 = std::min(nb, b0 + per);
        double s4[4] = {0, 0, 0, 0};
        for (uint32_t p = 0; p < 4; p++) for (uint32_t b = b0 + p; b < b1; b += 4) s4[p] += cx[b];
        out.part_cx[g] = ((s4[0] + s4[1]) + s4[2]) + s4[3];
    }
    double pcx[WAVE] = {0}, prc[WAVE] = {0}, pb[WAVE] = {0};
    out.n_budget = 0; out.max_steps = 0;
    for (uint32_t b = 0; b < nb; b++) {
        pcx[b % WAVE] += cx[b]; prc[b % WAVE] += rc[b]; pb[b % WAVE] += bnd[b];
        if (steps[b] & 0x80000000u) out.n_budget++;
        out.max_steps = std::max(out.max_steps, steps[b] & 0x7FFFFFFFu);
    }
    out.cx = out.rc = out.bnd = 0.0;
    for (int l = 0; l < WAVE; l++) { out.cx += pcx[l]; out.rc += prc[l]; out.bnd += pb[l]; }
}

bool ShardedSweeper::begin(const HostTables &t, uint32_t max_sweeps) {
    inner.budget = budget;
    const bool inner_ok = inner.begin(t, max_sweeps);
    T = nullptr; n_sweeps = 0;
    pass = ex.world <= 1 || t.n_blocks < min_blocks;   // (a function of the model alone: the same on every rank)
    if (pass) { if (inner_ok) T = &t; return inner_ok; }
    {   // A sharded model: every rank owes the others its part of every sweep's all-gather.  A rank whose device refused the model must not leave alone (the others
        // would wait for it in the first sweep): ONE status word is exchanged before anything else, and all ranks take the sweeps or leave them together — the
        // way ShardedBlocks (csrc/hqtick.cpp) treats a failed inner solver.
        uint64_t mine = inner_ok ? 1u : 0u;
        std::vector<uint64_t> all(ex.world, 0);
        const double t0 = now_s();
        if (!ex.allgather(&mine, all.data(), 8)) { if (inner_ok) inner.end(); return false; }
        ex.us += (now_s() - t0) * 1e6; ex.n_calls++; ex.n_bytes += 8 * (size_t)ex.world;
        bool everyone = true;
        for (uint64_t v : all) everyone = everyone && v == 1u;
        if (!everyone) { if (inner_ok) inner.end(); return false; }
    }
    T = &t;
    const uint32_t W = ex.world, nb = t.n_blocks, per = part_size(nb);
    rank_b0.assign(W, 0); rank_b1.assign(W, 0); rank_p0.assign(W, 0); rank_p1.assign(W, 0);
    max_blocks = max_parts = max_cols = 0;
    for (uint32_t r = 0; r < W; r++) {  // rank r owns the parts [r * 16 / world, (r + 1) * 16 / world): contiguous worker ranges, two per GPU on an 8-GPU node
        const uint32_t p0 = (uint32_t)((uint64_t)r * PARTS / W), p1 = (uint32_t)((uint64_t)(r + 1) * PARTS / W);
        rank_p0[r] = p0; rank_p1[r] = p1;
        rank_b0[r] = std::min(nb, p0 * per); rank_b1[r] = std::min(nb, p1 * per);
        max_blocks = std::max(max_blocks, rank_b1[r] - rank_b0[r]); max_parts = std::max(max_parts, p1 - p0);
        max_cols = std::max(max_cols, t.blk_off[rank_b1[r]] - t.blk_off[rank_b0[r]]);
    }
    cx.assign(nb, 0.0); rc.assign(nb, 0.0); bnd.assign(nb, 0.0); steps.assign(nb, 0); part_act.assign((size_t)PARTS * t.K, 0);
    return true;
}

// blob of one rank and sweep: [u64 clock flag][cx mb][rc mb][bnd mb] doubles, [part_act mp * K] i64, [steps mb] u32
bool ShardedSweeper::sweep(const double *pi, SweepTotals &out) {
    if (!T) return false;
    if (pass) {
        n_sweeps++;
        const bool ok = inner.sweep(pi, out);
        if (ex.world <= 1) return ok;
        // Replicas that sweep the whole (small) model each: the clock readings are still merged — one word per rank — so that all of them leave the sweeps at the
        // same sweep (a local reading here or in the master could split the replicas near the time guard).  A failed sweep takes part, like below.
        const uint64_t mine = (now_s() > guard_s ? 1u : 0u) | (ok ? 0u : 2u);
        std::vector<uint64_t> all(ex.world, 0);
        const double t0 = now_s();
        if (!ex.allgather(&mine, all.data(), 8)) return false;
        ex.us += (now_s() - t0) * 1e6; ex.n_calls++; ex.n_bytes += 8 * (size_t)ex.world;
        bool someone_failed = false;
        for (uint64_t v : all) { if (v & 1u) time_up = true; if (v & 2u) someone_failed = true; }
        return !someone_failed;
    }
    const HostTables &t = *T;
    const uint32_t me = ex.rank, W = ex.world, mb = max_blocks, mp = max_parts, K = t.K;
    RangeValues rv;
    const bool mine_ok = inner.sweep_range(pi, rank_b0[me], rank_b1[me], rv);   // a failed local sweep still takes part in the exchange (flag bit 1): the others are waiting in it
    const size_t o_cx = 8, o_rc = o_cx + (size_t)mb * 8, o_bnd = o_rc + (size_t)mb * 8, o_act = o_bnd + (size_t)mb * 8, o_st = o_act + (size_t)mp * K * 8, bytes = al8(o_st + (size_t)mb * 4);
    send.assign(bytes, 0); recv.resize(bytes * W);
    {
        const uint32_t b0 = rank_b0[me], n = rank_b1[me] - b0, p0 = rank_p0[me], np_ = rank_p1[me] - p0;
        const uint64_t flag = (now_s() > guard_s ? 1u : 0u) | (mine_ok ? 0u : 2u);
        memcpy(send.data(), &flag, 8);
        if (n && mine_ok) { memcpy(send.data() + o_cx, rv.cx + b0, (size_t)n * 8); memcpy(send.data() + o_rc, rv.rc + b0, (size_t)n * 8); memcpy(send.data() + o_bnd, rv.bnd + b0, (size_t)n * 8);
                 memcpy(send.data() + o_st, rv.steps + b0, (size_t)n * 4); }
        if (np_ && mine_ok) memcpy(send.data() + o_act, rv.part_act + (size_t)p0 * K, (size_t)np_ * K * 8);
    }
    const double t0 = now_s();
    if (!ex.allgather(send.data(), recv.data(), bytes)) return false;
    ex.us += (now_s() - t0) * 1e6; ex.n_calls++; ex.n_bytes += bytes * W;
    bool up = false, someone_failed = false;
    for (uint32_t r = 0; r < W; r++) {
        const unsigned char *b = recv.data() + (size_t)r * bytes;
        uint64_t flag; memcpy(&flag, b, 8); up = up || (flag & 1u) != 0; someone_failed = someone_failed || (flag & 2u) != 0;
        const uint32_t b0 = rank_b0[r], n = rank_b1[r] - b0, p0 = rank_p0[r], np_ = rank_p1[r] - p0;
        if (n) { memcpy(cx.data() + b0, b + o_cx, (size_t)n * 8); memcpy(rc.data() + b0, b + o_rc, (size_t)n * 8); memcpy(bnd.data() + b0, b + o_bnd, (size_t)n * 8); memcpy(steps.data() + b0, b + o_st, (size_t)n * 4); }
        if (np_) memcpy(part_act.data() + (size_t)p0 * K, b + o_act, (size_t)np_ * K * 8);
    }
    if (someone_failed) return false;   // every rank sees the same flags: all of them hand the model to the host search, at the same sweep
    if (up) time_up = true;
    totals_from_blocks(t.n_blocks, K, cx.data(), rc.data(), bnd.data(), steps.data(), part_act.data(), out);
    n_sweeps++;
    return true;
}

// the patterns of sweeps [first, first + count): every rank holds its own columns of each; one all-gather of [count][max_cols] u16 per rank completes them
const uint16_t *ShardedSweeper::patterns(uint32_t first, uint32_t count) {
    if (!T || first + count > n_sweeps) return nullptr;
    if (pass) return inner.patterns(first, count);
    const HostTables &t = *T;
    const uint32_t me = ex.rank, W = ex.world, mc = max_cols, nc = t.n_cols;
    pats.assign((size_t)count * nc + 1, 0);
    if (count == 0) return pats.data();
    const uint16_t *mine = inner.patterns(first, count);   // nullptr: this rank's copy failed — it still takes part in the exchange ([u64 ok][patterns]) and all ranks give up together
    const size_t bytes = al8(8 + (size_t)count * mc * 2);
    send.assign(bytes, 0); recv.resize(bytes * W);
    {
        const uint64_t ok = mine ? 1u : 0u;
        memcpy(send.data(), &ok, 8);
        const uint32_t c0 = t.blk_off[rank_b0[me]], n = t.blk_off[rank_b1[me]] - c0;
        for (uint32_t s = 0; s < count && n && mine; s++) memcpy(send.data() + 8 + (size_t)s * mc * 2, mine + (size_t)s * nc + c0, (size_t)n * 2);
    }
    const double t0 = now_s();
    if (!ex.allgather(send.data(), recv.data(), bytes)) return nullptr;
    ex.us += (now_s() - t0) * 1e6; ex.n_calls++; ex.n_bytes += bytes * W;
    bool everyone = true;
    for (uint32_t r = 0; r < W; r++) {
        const uint32_t c0 = t.blk_off[rank_b0[r]], n = t.blk_off[rank_b1[r]] - c0;
        const unsigned char *b = recv.data() + (size_t)r * bytes;
        uint64_t ok; memcpy(&ok, b, 8); everyone = everyone && ok == 1u;
        for (uint32_t s = 0; s < count && n; s++) memcpy(pats.data() + (size_t)s * nc + c0, b + 8 + (size_t)s * mc * 2, (size_t)n * 2);
    }
    return everyone ? pats.data() : nullptr;
}

}  // namespace hqprice
