// Sanitizer harness for the wire-encoding phases (csrc/wire_core.h): every input / output / scratch array in its own heap block of EXACTLY the
// size the ABI promises, the phases run under AddressSanitizer + UBSan -- an out-of-bounds access that the ctypes tests would survive
// silently (and that would be a memory fault on the GPU) aborts here.  Built and driven by tools/wire_asan.py; CPU only, test tooling.
//   wire_asan <scenario.bin> <capacity> <order> [limit]  ->  writes <scenario.bin>.out = header[4] u32 | slot_status | slot_off |
//   (limit given: slot_nfrag u32[S] | frag_end u64[S * HQWIRE_MAX_FRAGMENTS] |) bytes.  A limit turns the fragmentation on (hqwire ABI 2).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../hyperqueue_amd/csrc/wire_core.h"

static std::vector<void *> blocks;
static const uint8_t *cur;
static uint64_t take_u64() { uint64_t v; memcpy(&v, cur, 8); cur += 8; return v; }
// array record: u64 byte length + payload; copied into an exact-size malloc block (NULL-equivalent: a 1-byte block never dereferenced for length 0)
static void *take_array() {
    const uint64_t n = take_u64();
    void *p = malloc(n ? n : 1);
    memcpy(p, cur, n);
    cur += n;
    blocks.push_back(p);
    return p;
}

int main(int argc, char **argv) {
    if (argc < 4) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> buf(sz);
    if (fread(buf.data(), 1, sz, f) != (size_t)sz) return 2;
    fclose(f);
    cur = buf.data();
    const uint64_t capacity = strtoull(argv[2], nullptr, 10);
    const int order = atoi(argv[3]);
    const uint64_t limit = argc > 4 ? strtoull(argv[4], nullptr, 10) : 0;

    hqwire::Args a{};
    a.t.n_tasks = take_u64();
    a.t.n_configs = (uint32_t)take_u64();
    a.r.n_workers = (uint32_t)take_u64();
    a.r.n_records = (uint32_t)take_u64();
    a.r.n_mn = (uint32_t)take_u64();
    a.t.task_id = (const uint64_t *)take_array();
    a.t.task_rq = (const uint32_t *)take_array();
    a.t.task_instance = (const uint32_t *)take_array();
    a.t.task_priority = (const uint64_t *)take_array();
    a.t.task_config = (const uint32_t *)take_array();
    a.t.entry_some = (const uint8_t *)take_array();
    a.t.entry_off = (const uint64_t *)take_array();
    a.t.entry_blob = (const uint8_t *)take_array();
    a.t.config_time_some = (const uint8_t *)take_array();
    a.t.config_time_secs = (const uint64_t *)take_array();
    a.t.config_time_nanos = (const uint32_t *)take_array();
    a.t.body_off = (const uint64_t *)take_array();
    a.t.body_blob = (const uint8_t *)take_array();
    a.r.worker_id = (const uint32_t *)take_array();
    a.r.rec_off = (const uint32_t *)take_array();
    a.r.rec_task = (const uint64_t *)take_array();
    a.r.rec_variant = (const uint8_t *)take_array();
    a.r.rec_kind = (const uint8_t *)take_array();
    a.r.retract_off = (const uint32_t *)take_array();
    a.r.retract_task = (const uint64_t *)take_array();
    a.r.mn_task = (const uint64_t *)take_array();
    a.r.mn_worker_off = (const uint32_t *)take_array();
    a.r.mn_worker = (const uint32_t *)take_array();
    a.n_slots = a.r.n_workers + a.r.n_mn;

    const uint64_t S = a.n_slots, sb = hqwire::scratch_bytes((uint64_t)a.r.n_records + a.r.n_mn, S);
    a.o.bytes = (uint8_t *)malloc(capacity ? capacity : 1);
    a.o.capacity = capacity;
    a.o.slot_off = (uint64_t *)malloc(8 * (2 * S + 1));
    a.o.slot_status = (uint8_t *)malloc(S ? S : 1);
    a.o.header = (uint32_t *)malloc(16);
    a.o.scratch = malloc(sb);
    a.o.scratch_bytes = sb;
    if (limit) {
        a.o.slot_nfrag = (uint32_t *)malloc(S ? 4 * S : 1);
        a.o.frag_end = (uint64_t *)malloc(S ? 8 * S * HQWIRE_MAX_FRAGMENTS : 1);
        a.o.msg_size_limit = limit;
    }
    hqwire::bind_scratch(a);
    if (!hqwire::run_on_host(a, order)) return 3;

    std::string outp = std::string(argv[1]) + ".out";
    FILE *o = fopen(outp.c_str(), "wb");
    fwrite(a.o.header, 4, 4, o);
    fwrite(a.o.slot_status, 1, S, o);
    fwrite(a.o.slot_off, 8, 2 * S + 1, o);
    if (limit) {
        fwrite(a.o.slot_nfrag, 4, S, o);
        fwrite(a.o.frag_end, 8, S * HQWIRE_MAX_FRAGMENTS, o);
    }
    const uint64_t total = (uint64_t)a.o.header[2] | (uint64_t)a.o.header[3] << 32;
    if (a.o.header[0] == HQWIRE_OK) fwrite(a.o.bytes, 1, total, o);
    fclose(o);
    free(a.o.bytes); free(a.o.slot_off); free(a.o.slot_status); free(a.o.header); free(a.o.scratch);
    free(a.o.slot_nfrag); free(a.o.frag_end);
    for (void *p : blocks) free(p);
    return 0;
}
