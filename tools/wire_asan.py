"""Runs the wire-encoding phases under AddressSanitizer + UBSan (tools/wire_asan.cpp) on random ticks and compares the bytes with the oracle.
    python tools/wire_asan.py [--seeds 60]
Every array sits in a heap block of exactly the size the ABI promises (no padding element, unlike the ctypes path), so an out-of-bounds
access of a phase function -- a memory fault on the GPU -- aborts the run.  CPU only."""
import argparse
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def build(exe):
    src = os.path.join(ROOT, "tools", "wire_asan.cpp")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-o", exe, src])


def dump(path, t, r):
    with open(path, "wb") as f:
        f.write(struct.pack("<5Q", t.n_tasks, t.n_configs, r.n_workers, r.n_records, r.n_mn))
        # exact sizes: blobs built by WireTables.build carry one padding byte when empty -- strip it
        arrays = t.arrays() + r.arrays()
        exact = {7: int(t.entry_off[-1]), 12: int(t.body_off[-1])}
        for i, a in enumerate(arrays):
            b = np.ascontiguousarray(a).tobytes()
            if i in exact:
                b = b[: exact[i]]
            f.write(struct.pack("<Q", len(b)))
            f.write(b)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=60)
    a = ap.parse_args()
    import wire_cases as wc
    from hyperqueue_amd import wire

    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "wire_asan")
        build(exe)
        bad = 0
        for seed in range(a.seeds):
            sc = wc.random_scenario(2000 + seed, max_rec=400 if seed % 5 == 0 else 30)
            t, r = wc.tables_and_records(*sc)
            want = wc.oracle_messages(*sc)
            total = sum(len(b) for _, b in want)
            path = os.path.join(d, f"s{seed}.bin")
            dump(path, t, r)
            runs = [(total, seed % 3, 0, want), (max(0, total - 1), 0, 0, want)]  # exact-fit output buffer, then one byte short (CAPACITY: nothing written)
            if seed % 5 == 0:  # the bigger scenarios once more with the builder's limit low enough to cut messages (hqwire ABI 2 fragmentation)
                wf = wc.oracle_messages(*sc, limit=1500)
                runs.append((sum(len(b) for _, b in wf), 0, 1500, wf))
            for cap, order, limit, want_r in runs:
                p = subprocess.run([exe, path, str(cap), str(order)] + ([str(limit)] if limit else []), capture_output=True, text=True)
                if p.returncode != 0:
                    bad += 1
                    print("SANITIZER / failure", seed, cap, p.stderr[-1500:])
                    continue
                raw = open(path + ".out", "rb").read()
                S = r.n_workers + r.n_mn
                header = np.frombuffer(raw[:16], np.uint32)
                status = np.frombuffer(raw[16:16 + S], np.uint8)
                pos = 16 + S
                off = np.frombuffer(raw[pos:pos + 8 * (2 * S + 1)], np.uint64)
                pos += 8 * (2 * S + 1)
                nfrag = frag_end = None
                if limit:
                    nfrag = np.frombuffer(raw[pos:pos + 4 * S], np.uint32)
                    pos += 4 * S
                    frag_end = np.frombuffer(raw[pos:pos + 8 * S * wire.HQWIRE_MAX_FRAGMENTS], np.uint64)
                    pos += 8 * S * wire.HQWIRE_MAX_FRAGMENTS
                data = raw[pos:]
                if cap == sum(len(b) for _, b in want_r):
                    res = wire.WireResult(int(header[0]), cap, status, off, data, nfrag, frag_end)
                    if (status == wire.SLOT_OVERSIZE).any() and limit:
                        continue  # more than HQWIRE_MAX_FRAGMENTS cuts: the host builds that slot (covered by tests/test_wire.py)
                    if not (header[0] == 0 and (status == 0).all() and res.messages(r) == want_r):
                        bad += 1
                        print("MISMATCH", seed, "limit", limit)
                else:
                    if not (total == 0 or (header[0] == wire.HQWIRE_CAPACITY and data == b"")):
                        bad += 1
                        print("CAPACITY not reported", seed)
        # slots the device path refuses: an unknown task id, and one record more than HQWIRE_MAX_RECORDS for a worker
        attrs = {1: (0, 0, 0, 0, None), 3: (0, 0, 0, 0, b"e")}
        sc = (attrs, [(None, b"small")], [10, 11, 12], [[(1, 0, 1)], [(99, 0, 1)], [(3, 0, 1)] * (wire.HQWIRE_MAX_RECORDS + 1)], [[], [5], []], [])
        t, r = wc.tables_and_records(*sc)
        path = os.path.join(d, "special.bin")
        dump(path, t, r)
        p = subprocess.run([exe, path, "4096", "0"], capture_output=True, text=True)
        raw = open(path + ".out", "rb").read() if p.returncode == 0 else b""
        if p.returncode != 0 or list(raw[16:19]) != [wire.SLOT_OK, wire.SLOT_UNKNOWN, wire.SLOT_TOO_MANY]:
            bad += 1
            print("special scenario failed", p.returncode, p.stderr[-1500:])
        print(f"{a.seeds} scenarios x 2 runs + the refused-slot scenario under ASan+UBSan: {bad} problems")
        return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
