"""Randomised campaign of the wire encoding's phase functions (hqwire_debug_encode_host_order) against the bincode oracle and the
independent decoder, on all cores (CPU only).   python tools/wire_fuzz.py --seeds 3000 [--first 1000] [--jobs 8]"""
import argparse
import os
import sys
from multiprocessing import Pool

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def one(seed):
    import wire_cases as wc
    from hyperqueue_amd import wire

    try:
        sc = wc.random_scenario(seed, max_rec=300 if seed % 7 == 0 else 30)
        order = seed % 3
        res = wc.check_scenario(lambda t, r, cap: wire.encode_host_debug(t, r, cap, order), sc)
        t, r = wc.tables_and_records(*sc)
        wc.check_roundtrip(sc, res.messages(r))
        return seed, None, sum(len(x) for x in sc[3]), res.total_bytes
    except BaseException as e:  # noqa: BLE001
        return seed, repr(e)[:300], 0, 0


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=3000)
    ap.add_argument("--first", type=int, default=1000)
    ap.add_argument("--jobs", type=int, default=os.cpu_count())
    a = ap.parse_args()
    bad, recs, nbytes = [], 0, 0
    with Pool(a.jobs) as p:
        for seed, err, n, b in p.imap_unordered(one, range(a.first, a.first + a.seeds), chunksize=8):
            recs += n
            nbytes += b
            if err:
                bad.append(seed)
                print("MISMATCH", seed, err, flush=True)
    print(f"{a.seeds} scenarios, {recs} records, {nbytes} message bytes, {len(bad)} mismatches")
