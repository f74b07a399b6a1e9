"""Randomised allocate / release campaign of libhqalloc.so against the oracle on all cores (CPU only): the scenario family of
tests/test_alloc_capi.py::test_random_sequences_match_oracle over many more seeds.

    python tools/alloc_fuzz.py --seeds 2000 [--first 0] [--jobs 8]
"""
import argparse
import os
import sys
from multiprocessing import Pool

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def one(seed):
    import test_alloc_capi as t

    try:
        t.test_random_sequences_match_oracle(seed - 1000)  # the test adds 1000 to its parameter
        return seed, None
    except BaseException as e:  # noqa: BLE001
        return seed, repr(e)[:300]


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=2000)
    ap.add_argument("--first", type=int, default=5000)
    ap.add_argument("--jobs", type=int, default=os.cpu_count())
    a = ap.parse_args()
    bad = []
    with Pool(a.jobs) as p:
        for seed, err in p.imap_unordered(one, range(a.first, a.first + a.seeds), chunksize=4):
            if err:
                bad.append((seed, err))
                print("MISMATCH", seed, err, flush=True)
    print(f"{a.seeds} scenarios, {len(bad)} mismatches")
