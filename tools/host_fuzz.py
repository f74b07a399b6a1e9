#!/usr/bin/env python
"""Large CPU campaign of tests/test_host_stages.py: host stages (batches + placement) vs the canonical oracle on many more seeds, in parallel.
  python tools/host_fuzz.py <family: fuzz|idle|unsat> <first seed> <n seeds> [processes]"""
import multiprocessing as mp
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(args):
    family, seed = args
    import test_host_stages as t

    try:
        {"fuzz": t.test_host_stages_fuzz_scenarios, "idle": t.test_host_stages_idle_cluster, "unsat": t.test_host_stages_unsaturated}[family](seed)
        return (seed, "ok", "")
    except BaseException as e:  # pytest.skip raises a BaseException subclass
        kind = "skip" if type(e).__name__ == "Skipped" else "FAIL"
        return (seed, kind, f"{type(e).__name__}: {str(e)[:300]}")


if __name__ == "__main__":
    family, first, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    procs = int(sys.argv[4]) if len(sys.argv) > 4 else max(1, (os.cpu_count() or 2) - 1)
    with mp.get_context("spawn").Pool(procs) as pool:
        res = pool.map(run, [(family, s) for s in range(first, first + n)], chunksize=8)
    bad = [r for r in res if r[1] == "FAIL"]
    print(f"{family}: {n} seeds from {first}: ok {sum(r[1] == 'ok' for r in res)}, skipped {sum(r[1] == 'skip' for r in res)}, FAILED {len(bad)}")
    for b in bad[:20]:
        print(b)
