#!/usr/bin/env python
"""BASELINE-size COUPLED ticks as committed fixtures (run from the repo root: python tests/golden/make_coupled_fixtures.py).

Round 3: ticks whose placement model couples all workers (priority cuts, unsaturated batches) are solved by price sweeps (csrc/price.cpp; k_price_sweep on
the MI355X).  The reference's answer on such a tick is whatever incumbent HiGHS holds within its 1e-4 gap — nothing a fixture could pin — but the PRODUCT's
answer is deterministic, and it can be computed without a GPU: the host stages with the emulated wavefront (libhqtick_test.so) give the counts, and the
oracle's tick GIVEN those counts (decode, create_task_mapping, proactive filling — scheduler/mapping.rs:23-234 restated) gives every record.  The fixture
stores SHA-256 digests of both.  CPU suite: emulation + oracle reproduce the digests.  GPU suite: the HIP tick through the C ABI reproduces them — counts
bit-equal to the emulation's (same arithmetic on both sides), records bit-equal to the oracle's mapping of those counts.
"""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

from hyperqueue_amd import abi, workloads  # noqa: E402

COUPLED = {
    "c3p_full": dict(kind="c3p"),                  # BASELINE.md C3 with three priority levels: 1 M tasks x 1024 workers
    "c4p_full": dict(kind="c4p"),                  # BASELINE configs[3] as BASELINE.md §3 writes it ("as C3": three priority levels): 1 M tasks x 4096 workers, 2-variant OR-lists — 65 552 columns x 12 422 rows
    "c5_first_wave": dict(kind="wave"),            # BASELINE config 5, first tick: the 49 642 sources of the 1 M-node DAG on the idle cluster
    "unsaturated_1024_20": dict(kind="unsat", fill=0.2),
    "c3p_steady_256": dict(kind="steady", n_workers=256, n_tasks=400_000, seed=7),
}


def coupled_snapshot(gen: dict):
    if gen["kind"] == "c3p":
        return workloads.make("c3p", n_tasks=1_000_000, n_workers=1024), abi.make_config(time_limit_s=20.0)
    if gen["kind"] == "c4p":
        return workloads.make("c4p", n_tasks=1_000_000, n_workers=4096), abi.make_config(time_limit_s=60.0)
    if gen["kind"] == "steady":
        return workloads.make_steady("c3p", seed=gen["seed"], n_workers=gen["n_workers"], n_tasks=gen["n_tasks"]), abi.make_config(time_limit_s=20.0)
    ids, prio, rq, off, dep = workloads.make_dag(1_000_000, seed=0)
    src = np.nonzero((off[1:] - off[:-1]) == 0)[0]
    if gen["kind"] == "unsat":
        src = src[: int(len(src) * gen["fill"] / 0.45)]
    drv = workloads.DagChurn(n_workers=1024, churn=0.1, seed=0)
    return drv.snapshot(ids[src], prio[src], (rq[src] % 8).astype(np.uint32)), abi.make_config(time_limit_s=20.0)


def emulated_counts(snap, cfg):
    """the product's host stages with the price sweeps on the emulated wavefront: (result with counts, sweeps)"""
    from host_stages import HostStages

    hs = HostStages(cfg)
    hs.lib.hqtick_debug_set_price_emulation.argtypes = [C.c_int, C.c_uint32]
    hs.lib.hqtick_debug_last_price.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    hs.lib.hqtick_debug_set_price_emulation(1, 0)
    try:
        got = hs.stages(snap)
    finally:
        hs.lib.hqtick_debug_set_price_emulation(0, 0)
    sw = C.c_uint32()
    hs.lib.hqtick_debug_last_price(C.byref(sw), None)
    return got, sw.value


def expected(snap, cfg):
    from make_fixtures import big_digest
    from oracle.oracle import Oracle

    got, sweeps = emulated_counts(snap, cfg)
    assert got.status == abi.HQTICK_DONE and got.is_optimal
    full = Oracle(cfg).tick_given(snap, got.counts, is_optimal=True)
    assert full.counts == got.counts  # the decode's Map orders
    d = big_digest(full)
    d["price_sweeps"] = sweeps
    return d


def main(names=None):
    for name, gen in COUPLED.items():
        if names and name not in names:
            continue
        snap, cfg = coupled_snapshot(gen)
        with open(os.path.join(HERE, "coupled", name + ".json"), "w") as f:
            json.dump(dict(generator=gen, expect=expected(snap, cfg)), f, indent=1)
        print(name, "written")


if __name__ == "__main__":
    main(sys.argv[1:] or None)
