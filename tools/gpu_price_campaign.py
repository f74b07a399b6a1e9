#!/usr/bin/env python
"""One-off GPU campaign of the coupled path at cluster scale (needs a GPU): random clusters of 256-1024 workers mid-run, 1-3 priority levels, ready sets that do not
saturate — each through the HIP tick and through the emulated sweeps of the CPU hooks (the same path bit for bit: sweeps, configurations, status, counts), and each
certified tick through the oracle's mapping on the product's counts (T3 given counts: records, retracts, redirects, free vectors).
    python tools/gpu_price_campaign.py [first_seed] [count] [--big]        (--big: clusters of 2048 / 4096 workers, up to the 65 536 columns of BASELINE configs[3])"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401

from hyperqueue_amd import abi, workloads
from hyperqueue_amd.core import priority_from_user
from hyperqueue_amd.tick import Tick
from oracle.oracle import Oracle
from test_price import stages


BIG = "--big" in sys.argv


def scenario(seed):
    rng = np.random.default_rng(seed)
    W = int(rng.choice([2048, 4096] if BIG else [256, 384, 512, 768, 1024]))
    levels = int(rng.integers(1, 4))
    steady = rng.random() < 0.6
    n_ready = int(rng.integers(W * 4, W * 60))
    name = str(rng.choice(["c3", "c4"], p=[0.75, 0.25]))
    if steady:
        snap = workloads.make_steady(name, seed=seed, n_workers=W, n_tasks=n_ready, release=float(rng.choice([0.1, 0.3, 0.6])))
    else:
        snap = workloads.make(name, seed=seed, n_workers=W, n_tasks=n_ready)
    snap.task_priority = np.asarray([priority_from_user(int(p)) for p in rng.integers(0, levels, len(snap.task_id))], np.uint64)
    return snap, dict(W=W, levels=levels, steady=steady, n=n_ready, name=name)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    first = int(args[0]) if len(args) > 0 else 0
    count = int(args[1]) if len(args) > 1 else 40
    bad = 0; certified = 0; swept = 0; t_gpu = []
    for seed in range(first, first + count):
        snap, info = scenario(seed)
        t = Tick(abi.make_config(time_limit_s=5.0))
        try:
            t0 = time.perf_counter(); got = t.tick(snap); dt = time.perf_counter() - t0
            ks = t.kernel_stats()
        finally:
            t.close()
        want, sweeps, rounds = stages(snap, True, tl=60.0)  # (the emulated sweeps are ~100x slower than the kernel's: give them the time to reach what the GPU reaches in its 5 s)
        line = dict(seed=seed, **info, sweeps=int(ks["price_sweeps"]), rounds=int(ks["price_rounds"]), optimal=bool(got.is_optimal), tick_ms=round(dt * 1e3, 2), cols=int(ks["milp_cols"]))
        problems = []
        # (a tick that runs into its time limit is cut by the clock — on the emulation, which is 100x slower per sweep, much earlier: only certified ticks are compared sweep by sweep)
        if got.is_optimal and want.is_optimal and (int(ks["price_sweeps"]), int(ks["price_rounds"])) != (sweeps, rounds): problems.append(f"sweeps {ks['price_sweeps']}/{ks['price_rounds']} vs emulation {sweeps}/{rounds}")
        if got.batches != want.batches or (got.is_optimal and want.is_optimal and got.status != want.status): problems.append("status / batches differ from the emulation")
        if want.is_optimal and got.is_optimal and got.counts != want.counts: problems.append("counts differ from the emulation")
        if got.is_optimal:
            certified += 1
            o = Oracle(abi.make_config(time_limit_s=5.0))
            ref = o.tick_given(snap, got.counts, is_optimal=True)
            if not (got.records == ref.records and got.retracts == ref.retracts and got.redirects == ref.redirects and (got.new_free == ref.new_free).all() and got.counts == ref.counts):
                problems.append("T3 given counts: mapping differs from the oracle's")
        if ks["price_sweeps"]: swept += 1; t_gpu.append(dt * 1e3)
        if problems: bad += 1
        print(("PROBLEM " if problems else "ok      ") + str(line) + (" " + "; ".join(problems) if problems else ""), flush=True)
    print(f"{count} scenarios: {swept} went through k_price_sweep (median tick {np.median(t_gpu) if t_gpu else 0:.2f} ms incl. the first-use costs of a fresh context), {certified} certified, {bad} with problems")


if __name__ == "__main__":
    main()
