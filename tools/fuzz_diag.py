#!/usr/bin/env python
"""Diagnose fuzz seeds (needs a GPU): python tools/fuzz_diag.py <seed>...  — prints which result fields differ between the HIP path and the
canonical oracle, the snapshot, both count lists and the oracle's model size."""
import sys
import os
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, 'tests'))
import torch, numpy as np
import test_gpu_fuzz as f
from hyperqueue_amd.tick import Tick
from oracle.oracle import Oracle
for seed in [int(a) for a in sys.argv[1:]]:
    cfg, envs, rng = f.build(seed)
    g, o = Tick(cfg), Oracle(cfg, canonical=True)
    for round_ in range(3):
        snap = envs[0].snapshot()
        rg, ro = g.tick(snap), o.tick(envs[1].snapshot())
        diffs = [k for k in ("status","batches","counts","records","retracts","mn") if getattr(rg,k) != getattr(ro,k)]
        if not (rg.new_free == ro.new_free).all(): diffs.append("new_free")
        if diffs:
            print("seed", seed, "round", round_, "differs:", diffs, "W", len(snap.worker_id), "Q", len(snap.requests), "N", len(snap.task_id))
            print("  requests:", [[(v["entries"], v["n_nodes"], v["min_time_ns"], v["weight"]) for v in vs] for vs in snap.requests])
            print("  workers total:", snap.worker_total.tolist(), "free:", snap.worker_free.tolist(), "mu", snap.worker_min_utilization.tolist(), "rem", snap.worker_remaining_ns.tolist(), "group", snap.worker_group.tolist())
            print("  blocked", snap.blocked, "flags", snap.worker_flags.tolist())
            print("  gpu status/opt", rg.status, rg.is_optimal, "oracle", ro.status, ro.is_optimal)
            print("  gpu counts", rg.counts); print("  ora counts", ro.counts)
            print("  gpu mn", rg.mn, " ora mn", ro.mn)
            print("  batches equal:", rg.batches == ro.batches)
            if "batches" in diffs: print("  gpu batches", rg.batches); print("  ora batches", ro.batches)
            m = o.last_model()
            print("  model cols", len(m["obj"]), "rows", len(m["rhs"]), "oracle obj", m["objective"])
            break
        envs[0].apply(rg); envs[1].apply(ro)
        k = int(rng.integers(0, 4))
        for e in envs:
            done = 0
            for t in sorted(e.tasks.values(), key=lambda t: t.id):
                if done >= k: break
                if t.state == 1: e.finish_task(t.id, t.worker); done += 1
                elif t.state == 5 and t.mn_workers: e.finish_task(t.id, t.mn_workers[0]); done += 1
