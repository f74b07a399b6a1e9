#!/usr/bin/env python
"""Extended randomised parity campaign (needs a GPU): runs tests/test_gpu_fuzz.py::test_fuzz_scenario for seeds 120..2599 and ::test_fuzz_idle_cluster for 600 more seeds, and lists the failing ones."""
import sys
import os
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, 'tests'))
import torch
import test_gpu_fuzz as f
bad=[]
for seed in range(120, 2600):
    try:
        f.test_fuzz_scenario(seed)
    except Exception as e:
        bad.append((seed, type(e).__name__, str(e)[:200]))
        if len(bad) > 8: break
idle_skipped = 0
for seed in range(40, 640):  # the idle-cluster family (few tasks, many identical workers): provably-empty-worker elimination vs the full model
    try:
        f.test_fuzz_idle_cluster(seed)
    except BaseException as e:
        if type(e).__name__ == "Skipped":
            idle_skipped += 1
            continue
        bad.append((("idle", seed), type(e).__name__, str(e)[:200]))
        if len(bad) > 8: break
print("idle-cluster scenarios skipped (solver limit):", idle_skipped)
print("failures:", len(bad))
for b in bad: print(b)
