#!/usr/bin/env python
"""Extended randomised parity campaign (needs a GPU): runs tests/test_gpu_fuzz.py::test_fuzz_scenario for seeds 120..2599 and ::test_fuzz_idle_cluster for 600 more seeds, and lists the failing ones.
  python tools/fuzz_more.py [first last [idle_first idle_last]]   (other seed ranges)"""
import sys
import os
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, 'tests'))
import torch
import test_gpu_fuzz as f
bad=[]
A = [int(x) for x in sys.argv[1:5]]
S0, S1 = (A[0], A[1]) if len(A) >= 2 else (120, 2600)
I0, I1 = (A[2], A[3]) if len(A) >= 4 else (40, 640)
for seed in range(S0, S1):
    try:
        f.test_fuzz_scenario(seed)
    except Exception as e:
        bad.append((seed, type(e).__name__, str(e)[:200]))
        if len(bad) > 8: break
idle_skipped = 0
for seed in range(I0, I1):  # the idle-cluster family (few tasks, many identical workers): provably-empty-worker elimination vs the full model
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
