#!/usr/bin/env python
"""Extended randomised parity campaign (needs a GPU): runs tests/test_gpu_fuzz.py::test_fuzz_scenario for seeds 120..2599 and lists the failing ones."""
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
print("failures:", len(bad))
for b in bad: print(b)
