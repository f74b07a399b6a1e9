import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
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
