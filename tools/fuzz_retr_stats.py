#!/usr/bin/env python
"""Coverage of the prefill-disposal fuzz (needs a GPU): how often Retracting tasks were re-targeted / kept on their worker / hit the reference's
prefill assert over 600 scenarios, and whether the HIP path and the oracle ever disagreed."""
import sys
import os
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, 'tests'))
import torch, numpy as np
from hyperqueue_amd import abi
from hyperqueue_amd.core import SchedEnv, TaskBuilder as TB, WorkerBuilder as WB
from hyperqueue_amd.tick import Tick, HqTickError
from oracle.oracle import Oracle
import test_gpu_fuzz as f
stats = dict(scen=0, with_retracting=0, kind1=0, kind2=0, unsupported=0, mismatch=0)
for seed in range(0, 600):
    rng = np.random.default_rng(9000 + seed)
    cfg = abi.make_config(reserve=int(rng.integers(0, 2)), fill_max=int(rng.integers(1, 4)), time_limit_s=20.0)
    envs = [SchedEnv(cfg), SchedEnv(cfg)]
    g, o = Tick(cfg), Oracle(cfg, canonical=True)
    shapes = [TB().cpus(1), TB().cpus(2)]
    for e in envs:
        for c in [int(x) for x in np.random.default_rng(seed).integers(1, 5, size=3)]: e.new_worker(WB(c))
    prio = 0; stats["scen"] += 1
    try:
        for round_ in range(5):
            n_new = int(rng.integers(1, 7)) if round_ else int(rng.integers(8, 16)); which = [int(rng.integers(0, 2)) for _ in range(n_new)]
            if round_ and rng.random() < 0.7: prio += 1
            for e in envs:
                for c in which: e.new_task(shapes[c].user_priority(prio))
            snaps = [e.snapshot() for e in envs]
            if snaps[0].retracting: stats["with_retracting"] += 1
            try: rg = g.tick(snaps[0])
            except HqTickError as err:
                stats["unsupported"] += 1
                try: o.tick(snaps[1]); stats["mismatch"] += 1; print("seed", seed, "gpu unsupported but oracle ok")
                except RuntimeError: pass
                break
            ro = o.tick(snaps[1])
            f.assert_same(rg, ro)
            assert sorted(zip(rg.redirects, rg.redirect_kinds)) == sorted(zip(ro.redirects, ro.redirect_kinds))
            stats["kind1"] += rg.redirect_kinds.count(1); stats["kind2"] += rg.redirect_kinds.count(2)
            envs[0].apply(rg); envs[1].apply(ro)
            k = int(rng.integers(1, 7)); answer = rng.random() < 0.6
            for e in envs:
                done = 0
                for t in sorted(e.tasks.values(), key=lambda t: t.id):
                    if t.state == 1 and done < k: e.finish_task(t.id, t.worker); done += 1
                if answer:
                    for t in [t for t in sorted(e.tasks.values(), key=lambda t: t.id) if t.state == 4 and t.id not in e.retaken_variant][:2]: e.retract_response(t.worker, [t.id])
    except AssertionError as ex:
        stats["mismatch"] += 1; print("seed", seed, "MISMATCH", str(ex)[:200])
print(stats)
