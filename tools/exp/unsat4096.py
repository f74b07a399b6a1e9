"""bench.py's config4_unsaturated tick a few times (4096 workers, 65 536 columns, 22 sweeps of 4096 blocks) — a target for rocprofv3 counter passes"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401
from hyperqueue_amd import abi, workloads
from hyperqueue_amd.tick import Tick
s4 = workloads.make("c4", seed=8, n_workers=4096, n_tasks=56_761)
t = Tick(abi.make_config(time_limit_s=5.0))
t.upload_ready(s4.task_id, s4.task_priority, s4.task_rq, sorted_=True)
sc = s4.to_c()
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    t0 = time.perf_counter(); r = t.tick_raw(sc, resident=True); dt = time.perf_counter() - t0
    ks = t.kernel_stats()
    print(f"tick {1e3 * dt:.2f} ms, sweeps {int(ks['price_sweeps'])}, sweep time {ks['price_sweep_us'] / 1e3:.2f} ms, optimal {int(r.is_optimal)}", flush=True)
