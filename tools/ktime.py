#!/usr/bin/env python
"""Back-to-back kernel timing sweep (bench tooling; needs a GPU): python tools/ktime.py [workload] [iters] [n_tasks]
HQTICK_KTIME_GRAPH=1: the launches replayed from one captured graph (no host launch cost between them)."""
import ctypes as C, os, sys
import torch  # noqa: F401  (one HIP runtime in the process)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hyperqueue_amd import abi, workloads
from hyperqueue_amd.tick import Tick

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
n_tasks = int(sys.argv[3]) if len(sys.argv) > 3 else None
snap = workloads.make(name, n_tasks=n_tasks)
t = Tick(abi.make_config(time_limit_s=5.0), measure=True)
t.upload_ready(snap.task_id, snap.task_priority, snap.task_rq)
sc = snap.to_c()
for _ in range(3):
    t.tick_raw(sc, resident=True)
t._lib.hqtick_time_kernel.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]
for which, nm, nbytes in ((0, "level_hist", len(snap.task_id) * 12), (1, "select_scatter", len(snap.task_id) * 8)):
    us = C.c_double()
    for rep in range(3):
        rc = t._lib.hqtick_time_kernel(t._ctx, which, iters, C.byref(us))
        assert rc == 0, t._err()
    print(f"N={len(snap.task_id)} TPW={os.environ.get('HQTICK_TPW', '256')} {nm}: {us.value:.2f} us/launch {'in a graph' if os.environ.get('HQTICK_KTIME_GRAPH') else 'back-to-back'}  -> {nbytes / us.value / 1e3:.0f} GB/s on {nbytes / 1e6:.1f} MB")
r = t.tick_raw(sc, resident=True)
print("tick still consistent:", r.status, t.kernel_stats()["n_assigned"])
