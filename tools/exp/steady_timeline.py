import ctypes as C, sys, os, time
import numpy as np
sys.path.insert(0, os.getcwd())
from hyperqueue_amd import abi, workloads
from hyperqueue_amd.tick import Tick
snap = workloads.make("c3"); sc = snap.to_c(); W = len(snap.worker_id)
t = Tick(abi.make_config(time_limit_s=5.0, flags=6), measure=True)
t.upload_ready(snap.task_id, snap.task_priority, snap.task_rq); t.cluster_upload(sc); t.set_kernel_timing(False)
t._lib.hqtick_timeline.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int]
rq_of = snap.task_rq.copy(); res = t.tick_raw(sc, resident=True); gone = abi.record_task_ids(res, W); t.ready_consume_last()
next_id = int(snap.task_id[-1]) + 1
rows = []
for it in range(25):
    k = len(gone); new_rq = rq_of[(gone & np.uint64(0xFFFFFFFF)).astype(np.int64) - 1]; rq_of = np.concatenate([rq_of, new_rq])
    v_id, v_prio, v_rq = t.ready_add_stage(k); v_id[:] = np.arange(next_id, next_id + k, dtype=np.uint64); next_id += k; v_prio[:] = snap.task_priority[0]; v_rq[:] = new_rq
    t.ready_add_staged(k)
    res = t.tick_raw(sc, resident=True)
    buf = (C.c_double * 32)(); n = t._lib.hqtick_timeline(t._ctx, buf, 32)
    if it >= 5: rows.append([buf[j] for j in range(n)])
    t.ready_consume_last(); gone = abi.record_task_ids(res, W)
m = np.median(np.asarray(rows), axis=0)
labels = ["phaseA", "batches", "solve", "keytables", "prefillplan", "k5tables", "pack", "C_enqueued", "C_synced", "assembled"]
prev = 0
for l, v in zip(labels, m): print(f"{l:12s} at {v:8.1f} (+{v-prev:6.1f})"); prev = v
