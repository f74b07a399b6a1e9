#!/usr/bin/env python
"""Coupled ticks through the C ABI with the price sweeps (k_price_sweep) on and off: c3p at BASELINE size, the config-5 first wave and the
unsaturated 1024-worker probes of DESIGN.md §4.  Prints status, objective-free figures (assigned tasks), the solve's stage times and sweeps.
  python tools/price_probe.py [c3p 0.2 0.45 wave ...] [--repeat N] [--no-host]"""
import os, sys, time, json
import numpy as np
import torch  # noqa: F401  (one HIP runtime per process, as in bench.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hyperqueue_amd import abi, workloads
from hyperqueue_amd.tick import Tick

_dag = None


def snapshot(which):
    global _dag
    if which == "c3p":
        return workloads.make("c3p", n_tasks=1_000_000, n_workers=1024)
    if which == "c3ps":  # a busy cluster (every worker its own free vector) + three priority levels + 1 M ready tasks: the everyday production tick
        return workloads.make_steady("c3p", seed=0)
    if which == "c4u":   # BASELINE configs[3]'s cluster, unsaturated: 4096 blocks of 16 columns per sweep
        return workloads.make("c4", seed=8, n_workers=4096, n_tasks=56_761)
    if which == "c4p":   # BASELINE configs[3] as written: three priority levels x 4096 workers x 2-variant OR-lists
        return workloads.make("c4p")
    if which.startswith("c3p:"):
        w = int(which.split(":")[1])
        return workloads.make("c3p", n_tasks=1000 * w, n_workers=w)
    if _dag is None:
        ids, prio, rq, off, dep = workloads.make_dag(1_000_000, seed=0)
        _dag = (ids, prio, rq, np.nonzero((off[1:] - off[:-1]) == 0)[0])
    ids, prio, rq, src = _dag
    if which == "wave":  # config 5's first wave: every source of the DAG on the idle cluster
        sel = src
    else:
        fill = float(which)
        sel = src[: min(len(src), int(len(src) * fill / 0.45))]
    drv = workloads.DagChurn(n_workers=1024, churn=0.1, seed=0)
    return drv.snapshot(ids[sel], prio[sel], (rq[sel] % 8).astype(np.uint32))


def timeline(t, snap, which, n=12):
    """median stage marks of the resident tick (ready set + cluster tables in HBM, as the bench's loop)"""
    import ctypes as C
    t.upload_ready(snap.task_id, snap.task_priority, snap.task_rq)
    sc = snap.to_c()
    t.cluster_upload(sc)
    t.set_kernel_timing(False)
    t._lib.hqtick_timeline.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int]
    rows, ks = [], []
    for i in range(n + 3):
        r = t.tick_raw(sc, resident=True)
        buf = (C.c_double * 32)()
        k = t._lib.hqtick_timeline(t._ctx, buf, 32)
        if i >= 3:
            rows.append([buf[j] for j in range(k)] + [r.t_total_us]); ks.append(t.kernel_stats())
    m = np.median(np.asarray(rows), axis=0)
    labels = ["phaseA", "batches", "solve", "keytables", "prefillplan", "k5tables", "pack", "C_enqueued", "C_synced", "assembled", "total"]
    prev, parts = 0.0, []
    for l, v in zip(labels, m):
        parts.append(f"{l} +{v - prev:.0f}"); prev = v
    med = {k: round(float(np.median([s[k] for s in ks])), 1) for k in ("solve_pre_us", "model_us", "milp_us", "price_us", "price_sweep_us", "price_sweeps", "price_rounds")}
    print(which, "resident timeline (us):", " | ".join(parts), "| total", round(float(m[-1]), 1), med, flush=True)
    t.cluster_drop()


def main():
    args = [a for i, a in enumerate(sys.argv[1:], 1) if not a.startswith("--") and sys.argv[i - 1] != "--repeat"]
    repeat = int(sys.argv[sys.argv.index("--repeat") + 1]) if "--repeat" in sys.argv else 3
    cases = args or ["c3p", "wave", "0.2", "0.45"]
    out = {}
    for which in cases:
        snap = snapshot(which)
        for mode in (["price"] if "--no-host" in sys.argv else ["price", "host"]):
            os.environ["HQTICK_PRICE"] = "1" if mode == "price" else "0"
            t = Tick(abi.make_config(time_limit_s=5.0, flags=abi.HQTICK_FLAG_COMPACT_RECORDS | abi.HQTICK_FLAG_COMPACT_DELTA16), measure=True)  # as bench.py runs it
            if "--timeline" in sys.argv and mode == "price":
                timeline(t, snap, which)
            best = None
            for r in range(repeat if mode == "price" else 1):
                t0 = time.perf_counter()
                res = t.tick(snap)
                dt = time.perf_counter() - t0
                ks = t.kernel_stats()
                rec = dict(ms=round(dt * 1e3, 3), status=res.status, is_optimal=bool(res.is_optimal), assigned=int(sum(len(x) for x in res.records)),
                           sweeps=ks["price_sweeps"], rounds=ks["price_rounds"], price_ms=round(ks["price_us"] / 1e3, 3), sweep_ms=round(ks["price_sweep_us"] / 1e3, 3),
                           milp_ms=round(ks["milp_us"] / 1e3, 3), model_ms=round(ks["model_us"] / 1e3, 3), cols=ks["milp_cols"], rows=ks["milp_rows"])
                if best is None or rec["ms"] < best["ms"]:
                    best = rec
            counts = {(a, b, c): d for a, b, c, d in res.counts}
            best["n_counts"] = len(counts)
            out[f"{which}/{mode}"] = best
            print(which, mode, json.dumps(best), flush=True)
            t.close()
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/price_probe.json", "w"), indent=1)


if __name__ == "__main__":
    main()
