#!/usr/bin/env python
"""bench.py — tasks assigned/sec and tick latency of the MI355X-native tako scheduling tick.

A "step" is one scheduling tick (run_scheduling_inner, /root/reference/crates/tako/src/internal/scheduler/main.rs:50-72) over the synthetic
workload `c3p` = BASELINE.json configs[2] as SURVEY.md §8(d) / BASELINE.md §3 write it: 1 M ready tasks over 8 mixed {cpus, gpus, mem} request
classes incl. fractional GPUs, THREE user-priority levels at 80 / 15 / 5 %, 1024 workers; cold: every worker empty, every task ready.  The
priority cuts (scheduler/batches.rs:97-171) couple every worker's block through wide rows (scheduler/solver.rs:233-253,274-429): the placement
is ONE model of 8 205 columns, solved by price sweeps on the MI355X (k_price_sweep).  The ready-set columns are resident in HBM
when the timed region starts (hqtick_upload_ready), worker / request tables too (hqtick_cluster_upload).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3p]
  N > 1: launched by torch.distributed.run, one rank per GPU (or self-launched): ONE scheduler whose workers are hash-sharded over the ranks
  (hyperqueue_amd/sharded.py); weak scaling — 1024 workers and 1 M ready tasks per GPU; every rank sweeps its own worker range
  (one small all-gather per sweep), one RCCL all-gather merges the shards' assignment vectors; value = tasks assigned by the whole job per second.
  `--workload c3` is the one-level variant (every class saturated: the placement separates per worker) — round 1-4's headline, now a neighbour.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def cpu_model() -> str:
    """the host CPU as /proc/cpuinfo names it (state the CPU model and '1 core' next to every CPU baseline)"""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


LINE_LIMIT = 4096  # bytes: the driver's record keeps a parsed line only below this (BENCH_r05: a 20 KB line came back `parsed: None`)


def _num(x, digits=6):
    """a JSON-safe number: finite floats rounded to `digits` significant figures, anything else (NaN, inf, None) -> None"""
    if x is None or isinstance(x, bool):
        return x
    if isinstance(x, (int, np.integer)):
        return int(x)
    x = float(x)
    if not np.isfinite(x):
        return None
    return float(f"{x:.{digits}g}")


def headline_line(*, value, n_gpus, steps, warmup, ms_per_step, scaling, workload, priority_levels, parallelism, ranks_in_comm, seed, p50_tick_ms, assigned_per_tick,
                  all_done, caches_live, p50_warm_ms, model_columns, price_sweeps, gpu_busy_share, roofline, dominant, cpu):
    """THE line bench.py prints on stdout: exactly the keys VERDICT r05 (next 2) lists, nothing else, < LINE_LIMIT bytes.  Everything else goes to --extras-file."""
    line = {
        "metric": "tasks_assigned_per_sec", "value": _num(value, 9), "unit": "tasks/s", "n_gpus": int(n_gpus), "steps": int(steps), "warmup": int(warmup), "ms_per_step": _num(ms_per_step),
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": workload, "priority_levels": int(priority_levels), "parallelism": parallelism, "ranks": int(n_gpus), "ranks_in_the_library_communicator": int(ranks_in_comm),
                   "seed": int(seed), "p50_tick_ms": _num(p50_tick_ms), "assigned_per_tick": int(assigned_per_tick), "every_timed_tick_done_and_certified": bool(all_done),
                   "model_columns": _num(model_columns), "price_sweeps_per_tick": _num(price_sweeps), "price_sweeps_share_of_tick": _num(gpu_busy_share, 3),
                   "caches_live_in_timed_region": caches_live, "p50_tick_ms_identical_ticks_warm_caches": _num(p50_warm_ms)},
        "roofline": None, "cpu_baseline": None,
    }
    if roofline:
        line["roofline"] = {"bound": "hbm", "kernel": roofline.get("kernel"), "achieved": _num(roofline.get("achieved")), "peak": _num(roofline.get("peak")), "unit": "GB/s",
                            "frac": _num(roofline.get("frac"), 4), "algorithmic_bytes_per_launch": _num(roofline.get("algorithmic_bytes_per_launch")),
                            "avg_launch_us": _num(roofline.get("avg_launch_us"), 4), "traffic": _num(roofline.get("traffic")),
                            "dominant_kernel": ({"kernel": dominant.get("kernel"), "avg_us": _num(dominant.get("avg_us"), 4), "launches_per_tick": _num(dominant.get("launches_per_tick")),
                                                 "share_of_tick": _num(dominant.get("share_of_tick"), 3)} if dominant else None)}
    if cpu:
        line["cpu_baseline"] = ({"value": _num(cpu.get("value")), "unit": "tasks/s", "cores": int(cpu.get("cores", 1)), "cpu": str(cpu.get("cpu", ""))[:64], "kind": cpu.get("kind", "port"),
                                 "sample": str(cpu.get("sample", ""))[:400], "tick_s": _num(cpu.get("tick_s")), "is_optimal": cpu.get("is_optimal")} if "error" not in cpu else {"error": str(cpu["error"])[:300]})
    text = json.dumps(line, allow_nan=False)
    if len(text) >= LINE_LIMIT:  # (cannot happen with the fields above; if a string grows, the strings go first, never the numbers)
        line["config"]["workload"] = line["config"]["workload"][:200]; line["config"]["parallelism"] = line["config"]["parallelism"][:100]
        if line["cpu_baseline"] and "sample" in line["cpu_baseline"]:
            line["cpu_baseline"]["sample"] = line["cpu_baseline"]["sample"][:100]
        text = json.dumps(line, allow_nan=False)
    assert len(text) < LINE_LIMIT, len(text)
    return line, text


def write_extras(path: str, extras: dict):
    """everything that is not the headline line: one JSON file next to bench.py (copied under profiles/rNN/ by the rounds' GPU scripts), never stdout"""
    try:
        tmp = path + ".tmp"
        with open(tmp, "w") as f:
            json.dump(extras, f, indent=1, default=lambda o: o.item() if hasattr(o, "item") else repr(o))
        os.replace(tmp, path)
    except OSError as e:
        print(f"bench.py: could not write {path}: {e}", file=sys.stderr)


def cpu_baseline(snap, ticks: int):
    """The CPU oracle (restatement of the reference tick + HiGHS 1.8.0 for the MILP, with the reference's solver options) on this host, 1 core, same snapshot.
    On the three-level workload one tick runs into the reference's own 5 s time limit (scheduler/state.rs: mip_time_limit) and returns an uncertified incumbent."""
    from hyperqueue_amd import abi
    from oracle.oracle import Oracle

    o = Oracle(abi.make_config(time_limit_s=5.0), reference_solver_options=True)  # HiGHS with its default options, as the reference runs it
    lat, assigned, opt = [], 0, True
    for _ in range(ticks):
        t0 = time.perf_counter()
        r = o.tick(snap)
        lat.append(time.perf_counter() - t0)
        assigned = sum(1 for recs in r.records for (_, _, k) in recs if k == abi.HQ_REC_ASSIGN)
        opt = bool(r.is_optimal)
    st = o.stage_times_us()
    med = float(np.median(lat))
    try:
        model = o.last_model()
    except Exception:  # noqa: BLE001
        model = None
    return {
        "value": assigned / med, "unit": "tasks/s", "cores": 1, "cpu": cpu_model(), "kind": "port",
        "sample": f"{ticks} cold tick(s) of the full workload ({len(snap.task_id)} tasks x {len(snap.worker_id)} workers, {len(np.unique(snap.task_priority))} priority level(s)), "
                  f"median {med:.2f} s/tick, {assigned} tasks assigned/tick" + ("" if opt else "; HiGHS stopped at the reference's 5 s time limit with an uncertified incumbent (SchedulerResult::NeedMoreCompute)"),
        "tick_s": med, "assigned_per_tick": assigned, "is_optimal": opt,
        "stages_us": {k: round(v, 1) for k, v in st.items()},
        "note": "restatement of the reference (C++ -O2) + HiGHS 1.8.0 via scipy for the MILP; not the reference binary (no Rust toolchain)",
        "_model": model,
    }


def committed_profile(kernel_substr: str, stem: str = "bench_c3p"):
    """HBM bytes per launch and kernel-trace durations of one kernel, READ at run time from the newest committed rocprofv3 summaries of the bench command
    (profiles/rNN/<stem>*.summary.csv, written by profiles/summarize.py; FETCH_SIZE x2 + WRITE_SIZE per MI355X_MICROARCH.md §HBM, each counter in its own --pmc pass).
    A PMC pass cannot run inside this process (one profiler session per run), so the line quotes the committed passes — and says which files.  None: no such files."""
    import glob

    root = os.path.join(ROOT, "profiles")
    for d in sorted(glob.glob(os.path.join(root, "r[0-9][0-9]")), reverse=True):
        files = {k: os.path.join(d, f"{stem}{suf}.summary.csv") for k, suf in (("trace", ""), ("fetch", "_FETCH_SIZE"), ("write", "_WRITE_SIZE"))}
        if not os.path.exists(files["trace"]):
            continue

        def row(path):  # (kernel names carry commas — `k_level_hist<4, true, 8>` — and are not quoted: the numeric columns are split off from the right)
            if not os.path.exists(path):
                return None
            lines = [l.rstrip("\n") for l in open(path) if not l.startswith("==")]
            hdr = lines[0].split(",")
            for l in lines[1:]:
                parts = l.rsplit(",", len(hdr) - 1)
                if len(parts) == len(hdr) and kernel_substr in parts[0]:
                    return dict(zip(hdr, parts))
            return None

        tr, fe, wr = row(files["trace"]), row(files["fetch"]), row(files["write"])
        if not tr:
            continue
        out = {"kernel_trace": {"launches": int(tr["launches"]), "avg_ns": float(tr["avg_ns"]), "min_ns": float(tr["min_ns"]), "max_ns": float(tr["max_ns"])},
               "files": [os.path.relpath(f, ROOT) for f in files.values() if os.path.exists(f)],
               "note": "per launch, from the committed rocprofv3 passes of `bench.py --headline-only` (the same command, headline loop only): every launch of this kernel in "
                       "that run carries the same dispatch events as in the timed region, so the kernel-trace average and the live figure describe one population"}
        if fe and wr:
            fetch, write = int(float(fe["FETCH_SIZE_x2_bytes"])), int(float(wr["WRITE_SIZE_bytes"]))
            out.update({"traffic_bytes_per_launch": fetch + write, "fetch_x2_bytes": fetch, "write_bytes": write})
        return out
    return None


def dag_churn(cfg, steps: int, seed: int, n_classes: int, shape: str = "random", cpu_ticks: int = 1):
    """BASELINE config 5 on one GPU: the 1 M-node DAG lives in the device dependency graph (hqtick_graph_*); per tick the tasks handed out by
    the previous tick finish (their consumers are released into the resident ready set on the device), 10 % of the workers are lost (their
    tasks return to the ready set) and replaced, then hqtick_run_resident + hqtick_ready_consume_last."""
    from hyperqueue_amd import abi, workloads
    from hyperqueue_amd.tick import Tick

    n = 1_000_000
    ids, prio, rq, off, dep = workloads.make_dag(n, seed=seed) if shape == "random" else workloads.make_dag_layered(n, width=20_000, seed=seed)
    rq = (rq % np.uint32(n_classes)).astype(np.uint32)
    in_ready = np.zeros(n, bool)  # host mirror of the resident ready set's membership (for the CPU baseline's snapshots only; ids are job 1, task 1..n)
    ix = lambda a: (np.asarray(a, np.uint64) & np.uint64(0xFFFFFFFF)).astype(np.int64) - 1
    t = Tick(cfg)
    t.upload_ready(np.zeros(0, np.uint64), np.zeros(0, np.uint64), np.zeros(0, np.uint32))
    a = time.perf_counter(); ready0 = t.graph_add_tasks(ids, prio, rq, (off, dep)); t_add = time.perf_counter() - a
    add_kernel_us = t.graph_stats()["last_kernel_us"]
    in_ready[ix(ready0)] = True
    drv = workloads.DagChurn(n_workers=1024, churn=0.10, seed=seed)
    W = 1024
    rows, base_snaps = [], {}
    snap0 = drv.snapshot()
    t.cluster_upload(snap0)  # ONCE: afterwards the worker set changes through membership deltas only (ABI 7) and the snapshots carry no worker arrays
    total_row = np.asarray(drv.kw["worker_total"], np.uint64).reshape(W, -1)[0]
    t_delta = []
    for step in range(steps + 2):
        snap_now = drv.snapshot()  # keeps the arrays `sc` points into alive
        sc = snap_now.to_c(resident_workers=True)
        if cpu_ticks > 0 and step in (0, steps + 1):  # what the CPU baseline is timed on: the first wave, and the loop's last tick
            sel = np.nonzero(in_ready)[0]
            base_snaps[step] = drv.snapshot(ids[sel], prio[sel], rq[sel])
        a = time.perf_counter(); res = t.tick_raw(sc, resident=True)
        b = time.perf_counter(); t.ready_consume_last()
        c = time.perf_counter()
        ks_now = t.kernel_stats()
        rec_off = np.ctypeslib.as_array(res.rec_off, shape=(W + 1,)).astype(np.int64)
        rec_task = abi.record_task_ids(res, W)
        finished, returned = drv.after_tick(rec_off, rec_task)
        idx = (returned & np.uint64(0xFFFFFFFF)).astype(np.int64) - 1
        in_ready[ix(rec_task)] = False; in_ready[idx] = True
        d0 = time.perf_counter()
        t.cluster_remove_workers(drv.last_lost_ids)  # on_remove_worker / on_new_worker as deltas: one re-pack kernel each, no re-upload
        t.cluster_add_workers(drv.last_fresh_ids, np.tile(total_row, (len(drv.last_fresh_ids), 1)))
        t_delta.append(time.perf_counter() - d0)
        d = time.perf_counter()
        if len(returned):
            t.ready_add(returned, prio[idx], rq[idx])
        e = time.perf_counter()
        rel, unk = t.graph_finish(finished) if len(finished) else (np.zeros(0, np.uint64), 0)
        f = time.perf_counter()
        st = t.graph_stats()
        in_ready[ix(rel)] = True
        rows.append(dict(tick=b - a, consume=c - b, readd=e - d, finish=f - e, delta=t_delta[-1], n_out=len(rec_task), n_fin=len(finished), n_ret=len(returned), n_rel=len(rel),
                         finish_kernel_us=st["last_kernel_us"], ready=int(t.ready_count()), status=int(res.status), optimal=int(res.is_optimal),
                         sweeps=ks_now["price_sweeps"], sweep_us=ks_now["price_sweep_us"], milp_us=ks_now["milp_us"], cols=ks_now["milp_cols"]))
        if len(rec_task) == 0:
            break
    # EXTENSION (hqtick_graph_blevel, include/hqtick.h): BASELINE config 5 names a "dynamic b-level recompute"; the reference has none, so this is measured
    # AFTER the loop — no tick above saw a b-level — on what the loop left of the graph: longest path to a sink for every task, into the low 32 bits of its priority.
    blevel = None
    try:
        n_left = int(t.graph_stats()["n_tasks"])
        b0 = time.perf_counter(); bi = t.graph_blevel(update_ready=True); b1 = time.perf_counter()
        gs = t.graph_stats()
        edges_left = int(gs["n_edges_live"])
        blevel = {"what": "EXTENSION, no reference counterpart, parity unpinned (checker: oracle/graph_oracle.py blevels()); not used by any tick of this loop",
                  "tasks_in_graph": n_left, "live_edges": edges_left, "sweeps": bi["sweeps"], "max_level": bi["max_level"], "ready_tasks_updated": bi["ready_updated"],
                  "call_ms": 1e3 * (b1 - b0), "kernels_us": gs["last_kernel_us"],
                  # per sweep every live slot walks its consumer list: 16 B of slot state + 8 B per edge + 4 B per consumer value read, 4 B written
                  "algorithmic_bytes": int(bi["sweeps"] * (n_left * 20 + edges_left * 12)),
                  "GBps": (bi["sweeps"] * (n_left * 20 + edges_left * 12)) / (gs["last_kernel_us"] * 1e-6) / 1e9 if gs["last_kernel_us"] > 0 else None}
    except Exception as e:  # noqa: BLE001
        blevel = {"error": repr(e)}
    st = t.graph_stats()
    t.close()
    first = rows[0]  # the first wave: every source of the DAG that the cold tick could place finishes at once
    first_edges = 3.0 * first["n_fin"]
    first_bytes = first["n_fin"] * 44 + first_edges * 16 + first["n_rel"] * 20
    use = rows[2:] if len(rows) > 4 else rows
    med = lambda k: float(np.median([r[k] for r in use]))
    step_s = np.asarray([r["tick"] + r["consume"] + r["readd"] + r["finish"] + r["delta"] for r in use])
    handed = np.asarray([r["n_out"] for r in use])
    # algorithmic bytes of the release kernel per step: per finished task id 8 + hash bucket 12 + state 4 + gen 4 + head 4 + run record 12,
    # per consumer edge 8 + gen 4 + counter 4, per released task id 8 + output pair 12
    fin, rel = med("n_fin"), med("n_rel")
    edges = 3.0 * fin  # mean fan-out = mean fan-in
    fin_bytes = fin * 44 + edges * 16 + rel * 20
    ku = med("finish_kernel_us")
    base = {}
    for step_no, bs in base_snaps.items():  # the reference-configured oracle (HiGHS, 5 s limit as in the reference) on the snapshots the GPU ticked
        try:
            from oracle.oracle import Oracle

            o = Oracle(abi.make_config(time_limit_s=5.0), reference_solver_options=True)
            t0 = time.perf_counter(); r = o.tick(bs); dt = time.perf_counter() - t0
            n_asg = sum(1 for recs in r.records for (_, _, k) in recs if k == abi.HQ_REC_ASSIGN)
            base["first_wave" if step_no == 0 else "last_tick"] = {"value": n_asg / dt, "unit": "tasks/s", "cores": 1, "cpu": cpu_model(), "kind": "port", "tick_s": dt, "is_optimal": bool(r.is_optimal),
                                                                   "assigned_per_tick": n_asg, "sample": f"1 tick of the snapshot the GPU ticked ({len(bs.task_id)} ready tasks x 1024 workers)",
                                                                   "gpu_tick_s": rows[step_no]["tick"] if step_no < len(rows) else None}
        except Exception as e:
            base["first_wave" if step_no == 0 else "last_tick"] = {"error": repr(e)}
    shape_txt = "random DAG (fan-in ~ Poisson(3) from lower ids" if shape == "random" else "layered DAG (layers of 20 000 tasks, fan-in 3 from the previous layer"
    return {
        "workload": f"c5: {n}-node {shape_txt}, {len(dep)} edges) over the first {n_classes} c3 classes, 1024 workers, 10 % of the workers lost and replaced per tick",
        "note": "the frontier stays below the cluster's capacity, so no batch is saturated and the placement model couples all workers through the batch-size rows: every tick with "
                "at least ~1000 model columns goes through k_price_sweep (DESIGN.md §4b); smaller ones (a few dozen ready tasks) stay with the host search",
        "cpu_baseline": base,
        "worker_churn": {"how": "hqtick_cluster_remove_workers + hqtick_cluster_add_workers per tick (ABI 7): the lost rows leave and the fresh ones join the HBM-resident tables through one re-pack "
                                "kernel each; the snapshots of the loop carry no worker arrays (worker_id == NULL)", "workers_replaced_per_tick": int(drv.n_lost),
                         "p50_membership_deltas_us": 1e6 * float(np.median(t_delta)) if t_delta else None},
        "p50_price_sweeps_per_tick": int(med("sweeps")), "p50_model_columns": int(med("cols")), "p50_coupled_solve_us": med("milp_us"), "p50_sweeps_us": med("sweep_us"),
        "graph_add_ms": 1e3 * t_add, "graph_add_link_kernels_us": add_kernel_us, "initially_ready": int(len(ready0)),
        "graph_bytes_hbm": int(st["bytes_hbm"]),
        "steps": len(use), "p50_step_ms": 1e3 * float(np.median(step_s)), "tasks_handed_out_per_step": int(np.median(handed)),
        "tasks_per_s": float(handed.sum() / step_s.sum()),
        "p50_tick_us": 1e6 * med("tick"), "p50_consume_us": 1e6 * med("consume"), "p50_return_lost_workers_tasks_us": 1e6 * med("readd"),
        "p50_graph_finish_us": 1e6 * med("finish"), "p50_finished_per_step": int(fin), "p50_released_per_step": int(rel), "p50_returned_per_step": int(med("n_ret")),
        "finish_kernel": {"avg_us": ku, "algorithmic_bytes": int(fin_bytes), "GBps": fin_bytes / (ku * 1e-6) / 1e9 if ku > 0 else None,
                          "note": "dependent random 4-16 B accesses (hash probe -> slot -> run -> edge -> counter RMW): latency-bound, not a streaming kernel"},
        "first_wave": {"finished": first["n_fin"], "released": first["n_rel"], "graph_finish_call_us": 1e6 * first["finish"], "finish_kernel_us": first["finish_kernel_us"],
                       "algorithmic_bytes": int(first_bytes), "GBps": first_bytes / (first["finish_kernel_us"] * 1e-6) / 1e9 if first["finish_kernel_us"] > 0 else None,
                       "tick_us": 1e6 * first["tick"], "handed_out": first["n_out"], "price_sweeps": int(first["sweeps"]), "model_columns": int(first["cols"]), "is_optimal": bool(first["optimal"])},
        "ready_set_p50": int(med("ready")), "all_ticks_optimal": bool(all(r["optimal"] for r in use)),
        "blevel_recompute": blevel,
    }


def steady_hetero(cfg, snap, steps: int, seed: int, cpu_ticks: int, release: float = 0.10):
    """SURVEY.md §8(d) steady state on one GPU: after the cold tick, every running task finishes with probability `release` before the next tick
    ("free a random 10 % of assigned tasks per tick and re-run"), so the workers' free vectors all differ (about one worker class per worker) and
    the placement is ~1000 different bounded knapsacks per tick — solved by k_block_solve, one workgroup of four wavefronts per class.  The ready set stays
    resident and saturated (arrivals replace what was handed out, class by class).  Prefilled tasks leave the queue but are not tracked as
    running (the sleep-0 model of benchmarks/experiment-per-task-overhead.py: they are done before the next tick)."""
    import dataclasses

    from hyperqueue_amd import abi
    from hyperqueue_amd.tick import Tick

    W, R, Q = len(snap.worker_id), snap.n_resources, len(snap.requests)
    need = np.zeros((Q, R), np.int64)
    for q, variants in enumerate(snap.requests):
        for (r, _k, a) in variants[0]["entries"]:
            need[q, r] = int(a)
    total = np.asarray(snap.worker_total, np.int64).reshape(W, R)
    running = np.zeros((W, Q), np.int64)
    rng = np.random.default_rng(seed)
    ts = Tick(cfg)
    ts.upload_ready(snap.task_id, snap.task_priority, snap.task_rq, sorted_=True)
    rq_of = snap.task_rq.copy()
    alive = np.ones(len(rq_of), bool)
    next_id = int(snap.task_id[-1]) + 1
    n_staged = 0
    rows, last_snap, prev_free, n_changed = [], None, None, []
    # the loop's times come from steps without per-kernel timing events (as the headline's: ~6 runtime calls per tick less); three more steps with the events on give the
    # kernels' durations
    timing_was_on = not (cfg.flags & abi.HQTICK_FLAG_NO_KERNEL_TIMING)
    ts.set_kernel_timing(False)
    for step in range(steps + 3 + (3 if timing_was_on else 0)):
        if step == steps + 3:
            ts.set_kernel_timing(True)
        free = total - running @ need
        assert (free >= 0).all()
        assigned = [[(int(q), 0) for q in np.repeat(np.arange(Q), running[w])] for w in range(W)]
        cur = dataclasses.replace(snap, _keep=[], worker_free=free.astype(np.uint64), assigned=assigned, task_id=np.zeros(0, np.uint64), task_priority=np.zeros(0, np.uint64), task_rq=np.zeros(0, np.uint32))
        sc = cur.to_c()
        a = time.perf_counter()
        if n_staged:
            ts.ready_add_staged(n_staged)  # the arrivals were written straight into the library's pinned staging buffer (hqtick_ready_add_stage): no copy on the host
        if prev_free is None:
            ts.cluster_upload(sc)
        else:  # the rows the reactor's handlers touched since the last tick (tasks started by it, tasks finished since): deltas into the HBM tables
            changed = np.nonzero((free != prev_free).any(axis=1))[0].astype(np.uint32)
            ts.cluster_update_workers(changed, free[changed].astype(np.uint64))
            n_changed.append(len(changed))
        prev_free = free.copy()
        b = time.perf_counter()
        res = ts.tick_raw(sc, resident=True)
        c = time.perf_counter()
        ts.ready_consume_last()
        d = time.perf_counter()
        ks = ts.kernel_stats()
        n_cnt = int(res.n_counts)
        cw = np.ctypeslib.as_array(res.count_worker, shape=(n_cnt,)).astype(np.int64) if n_cnt else np.zeros(0, np.int64)
        cq = np.ctypeslib.as_array(res.count_rq, shape=(n_cnt,)).astype(np.int64) if n_cnt else np.zeros(0, np.int64)
        cv = np.ctypeslib.as_array(res.count_value, shape=(n_cnt,)).astype(np.int64) if n_cnt else np.zeros(0, np.int64)
        last_snap = (free.copy(), [list(x) for x in assigned], alive.copy(), rq_of.copy(), (cw.copy(), cq.copy(), cv.copy()))
        np.add.at(running, (cw, cq), cv)
        n_rec = int(np.ctypeslib.as_array(res.rec_off, shape=(W + 1,))[W])
        gone = abi.record_task_ids(res, W)
        idx = (gone & np.uint64(0xFFFFFFFF)).astype(np.int64) - 1
        alive[idx] = False
        n_staged = len(idx)
        v_id, v_prio, v_rq = ts.ready_add_stage(n_staged)  # the reactor writes the new ready tasks where the merge kernel's upload starts from
        new_rq = rq_of[idx]
        v_rq[:] = new_rq
        rq_of = np.concatenate([rq_of, new_rq]); alive = np.concatenate([alive, np.ones(len(idx), bool)])
        v_id[:] = np.arange(next_id, next_id + len(idx), dtype=np.uint64); next_id += len(idx)
        v_prio[:] = snap.task_priority[0]
        rows.append(dict(add=b - a, tick=c - b, consume=d - c, assigned=int(cv.sum()), handed=n_rec, status=int(res.status), optimal=int(res.is_optimal), canonical=int(res.is_canonical),
                         t_scan=res.t_scan_us, t_batches=res.t_batches_us, t_solve=res.t_solve_us, t_map=res.t_mapping_us, **{k: ks[k] for k in
                         ("n_classes", "n_classes_device", "n_classes_host", "block_solve_us", "block_steps_max", "solve_classify_us", "solve_blocks_us", "solve_decode_us", "level_hist_us", "select_us", "other_us")}))
        running -= rng.binomial(running, release)
    ts.close()
    use, kuse = rows[3:steps + 3], (rows[steps + 3:] or rows[3:])
    med = lambda k: float(np.median([r[k] for r in (kuse if k in ("block_solve_us", "level_hist_us", "select_us", "other_us") else use)]))
    step_s = np.asarray([r["add"] + r["tick"] + r["consume"] for r in use])
    out = {
        "workload": f"c3 steady state: {len(snap.task_id)} ready tasks (resident, refilled), {W} workers each running a packed mix of which a random {int(release * 100)} % finishes per tick",
        "steps": len(use), "p50_step_ms": 1e3 * float(np.median(step_s)), "p50_tick_ms": 1e3 * med("tick"), "p95_tick_ms": 1e3 * float(np.percentile([r["tick"] for r in use], 95)),
        "p50_add_us": 1e6 * med("add"), "p50_consume_us": 1e6 * med("consume"), "worker_rows_sent_per_tick": int(np.median(n_changed)) if n_changed else 0,
        "assigned_per_tick": int(med("assigned")), "handed_out_per_tick": int(med("handed")), "tasks_assigned_per_sec": med("assigned") / float(np.median(step_s)),
        "worker_classes_per_tick": int(med("n_classes")), "classes_solved_on_device": int(med("n_classes_device")), "classes_solved_on_host": int(med("n_classes_host")),
        "all_ticks_optimal_and_canonical": bool(all(r["optimal"] and r["canonical"] for r in use)),
        "block_solve_kernel": {"avg_us": med("block_solve_us"), "classes_per_launch": int(med("n_classes_device")), "classes_per_s": med("n_classes_device") / (med("block_solve_us") * 1e-6) if med("block_solve_us") > 0 else None,
                               "max_search_steps": int(max(r["block_steps_max"] for r in use)),
                               "bound": "latency / integer-f64 ALU in LDS: a workgroup of four wavefronts per class (one runs the chain, the others share its dual pool and run its greedy fills), 36.9 KB of LDS per block = four blocks resident per CU; not an HBM-bound kernel (a class reads ~60 B)"},
        "tick_stages_us": {"gpu_phase_a_scans": med("t_scan"), "batches": med("t_batches"), "placement": med("t_solve"), "placement_worker_classes": med("solve_classify_us"),
                           "placement_block_solves_incl_launch_and_wait": med("solve_blocks_us"), "placement_counts_in_map_order": med("solve_decode_us"), "mapping_plan_gpu_phase_c": med("t_map")},
    }
    if cpu_ticks > 0 and last_snap is not None:
        try:
            from oracle.oracle import Oracle

            free, assigned, alive_m, rq_all, gpu_counts = last_snap
            ids_all = (np.uint64(1) << np.uint64(32)) | np.arange(1, len(rq_all) + 1, dtype=np.uint64)
            keep = np.nonzero(alive_m)[0]
            full = dataclasses.replace(snap, _keep=[], worker_free=free.astype(np.uint64), assigned=assigned, task_id=ids_all[keep], task_priority=np.full(len(keep), snap.task_priority[0], np.uint64), task_rq=rq_all[keep])
            o = Oracle(abi.make_config(time_limit_s=5.0), reference_solver_options=True)
            lat, n_asg, opt = [], 0, True
            for _ in range(cpu_ticks):
                t0 = time.perf_counter(); r = o.tick(full); lat.append(time.perf_counter() - t0)
                n_asg = sum(1 for recs in r.records for (_, _, k) in recs if k == abi.HQ_REC_ASSIGN); opt = bool(r.is_optimal)
            tick_s = float(np.median(lat))
            # both sides maximise the same objective (scheduler/solver.rs:542-571): c.x of the GPU tick's counts in the oracle's own model of the snapshot
            mdl = o.last_model()
            gd = {(int(q), int(w)): int(v) for w, q, v in zip(*gpu_counts)}
            xg = np.asarray([gd.get((int(mdl["crq"][j]), int(mdl["cworker"][j])), 0) if mdl["ctype"][j] == 0 else 0 for j in range(len(mdl["obj"]))], np.float64)
            out["objective"] = {"gpu_tick": float(np.dot(mdl["obj"], xg)), "cpu_baseline": float(mdl["objective"]),
                                "note": "same snapshot; equal objective = both optimal (the reference's answer is HiGHS's optimum; where optima tie, which one it returns is an artefact of HiGHS — DESIGN.md §4)"}
            out["cpu_baseline"] = {"value": n_asg / tick_s, "unit": "tasks/s", "cores": 1, "cpu": cpu_model(), "kind": "port", "tick_s": tick_s, "assigned_per_tick": n_asg, "is_optimal": opt,
                                   "sample": f"{cpu_ticks} tick(s) of the snapshot the last timed GPU tick saw ({len(keep)} ready tasks x {W} heterogeneous workers)",
                                   "stages_us": {k: round(v, 1) for k, v in o.stage_times_us().items()},
                                   "note": "restatement of the reference (C++ -O2) + HiGHS 1.8.0 with default options on the un-reduced model (8 k columns); both sides optimal, tied optima may differ"}
            out["speedup_vs_cpu_baseline"] = out["tasks_assigned_per_sec"] / out["cpu_baseline"]["value"] if out["cpu_baseline"]["value"] > 0 else None
            out["tick_latency_ratio_vs_cpu"] = tick_s / float(np.median([r["tick"] for r in use]))
        except Exception as e:
            out["cpu_baseline"] = {"error": repr(e)}
    return out


def wire_block(iters: int):
    """Row f3: the wire encoding of a C3-shaped mapping, measured in a SUBPROCESS (tools/wire_bench.py) -- these kernels
    had not run on hardware when round 1 ended, and whatever happens there must not cost the headline line."""
    import subprocess

    try:
        p = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "wire_bench.py"), "--iters", str(iters)],
                           capture_output=True, text=True, timeout=180)
        lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
        if p.returncode == 0 and lines:
            return json.loads(lines[-1])
        return {"error": f"exit {p.returncode}", "stderr_tail": p.stderr[-400:]}
    except Exception as e:
        return {"error": repr(e)}


def multi_rank_extras(cfg, rank: int, world: int, local_rank: int, ticks: int = 5):
    """N > 1 only, EVERY rank calls it (collectives inside): what the ranks add beyond the expansion of the records.
      config4_unsaturated  BASELINE configs[3]'s cluster (4096 workers x 2-variant requests) with a ready set that saturates nothing: ONE coupled model of 65 536 columns, 4096
                           worker blocks per price sweep = four rounds of resident workgroups on one MI355X.  Sharded solve (include/hqtick.h, hqtick_set_exchange / the
                           library's RCCL communicator): every rank sweeps its 4096 / N blocks, one small all-gather per sweep — against the same tick with the solve replicated
                           on every rank (HQTICK_SHARD_SOLVE=0).  Equal counts on both (checked here through the placement checksum of the sink headers).
      config4_strong       configs[3] as written: c4, 1 M ready tasks x 4096 workers hash-sharded over the ranks (strong scaling; the placement is one class block)."""
    import torch

    from hyperqueue_amd import abi, workloads
    from hyperqueue_amd.sharded import ShardedTick

    out = {}

    def run(snap, shard_solve: bool, n_ticks: int):
        old = os.environ.get("HQTICK_SHARD_SOLVE")
        os.environ["HQTICK_SHARD_SOLVE"] = "1" if shard_solve else "0"  # (read once, in hqtick_create)
        try:
            st = ShardedTick(cfg, rank=rank, world=world, records_per_shard=int(1.2 * 300 * len(snap.worker_id) / world) + 8192)
        finally:
            if old is None:
                os.environ.pop("HQTICK_SHARD_SOLVE", None)
            else:
                os.environ["HQTICK_SHARD_SOLVE"] = old
        st.upload_ready(snap.task_id, snap.task_priority, snap.task_rq)
        sc = snap.to_c()
        st.t.cluster_upload(sc)
        lat, ks, chk = [], None, None
        for _ in range(n_ticks + 1):
            torch.distributed.barrier(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            res, merged = st.tick_device(sc, len(snap.worker_id), resident=True)
            torch.cuda.synchronize()
            lat.append(time.perf_counter() - t0)
            ks = st.t.kernel_stats()
            chk = (int(res.status), int(res.is_optimal), int(merged[4:8].cpu().numpy().view(np.uint32)[0]))  # the placement checksum of this rank's sink header
        t = torch.tensor([float(np.median(lat[1:]))], dtype=torch.float64, device=f"cuda:{local_rank}")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        st.t.close()
        return float(t.item()), ks, chk, st.collective

    try:
        s4 = workloads.make("c4", seed=8, n_workers=4096, n_tasks=56_761)
        a_s, a_ks, a_chk, coll = run(s4, True, ticks)
        b_s, b_ks, b_chk, _ = run(s4, False, max(2, ticks // 2))
        out["config4_unsaturated"] = {
            "workload": "c4's cluster (4096 workers x 2-variant requests) with 56 761 ready tasks: one coupled model of all workers", "merge_collective": coll,
            "sharded_solve": {"p50_tick_ms": 1e3 * a_s, "status": a_chk[0], "is_optimal": bool(a_chk[1]), "price_sweeps": int(a_ks["price_sweeps"]), "blocks_per_sweep_and_rank": 4096 // world,
                              "avg_sweep_us_incl_exchange": (a_ks["price_sweep_us"] / a_ks["price_sweeps"]) if a_ks["price_sweeps"] else None,
                              "exchange_calls": int(a_ks["exchange_calls"]), "exchange_us_per_call": (a_ks["exchange_us"] / a_ks["exchange_calls"]) if a_ks["exchange_calls"] else None,
                              "exchange_bytes_per_call": (a_ks["exchange_bytes"] / a_ks["exchange_calls"]) if a_ks["exchange_calls"] else None, "coupled_solve_ms": a_ks["milp_us"] / 1e3},
            "replicated_solve": {"p50_tick_ms": 1e3 * b_s, "status": b_chk[0], "is_optimal": bool(b_chk[1]), "price_sweeps": int(b_ks["price_sweeps"]),
                                 "avg_sweep_us": (b_ks["price_sweep_us"] / b_ks["price_sweeps"]) if b_ks["price_sweeps"] else None, "coupled_solve_ms": b_ks["milp_us"] / 1e3},
            "same_placement": bool(a_chk == b_chk),
        }
    except Exception as e:  # noqa: BLE001
        out["config4_unsaturated"] = {"error": repr(e)}
    try:
        sf = workloads.make("c4", seed=0)
        c_s, c_ks, c_chk, coll = run(sf, True, ticks)
        out["config4_strong"] = {"workload": "c4: 1 M ready tasks x 4096 workers (BASELINE configs[3] as written), hash-sharded over the ranks", "scaling": "strong", "merge_collective": coll,
                                 "p50_tick_ms": 1e3 * c_s, "assigned_per_tick": int(c_ks["n_assigned"]), "tasks_assigned_per_sec": int(c_ks["n_assigned"]) / c_s if c_s > 0 else None,
                                 "note": "timed: sharded tick + the all-gather of the record sinks (no D2H of the merged vector in this block)"}
    except Exception as e:  # noqa: BLE001
        out["config4_strong"] = {"error": repr(e)}
    return out


def watchdog(seconds: float, last_words):
    """a timer that ends the process if what follows does not come back: the extras of a multi-rank run must never cost the line of the timed region"""
    import threading

    def fire():
        try:
            last_words()
        finally:
            sys.stdout.flush()
            os._exit(0)

    t = threading.Timer(seconds, fire)
    t.daemon = True
    t.start()
    return t


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: re-run this file under torch.distributed.run, N ranks on this node, rendezvous on 127.0.0.1 (a free port)."""
    import socket
    import subprocess

    import torch

    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < n:
        print(f"bench.py: --gpus {n} but this node shows {have} GPU(s)", file=sys.stderr)
        return 2
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def one_level_cold_tick(cfg, args, rec_bytes: int):
    """The one-level variant c3 (every class saturated: the placement separates per worker into one 8-column block, solved on the host in ~12 us): round 1-4's headline,
    now a neighbour of the three-level tick.  Cold tick repeated on the resident set; per-kernel durations from a second pass with dispatch events on."""
    from hyperqueue_amd import abi, workloads
    from hyperqueue_amd.tick import Tick

    snap = workloads.make("c3", seed=args.seed)
    sc = snap.to_c()
    t = Tick(cfg)
    t.upload_ready(snap.task_id, snap.task_priority, snap.task_rq, sorted_=True)
    if not args.no_resident_cluster:
        t.cluster_upload(sc)
    t.set_kernel_timing(False)
    for _ in range(5):
        t.tick_raw(sc, resident=True)
    lat, stages = [], []
    for _ in range(50):
        t0 = time.perf_counter(); res = t.tick_raw(sc, resident=True); lat.append(time.perf_counter() - t0)
        stages.append((res.t_scan_us, res.t_batches_us, res.t_solve_us, res.t_mapping_us, res.t_total_us))
    t.set_kernel_timing(True)
    kstats = []
    for _ in range(20):
        t.tick_raw(sc, resident=True); kstats.append(t.kernel_stats())
    ks = kstats[-1]
    t.close()
    n_ready, assigned, prefilled = len(snap.task_id), int(ks["n_assigned"]), int(ks["n_prefilled"])
    sel = assigned + prefilled
    mean = lambda k: float(np.mean([x[k] for x in kstats]))
    G = len(snap.requests)
    kernels = {
        "level_hist": dict(us=mean("level_hist_us"), bytes=n_ready * 12, bound="hbm", what="K1: priority u64 + rq u32 of every ready task"),
        "scan_waves": dict(us=mean("scan_us"), bytes=G * ((n_ready + 255) // 256) * 8, bound="latency", what="K1b: per-slice counts -> offsets"),
        "select_scatter": dict(us=mean("select_us"), bytes=n_ready * 2 + sel * 18, bound="hbm", what="K4: group key u16 of every ready task + id u64 of the slices that still feed a group + (id, key) of the taken ones"),
        "sweep_bits": dict(us=mean("sweep_us"), bytes=0, bound="latency", what="K5a: round-robin bit rows"),
        "expand_mapping": dict(us=mean("other_us"), bytes=sel * 10 + sel * rec_bytes, bound="pcie", what="K5b: gathers (id, level) and writes the records straight into pinned host memory"),
    }
    for k in kernels.values():
        k["GBps"] = k["bytes"] / (k["us"] * 1e-6) / 1e9 if k["us"] > 0 else 0.0
        k["us"] = round(k["us"], 2)
    em = kernels["expand_mapping"]
    pcie_bytes, pcie_peak = sel * rec_bytes, 63.0
    med = float(np.median(lat))
    return {
        "workload": f"c3: {n_ready} ready tasks x {len(snap.worker_id)} workers, 8 request classes, ONE priority level, cold tick repeated on the resident ready set (best case: one host-solved class block)",
        "tasks_assigned_per_sec": assigned / med, "p50_tick_ms": 1e3 * med, "p95_tick_ms": 1e3 * float(np.percentile(lat, 95)), "assigned_per_tick": assigned, "prefilled_per_tick": prefilled,
        "tick_algorithmic_bytes": int(ks["algorithmic_bytes"]), "tick_bytes_per_s_end_to_end_GBps": ks["algorithmic_bytes"] / med / 1e9,
        "kernels": kernels,
        "tick_stages_us": dict(zip(["gpu_phase_a_scans", "batches", "solve", "mapping_plan_gpu_phase_c", "total_in_library"], [round(float(x), 1) for x in np.median(np.asarray(stages), axis=0)])),
        "time_dominant_kernel": {"kernel": "expand_mapping", "bound": "pcie", "achieved": pcie_bytes / (em["us"] * 1e-6) / 1e9 if em["us"] > 0 else 0.0, "peak": pcie_peak, "unit": "GB/s",
                                 "frac": (pcie_bytes / (em["us"] * 1e-6) / 1e9 / pcie_peak) if em["us"] > 0 else 0.0, "bytes_over_pcie_per_launch": pcie_bytes, "avg_launch_us": em["us"], "bytes_per_record": rec_bytes},
    }, snap


def preflight(rank: int, world: int, local_rank: int) -> int:
    """`--preflight`: does the library's own RCCL communicator come up on this node?  Builds it (hqtick_comm_unique_id on rank 0, the 128-byte id over torch.distributed,
    hqtick_comm_init everywhere), runs ONE 4 KB all-gather through hqtick_shard_allgather and checks every rank's block — nothing else.  Rank 0 prints one JSON line."""
    import torch

    from hyperqueue_amd import abi
    from hyperqueue_amd.sharded import ShardedTick, sink_layout

    out = {"preflight": "hqtick_comm_init + one all-gather of 4 KB per rank", "ranks": world}
    wd = watchdog(120.0, lambda: (rank == 0) and print(json.dumps(dict(out, error="did not come back within 120 s"))))
    try:
        st = ShardedTick(abi.make_config(device_index=local_rank), rank=rank, world=world, records_per_shard=256, collective="library")  # (the library's own communicator, also with one rank)
        out["collective"] = st.collective; out["ranks_in_the_library_communicator"] = int(st.comm_world)
        dev = torch.device("cuda", local_rank)
        total = 4096
        sink = torch.full((total,), rank + 1, dtype=torch.uint8, device=dev)
        merged = torch.zeros(total * world, dtype=torch.uint8, device=dev)
        ok = False
        if st.collective == "library":
            lib = st.t._lib
            rc = lib.hqtick_set_record_sink(st.t._ctx, C.c_void_p(sink.data_ptr()), C.c_size_t(total))
            t0 = time.perf_counter()
            rc = rc or lib.hqtick_shard_allgather(st.t._ctx, C.c_void_p(merged.data_ptr()), C.c_size_t(merged.numel()))
            torch.cuda.synchronize()
            out["allgather_ms_first_call"] = 1e3 * (time.perf_counter() - t0)
            t0 = time.perf_counter()
            rc = rc or lib.hqtick_shard_allgather(st.t._ctx, C.c_void_p(merged.data_ptr()), C.c_size_t(merged.numel()))
            torch.cuda.synchronize()
            out["allgather_ms_second_call"] = 1e3 * (time.perf_counter() - t0)
            got = merged.cpu().numpy().reshape(world, total)
            ok = rc == 0 and all((got[r] == r + 1).all() for r in range(world))
            out["rc"] = int(rc)
            if rc:
                out["error"] = st.t._err()
        out["ranks_seen"] = int(sum(1 for r in range(world) if (merged.cpu().numpy().reshape(world, total)[r] == r + 1).all()))
        out["ok"] = bool(ok)
        st.t.close()
    except Exception as e:  # noqa: BLE001
        out["error"] = repr(e); out["ok"] = False
    wd.cancel()
    flag = torch.tensor([1 if out.get("ok") else 0], dtype=torch.int32, device=f"cuda:{local_rank}")
    if world > 1:
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
    out["ok_on_every_rank"] = bool(int(flag.item()))
    if rank == 0:
        print(json.dumps(out)); sys.stdout.flush()
    if world > 1:
        torch.distributed.destroy_process_group()
    return 0 if out["ok_on_every_rank"] else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default=None, help="c3p (default: BASELINE configs[2] as SURVEY 8d writes it, three priority levels) / c3 (one level) / c2 / c4; on N > 1 GPUs per rank "
                                                   "(weak scaling: 1024 workers and 1 M ready tasks per GPU, hash-sharded + one RCCL all-gather); c4 = BASELINE configs[3] as written (strong scaling)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None, help="N > 1: weak = the workload grows with N (default for c2 / c3 / c3p), strong = the configuration as written (default for c4 = BASELINE configs[3])")
    ap.add_argument("--cpu-ticks", type=int, default=1, help="ticks of the CPU baseline (0 = skip): one tick of the reference-configured HiGHS on c3p runs into its 5 s limit")
    ap.add_argument("--headline-only", action="store_true", help="the timed region and its line only: no neighbours, loops, extras, CPU baseline (what profiles/rNN/bench_c3p*.summary.csv are taken from)")
    ap.add_argument("--no-resident-cluster", action="store_true", help="pack the worker tables per tick instead of keeping them in HBM (hqtick_cluster_*)")
    ap.add_argument("--priority-ticks", type=int, default=5, help="ticks of the other coupled workloads (busy cluster with three levels, configs[3] unsaturated), 0 = skip")
    ap.add_argument("--steady-steps", type=int, default=20, help="steps of the steady-state (delta-updated resident set) measurement, 0 = skip")
    ap.add_argument("--hetero-steps", type=int, default=25, help="ticks of the heterogeneous-worker steady state (SURVEY 8d: 10 %% of the running tasks finish per tick), 0 = skip")
    ap.add_argument("--dag-steps", type=int, default=12, help="ticks of the config-5 loop (1 M-node DAG in the device graph + 10 %% worker churn per tick), 0 = skip")
    ap.add_argument("--dag-classes", type=int, default=8, help="request classes of the config-5 DAG (first N of the c3 classes; 8 = all, as BASELINE config 5 names them)")
    ap.add_argument("--wire-iters", type=int, default=50, help="launch triples of the wire-encoding measurement (row f3, in a subprocess), 0 = skip")
    ap.add_argument("--full-records", action="store_true", help="10-byte records (u64 id, variant, kind) instead of the compact emission (HQTICK_FLAG_COMPACT_RECORDS)")
    ap.add_argument("--u32-records", action="store_true", help="compact emission with 4-byte low halves (ABI 4/5) instead of the 16-bit differences of ABI 6 (HQTICK_FLAG_COMPACT_DELTA16)")
    ap.add_argument("--no-roofline-sweep", dest="roofline_sweep", action="store_false", help="skip the K1 bandwidth measurement on 4 M / 16 M task ready sets")
    ap.add_argument("--no-multi-extras", dest="multi_extras", action="store_false", help="N > 1: skip the blocks after the timed region (configs[3]'s coupled tick with the solve split over the ranks vs replicated; c4 strong scaling)")
    ap.add_argument("--extras-timeout", type=float, default=240.0, help="N > 1: seconds the extra blocks may take before the line is printed without them")
    ap.add_argument("--run-timeout", type=float, default=900.0, help="N > 1: seconds the whole run may take before rank 0 prints what it has and every rank ends (a lost rank must not hang the node)")
    ap.add_argument("--preflight", action="store_true", help="N > 1: only build the library's RCCL communicator (hqtick_comm_init), run one 4 KB all-gather through it and print the ranks seen")
    ap.add_argument("--plain-adds", action="store_true", help="steady-state loop: new tasks as three full columns (hqtick_ready_add_staged, 20 B per task) instead of the packed form")
    ap.add_argument("--two-call-consume", action="store_true", help="the loops: hqtick_run_resident + hqtick_ready_consume_last as two calls (up to round 4) instead of HQTICK_FLAG_CONSUME_IN_TICK")
    ap.add_argument("--force-sharded", action="store_true", help="use the sharded code path (device record sink + merge + D2H) even with one rank")
    ap.add_argument("--warm-caches", action="store_true", help="the timed region with the product's default cross-tick caches live (level table, Map-order memo): NOT the headline")
    ap.add_argument("--extras-file", default=os.path.join(ROOT, "bench_extras.json"), help="where everything that is not the headline line goes (neighbour workloads, loops, DAG, wire, notes)")
    ap.add_argument("--no-kernel-timing", action="store_true", help="HQTICK_FLAG_NO_KERNEL_TIMING: no HIP events inside the tick at all (the roofline object is then empty)")
    args = ap.parse_args()
    if args.headline_only:
        args.cpu_ticks = args.priority_ticks = args.steady_steps = args.hetero_steps = args.dag_steps = args.wire_iters = 0
        args.roofline_sweep = False

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: launch the N ranks ourselves (one process per GPU, the same command line the driver uses) and hand their
        # line through.  With WORLD_SIZE in the environment this process IS one of the ranks (torch.distributed.run started it).
        raise SystemExit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if rank == 0 and world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: running (and reporting n_gpus =) {world} rank(s)", file=sys.stderr)
    if args.workload is None:
        # one curve for N = 1, 2, 4, 8: the headline workload (BASELINE configs[2] with its three priority levels) per GPU — weak scaling, 1024 workers and 1 M ready
        # tasks per rank.  configs[3] as written (c4: 4096 workers hash-sharded over the ranks, strong scaling) is `--workload c4`, and rides along in the N > 1 line.
        args.workload = "c3p"
    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the tick has no CPU path (libhqtick.so fails with HQTICK_E_NO_DEVICE)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import __graft_entry__ as ge

    if rank == 0:
        ge.build()
    if dist is not None:
        dist.barrier()
    from hyperqueue_amd import abi, workloads
    from hyperqueue_amd.tick import Tick

    if args.preflight:
        raise SystemExit(preflight(rank, world, local_rank))
    partial = {"metric": "tasks_assigned_per_sec", "value": None, "unit": "tasks/s", "n_gpus": world, "error": "run timed out before the timed region finished"}
    run_wd = watchdog(args.run_timeout, lambda: (rank == 0) and print(json.dumps(partial))) if world > 1 else None

    # N = 1: the plain tick.  N > 1: ONE scheduler whose workers are hash-sharded over the ranks (hyperqueue_amd/sharded.py): every rank holds the same snapshot (ready set
    # replicated in its HBM), sweeps its own worker range of the coupled model (one small all-gather per sweep), expands the records of its own workers, and one
    # RCCL all-gather merges the shards' assignment vectors; rank 0 then pulls the merged vector to the host.  Weak scaling: 1024 workers and 1 M ready tasks per GPU.
    n_workers_per_gpu, n_tasks_per_gpu = {"c2": (256, 100_000), "c3": (1024, 1_000_000), "c3p": (1024, 1_000_000), "c4": (4096, 1_000_000)}.get(args.workload, (1024, 1_000_000))
    # --scaling strong: the configuration as BASELINE.json writes it, whatever N is (configs[3]: c4 = 4096 workers, 1 M tasks, hash-sharded over the GPUs)
    scaling = args.scaling or ("strong" if args.workload == "c4" else "weak")
    mult = world if scaling == "weak" else 1
    snap = workloads.make(args.workload, seed=args.seed, n_tasks=n_tasks_per_gpu * mult, n_workers=n_workers_per_gpu * mult)
    cfg = abi.make_config(time_limit_s=5.0, device_index=local_rank)
    if args.no_kernel_timing:
        cfg.flags |= abi.HQTICK_FLAG_NO_KERNEL_TIMING
    if not args.full_records:
        cfg.flags |= abi.HQTICK_FLAG_COMPACT_RECORDS  # records cross PCIe as u32 low halves + runs of (job, variant, kind): include/hqtick.h
        if not args.u32_records:
            cfg.flags |= abi.HQTICK_FLAG_COMPACT_DELTA16  # ... as 16-bit differences of the low halves (ABI 6): 2 bytes per record
    # The product keeps the answers of its last host-solved class blocks (hqtick.h: HQTICK_FLAG_NO_BLOCK_MEMO).  The headline repeats ONE tick: with the table on, a block
    # solve would be a lookup — so the headline context switches it off (nothing cached inside the timed region); the loops below, whose ticks see a changing ready set,
    # run the product's default and say how often the table answered.
    loop_cfg = type(cfg).from_buffer_copy(cfg)
    if not args.two_call_consume:
        loop_cfg.flags |= abi.HQTICK_FLAG_CONSUME_IN_TICK  # the loops' ticks take what they hand out themselves, as take_tasks does inside the reference's tick (hqtick_ready_consume_last: a no-op)
    cfg.flags |= abi.HQTICK_FLAG_NO_BLOCK_MEMO
    # ... and, since the timed loop repeats ONE identical tick, nothing else derived may survive from tick to tick either: HQTICK_FLAG_NO_TICK_CACHES
    # makes every tick rediscover the priority-level table of the resident ready set and recompute every hashbrown iteration order (the two memos an identical tick
    # would otherwise hit).  `value` is that loop; the same loop with the product's default caches rides along as config.p50_tick_ms_identical_ticks_warm_caches.
    warm_cfg = type(cfg).from_buffer_copy(cfg)
    if not args.warm_caches:
        cfg.flags |= abi.HQTICK_FLAG_NO_TICK_CACHES
    rec_bytes = 10 if args.full_records else (4 if args.u32_records else 2)  # what one record costs on PCIe (runs and spans on top in the compact forms)
    sc = snap.to_c()
    W_all = len(snap.worker_id)
    st = None
    if world == 1 and not args.force_sharded:
        tick = Tick(cfg)
        tick.upload_ready(snap.task_id, snap.task_priority, snap.task_rq, sorted_=True)
        if not args.no_resident_cluster:
            tick.cluster_upload(sc)  # worker rows + request tables resident in HBM (ABI 5); between the cold ticks of this loop no worker row changes: no delta to send
        step = lambda: tick.tick_raw(sc, resident=True)  # returns after the assignment vector is in host memory
    else:
        from hyperqueue_amd.sharded import ShardedTick

        st = ShardedTick(cfg, rank=rank, world=world, records_per_shard=int(1.5 * 200 * n_workers_per_gpu * mult / world) + 4096)
        st.upload_ready(snap.task_id, snap.task_priority, snap.task_rq)
        tick = st.t
        if not args.no_resident_cluster:
            tick.cluster_upload(sc)
        host_merged = None

        def step():
            nonlocal host_merged
            res, merged = st.tick_device(sc, W_all, resident=True)  # sharded tick + the one all-gather (RCCL)
            if rank == 0:  # the scheduler process pulls the merged assignment vector to the host
                if host_merged is None:
                    host_merged = torch.empty(merged.shape, dtype=merged.dtype, pin_memory=True)
                host_merged.copy_(merged, non_blocking=True)
            torch.cuda.synchronize()
            return res

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if st is not None:
        # The sinks are fixed-capacity blocks (one all-gather, one D2H of the merged vector): size them to what the workload really emits — the largest shard
        # of a first tick plus 10 % — instead of the a-priori bound above, which would travel over xGMI and PCIe every tick.
        # The first multi-rank tick also proves the merge path: if the library's RCCL communicator fails its first all-gather on ANY rank, every rank falls back
        # to torch.distributed's collective for the rest of the run (and the line says which one carried the timed ticks).
        first_err = None
        try:
            r0 = step()
        except Exception as e:  # noqa: BLE001
            first_err, r0 = repr(e), None
        if dist is not None:
            t_ok = torch.tensor([0 if first_err else 1], dtype=torch.int32, device=f"cuda:{local_rank}")
            dist.all_reduce(t_ok, op=dist.ReduceOp.MIN)
            if int(t_ok.item()) == 0:
                if st.collective != "library":
                    raise SystemExit(f"bench.py: the first sharded tick failed: {first_err}")
                print(f"bench.py rank {rank}: the library's RCCL all-gather failed on some rank ({first_err}); merging through torch.distributed instead", file=sys.stderr)
                st.collective, st.comm_world = "torch", 0
                st._install_exchange()
                r0 = step()
        elif first_err:
            raise SystemExit(f"bench.py: the first sharded tick failed: {first_err}")
        n_mine = int(np.ctypeslib.as_array(r0.rec_off, shape=(W_all + 1,))[W_all])
        n_max = n_mine
        if dist is not None:
            t_n = torch.tensor([n_mine], dtype=torch.int64, device=f"cuda:{local_rank}")
            dist.all_reduce(t_n, op=dist.ReduceOp.MAX)
            n_max = int(t_n.item())
        st.set_capacity(int(1.1 * n_max) + 1024)
        host_merged = None

    # The timed region carries dispatch events around ONE kernel, K1 (k_level_hist, the launch that streams the ready set: hqtick_set_kernel_timing(ctx, 2)) — the roofline
    # figure is the average over exactly these launches, and a rocprofv3 kernel trace of this command (`--headline-only`) sees the same population.
    tick.set_kernel_timing(False if args.no_kernel_timing else 2)
    for _ in range(args.warmup):
        step()
    barrier()
    lat, kstats, stages = [], [], []
    t_begin = time.perf_counter()
    for _ in range(args.steps):
        t0 = time.perf_counter()
        res = step()
        lat.append(time.perf_counter() - t0)
        stages.append((res.t_scan_us, res.t_batches_us, res.t_solve_us, res.t_mapping_us, res.t_total_us, int(res.status), int(res.is_optimal)))
        kstats.append(tick.kernel_stats())  # (one ctypes call per tick, inside the timed region: ~3 us of a ~4 ms step)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t_begin
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ks = kstats[-1]
    assigned, prefilled = int(ks["n_assigned"]), int(ks["n_prefilled"])  # whole-job counts: the placement is replicated on every rank
    # (rq, variant, worker index) -> count of the last timed tick, for the objective comparison with the CPU baseline further down (the result arrays belong to the
    # context and live until its next tick; the snapshot's C view is re-made by whoever calls snap.to_c() next — so this is read here)
    ncp = int(res.n_counts)
    gdp = dict(zip(zip(abi._np(res.count_rq, ncp, np.uint32).tolist(), abi._np(res.count_variant, ncp, np.uint8).tolist(), abi._np(res.count_worker, ncp, np.uint32).tolist()),
                   abi._np(res.count_value, ncp, np.uint32).tolist())) if (ncp and rank == 0) else {}
    if rank != 0:
        if dist is not None:
            if args.multi_extras and not args.headline_only:
                watchdog(args.extras_timeout, lambda: None)
                multi_rank_extras(cfg, rank, world, local_rank)
            dist.destroy_process_group()
        return

    n_ready, W, R = len(snap.task_id), len(snap.worker_id), snap.n_resources
    n_levels = int(len(np.unique(snap.task_priority)))
    med = lambda k: float(np.median([x[k] for x in kstats]))
    mean = lambda k: float(np.mean([x[k] for x in kstats]))
    value = assigned * args.steps / elapsed
    p50 = float(np.median(lat))
    # Roofline kernel = K1, the pass that streams the ready set (largest algorithmic byte count per launch: 12 B per ready task).  Duration = average over the launches
    # of the TIMED REGION itself (start / stop events at the dispatch, hipExtLaunchKernel).
    k1_us = mean("level_hist_us")
    k1_bytes, peak = n_ready * 12, 8000.0
    achieved = k1_bytes / (k1_us * 1e-6) / 1e9 if k1_us > 0 else 0.0
    prof_k1 = committed_profile("k_level_hist") if args.workload == "c3p" else None
    prof_sw = committed_profile("k_price_sweep") if args.workload == "c3p" else None
    sweeps, sweep_us = med("price_sweeps"), med("price_sweep_us")
    coupled = sweeps > 0
    blocks = W_all // world if (st is not None and world > 1) else W_all
    all_done = bool(all(s[5] == abi.HQTICK_DONE and s[6] for s in stages))
    stage_med = [round(float(x), 1) for x in np.median(np.asarray([s[:5] for s in stages]), axis=0)]
    par = "single" if world == 1 else (f"worker-shards x{world}: FxHash(worker_id) % {world}, ready set replicated, every rank sweeps its own worker range (one small all-gather per price sweep), "
                                        "one RCCL all-gather of the record sinks " + (f"inside libhqtick.so (hqtick_shard_allgather; communicator of {getattr(st, 'comm_world', world)} ranks)"
                                                                                    if getattr(st, "collective", "") == "library" else f"through torch.distributed ({getattr(st, 'collective', '?')})") + ", merged vector D2H on rank 0")
    out = {
        "metric": "tasks_assigned_per_sec", "value": value, "unit": "tasks/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "dtype": "u64", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {n_ready} ready tasks x {W} workers x {R} resource kinds, {len(snap.requests)} request classes, {n_levels} priority level(s)"
                               + (" at 80/15/5 % (BASELINE configs[2] as SURVEY 8d / BASELINE.md 3 write it), cold tick: the priority cuts couple every worker — ONE placement model of "
                                  f"{int(med('milp_cols'))} columns x {int(med('milp_rows'))} rows solved by price sweeps on the MI355X (k_price_sweep) under a host master" if coupled else
                                  ", cold tick (every class saturated: the placement separates per worker)"),
                   "priority_levels": n_levels, "parallelism": par, "ranks": world,
                   "ranks_in_the_library_communicator": int(getattr(st, "comm_world", 0)) if st is not None else 0,
                   "ready_set": "resident in HBM", "seed": args.seed,
                   # the headline's own numbers, here as well (the driver's record keeps `config` and `roofline` whole)
                   "p50_tick_ms": 1e3 * p50, "p95_tick_ms": 1e3 * float(np.percentile(lat, 95)), "assigned_per_tick": assigned, "prefilled_per_tick": prefilled,
                   "every_timed_tick_done_and_certified": all_done, "price_sweeps_per_tick": int(sweeps), "flag_configurations_per_tick": int(med("price_rounds")),
                   "tick_stages_us": dict(zip(["gpu_phase_a_scans", "batches", "solve", "mapping_plan_gpu_phase_c", "total_in_library"], stage_med)),
                   "coupled_solve_us": {"build_model": med("model_us"), "solve": med("milp_us"), "of_which_price_path": med("price_us"), "of_which_inside_sweeps_launch_to_totals": sweep_us},
                   "price_sweeps_share_of_tick": sweep_us / (1e6 * p50) if p50 > 0 else None,
                   "is_optimal_means": "certified within HiGHS's default mip_rel_gap = 1e-4, which is all the reference's solve_bounded asks for (solver/highs.rs:65-68)"},
        "p50_tick_ms": 1e3 * p50, "p95_tick_ms": 1e3 * float(np.percentile(lat, 95)),
        "assigned_per_tick": assigned, "prefilled_per_tick": prefilled,
        "tick_algorithmic_bytes": int(ks["algorithmic_bytes"]),
        "tick_bytes_per_s_end_to_end_GBps": ks["algorithmic_bytes"] / p50 / 1e9,
        "roofline": {"bound": "hbm", "kernel": "k_level_hist (K1)", "achieved": achieved, "peak": peak, "unit": "GB/s", "frac": achieved / peak,
                     "algorithmic_bytes_per_launch": k1_bytes, "avg_launch_us": k1_us, "launches_averaged": len(kstats),
                     "traffic": (prof_k1 or {}).get("traffic_bytes_per_launch"),
                     "rocprofv3": prof_k1,
                     "timing": "start / stop events at the dispatch (hipExtLaunchKernel) of EVERY K1 launch of the timed region (hqtick_set_kernel_timing(ctx, 2): K1 alone carries events), "
                               "averaged over those launches; `rocprofv3` quotes the committed kernel trace of `bench.py --headline-only`, the same launches",
                     "note": "K1 streams the whole ready set (12 B/task) and is the tick's only pass over it.  At 1 M tasks the set (20 MB) lives in the 256 MiB Infinity Cache across "
                             "ticks and a launch is latency-bound (12 MB = 1.9 us at 6.3 TB/s achievable): `roofline_vs_n` has the same kernel beyond the cache.  It is NOT where a "
                             "three-level tick spends its GPU time: see `dominant_kernel`",
                     # the kernel the three-level tick's GPU time goes to: not an HBM kernel, priced by its own figure of merit
                     "dominant_kernel": ({"kernel": "k_price_sweep", "launches_per_tick": int(sweeps), "blocks_per_launch": int(blocks), "avg_us_launch_to_totals_on_host": sweep_us / sweeps,
                                          "block_solves_per_s": blocks * sweeps / (sweep_us * 1e-6) if sweep_us > 0 else None, "share_of_tick": sweep_us / (1e6 * p50),
                                          "rocprofv3": prof_sw,
                                          "bound": "latency: a workgroup of four wavefronts per worker block — one walks the block's dependent chain on LDS-resident data (exact bounded knapsack under the current "
                                                   "prices, <= 32 columns x 4 rows), the others share its dual pool and run its greedy fills — 17.5 / 23.9 / 36.9 KB of LDS at 8 / 16 / 32 columns, six blocks per CU "
                                                   "(wavefront slots): a sweep of 1024 blocks is ONE round whose length is the slowest block's chain.  A block reads 0.5-2 KB: not an HBM kernel, not MFMA work.  "
                                                   "Counters: profiles/r06/sweepctr_*.csv (--pmc passes of the headline command)",
                                          "figure_of_merit": "exact block solves per second, launch -> totals visible on the host"} if coupled else None)},
    }
    if run_wd is not None:
        run_wd.cancel()
    # ---- the same loop with the product's default caches live (level table + Map-order memo): rides along in the line, is NOT `value` ----
    p50_warm = None
    if world == 1 and not args.force_sharded and not args.warm_caches and not args.headline_only:   # (--headline-only is what the rocprofv3 passes run: the timed loop's launches alone)
        try:
            tw = Tick(warm_cfg)
            tw.upload_ready(snap.task_id, snap.task_priority, snap.task_rq, sorted_=True)
            if not args.no_resident_cluster:
                tw.cluster_upload(sc)
            tw.set_kernel_timing(False if args.no_kernel_timing else 2)
            for _ in range(args.warmup):
                tw.tick_raw(sc, resident=True)
            lw = []
            for _ in range(args.steps):
                t0 = time.perf_counter(); tw.tick_raw(sc, resident=True); lw.append(time.perf_counter() - t0)
            tw.close()
            p50_warm = 1e3 * float(np.median(lw))
        except Exception as e:  # noqa: BLE001
            print(f"bench.py: warm-cache loop failed: {e!r}", file=sys.stderr)
    # ---- CPU baseline (N = 1 only): before the line, which carries it ----
    cpu = None
    if world == 1 and args.cpu_ticks > 0:
        try:
            cpu = cpu_baseline(snap, args.cpu_ticks)
        except Exception as e:  # the baseline is a reported extra; never lose the GPU line over it
            cpu = {"error": repr(e)}
    cpu_mdl = cpu.pop("_model", None) if cpu else None   # (the oracle's model of the snapshot: arrays, for the objective comparison below — not for any JSON)
    dom = out["roofline"].get("dominant_kernel")
    caches = ("none: HQTICK_FLAG_NO_TICK_CACHES (level table rediscovered, Map iteration orders recomputed, no block memo); resident inputs only (ready set, cluster tables)"
              if not args.warm_caches else "level table of the resident ready set + Map-order memo (product default); block memo off")
    gpu_busy = sweep_us / (1e6 * p50) if p50 > 0 else None   # (launch -> totals on the host, all sweeps of a tick; the kernel-trace share is in profiles/rNN/)
    line, text = headline_line(
        value=value, n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=1e3 * elapsed / args.steps, scaling=scaling,
        workload=f"{args.workload}: {n_ready} ready tasks x {W} workers x {R} resource kinds, {len(snap.requests)} request classes, {n_levels} priority level(s)"
                 + (" at 80/15/5 % (BASELINE configs[2] as SURVEY 8d writes it), cold tick on the HBM-resident ready set" if args.workload == "c3p" else ", cold tick on the HBM-resident ready set"),
        priority_levels=n_levels, parallelism=("single" if world == 1 else f"worker-shards x{world}: FxHash(worker_id) % {world}, one RCCL all-gather of the record sinks ({getattr(st, 'collective', '?')})"),
        ranks_in_comm=int(getattr(st, "comm_world", 0)) if st is not None else 0, seed=args.seed, p50_tick_ms=1e3 * p50, assigned_per_tick=assigned, all_done=all_done,
        caches_live=caches, p50_warm_ms=p50_warm, model_columns=med("milp_cols"), price_sweeps=sweeps, gpu_busy_share=gpu_busy,
        roofline=out["roofline"], dominant=({"kernel": "k_price_sweep", "avg_us": dom["avg_us_launch_to_totals_on_host"], "launches_per_tick": dom["launches_per_tick"], "share_of_tick": dom["share_of_tick"]} if dom else None),
        cpu=cpu)
    print(text)
    sys.stdout.flush()
    # ================= everything below is EXTRAS: written to --extras-file, summarised on stderr, never on stdout =================
    out["headline"] = line
    if cpu is not None:
        out["cpu_baseline"] = cpu
    write_extras(args.extras_file, out)
    nb = {}
    if world == 1 and not args.force_sharded and not args.no_kernel_timing and args.roofline_sweep:
        sweep = []
        for n_big in (4_000_000, 16_000_000):
            s2 = workloads.make(args.workload, seed=args.seed, n_tasks=n_big)
            t2 = Tick(cfg)
            t2.upload_ready(s2.task_id, s2.task_priority, s2.task_rq, sorted_=True)
            sc2 = s2.to_c()
            t2.cluster_upload(sc2)
            t2.set_kernel_timing(2)
            for _ in range(2):
                t2.tick_raw(sc2, resident=True)
            us = []
            for _ in range(8):  # in-tick launches under per-dispatch events
                t2.tick_raw(sc2, resident=True); us.append(t2.kernel_stats()["level_hist_us"])
            u = float(np.mean(us))
            sweep.append({"kernel": "level_hist", "n_ready": n_big, "avg_launch_us": round(u, 2), "GBps": n_big * 12 / (u * 1e-6) / 1e9, "frac": n_big * 12 / (u * 1e-6) / 1e9 / peak})
            t2.close()
        out["roofline"]["roofline_vs_n"] = sweep
    snap1 = None
    if world == 1 and not args.force_sharded and not args.headline_only and args.workload == "c3p":
        try:
            out["one_level_cold_tick"], snap1 = one_level_cold_tick(cfg, args, rec_bytes)
            nb["c3_one_priority_level_cold_tick"] = {k: out["one_level_cold_tick"][k] for k in ("tasks_assigned_per_sec", "p50_tick_ms", "assigned_per_tick")}
        except Exception as e:  # noqa: BLE001
            out["one_level_cold_tick"] = {"error": repr(e)}
    if snap1 is None and args.workload == "c3":
        snap1 = snap
    sc1 = snap1.to_c() if snap1 is not None else None
    if world == 1 and not args.force_sharded and args.steady_steps > 0 and snap1 is not None:
        # Steady state of the reference's own throughput benchmark shape (benchmarks/experiment-per-task-overhead.py: zero-worker, `sleep 0`):
        # everything a tick hands out has finished before the next one, and as many new tasks have become ready.  The ready set stays in HBM
        # and is updated by deltas (hqtick_ready_consume_last / hqtick_ready_add) — nothing is re-uploaded but the new tasks.
        ts = Tick(loop_cfg)
        ts.set_kernel_timing(False)  # no per-kernel timing events
        ts.upload_ready(snap1.task_id, snap1.task_priority, snap1.task_rq, sorted_=True)
        if not args.no_resident_cluster:
            ts.cluster_upload(sc1)  # the workers are empty again before every tick, no row changes
        def handed_out(res):  # ids of the records of a tick (assigned + prefilled)
            return abi.record_task_ids(res, W)

        rq_of = snap1.task_rq.copy()  # rq by (job_task_id - 1): every id here is job 1, task 1..n
        res = ts.tick_raw(sc1, resident=True)
        gone = handed_out(res)
        next_id = int(snap1.task_id[-1]) + 1
        t_delta, t_cons, t_tick, per_step, memo_hits = [], [], [], [], 0
        for _ in range(args.steady_steps + 2):
            k = len(gone)
            new_rq = rq_of[(gone & np.uint64(0xFFFFFFFF)).astype(np.int64) - 1]  # arrivals replace exactly what left, class by class
            rq_of = np.concatenate([rq_of, new_rq])
            # the arrivals in the PACKED form (hqtick_ready_add_packed, ABI 8): freshly minted ids are one consecutive run, one priority, u16 request ids — 2 bytes per
            # task over PCIe instead of 20 (--plain-adds: the three full columns through the pinned staging buffer, as up to round 3)
            rq16 = new_rq.astype(np.uint16)
            if args.plain_adds:
                v_id, v_prio, v_rq = ts.ready_add_stage(k)
                v_id[:] = np.arange(next_id, next_id + k, dtype=np.uint64)
                v_prio[:] = snap1.task_priority[0]
                v_rq[:] = new_rq
            # The step's delta: what the last tick handed out leaves the resident set (hqtick_ready_consume_last queues one kernel and returns), the arrivals join it
            # (one kernel when they are appended; the add returns when the set is ready, i.e. it waits for both).  The GPU has been idle since the tick returned —
            # the driver's own bookkeeping above is not part of the step, and none of the device's work hides behind it.
            a = time.perf_counter(); ts.ready_consume_last()
            a2 = time.perf_counter()
            if args.plain_adds:
                ts.ready_add_staged(k)
            else:
                ts.ready_add_packed([(next_id, k)], [(int(snap1.task_priority[0]), k)], rq16)
            next_id += k
            b = time.perf_counter(); res = ts.tick_raw(sc1, resident=True)
            c = time.perf_counter()
            gone = handed_out(res)
            memo_hits += int(ts.kernel_stats()["n_classes_memo"])
            t_delta.append(b - a); t_cons.append(a2 - a); t_tick.append(c - b); per_step.append(len(gone))
        per_step = int(np.median(per_step[2:]))
        t_delta, t_cons, t_tick = (np.asarray(x[2:]) for x in (t_delta, t_cons, t_tick))
        stp = t_delta + t_tick
        out["steady_state"] = {
            "what": "one priority level (c3).  Per step: hqtick_ready_consume_last (a no-op under HQTICK_FLAG_CONSUME_IN_TICK, the loops' default: the tick's selection kernel takes what it hands out; "
                    "--two-call-consume: its own kernel) + hqtick_ready_add_packed / _staged (the new tasks; returns once the resident set is ready) + hqtick_run_resident; workers empty again before every tick (sleep-0 tasks)",
            "consume": "two calls (hqtick_ready_consume_last runs the selection once more in mark mode)" if args.two_call_consume else "inside the tick (HQTICK_FLAG_CONSUME_IN_TICK)",
            "steps": args.steady_steps, "ready_set_before_each_tick": int(ts.ready_count()), "tasks_handed_out_per_step": per_step,
            "p50_step_ms": 1e3 * float(np.median(stp)), "tasks_per_s": per_step / float(np.median(stp)),
            "p50_consume_plus_add_us": 1e6 * float(np.median(t_delta)), "of_which_consume_call_us": 1e6 * float(np.median(t_cons)), "p50_tick_us": 1e6 * float(np.median(t_tick)),
            "add_batches_appended_behind_the_resident_columns": int(ts.kernel_stats()["ready_appends"]), "of_add_batches": args.steady_steps + 2,
            "host_class_blocks_answered_from_the_contexts_table": memo_hits,
            "delta_bytes_host_to_device_per_step": per_step * (20 if args.plain_adds else 2), "adds": "plain columns (20 B per task)" if args.plain_adds else "packed (hqtick_ready_add_packed: 2 B per task)",
        }
        ts.close()
        nb["steady_state_add_tick_consume_one_level"] = {"tasks_handed_out_per_sec": out["steady_state"]["tasks_per_s"], "p50_step_ms": out["steady_state"]["p50_step_ms"], "p50_tick_us_inside_the_loop": out["steady_state"]["p50_tick_us"]}
    if world == 1 and not args.force_sharded and snap1 is not None and args.hetero_steps > 0:
        out["steady_hetero"] = steady_hetero(loop_cfg, snap1, args.hetero_steps, args.seed, min(args.cpu_ticks, 1))
    if world == 1 and not args.force_sharded and snap1 is not None and args.dag_steps > 0:
        out["dag_churn"] = dag_churn(loop_cfg, args.dag_steps, args.seed, args.dag_classes, "random", min(args.cpu_ticks, 1))
        # the same loop with the coupled solve stopping where the reference's stops (HQTICK_FLAG_CERTIFICATE_ONLY: the 1e-4 certificate, no exact / canonical pass) —
        # what a single scheduler would run; the default above pays for an answer that is a function of the snapshot alone
        try:
            quick_cfg = type(cfg).from_buffer_copy(loop_cfg); quick_cfg.flags |= abi.HQTICK_FLAG_CERTIFICATE_ONLY
            qd = dag_churn(quick_cfg, args.dag_steps, args.seed, args.dag_classes, "random", 0)
            out["dag_churn"]["certificate_only"] = {"flag": "HQTICK_FLAG_CERTIFICATE_ONLY (is_optimal = 1, is_canonical = 0: the reference's own stopping rule, solver/highs.rs:65-88)",
                                                    **{k: qd[k] for k in ("p50_tick_us", "p50_coupled_solve_us", "p50_model_columns", "p50_step_ms", "tasks_handed_out_per_step", "tasks_per_s", "all_ticks_optimal", "steps")}}
        except Exception as e:  # noqa: BLE001
            out["dag_churn"]["certificate_only"] = {"error": repr(e)}
        out["dag_churn_layered"] = dag_churn(loop_cfg, args.dag_steps, args.seed, args.dag_classes, "layered", min(args.cpu_ticks, 1))
    if world == 1 and not args.force_sharded and args.workload == "c3p" and args.priority_ticks > 0:
        # the same three priority levels on a BUSY cluster (workloads.make_steady: every worker runs a packed mix of which 10 % just finished, ~930 distinct free
        # vectors): the everyday production tick — priorities AND heterogeneous workers.  The cuts make it one coupled model of all 1024 workers.
        try:
            ss = workloads.make_steady("c3p", seed=args.seed)
            tq = Tick(cfg)
            tq.upload_ready(ss.task_id, ss.task_priority, ss.task_rq, sorted_=True)
            scq = ss.to_c()
            tl2, inf2 = [], None
            for _ in range(args.priority_ticks + 1):
                t0 = time.perf_counter(); rq_ = tq.tick_raw(scq, resident=True); tl2.append(time.perf_counter() - t0)
                inf2 = (int(rq_.status), int(rq_.is_optimal), tq.kernel_stats())
            tq.close()
            out["multi_priority_busy_cluster"] = {"workload": "c3p on a cluster mid-run (workloads.make_steady('c3p')): 1 M ready tasks at three priority levels, 1024 workers with ~930 distinct free vectors",
                                                  "p50_tick_ms": 1e3 * float(np.median(tl2[1:])), "status": inf2[0], "is_optimal": bool(inf2[1]), "assigned_per_tick": int(inf2[2]["n_assigned"]),
                                                  "prefilled_per_tick": int(inf2[2]["n_prefilled"]), "model_columns": int(inf2[2]["milp_cols"]), "model_rows": int(inf2[2]["milp_rows"]),
                                                  "price_sweeps": int(inf2[2]["price_sweeps"]), "coupled_solve_ms": inf2[2]["milp_us"] / 1e3, "build_model_ms": inf2[2]["model_us"] / 1e3}
            if args.cpu_ticks > 0:
                from oracle.oracle import Oracle
                oq = Oracle(abi.make_config(time_limit_s=5.0), reference_solver_options=True)
                t0 = time.perf_counter(); wq = oq.tick(ss); tcq = time.perf_counter() - t0
                out["multi_priority_busy_cluster"]["cpu_baseline"] = {"tick_s": tcq, "is_optimal": bool(wq.is_optimal), "kind": "port", "cores": 1, "cpu": cpu_model(),
                                                                        "assigned_per_tick": sum(1 for recs in wq.records for (_, _, k) in recs if k == abi.HQ_REC_ASSIGN),
                                                                        "sample": "1 tick of the same snapshot, HiGHS 1.8.0 with the reference's options (time_limit = 5 s only)"}
        except Exception as e:
            out["multi_priority_busy_cluster"] = {"error": repr(e)}
        # ... and BASELINE configs[3]'s cluster (4096 workers, every class a 2-variant OR-list) with a ready set that does NOT saturate it: one coupled model of 65 536
        # columns through the batch-size rows — the model HiGHS holds an unproven incumbent on after minutes (no CPU baseline here for that reason)
        try:
            s4 = workloads.make("c4", seed=8, n_workers=4096, n_tasks=56_761)
            t4 = Tick(cfg)
            t4.upload_ready(s4.task_id, s4.task_priority, s4.task_rq, sorted_=True)
            sc4 = s4.to_c()
            tl4, inf4 = [], None
            for _ in range(max(3, args.priority_ticks // 2) + 1):
                t0 = time.perf_counter(); r4 = t4.tick_raw(sc4, resident=True); tl4.append(time.perf_counter() - t0)
                inf4 = (int(r4.status), int(r4.is_optimal), t4.kernel_stats())
            t4.close()
            out["config4_unsaturated"] = {"workload": "c4's cluster (4096 workers x 2-variant requests) with 56 761 ready tasks: no batch saturated, one coupled model of all workers",
                                          "p50_tick_ms": 1e3 * float(np.median(tl4[1:])), "status": inf4[0], "is_optimal": bool(inf4[1]), "assigned_per_tick": int(inf4[2]["n_assigned"]),
                                          "model_columns": int(inf4[2]["milp_cols"]), "model_rows": int(inf4[2]["milp_rows"]), "price_sweeps": int(inf4[2]["price_sweeps"]),
                                          "sweeps_ms": inf4[2]["price_sweep_us"] / 1e3, "avg_sweep_us": (inf4[2]["price_sweep_us"] / inf4[2]["price_sweeps"]) if inf4[2]["price_sweeps"] else None,
                                          "coupled_solve_ms": inf4[2]["milp_us"] / 1e3, "build_model_ms": inf4[2]["model_us"] / 1e3,
                                          "tasks_assigned_per_sec": int(inf4[2]["n_assigned"]) / float(np.median(tl4[1:])),
                                          "note": "4096 blocks of 16 columns per sweep; parity: tests/test_gpu_price.py::test_config4_unsaturated_full_tick_on_the_gpu"}
        except Exception as e:
            out["config4_unsaturated"] = {"error": repr(e)}
        # ... and BASELINE configs[3] AS WRITTEN ("C4 as C3 but 4096 workers ... 2-variant OR-list"): c4p = 1 M tasks at three priority levels x 4096 workers —
        # the largest model the contract names (65 536 placement columns + cut / blocker rows over 4096 blocks); parity: tests/test_gpu_price.py::test_c4p_full_tick_on_the_gpu
        try:
            s4p = workloads.make("c4p", seed=args.seed)
            t4p = Tick(cfg)
            t4p.upload_ready(s4p.task_id, s4p.task_priority, s4p.task_rq, sorted_=True)
            sc4p = s4p.to_c()
            t4p.cluster_upload(sc4p)
            tl4p, inf4p = [], None
            for _ in range(max(3, args.priority_ticks) + 1):
                t0 = time.perf_counter(); r4p = t4p.tick_raw(sc4p, resident=True); tl4p.append(time.perf_counter() - t0)
                inf4p = (int(r4p.status), int(r4p.is_optimal), t4p.kernel_stats())
            t4p.close()
            out["config4_three_levels"] = {"workload": "c4p: 1 M ready tasks at three priority levels (80/15/5 %) x 4096 workers, every class a 2-variant OR-list (BASELINE configs[3] as BASELINE.md 3 writes it), cold tick, no cross-tick caches",
                                           "p50_tick_ms": 1e3 * float(np.median(tl4p[1:])), "status": inf4p[0], "is_optimal": bool(inf4p[1]), "assigned_per_tick": int(inf4p[2]["n_assigned"]),
                                           "prefilled_per_tick": int(inf4p[2]["n_prefilled"]), "model_columns": int(inf4p[2]["milp_cols"]), "model_rows": int(inf4p[2]["milp_rows"]),
                                           "price_sweeps": int(inf4p[2]["price_sweeps"]), "sweeps_ms": inf4p[2]["price_sweep_us"] / 1e3, "coupled_solve_ms": inf4p[2]["milp_us"] / 1e3,
                                           "build_model_ms": inf4p[2]["model_us"] / 1e3, "tasks_assigned_per_sec": int(inf4p[2]["n_assigned"]) / float(np.median(tl4p[1:]))}
        except Exception as e:
            out["config4_three_levels"] = {"error": repr(e)}
        write_extras(args.extras_file, out)
    if cpu is not None and "error" not in cpu:
        # ratios against a baseline that STOPPED at the reference's 5 s time limit with an uncertified incumbent measure that limit, not equivalent work:
        # reported as lower bounds, next to the objective both sides reached on the same snapshot
        capped = not cpu.get("is_optimal")
        out["vs_cpu_baseline"] = {"throughput_ratio": value / cpu["value"] if cpu["value"] else None, "tick_latency_ratio": cpu["tick_s"] / p50 if p50 > 0 else None,
                                  "capped_by_the_baselines_time_limit": bool(capped),
                                  "read_as": "lower bounds (the baseline tick ended at the reference's mip_time_limit without a certificate)" if capped else "both sides certified"}
        if coupled:
            try:  # both sides maximise the same objective: c.x of the GPU tick's counts in the oracle's own model of the snapshot
                mp = cpu_mdl
                xg = np.asarray([gdp.get((int(mp["crq"][j]), int(mp["cvariant"][j]), int(mp["cworker"][j])), 0) if mp["ctype"][j] == 0 else 0 for j in range(len(mp["obj"]))], np.float64)
                out["objective"] = {"gpu_tick": float(np.dot(mp["obj"], xg)), "cpu_baseline": float(mp["objective"]), "cpu_baseline_is_optimal": cpu.get("is_optimal"),
                                    "note": "same snapshot, same objective (scheduler/solver.rs:542-571); the reference-configured HiGHS stops at its 5 s time limit on this model"}
            except Exception as e:  # noqa: BLE001
                out["objective"] = {"error": repr(e)}
    if world == 1 and not args.force_sharded and snap1 is not None and args.wire_iters > 0:
        out["wire"] = wire_block(args.wire_iters)
    out["neighbours"] = nb
    if dist is not None and args.multi_extras and not args.headline_only:
        # after everything the line is quoted on: if a rank gets lost in there, the watchdog writes the extras as they stand and ends the process
        wd = watchdog(args.extras_timeout, lambda: write_extras(args.extras_file, dict(out, multi_rank={"error": f"did not come back within {args.extras_timeout:.0f} s"})))
        out["multi_rank"] = multi_rank_extras(cfg, rank, world, local_rank)
        wd.cancel()
    write_extras(args.extras_file, out)
    summary = {k: (v.get("p50_tick_ms", v.get("p50_step_ms")) if isinstance(v, dict) else None) for k, v in out.items()
               if k in ("one_level_cold_tick", "steady_state", "steady_hetero", "dag_churn", "dag_churn_layered", "multi_priority_busy_cluster", "config4_unsaturated", "config4_three_levels")}
    print(f"bench.py: extras -> {args.extras_file}; p50 ms per block: {json.dumps(summary)}", file=sys.stderr)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
