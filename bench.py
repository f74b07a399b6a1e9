#!/usr/bin/env python
"""bench.py — tasks assigned/sec and tick latency of the MI355X-native tako scheduling tick.

A "step" is one scheduling tick (run_scheduling_inner, /root/reference/crates/tako/src/internal/scheduler/main.rs:50-72) over
the synthetic workload `c3` (BASELINE.json configs[2]: 1 M ready tasks over 8 mixed {cpus, gpus, mem} request classes incl.
fractional GPUs, 1024 workers), cold: every worker empty, every task ready.  The ready-set columns are resident in HBM when
the timed region starts (hqtick_upload_ready); worker/request tables (50 KB) are part of the snapshot handed over each tick.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3]
  N > 1: launched by torch.distributed.run, one rank per GPU.  The tick of ONE server does not shard without an exchange
  step that this round does not implement, so ranks run independent scheduler replicas
  ("replicas only"): value = N x per-replica rate, scaling "weak", no data-path collective.

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


def cpu_baseline(snap, ticks: int):
    """The CPU oracle (restatement of the reference tick + HiGHS 1.8.0 for the MILP) on this host, 1 core, same snapshot."""
    from hyperqueue_amd import abi
    from oracle.oracle import Oracle

    o = Oracle(abi.make_config(time_limit_s=5.0))
    lat, assigned = [], 0
    for _ in range(ticks):
        t0 = time.perf_counter()
        r = o.tick(snap)
        lat.append(time.perf_counter() - t0)
        assigned = sum(1 for recs in r.records for (_, _, k) in recs if k == abi.HQ_REC_ASSIGN)
    st = o.stage_times_us()
    med = float(np.median(lat))
    return {
        "value": assigned / med, "unit": "tasks/s", "cores": 1, "kind": "port",
        "sample": f"{ticks} cold tick(s) of the full workload ({len(snap.task_id)} tasks x {len(snap.worker_id)} workers), "
                  f"median {med:.2f} s/tick, {assigned} tasks assigned/tick",
        "tick_s": med, "assigned_per_tick": assigned,
        "stages_us": {k: round(v, 1) for k, v in st.items()},
        "note": "restatement of the reference (C++ -O2) + HiGHS 1.8.0 via scipy for the MILP; not the reference binary (no Rust toolchain)",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cpu-ticks", type=int, default=3, help="ticks of the CPU baseline (0 = skip)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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

    snap = workloads.make(args.workload, seed=args.seed + rank)  # replicas: every rank schedules its own (differently seeded) ready set
    tick = Tick(abi.make_config(time_limit_s=5.0, device_index=local_rank))
    tick.upload_ready(snap.task_id, snap.task_priority, snap.task_rq, sorted_=True)
    sc = snap.to_c()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        tick.tick_raw(sc, resident=True)
    barrier()
    lat, kstats, stages = [], [], []
    t_begin = time.perf_counter()
    for _ in range(args.steps):
        t0 = time.perf_counter()
        res = tick.tick_raw(sc, resident=True)  # returns after the assignment vector is back in host memory
        lat.append(time.perf_counter() - t0)
        kstats.append(tick.kernel_stats())
        stages.append((res.t_scan_us, res.t_batches_us, res.t_solve_us, res.t_mapping_us, res.t_total_us))
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t_begin
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ks = kstats[-1]
    assigned, prefilled = int(ks["n_assigned"]), int(ks["n_prefilled"])
    total_assigned = assigned
    if dist is not None:
        t = torch.tensor([assigned], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        total_assigned = int(t.item())
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    n_ready, W, R = len(snap.task_id), len(snap.worker_id), snap.n_resources
    mean = lambda k: float(np.mean([s[k] for s in kstats]))
    k_us = {"distinct_priorities": mean("distinct_us"), "level_hist": mean("level_hist_us"), "select_scatter": mean("select_us"), "expand_mapping": mean("other_us")}
    # algorithmic bytes each streaming kernel has to move per launch
    sel = assigned + prefilled
    k_bytes = {"distinct_priorities": n_ready * 8, "level_hist": n_ready * 12, "select_scatter": n_ready * 12 + sel * (8 + 8 + 2), "expand_mapping": sel * (8 + 2 + 8 + 2)}
    dom = max(k_us, key=lambda k: k_us[k])
    achieved = k_bytes[dom] / (k_us[dom] * 1e-6) / 1e9 if k_us[dom] > 0 else 0.0
    peak = 8000.0
    value = total_assigned * args.steps / elapsed
    out = {
        "metric": "tasks_assigned_per_sec", "value": value, "unit": "tasks/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {n_ready} ready tasks x {W} workers x {R} resource kinds, {len(snap.requests)} request classes, cold tick",
                   "parallelism": "single" if world == 1 else f"replicas{world}", "ready_set": "resident in HBM", "seed": args.seed},
        "p50_tick_ms": 1e3 * float(np.median(lat)), "p95_tick_ms": 1e3 * float(np.percentile(lat, 95)),
        "assigned_per_tick": assigned, "prefilled_per_tick": prefilled,
        "tick_algorithmic_bytes": int(ks["algorithmic_bytes"]),
        "tick_bytes_per_s_end_to_end_GBps": ks["algorithmic_bytes"] / float(np.median(lat)) / 1e9,
        "kernels_us": {k: round(v, 2) for k, v in k_us.items()},
        "tick_stages_us": dict(zip(["scan_gpu_phase", "batches", "solve", "mapping_plan_gpu_d2h", "total_in_library"], [round(float(x), 1) for x in np.median(np.asarray(stages), axis=0)])),
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": peak, "unit": "GB/s", "frac": achieved / peak,
                     "algorithmic_bytes_per_launch": k_bytes[dom], "avg_launch_us": k_us[dom], "traffic": None},
    }
    if world == 1 and args.cpu_ticks > 0:
        try:
            out["cpu_baseline"] = cpu_baseline(snap, args.cpu_ticks)
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
            out["tick_latency_ratio_vs_cpu"] = out["cpu_baseline"]["tick_s"] / float(np.median(lat))
        except Exception as e:  # the baseline is a reported extra; never lose the GPU line over it
            out["cpu_baseline"] = {"error": repr(e)}
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
