#!/usr/bin/env python
"""The figures of a bench.py line a human wants to see after a GPU call:  python tools/bench_digest.py <file with the JSON line>"""
import json
import sys

line = [l for l in open(sys.argv[1]) if l.startswith("{")]
if not line:
    print("no JSON line in", sys.argv[1]); sys.exit(0)
d = json.loads(line[-1])
c, r = d.get("config", {}), d.get("roofline", {})
print(f"value {d.get('value'):.4g} {d.get('unit')}  ms/step {d.get('ms_per_step'):.4f}  p50 {c.get('p50_tick_ms')}  assigned {c.get('assigned_per_tick')}  certified {c.get('every_timed_tick_done_and_certified')}  sweeps {c.get('price_sweeps_per_tick')}")
print("stages", c.get("tick_stages_us"), "coupled", c.get("coupled_solve_us"))
print(f"roofline K1: {r.get('avg_launch_us')} us  frac {r.get('frac')}  rocprof {((r.get('rocprofv3') or {}).get('kernel_trace'))}")
dk = r.get("dominant_kernel") or {}
print("sweep kernel:", {k: dk.get(k) for k in ("launches_per_tick", "avg_us_launch_to_totals_on_host", "block_solves_per_s", "share_of_tick")})
print("roofline_vs_n", r.get("roofline_vs_n"))
print("neighbours", json.dumps(d.get("neighbours")))
for k in ("steady_hetero", "dag_churn", "dag_churn_layered", "multi_priority_busy_cluster", "config4_unsaturated", "wire"):
    v = d.get(k)
    if isinstance(v, dict):
        print(k, {kk: v[kk] for kk in ("p50_tick_ms", "p50_step_ms", "p50_tick_us", "tasks_per_s", "error", "is_optimal", "price_sweeps", "p50_price_sweeps_per_tick", "tasks_assigned_per_sec") if kk in v})
        if isinstance(v.get("certificate_only"), dict):
            print("  certificate_only", {kk: v["certificate_only"].get(kk) for kk in ("p50_tick_us", "p50_coupled_solve_us", "p50_step_ms", "all_ticks_optimal", "error") if kk in v["certificate_only"]})
        if isinstance(v.get("tick_stages_us"), dict):
            print("  stages", {kk: round(x, 1) for kk, x in v["tick_stages_us"].items()})
cb = d.get("cpu_baseline") or {}
print("cpu_baseline", {k: cb.get(k) for k in ("value", "tick_s", "assigned_per_tick", "is_optimal", "error")}, "speedup", d.get("speedup_vs_cpu_baseline"), "objective", d.get("objective"))
