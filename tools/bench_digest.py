#!/usr/bin/env python
"""The figures of a bench.py run a human wants to see after a GPU call:  python tools/bench_digest.py <file with the JSON line> [<extras file>]
The line (stdout of bench.py, < 4 KB) is printed whole with its size; the blocks of the extras file (bench_extras.json) as one-liners."""
import json
import sys

line = [l for l in open(sys.argv[1]) if l.startswith("{")]
if not line:
    print("no JSON line in", sys.argv[1]); sys.exit(0)
print(f"line: {len(line[-1].strip())} bytes, {len(line)} JSON line(s) on stdout")
print(line[-1].strip())
if len(sys.argv) < 3:
    sys.exit(0)
try:
    d = json.load(open(sys.argv[2]))
except Exception as e:  # noqa: BLE001
    print("no extras:", e); sys.exit(0)
c, r = d.get("config", {}), d.get("roofline", {})
print("stages", c.get("tick_stages_us"), "coupled", c.get("coupled_solve_us"))
print(f"roofline K1: {r.get('avg_launch_us')} us  frac {r.get('frac')}  rocprof {((r.get('rocprofv3') or {}).get('kernel_trace'))}")
dk = r.get("dominant_kernel") or {}
print("sweep kernel:", {k: dk.get(k) for k in ("launches_per_tick", "avg_us_launch_to_totals_on_host", "block_solves_per_s", "share_of_tick")})
print("roofline_vs_n", r.get("roofline_vs_n"))
print("neighbours", json.dumps(d.get("neighbours")))
for k in ("steady_hetero", "dag_churn", "dag_churn_layered", "multi_priority_busy_cluster", "config4_unsaturated", "config4_three_levels", "wire"):
    v = d.get(k)
    if isinstance(v, dict):
        print(k, {kk: v[kk] for kk in ("p50_tick_ms", "p50_step_ms", "p50_tick_us", "tasks_per_s", "error", "is_optimal", "status", "price_sweeps", "sweeps_ms", "coupled_solve_ms", "build_model_ms", "p50_price_sweeps_per_tick", "tasks_assigned_per_sec") if kk in v})
        if isinstance(v.get("certificate_only"), dict):
            print("  certificate_only", {kk: v["certificate_only"].get(kk) for kk in ("p50_tick_us", "p50_coupled_solve_us", "p50_step_ms", "all_ticks_optimal", "error") if kk in v["certificate_only"]})
        if isinstance(v.get("tick_stages_us"), dict):
            print("  stages", {kk: round(x, 1) for kk, x in v["tick_stages_us"].items()})
print("vs_cpu_baseline", d.get("vs_cpu_baseline"), "objective", d.get("objective"))
