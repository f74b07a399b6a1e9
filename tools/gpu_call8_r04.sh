#!/usr/bin/env bash
# after the host-side work on coupled ticks: the bench's loops and coupled blocks, async greedy A/B, resident + price tests
set -u
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r04_call8
mkdir -p "$OUT"
timeout 300 python -m pytest tests/test_gpu_resident.py -m gpu -q -x 2>&1 | tail -4 | tee "$OUT/tests_resident.log"
timeout 400 python bench.py --steps 30 --warmup 5 --no-roofline-sweep --cpu-ticks 0 --wire-iters 0 > "$OUT/bench.json" 2> "$OUT/bench.err"
C="python bench.py --steps 10 --warmup 3 --no-roofline-sweep --cpu-ticks 0 --wire-iters 0 --steady-steps 0 --hetero-steps 0 --dag-steps 0 --priority-ticks 7 --no-b2b"
HQMILP_ASYNC_GREEDY=1 timeout 300 $C > "$OUT/bench_async.json" 2> "$OUT/bench_async.err"
timeout 300 $C > "$OUT/bench_sync.json" 2> "$OUT/bench_sync.err"
python - <<'PY'
import json
def load(f):
    try: return json.loads(open(f"gpurun_out/r04_call8/{f}.json").read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "failed", e); print(open(f"gpurun_out/r04_call8/{f}.err").read()[-1500:]); return None
d = load("bench")
if d:
    print("value", d["value"], "neighbours", json.dumps(d["neighbours"]))
    print("steady_state", json.dumps(d.get("steady_state"))[:1200])
    h = d.get("steady_hetero") or {}
    print("hetero", {k: h.get(k) for k in ("p50_step_ms", "p50_tick_ms", "p50_add_us", "worker_classes_per_tick", "tick_stages_us")})
    for k in ("multi_priority", "multi_priority_busy_cluster", "config4_unsaturated"):
        m = d.get(k) or {}
        print(k, {kk: m.get(kk) for kk in ("p50_tick_ms", "coupled_solve", "coupled_solve_ms", "build_model_ms", "sweeps_ms", "price_sweeps", "is_optimal")})
    for k in ("dag_churn", "dag_churn_layered"):
        m = d.get(k) or {}
        print(k, {kk: m.get(kk) for kk in ("p50_step_ms", "p50_tick_us", "p50_price_sweeps_per_tick", "p50_coupled_solve_us", "p50_sweeps_us", "tasks_per_s")})
for f in ("bench_async", "bench_sync"):
    d = load(f)
    if d:
        for k in ("multi_priority", "config4_unsaturated"):
            m = d.get(k) or {}
            print(f, k, {kk: m.get(kk) for kk in ("p50_tick_ms", "coupled_solve", "coupled_solve_ms", "build_model_ms")})
PY
timeout 500 python -m pytest tests/test_gpu_price.py -m gpu -q -x 2>&1 | tail -4 | tee "$OUT/tests_price.log"
