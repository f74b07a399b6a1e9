#!/usr/bin/env bash
set -u
OUT=gpurun_out/r04_call13; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_price.py tests/test_gpu_blocks.py tests/test_fixtures.py -m gpu -q 2>&1 | grep -E "passed|failed|FAILED" | tee $OUT/tests.log
timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04_call13/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], json.dumps(d["neighbours"]))
for k in ("multi_priority","multi_priority_busy_cluster","config4_unsaturated"):
    m=d.get(k) or {}
    print(" ",k,{kk:m.get(kk) for kk in ("p50_tick_ms","coupled_solve","coupled_solve_ms","build_model_ms","sweeps_ms","price_sweeps","avg_sweep_us")}, (m.get("price_sweep_kernel") or {}).get("avg_sweep_us"))
for k in ("dag_churn","dag_churn_layered","steady_hetero"):
    m=d.get(k) or {}
    print(" ",k,{kk:m.get(kk) for kk in ("p50_step_ms","p50_tick_ms","p50_tick_us","p50_price_sweeps_per_tick","p50_sweeps_us","tasks_per_s")}, (m.get("block_solve_kernel") or {}).get("avg_us"))
PY
timeout 200 python tools/steady_probe.py c4 12 2>&1 | grep -v amdgpu | tail -4 | cut -c1-400
