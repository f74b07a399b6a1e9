#!/usr/bin/env bash
set -u
OUT=gpurun_out/r04_call11; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_price.py tests/test_fixtures.py tests/test_gpu_fuzz.py tests/test_gpu_blocks.py tests/test_gpu_graph.py -m gpu -q 2>&1 | tail -4 | tee $OUT/tests.log
timeout 900 python tools/gpu_price_campaign.py 400 80 > $OUT/price_campaign.txt 2>&1; tail -1 $OUT/price_campaign.txt
timeout 600 python tools/gpu_price_campaign.py 0 24 --big > $OUT/price_campaign_big.txt 2>&1; tail -1 $OUT/price_campaign_big.txt
timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04_call11/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], json.dumps(d["neighbours"]))
for k in ('multi_priority','multi_priority_busy_cluster','config4_unsaturated'):
    m=d.get(k) or {}
    print(' ',k,{kk:m.get(kk) for kk in ('p50_tick_ms','coupled_solve','coupled_solve_ms','build_model_ms','sweeps_ms','price_sweeps')})
for k in ('dag_churn','dag_churn_layered'):
    m=d.get(k) or {}
    print(' ',k,{kk:m.get(kk) for kk in ('p50_step_ms','p50_tick_us','p50_price_sweeps_per_tick','p50_coupled_solve_us','tasks_per_s')})
PY
