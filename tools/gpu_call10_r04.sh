#!/usr/bin/env bash
set -u
OUT=gpurun_out/r04_call10; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_price.py tests/test_fixtures.py tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | tail -4 | tee $OUT/tests.log
C="python bench.py --steps 10 --warmup 3 --no-roofline-sweep --cpu-ticks 0 --wire-iters 0 --steady-steps 0 --hetero-steps 0 --dag-steps 12 --priority-ticks 7 --no-b2b"
for lz in 1 0; do
HQMILP_LAZY_GREEDY=$lz timeout 400 $C 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('lazy $lz')
for k in ('multi_priority','multi_priority_busy_cluster','config4_unsaturated'):
    m=d.get(k) or {}
    print(' ',k,{kk:m.get(kk) for kk in ('p50_tick_ms','coupled_solve','coupled_solve_ms','build_model_ms','sweeps_ms','price_sweeps')})
for k in ('dag_churn','dag_churn_layered'):
    m=d.get(k) or {}
    print(' ',k,{kk:m.get(kk) for kk in ('p50_step_ms','p50_tick_us','p50_price_sweeps_per_tick','p50_coupled_solve_us','tasks_per_s')})
"
done
