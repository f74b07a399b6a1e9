#!/usr/bin/env bash
set -u
OUT=gpurun_out/r04_call19; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_resident.py tests/test_gpu_graph.py tests/test_gpu_cluster.py -m gpu -q 2>&1 | grep -E "passed|failed|FAILED" | tee $OUT/tests.log
timeout 600 python tools/gpu_resident_campaign.py 300 40 2>&1 | tail -1
L="python bench.py --steps 30 --warmup 5 --no-roofline-sweep --steady-steps 60 --hetero-steps 10 --dag-steps 12 --priority-ticks 0 --wire-iters 0 --cpu-ticks 0 --no-b2b"
for v in "" "--two-call-consume"; do
timeout 300 $L $v 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['steady_state']
print('$v', {k:(round(s[k],1) if isinstance(s[k],float) else s[k]) for k in ('p50_step_ms','tasks_per_s','p50_consume_plus_add_us','of_which_consume_call_us','p50_tick_us','consume')})
h=d['steady_hetero']; print('   hetero', round(h['p50_step_ms'],3), round(h['p50_tick_ms'],3), round(h['p50_consume_us'],1)); g=d['dag_churn_layered']; print('   dag layered', round(g['p50_step_ms'],3), round(g['p50_tick_us'],1), round(g['p50_consume_us'],1))"
done
