#!/usr/bin/env bash
# experiment: an early marker packet in event-less ticks (does K1's recorded span, and the tick, get shorter?)
set -u
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r04_call9
mkdir -p "$OUT"
B="python $ROOT/bench.py --steps 200 --warmup 10 --no-roofline-sweep --steady-steps 0 --hetero-steps 0 --dag-steps 0 --priority-ticks 0 --wire-iters 0 --cpu-ticks 0 --no-b2b"
for rep in 1 2; do
  for m in 0 1; do
    HQTICK_EARLY_MARKER=$m timeout 200 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('marker $m rep $rep: value', round(d['value']/1e6,1), 'p50', round(1e3*d['p50_tick_ms'],2), 'p95', round(1e3*d['p95_tick_ms'],2), 'stages', d['tick_stages_us'])"
  done
done
for m in 0 1; do
  ( cd /tmp && HQTICK_EARLY_MARKER=$m timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/m$m" -- $B > "$OUT/m$m.log" 2>&1 )
  python profiles/per_launch.py "$OUT/m$m" k_level_hist 2>&1 | head -6
done
find "$OUT" -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
