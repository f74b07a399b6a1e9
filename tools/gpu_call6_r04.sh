#!/usr/bin/env bash
# the append path and the block memo: tests, then the steady loop with each switched off in turn
set -u
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r04_call6
mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_resident.py tests/test_gpu_parity.py tests/test_gpu_cluster.py -m gpu -q -x 2>&1 | tail -5 | tee "$OUT/tests.log"
LOOP="python $ROOT/bench.py --steps 30 --warmup 5 --no-roofline-sweep --steady-steps 60 --hetero-steps 10 --dag-steps 0 --priority-ticks 0 --wire-iters 0 --cpu-ticks 0 --no-b2b"
timeout 300 $LOOP > "$OUT/loop.json" 2> "$OUT/loop.err"
HQTICK_APPEND=0 timeout 300 $LOOP > "$OUT/loop_noappend.json" 2> "$OUT/loop_noappend.err"
timeout 300 $LOOP --plain-adds > "$OUT/loop_plain.json" 2> "$OUT/loop_plain.err"
python - <<'PY'
import json, os
out = os.environ.get("OUT", "gpurun_out/r04_call6")
for f in ("loop", "loop_noappend", "loop_plain"):
    try:
        d = json.loads(open(f"gpurun_out/r04_call6/{f}.json").read().strip().splitlines()[-1])
        print(f, "value", d["value"], "p50", d.get("p50_tick_us"), json.dumps(d.get("steady_state"))[:900])
        print("   hetero", json.dumps(d.get("steady_hetero"))[:500])
    except Exception as e:
        print(f, "failed", e); print(open(f"gpurun_out/r04_call6/{f}.err").read()[-1500:])
PY
timeout 300 python tools/loop_timeline.py > "$OUT/loop_timeline.txt" 2>&1; tail -16 "$OUT/loop_timeline.txt" | cut -c1-300
