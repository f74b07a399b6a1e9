#!/usr/bin/env bash
# round 4, first GPU call: the whole GPU suite on the round's first commits, the bench line, and K1 launch by launch under the kernel trace
set -u
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r04_call1
mkdir -p "$OUT"
timeout 1200 python -m pytest tests -m gpu -x -q -rs > "$OUT/pytest.log" 2>&1; echo "pytest rc $?" >> "$OUT/pytest.log"
tail -15 "$OUT/pytest.log"
timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc $?"
BENCH="python $ROOT/bench.py --steps 50 --warmup 5 --no-roofline-sweep --steady-steps 0 --hetero-steps 0 --dag-steps 0 --priority-ticks 0 --wire-iters 0 --cpu-ticks 0 --no-b2b"
( cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/k1trace" -- $BENCH > "$OUT/k1trace.log" 2>&1 )
python profiles/summarize.py "$OUT/k1trace" > "$OUT/k1trace.summary.csv" 2>> "$OUT/k1trace.log"
python profiles/per_launch.py "$OUT/k1trace" k_level_hist > "$OUT/k1_per_launch.txt" 2>&1
python profiles/per_launch.py "$OUT/k1trace" k_expand_mapping > "$OUT/k5b_per_launch.txt" 2>&1
find "$OUT" -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
head -12 "$OUT/k1_per_launch.txt"
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r04_call1/bench.json") if l.startswith("{")][-1])
print({k: d[k] for k in ("value","ms_per_step","p50_tick_ms","n_gpus")}, d["roofline"]["frac"], d["kernels"])
PY
