#!/usr/bin/env bash
# round 4, third GPU call: worker-major selection (K4 -> K5b) — the GPU suite, the bench line, and the headline loop under the kernel trace + FETCH / WRITE passes
set -u
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r04_call3
mkdir -p "$OUT"
timeout 1500 python -m pytest tests -m gpu -q -rs -rf -x > "$OUT/pytest.log" 2>&1; echo "pytest rc $?" >> "$OUT/pytest.log"
grep -E "passed|failed|FAILED|solver limits|RCCL exchange" "$OUT/pytest.log" | tail -30
timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc $?"
HQTICK_TRANSPOSE=0 timeout 300 python bench.py --steps 50 --warmup 5 --no-roofline-sweep --steady-steps 0 --hetero-steps 0 --dag-steps 0 --priority-ticks 0 --wire-iters 0 --cpu-ticks 0 > "$OUT/bench_queue_order.json" 2> /dev/null
BENCH="python $ROOT/bench.py --steps 50 --warmup 5 --no-roofline-sweep --steady-steps 0 --hetero-steps 0 --dag-steps 0 --priority-ticks 0 --wire-iters 0 --cpu-ticks 0"
run_trace() { local name=$1; shift; ( cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name" -- "$@" > "$OUT/$name.log" 2>&1 ); python profiles/summarize.py "$OUT/$name" > "$OUT/$name.summary.csv" 2>> "$OUT/$name.log"; }
run_pmc() { local name=$1 c=$2; shift 2; ( cd /tmp && timeout 500 rocprofv3 --pmc "$c" --output-format csv -d "$OUT/${name}_$c" -- "$@" > "$OUT/${name}_$c.log" 2>&1 ); python profiles/summarize.py "$OUT/${name}_$c" > "$OUT/${name}_$c.summary.csv" 2>> "$OUT/${name}_$c.log"; }
run_trace bench_c3 $BENCH
python profiles/per_launch.py "$OUT/bench_c3" k_level_hist > "$OUT/k1_per_launch.txt" 2>&1
for c in FETCH_SIZE WRITE_SIZE; do run_pmc bench_c3 $c $BENCH; done
find "$OUT" -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
cat "$OUT/bench_c3.summary.csv" "$OUT/bench_c3_FETCH_SIZE.summary.csv" | head -30
python - <<'PY'
import json
for f in ("bench.json", "bench_queue_order.json"):
    try:
        d=json.loads([l for l in open("gpurun_out/r04_call3/"+f) if l.startswith("{")][-1])
        print(f, {k: d.get(k) for k in ("value","ms_per_step","p50_tick_ms","neighbours")}, d["roofline"]["frac"], {k:v["us"] for k,v in d["kernels"].items()}, d.get("tick_stages_us"))
    except Exception as e: print(f, "ERR", e)
PY
