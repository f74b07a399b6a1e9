#!/usr/bin/env bash
# round 4, second GPU call: the whole GPU suite (no -x: every uncertified seed is wanted), the multi-process tests in the clear, the bench line
set -u
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r04_call2
mkdir -p "$OUT"
timeout 1500 python -m pytest tests -m gpu -q -rs -rf > "$OUT/pytest.log" 2>&1; echo "pytest rc $?" >> "$OUT/pytest.log"
grep -E "passed|failed|FAILED|solver limits|RCCL exchange" "$OUT/pytest.log" | tail -30
timeout 600 python -m pytest tests/test_gpu_multi.py -m gpu -q -s > "$OUT/multi.log" 2>&1; grep -E "RCCL exchange|passed|failed" "$OUT/multi.log"
timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc $?"
timeout 300 python bench.py --gpus 1 --force-sharded --steps 20 --cpu-ticks 0 > "$OUT/bench_force_sharded.json" 2> "$OUT/bench_force_sharded.err"; echo "force-sharded rc $?"
python - <<'PY'
import json
for f in ("bench.json", "bench_force_sharded.json"):
    try:
        d=json.loads([l for l in open("gpurun_out/r04_call2/"+f) if l.startswith("{")][-1])
        print(f, {k: d.get(k) for k in ("value","ms_per_step","p50_tick_ms","n_gpus","neighbours")}, d["roofline"]["frac"], d["roofline"].get("traffic"))
    except Exception as e: print(f, "ERR", e)
PY
