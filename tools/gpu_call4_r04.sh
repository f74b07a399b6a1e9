#!/usr/bin/env bash
# round 4, fourth GPU call: packed adds (tests + the steady loop both ways), coupled-tick timelines of this build
set -u
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r04_call4
mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_resident.py tests/test_gpu_parity.py tests/test_gpu_cluster.py -m gpu -q -x > "$OUT/pytest.log" 2>&1; tail -3 "$OUT/pytest.log"
timeout 600 python bench.py --steps 50 --warmup 5 --no-roofline-sweep --hetero-steps 0 --dag-steps 0 --priority-ticks 0 --wire-iters 0 --cpu-ticks 0 > "$OUT/bench_loop_packed.json" 2> "$OUT/bench_loop_packed.err"
timeout 600 python bench.py --steps 50 --warmup 5 --no-roofline-sweep --hetero-steps 0 --dag-steps 0 --priority-ticks 0 --wire-iters 0 --cpu-ticks 0 --plain-adds > "$OUT/bench_loop_plain.json" 2> /dev/null
timeout 600 python tools/price_probe.py c3p wave 0.2 --no-host --timeline --repeat 1 > "$OUT/coupled_ticks.txt" 2>&1
timeout 600 python tools/steady_probe.py c3 20 > "$OUT/steady_c3.txt" 2>&1
python - <<'PY'
import json
for f in ("bench_loop_packed.json", "bench_loop_plain.json"):
    try:
        d=json.loads([l for l in open("gpurun_out/r04_call4/"+f) if l.startswith("{")][-1])
        print(f, d["value"], d["p50_tick_ms"], d.get("steady_state"))
    except Exception as e: print(f, "ERR", e)
PY
grep -E "timeline|price " "$OUT/coupled_ticks.txt" | cut -c1-700 | head -8
tail -5 "$OUT/steady_c3.txt" | cut -c1-600
