#!/usr/bin/env bash
# First gpurun call of round 2 (row f3): run it as
#     gpurun --timeout 900 -- 'bash tools/gpu_first_call_r02.sh'
# Everything lands under gpurun_out/r02_wire/ ; copy the summaries into profiles/r02/ afterwards.
# Order: cheapest evidence first (a fault later must not cost what is already on disk).
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02_wire
mkdir -p "$OUT"
# 1. the wire kernels' GPU tests (subprocess canary first) + the late end-to-end pins
timeout 600 python -m pytest tests/test_zz_gpu_wire.py -q > "$OUT/pytest_wire.log" 2>&1; echo "pytest exit $?" >> "$OUT/pytest_wire.log"
# 2. the bench tool: parity + events around the three launches
timeout 300 python tools/wire_bench.py --iters 200 > "$OUT/wire_bench.json" 2> "$OUT/wire_bench.err"
# 3. kernel trace of the same command (per-kernel durations of k_wire_plan / k_wire_scan / k_wire_emit)
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/trace" -- python "$OLDPWD/tools/wire_bench.py" --iters 50 > "$OLDPWD/$OUT/trace.log" 2>&1 )
python profiles/summarize.py "$OUT/trace" > "$OUT/summary_trace.csv" 2>> "$OUT/trace.log"
# 4. HBM traffic counters, each in its own pass, counters only (no trace domains next to --pmc: gpurun refuses that combination)
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 300 rocprofv3 --pmc $c -d "$OLDPWD/$OUT/pmc_$c" -- python "$OLDPWD/tools/wire_bench.py" --iters 20 > "$OLDPWD/$OUT/pmc_$c.log" 2>&1 )
  python profiles/summarize.py "$OUT/pmc_$c" > "$OUT/summary_$c.csv" 2>> "$OUT/pmc_$c.log"
done
# 5. the whole GPU suite and the bench line (the driver runs these too; here for the per-file log)
timeout 1500 python -m pytest tests -m gpu -q > "$OUT/pytest_gpu.log" 2>&1; echo "pytest exit $?" >> "$OUT/pytest_gpu.log"
timeout 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -3 "$OUT/pytest_wire.log"; cat "$OUT/wire_bench.json"
