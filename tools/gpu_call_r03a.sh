#!/usr/bin/env bash
# First gpurun call of round 3: the price sweeps of the coupled solve on the MI355X.
#   gpurun --timeout 1200 -- 'bash tools/gpu_call_r03a.sh'
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r03a
mkdir -p "$OUT"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke exit $?" >> "$OUT/smoke.log"
timeout 600 python tools/price_probe.py c3p wave 0.2 0.45 c3p:64 c3p:256 > "$OUT/price_probe.log" 2>&1; echo "probe exit $?" >> "$OUT/price_probe.log"
cp gpurun_out/price_probe.json "$OUT/" 2>/dev/null
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/trace" -- python "$OLDPWD/tools/price_probe.py" c3p 0.45 --no-host --repeat 2 > "$OLDPWD/$OUT/trace.log" 2>&1 )
python profiles/summarize.py "$OUT/trace" > "$OUT/summary_trace.csv" 2>> "$OUT/trace.log"
timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; echo "pytest exit $?" >> "$OUT/pytest_gpu.log"
tail -5 "$OUT/smoke.log"; cat "$OUT/price_probe.log"; cat "$OUT/summary_trace.csv" | head -20; tail -5 "$OUT/pytest_gpu.log"
