#!/usr/bin/env bash
# Second gpurun call of round 3: stage timeline of the coupled ticks, kernel trace of the sweeps, the new GPU tests.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r03b
mkdir -p "$OUT"
timeout 600 python tools/price_probe.py c3p wave --timeline --no-host > "$OUT/price_probe.log" 2>&1; echo "probe exit $?" >> "$OUT/price_probe.log"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/trace" -- python "$OLDPWD/tools/price_probe.py" c3p wave --no-host --repeat 2 > "$OLDPWD/$OUT/trace.log" 2>&1 )
python profiles/summarize.py "$OUT/trace" > "$OUT/summary_trace.csv" 2>> "$OUT/trace.log"
timeout 1200 python -m pytest tests/test_gpu_price.py -x -q > "$OUT/pytest_price.log" 2>&1; echo "pytest exit $?" >> "$OUT/pytest_price.log"
cat "$OUT/price_probe.log"; cat "$OUT/summary_trace.csv" | head -20; tail -15 "$OUT/pytest_price.log"
