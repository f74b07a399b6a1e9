#!/usr/bin/env bash
set -u
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r04_call14
mkdir -p "$OUT"
( cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/coupled" -- python $ROOT/tools/price_probe.py c3p wave --no-host --repeat 2 > "$OUT/coupled.log" 2>&1 )
python profiles/summarize.py "$OUT/coupled" | head -4
f=$(find "$OUT/coupled" -name "*kernel_trace.csv" | head -1); head -1 "$f" | cut -c1-600; grep price_sweep "$f" | head -2 | cut -c1-600
find "$OUT" -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
