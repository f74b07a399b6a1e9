#!/usr/bin/env bash
# the coupled part of tools/profile_r04.sh once more, on the final build
set -u
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r04_prof2
mkdir -p "$OUT"
( cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/coupled" -- python $ROOT/tools/price_probe.py c3p wave 0.2 0.45 --no-host --timeline --repeat 1 > "$OUT/coupled.log" 2>&1 )
python profiles/summarize.py "$OUT/coupled" > "$OUT/coupled.summary.csv" 2>> "$OUT/coupled.log"
grep -v amdgpu.ids "$OUT/coupled.log" | grep -E "timeline|price " > "$OUT/coupled_ticks.txt"
find "$OUT" -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
cut -c1-330 "$OUT/coupled_ticks.txt"; head -4 "$OUT/coupled.summary.csv"
