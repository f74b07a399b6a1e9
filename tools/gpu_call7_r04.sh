#!/usr/bin/env bash
set -u
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r04_call7
mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_gpu_resident.py -m gpu -q -x 2>&1 | tail -8 | tee "$OUT/tests.log"
HQTICK_TRACE_ADD=1 timeout 300 python tools/loop_timeline.py 20 --packed > "$OUT/loop_timeline_packed.txt" 2>&1; grep "hqtick add" "$OUT/loop_timeline_packed.txt" | tail -8; grep -v "hqtick add" "$OUT/loop_timeline_packed.txt" | tail -4 | cut -c1-300
