#!/usr/bin/env bash
# the steady-state loop (add -> tick -> consume) under the kernel trace: what the 140 us of an add are made of
set -u
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r04_call5
mkdir -p "$OUT"
LOOP="python $ROOT/bench.py --steps 5 --warmup 2 --no-roofline-sweep --steady-steps 40 --hetero-steps 0 --dag-steps 0 --priority-ticks 0 --wire-iters 0 --cpu-ticks 0 --no-b2b"
( cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/loop" -- $LOOP > "$OUT/loop.log" 2>&1 )
python profiles/summarize.py "$OUT/loop" > "$OUT/loop.summary.csv" 2>> "$OUT/loop.log"
( cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/loop_plain" -- $LOOP --plain-adds > "$OUT/loop_plain.log" 2>&1 )
python profiles/summarize.py "$OUT/loop_plain" > "$OUT/loop_plain.summary.csv" 2>> "$OUT/loop_plain.log"
find "$OUT" -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
cat "$OUT/loop.summary.csv"; echo; cat "$OUT/loop_plain.summary.csv"
timeout 300 python tools/loop_timeline.py > "$OUT/loop_timeline.txt" 2>&1; tail -40 "$OUT/loop_timeline.txt" | cut -c1-300
timeout 200 python -m pytest tests/test_gpu_resident.py -m gpu -q -x 2>&1 | tail -2
