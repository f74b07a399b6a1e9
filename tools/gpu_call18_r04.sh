#!/usr/bin/env bash
# the steady loop (consume -> add -> tick) under the kernel trace, final build: which kernels a step is made of now
set -u
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r04_call18
mkdir -p "$OUT"
LOOP="python $ROOT/bench.py --steps 5 --warmup 2 --no-roofline-sweep --steady-steps 60 --hetero-steps 0 --dag-steps 0 --priority-ticks 0 --wire-iters 0 --cpu-ticks 0 --no-b2b"
( cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/loop" -- $LOOP > "$OUT/loop.log" 2>&1 )
python profiles/summarize.py "$OUT/loop" > "$OUT/loop.summary.csv" 2>> "$OUT/loop.log"
( cd /tmp && HQTICK_APPEND=0 timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/loop_merge" -- $LOOP > "$OUT/loop_merge.log" 2>&1 )
python profiles/summarize.py "$OUT/loop_merge" > "$OUT/loop_merge.summary.csv" 2>> "$OUT/loop_merge.log"
find "$OUT" -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
cat "$OUT/loop.summary.csv"; echo; cat "$OUT/loop_merge.summary.csv"
