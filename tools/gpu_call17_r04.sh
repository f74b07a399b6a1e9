#!/usr/bin/env bash
set -u
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r04_call17; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_price.py tests/test_gpu_blocks.py tests/test_fixtures.py tests/test_gpu_fuzz.py -m gpu -q 2>&1 | grep -E "passed|failed|FAILED" | tee $OUT/tests.log
timeout 100 python tools/exp/unsat4096.py 4 2>&1 | grep "^tick"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python $ROOT/tools/exp/unsat4096.py 3 > "$OUT/trace.log" 2>&1 ); python profiles/summarize.py "$OUT/trace" | grep price_sweep
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace2" -- python $ROOT/tools/price_probe.py c3p wave --no-host --repeat 2 > "$OUT/trace2.log" 2>&1 ); python profiles/summarize.py "$OUT/trace2" | grep price_sweep
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace3" -- python $ROOT/tools/steady_probe.py c4 12 > "$OUT/trace3.log" 2>&1 ); python profiles/summarize.py "$OUT/trace3" | grep block_solve
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace4" -- python $ROOT/tools/steady_probe.py c3 12 > "$OUT/trace4.log" 2>&1 ); python profiles/summarize.py "$OUT/trace4" | grep block_solve
find "$OUT" -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
