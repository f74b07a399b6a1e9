#!/usr/bin/env bash
# gpurun call: the new GPU tests (price path, membership deltas), the whole GPU suite, the bench line.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r03d
mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_price.py tests/test_gpu_cluster.py -x -q > "$OUT/pytest_new.log" 2>&1; echo "pytest exit $?" >> "$OUT/pytest_new.log"
timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench exit $?" >> "$OUT/bench.err"
timeout 1500 python -m pytest tests -m gpu -q > "$OUT/pytest_gpu.log" 2>&1; echo "pytest exit $?" >> "$OUT/pytest_gpu.log"
tail -25 "$OUT/pytest_new.log"; tail -5 "$OUT/bench.err"; head -c 3000 "$OUT/bench.json"; echo; tail -8 "$OUT/pytest_gpu.log"
