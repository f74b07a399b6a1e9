#!/usr/bin/env bash
# round-3 verification call: whole GPU suite, the price probe, the default bench
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r03_verify
mkdir -p "$OUT"
timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; echo "pytest exit $?" >> "$OUT/pytest_gpu.log"
timeout 600 python tools/price_probe.py c3p wave 0.2 0.45 --no-host --timeline > "$OUT/price_probe.log" 2>&1
timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench exit $?" >> "$OUT/bench.err"
tail -4 "$OUT/pytest_gpu.log"; grep -v amdgpu.ids "$OUT/price_probe.log" | tail -12; tail -c 3000 "$OUT/bench.json"; tail -2 "$OUT/bench.err"
