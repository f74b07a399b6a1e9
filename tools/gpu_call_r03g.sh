#!/usr/bin/env bash
# price path after the disaggregated master: device == emulation tests, fixtures, probe
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r03g
mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_price.py tests/test_fixtures.py -m gpu -x -q > "$OUT/pytest_price.log" 2>&1; echo "pytest exit $?" >> "$OUT/pytest_price.log"
timeout 600 python tools/price_probe.py c3p wave 0.2 0.45 --no-host --timeline > "$OUT/price_probe.log" 2>&1
tail -4 "$OUT/pytest_price.log"; grep -v amdgpu.ids "$OUT/price_probe.log" | tail -12
