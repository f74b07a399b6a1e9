#!/usr/bin/env bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r03e
mkdir -p "$OUT"
HQTICK_PRICE_PROFILE=1 timeout 600 python tools/price_probe.py c3p wave 0.2 --no-host --repeat 2 > "$OUT/price_probe_profile.log" 2>&1
timeout 600 python tools/price_probe.py c3p wave 0.2 0.45 --no-host --timeline > "$OUT/price_probe.log" 2>&1
timeout 600 python -m pytest tests/test_gpu_price.py -x -q > "$OUT/pytest_price.log" 2>&1; echo "pytest exit $?" >> "$OUT/pytest_price.log"
grep -v amdgpu.ids "$OUT/price_probe_profile.log"; grep -v amdgpu.ids "$OUT/price_probe.log"; tail -4 "$OUT/pytest_price.log"
