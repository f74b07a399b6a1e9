#!/usr/bin/env bash
# steady-state tick: stage timeline with device blocks (c3, c4)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r03h
mkdir -p "$OUT"
timeout 600 python tools/steady_probe.py c3 30 > "$OUT/steady_c3.log" 2>&1
timeout 600 python tools/steady_probe.py c4 20 > "$OUT/steady_c4.log" 2>&1
grep -v amdgpu.ids "$OUT/steady_c3.log"; grep -v amdgpu.ids "$OUT/steady_c4.log"
