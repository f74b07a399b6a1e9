#!/usr/bin/env bash
# rocprofv3 evidence of round 4, one gpurun call at the END of the round:   gpurun --timeout 1800 -- 'bash tools/profile_r04.sh'
# Everything lands under gpurun_out/r04_prof/; the summaries are copied into profiles/r04/ afterwards.  Counters in their own passes (--pmc only).
set -u
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r04_prof
mkdir -p "$OUT"
BENCH="python $ROOT/bench.py --steps 50 --warmup 5 --no-roofline-sweep --steady-steps 0 --hetero-steps 0 --dag-steps 0 --priority-ticks 0 --wire-iters 0 --cpu-ticks 0"
run_trace() {  # name, command...
  local name=$1; shift
  ( cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name" -- "$@" > "$OUT/$name.log" 2>&1 )
  python profiles/summarize.py "$OUT/$name" > "$OUT/$name.summary.csv" 2>> "$OUT/$name.log"
}
run_pmc() {  # name, counter, command...
  local name=$1 c=$2; shift 2
  ( cd /tmp && timeout 500 rocprofv3 --pmc "$c" --output-format csv -d "$OUT/${name}_$c" -- "$@" > "$OUT/${name}_$c.log" 2>&1 )
  python profiles/summarize.py "$OUT/${name}_$c" > "$OUT/${name}_$c.summary.csv" 2>> "$OUT/${name}_$c.log"
}
# 0. the unprofiled bench line of the same build (the driver's command), and the dispatch-floor micro-benchmark
timeout 900 python bench.py > "$OUT/bench_unprofiled.json" 2> "$OUT/bench_unprofiled.err"
[ -x tools/exp/bin/dispatch_floor ] && tools/exp/bin/dispatch_floor > "$OUT/dispatch_floor.txt" 2>&1
# 1. the headline command (c3 cold ticks): kernel trace + HBM traffic
run_trace bench_c3 $BENCH
python profiles/per_launch.py "$OUT/bench_c3" k_level_hist > "$OUT/k1_per_launch.txt" 2>&1
python profiles/per_launch.py "$OUT/bench_c3" k_expand_mapping > "$OUT/k5b_per_launch.txt" 2>&1
grep '^{' "$OUT/bench_c3.log" | tail -1 > "$OUT/bench_c3_under_rocprof.json"
for c in FETCH_SIZE WRITE_SIZE; do run_pmc bench_c3 $c $BENCH; done
# 2. the coupled ticks of round 4 (k_price_sweep): c3p at BASELINE size, the config-5 first wave, the unsaturated probes — resident ticks, 12 each
run_trace coupled python $ROOT/tools/price_probe.py c3p wave 0.2 0.45 --no-host --timeline --repeat 1
grep -v amdgpu.ids "$OUT/coupled.log" | grep -E "timeline|price " > "$OUT/coupled_ticks.txt"
for c in FETCH_SIZE WRITE_SIZE; do run_pmc coupled $c python $ROOT/tools/price_probe.py c3p wave --no-host --repeat 2; done
HQTICK_PRICE_PROFILE=1 timeout 300 python tools/price_probe.py c3p wave 0.2 --no-host --repeat 2 2>&1 | grep -E "price profile|price \{" > "$OUT/price_sweep_stage_profile.txt"
# 3. the steady-state tick (k_block_solve) and the wire encoding, unchanged kernels: one trace each for the record
run_trace steady_c3 python $ROOT/tools/steady_probe.py c3 20
run_trace wire python $ROOT/tools/wire_bench.py --iters 50
find "$OUT" -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
ls -la "$OUT"
# 4. the whole GPU suite on this build (what the driver runs at round end)
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > "$OUT/gpu_suite.log"; cat "$OUT/gpu_suite.log"
