#!/usr/bin/env bash
# rocprofv3 evidence of round 2, one gpurun call:   gpurun --timeout 1500 -- 'bash tools/profile_r02.sh'
# Everything lands under gpurun_out/r02_prof/; the summaries are copied into profiles/r02/ afterwards.
# Counters are collected in their own passes (--pmc only; no trace domain next to it).
set -u
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r02_prof
mkdir -p "$OUT"
# the driver's bench command, minus the extra blocks (so that rocprofv3's per-kernel averages are over the ticks of the timed region and the stats pass only)
BENCH="python $ROOT/bench.py --steps 50 --warmup 5 --no-b2b --no-roofline-sweep --steady-steps 0 --hetero-steps 0 --dag-steps 0 --priority-ticks 0 --wire-iters 0 --cpu-ticks 0"
run_trace() {  # name, command...
  local name=$1; shift
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name" -- "$@" > "$OUT/$name.log" 2>&1 )
  python profiles/summarize.py "$OUT/$name" > "$OUT/$name.summary.csv" 2>> "$OUT/$name.log"
}
run_pmc() {  # name, counter, command...
  local name=$1 c=$2; shift 2
  ( cd /tmp && timeout 400 rocprofv3 --pmc "$c" --output-format csv -d "$OUT/${name}_$c" -- "$@" > "$OUT/${name}_$c.log" 2>&1 )
  python profiles/summarize.py "$OUT/${name}_$c" > "$OUT/${name}_$c.summary.csv" 2>> "$OUT/${name}_$c.log"
}
# 1. the bench command: kernel trace, then HBM traffic counters
run_trace bench_c3 $BENCH
grep '^{' "$OUT/bench_c3.log" | tail -1 > "$OUT/bench_c3_under_rocprof.json"
for c in FETCH_SIZE WRITE_SIZE; do run_pmc bench_c3 $c $BENCH; done
# 1b. 200 cold ticks that ALL carry the dispatch events (what bench.py's stats pass does), and 200 that carry none (what its timed region does): the bench
# command above mixes 55 ticks without events and 50 with them, and a K1 launched without a start event is the first packet the GPU sees after the host's
# writes — its recorded duration then includes the acquire at the head of the queue (7.5 us against 4.6 us)
run_trace ticks_with_events python $ROOT/tools/timeline.py c3 200 --timed
run_trace ticks_without_events python $ROOT/tools/timeline.py c3 200
# (PROFILE_PARTS=tick: only the tick's own traces above — enough after a change to one of the tick's kernels)
if [ "${PROFILE_PARTS:-all}" = "tick" ]; then find "$OUT" -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +; ls -la "$OUT"; exit 0; fi
# 2. the streaming kernels beyond the Infinity Cache: 16 M and 64 M ready tasks (tools/ktime.py: 3 ticks + back-to-back launches of K1 / K4)
for n in 16000000 64000000; do
  run_trace ktime_$n python $ROOT/tools/ktime.py c3 20 $n
  for c in FETCH_SIZE WRITE_SIZE; do run_pmc ktime_$n $c python $ROOT/tools/ktime.py c3 5 $n; done
done
# 3. the steady-state tick (k_block_solve: one wavefront per worker class)
run_trace steady_c3 python $ROOT/tools/steady_probe.py c3 20
run_trace steady_c4 python $ROOT/tools/steady_probe.py c4 10
# 4. the wire encoding (row f3)
run_trace wire python $ROOT/tools/wire_bench.py --iters 50
for c in FETCH_SIZE WRITE_SIZE; do run_pmc wire $c python $ROOT/tools/wire_bench.py --iters 10; done
# keep only the condensed files (the raw traces are hundreds of MB)
find "$OUT" -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
ls -la "$OUT"
