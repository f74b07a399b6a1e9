#!/usr/bin/env bash
# GPU-side steps of round 6, one script:   gpurun --timeout N -- 'bash tools/gpu_r06.sh <step> [...]'
# Everything lands under gpurun_out/r06/; the summaries worth keeping are copied into profiles/r06/ afterwards (profiles/r06/README.md says which call made which file).
set -u
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r06
mkdir -p "$OUT"
HEAD="python $ROOT/bench.py --steps 50 --warmup 5 --headline-only --extras-file $OUT/bench_headline_extras.json"
run_trace() {  # name, command...
  local name=$1; shift
  ( cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name" -- "$@" > "$OUT/$name.log" 2>&1 )
  python profiles/summarize.py "$OUT/$name" > "$OUT/$name.summary.csv" 2>> "$OUT/$name.log"
}
run_pmc() {  # name, counter(s), command...
  local name=$1 c=$2; shift 2
  ( cd /tmp && timeout 500 rocprofv3 --pmc $c --output-format csv -d "$OUT/${name}_${c// /_}" -- "$@" > "$OUT/${name}_${c// /_}.log" 2>&1 )
  python profiles/summarize.py "$OUT/${name}_${c// /_}" > "$OUT/${name}_${c// /_}.summary.csv" 2>> "$OUT/${name}_${c// /_}.log"
}
for step in "$@"; do
case $step in
  tests)     timeout 1500 python -m pytest tests -m gpu -q -x > "$OUT/gpu_suite.full.log" 2>&1; grep -E "passed|failed|error|Error|solver limits" "$OUT/gpu_suite.full.log" | tail -12 | tee "$OUT/gpu_suite.log" ;;
  smoke)     timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee "$OUT/smoke.log" ;;
  bench)     timeout 900 python bench.py --steps 20 --warmup 5 --extras-file "$OUT/bench_extras.json" > "$OUT/bench.json" 2> "$OUT/bench.err"; tail -c 600 "$OUT/bench.err"; python tools/bench_digest.py "$OUT/bench.json" "$OUT/bench_extras.json" ;;
  headline)  timeout 300 $HEAD > "$OUT/bench_headline.json" 2> "$OUT/bench_headline.err"; tail -c 400 "$OUT/bench_headline.err"; python tools/bench_digest.py "$OUT/bench_headline.json" ;;
  trace)     # the headline command under the kernel trace: K1 / k_price_sweep per launch, and the line it printed while traced
             run_trace bench_c3p $HEAD
             python profiles/per_launch.py "$OUT/bench_c3p" k_level_hist > "$OUT/k1_per_launch.txt" 2>&1
             python profiles/per_launch.py "$OUT/bench_c3p" k_price_sweep > "$OUT/sweep_per_launch.txt" 2>&1
             grep '^{' "$OUT/bench_c3p.log" | tail -1 > "$OUT/bench_c3p_under_rocprof.json"; python tools/bench_digest.py "$OUT/bench_c3p_under_rocprof.json"
             cat "$OUT/bench_c3p.summary.csv" ;;
  pmc)       for c in FETCH_SIZE WRITE_SIZE; do run_pmc bench_c3p $c $HEAD; done; head -12 "$OUT"/bench_c3p_FETCH_SIZE.summary.csv "$OUT"/bench_c3p_WRITE_SIZE.summary.csv ;;
  sweepctr)  # what the sweep kernel's waves do: occupancy, VALU / LDS activity of k_price_sweep (own passes, --pmc only)
             run_pmc sweepctr "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS" $HEAD
             run_pmc sweepctr2 "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" $HEAD
             grep -h "price_sweep\|^kernel" "$OUT"/sweepctr*.summary.csv ;;
  stage)     HQTICK_PRICE_PROFILE=1 timeout 300 python tools/price_probe.py c3p wave c4u c4p --no-host --repeat 2 2>&1 | grep -E "price profile|price \{" | tee "$OUT/price_sweep_stage_profile.txt" ;;
  waves)     # k_price_sweep with 1 / 2 / 4 wavefronts per block (HQTICK_PRICE_WAVES): the stage profile of each, then one run without the activity atomics (timing only)
             for w in ${WAVES_LIST:-1 2 4}; do echo "== HQTICK_PRICE_WAVES=$w"; HQTICK_PRICE_WAVES=$w HQTICK_PRICE_PROFILE=1 timeout 300 python tools/price_probe.py c3p c4u c4p --no-host --repeat 2 2>&1 | grep -E "price profile|price \{"; done | tee "$OUT/price_sweep_waves.txt"
             echo "== HQTICK_PRICE_WAVES=4, no activity atomics (timing only)" | tee -a "$OUT/price_sweep_waves.txt"
             HQTICK_PRICE_DBG=1 HQTICK_PRICE_WAVES=4 HQTICK_PRICE_PROFILE=1 timeout 300 python tools/price_probe.py c3p --no-host --repeat 2 2>&1 | grep -E "price profile" | tee -a "$OUT/price_sweep_waves.txt" ;;
  blocktests) timeout 900 python -m pytest tests/test_gpu_blocks.py -m gpu -q -x 2>&1 | grep -E "passed|failed|error|Error" | tail -6 | tee "$OUT/gpu_block_tests.log" ;;
  heterotrace) # the kernels of the heterogeneous steady-state loop (136 worker classes per tick through k_block_solve)
             run_trace bench_hetero python $ROOT/bench.py --steps 2 --warmup 1 --hetero-steps 25 --dag-steps 0 --priority-ticks 0 --cpu-ticks 0 --steady-steps 0 --wire-iters 0
             cat "$OUT/bench_hetero.summary.csv" ;;
  hosttrace) # the host side of the coupled tick, stage by stage, on this box's cores (HQMILP_TRACE marks; the last of three ticks)
             HQMILP_TRACE=1 timeout 300 python tools/price_probe.py c3p --no-host --repeat 3 2>&1 | grep -E "^\[(price|milp|model)\]" > "$OUT/host_trace_c3p.txt"; tail -24 "$OUT/host_trace_c3p.txt" ;;
  coupled)   timeout 600 python tools/price_probe.py c3p wave 0.2 0.45 --no-host --timeline --repeat 1 2>&1 | grep -E "timeline|price " | tee "$OUT/coupled_ticks.txt" ;;
  pricetests) timeout 900 python -m pytest tests/test_gpu_price.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | grep -E "passed|failed|error|Error|solver limits" | tail -6 | tee "$OUT/gpu_price_tests.log" ;;
  sharded)   # the sharded code path with one rank (record sink + the library's collective + D2H of the merged vector), on the headline workload
             timeout 300 python bench.py --steps 10 --warmup 3 --headline-only --force-sharded > "$OUT/bench_force_sharded.json" 2> "$OUT/bench_force_sharded.err"; tail -c 300 "$OUT/bench_force_sharded.err"; python tools/bench_digest.py "$OUT/bench_force_sharded.json" | head -4 ;;
  preflight) timeout 200 python bench.py --gpus 1 --preflight 2>&1 | grep -E "^\{|Error|error" | tail -3 | tee "$OUT/preflight.log" ;;
  campaign)  # fresh seeds for the families of rounds 3 / 4 on this round's build: coupled ticks at cluster scale (GPU tick vs emulated sweeps vs oracle mapping),
             # resident-delta scenarios, class blocks on the device, the fuzz family through the tick
             timeout 500 python tools/gpu_price_campaign.py 900 60 > "$OUT/price_campaign.txt" 2>&1; tail -2 "$OUT/price_campaign.txt"
             timeout 400 python tools/gpu_resident_campaign.py 500 60 > "$OUT/resident_campaign.txt" 2>&1; tail -1 "$OUT/resident_campaign.txt"
             timeout 300 python tools/gpu_block_campaign.py 900 60 > "$OUT/block_campaign.txt" 2>&1; tail -1 "$OUT/block_campaign.txt"
             timeout 400 python tools/fuzz_more.py 12000 12800 > "$OUT/fuzz_campaign.txt" 2>&1; tail -3 "$OUT/fuzz_campaign.txt" ;;
  clean)     find "$OUT" -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} + ;;
  *)         echo "unknown step $step" ;;
esac
done
find "$OUT" -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} + 2>/dev/null
true
