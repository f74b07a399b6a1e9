set -u
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06; mkdir -p $OUT
for mode in spec nospec; do
  if [ $mode = nospec ]; then export HQTICK_NO_SPEC_SCAN=1; else unset HQTICK_NO_SPEC_SCAN; fi
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/cold_$mode" -- python $ROOT/bench.py --steps 50 --warmup 5 --headline-only --extras-file $OUT/x.json > "$OUT/cold_$mode.log" 2>&1 )
  python profiles/summarize.py "$OUT/cold_$mode" | grep "level_hist\|distinct\|sort_levels"
  python profiles/per_launch.py "$OUT/cold_$mode" k_level_hist | sed -n 2,4p
  rm -rf $OUT/cold_$mode
  for i in 1 2; do python bench.py --steps 100 --warmup 10 --headline-only --extras-file $OUT/x.json 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$mode', 'p50', d['config']['p50_tick_ms'], 'ms/step', d['ms_per_step'], 'K1 us', d['roofline']['avg_launch_us'])"; done
done
