set -u
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06; mkdir -p $OUT
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/cold_trace" -- python $ROOT/bench.py --steps 50 --warmup 5 --headline-only --extras-file $OUT/x.json > "$OUT/cold_trace.log" 2>&1 )
python profiles/summarize.py "$OUT/cold_trace" > "$OUT/cold_trace.summary.csv"
python profiles/per_launch.py "$OUT/cold_trace" k_level_hist | head -12
cat $OUT/cold_trace.summary.csv
rm -rf $OUT/cold_trace
