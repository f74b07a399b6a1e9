set -u
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06; mkdir -p $OUT
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/warm_trace" -- python $ROOT/bench.py --steps 50 --warmup 5 --headline-only --warm-caches --extras-file $OUT/x.json > "$OUT/warm_trace.log" 2>&1 )
python profiles/summarize.py "$OUT/warm_trace" > "$OUT/warm_trace.summary.csv"
python profiles/per_launch.py "$OUT/warm_trace" k_level_hist | head -12
cat $OUT/warm_trace.summary.csv
rm -rf $OUT/warm_trace
