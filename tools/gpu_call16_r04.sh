#!/usr/bin/env bash
# occupancy experiment: the same counters on the library as committed (40.8 KB of LDS per block) and on a variant with 32 KB (built beside it as libhqtick_v32.so)
set -u
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r04_call16
mkdir -p "$OUT"
run() {  # tag
  for c in "SQ_LEVEL_WAVES SQ_BUSY_CYCLES" "SQ_WAVES GRBM_GUI_ACTIVE" "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES"; do
    n=$1_$(echo $c | tr ' ' '_')
    ( cd /tmp && timeout 300 rocprofv3 --pmc $c --output-format csv -d "$OUT/$n" -- python $ROOT/tools/exp/unsat4096.py 3 > "$OUT/$n.log" 2>&1 )
    f=$(find "$OUT/$n" -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python - "$f" "$1" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    if "price_sweep" in r["Kernel_Name"] and int(r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", 0)) >= 200000:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(sys.argv[2], k, "launches", len(v), "mean per launch", round(sum(v) / len(v), 1))
PY
  done
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$1_trace" -- python $ROOT/tools/exp/unsat4096.py 3 > "$OUT/$1_trace.log" 2>&1 )
  python profiles/summarize.py "$OUT/$1_trace" | grep price_sweep
}
run base
cp hyperqueue_amd/libhqtick.so /tmp/libhqtick_base.so; cp hyperqueue_amd/libhqtick_v32.so hyperqueue_amd/libhqtick.so
run v32
cp /tmp/libhqtick_base.so hyperqueue_amd/libhqtick.so
find "$OUT" -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
