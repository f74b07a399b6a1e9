#!/usr/bin/env bash
set -u
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r04_call15
mkdir -p "$OUT"
for c in "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "SQ_INSTS_LDS SQ_INSTS_VALU" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" "SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE"; do
  n=$(echo $c | tr ' ' '_')
  ( cd /tmp && timeout 300 rocprofv3 --pmc $c --output-format csv -d "$OUT/$n" -- python $ROOT/tools/price_probe.py c3p --no-host --repeat 1 > "$OUT/$n.log" 2>&1 )
  f=$(find "$OUT/$n" -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    if "price_sweep" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k, "launches", len(v), "mean per launch", sum(v) / len(v))
PY
done
find "$OUT" -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
