# slice size (HQTICK_TPW) against the cold tick of c3: K1 / K1b like larger slices, K4 smaller ones (tools/exp/README.md)
mkdir -p gpurun_out
for t in 256 512 1024 2048; do
  echo "=== TPW $t timed"; HQTICK_TPW=$t timeout 120 python tools/timeline.py c3 300 --timed 2>&1 | tail -3
  echo "=== TPW $t untimed"; HQTICK_TPW=$t timeout 120 python tools/timeline.py c3 300 2>&1 | grep -E "phaseA|C_synced|total"
done
