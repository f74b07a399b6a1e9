# A/B of the slowest-first order of a > 1536-block sweep (price.hip: k_order_blocks): bash tools/exp/lpt_ab.sh
for v in 0 1; do
  if [ $v = 1 ]; then export HQTICK_PRICE_NO_LPT=1; echo "== no LPT"; else unset HQTICK_PRICE_NO_LPT; echo "== LPT"; fi
  python tools/price_probe.py c4u c4p --no-host --repeat 3 2>&1 | grep -E "price \{" | cut -c1-260
  HQTICK_PRICE_PROFILE=1 python tools/price_probe.py c4u c4p --no-host --repeat 2 2>&1 | grep -E "price profile" | sed -E 's/.*(first block.s start -> last block.s results [0-9.]+, -> completion word stored [0-9.]+ more).*(sweep as the host saw it [0-9.]+ us).*/\1; \2/'
done
