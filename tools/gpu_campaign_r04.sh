#!/usr/bin/env bash
# closing campaigns of round 4 on the final build: fresh seeds for the three r03 families + the resident-delta family
set -u
OUT=gpurun_out/r04_campaign; mkdir -p $OUT
timeout 900 python tools/gpu_resident_campaign.py 0 80 > $OUT/resident_campaign.txt 2>&1; tail -1 $OUT/resident_campaign.txt
timeout 900 python tools/fuzz_more.py 8000 9600 3000 3400 > $OUT/fuzz_campaign.txt 2>&1; tail -4 $OUT/fuzz_campaign.txt
timeout 600 python tools/gpu_block_campaign.py 500 80 > $OUT/block_campaign.txt 2>&1; tail -1 $OUT/block_campaign.txt
timeout 900 python tools/gpu_price_campaign.py 400 80 > $OUT/price_campaign.txt 2>&1; tail -1 $OUT/price_campaign.txt
