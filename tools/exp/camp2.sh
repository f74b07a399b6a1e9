set -u
export TMPDIR=/tmp
OUT=gpurun_out/r06/camp2; mkdir -p $OUT
timeout 700 python tools/gpu_price_campaign.py ${CAMP_SEED:-3100} 80 > $OUT/price_campaign.txt 2>&1; tail -2 $OUT/price_campaign.txt
timeout 500 python tools/gpu_resident_campaign.py ${CAMP_SEED:-3100} 80 > $OUT/resident_campaign.txt 2>&1; tail -1 $OUT/resident_campaign.txt
timeout 400 python tools/gpu_block_campaign.py ${CAMP_SEED:-3100} 80 > $OUT/block_campaign.txt 2>&1; tail -1 $OUT/block_campaign.txt
timeout 500 python tools/fuzz_more.py ${CAMP_FUZZ:-31000} $(( ${CAMP_FUZZ:-31000} + 1000 )) > $OUT/fuzz_campaign.txt 2>&1; tail -3 $OUT/fuzz_campaign.txt
