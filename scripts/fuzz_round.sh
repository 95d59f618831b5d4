#!/bin/bash
# The randomized oracle campaigns on the round's final tree, product library and race-stress build (one GPU box; ~17 min).
#   bash scripts/fuzz_round.sh r04 [part]      part 1: product library, part 2: jitter build + threads
R=${1:-r04}; PART=${2:-1}
OUT=gpurun_out/$R; mkdir -p $OUT
P=tests/probes
if [ "$PART" = "1" ]; then
{ echo "== product library"; timeout -k 5 400 python $P/fuzz_campaign.py 240 | tail -1; FUZZ_BIG=1 timeout -k 5 300 python $P/fuzz_campaign.py 150 | tail -1;
  timeout -k 5 300 python $P/fuzz_campaign_lk.py 150 | tail -1; FUZZ_DEEP=1 timeout -k 5 300 python $P/fuzz_campaign_lk.py 120 | tail -1;
  timeout -k 5 300 python $P/fuzz_campaign_oropt.py 100 | tail -1; timeout -k 5 400 python $P/fuzz_campaign_trace.py 120 | tail -1; } > $OUT/fuzz_part1.txt 2>&1
cat $OUT/fuzz_part1.txt
else
export TEELINE_GPU_LIB=$PWD/teeline_amd/libteeline_gpu_jitter.so
{ echo "== race-stress build (libteeline_gpu_jitter.so)"; timeout -k 5 400 python $P/fuzz_campaign.py 240 | tail -1; timeout -k 5 300 python $P/fuzz_campaign_lk.py 150 | tail -1;
  FUZZ_DEEP=1 timeout -k 5 300 python $P/fuzz_campaign_lk.py 90 | tail -1; timeout -k 5 300 python $P/fuzz_campaign_oropt.py 90 | tail -1;
  timeout -k 5 300 python $P/thread_campaign.py 90 8 | tail -3; } > $OUT/fuzz_part2.txt 2>&1
unset TEELINE_GPU_LIB
{ echo "== threads, product library"; timeout -k 5 300 python $P/thread_campaign.py 90 8 | tail -3; } >> $OUT/fuzz_part2.txt 2>&1
cat $OUT/fuzz_part2.txt
fi
