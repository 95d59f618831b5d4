#!/bin/bash
# rocprofv3 kernel stats of the two forms of the LK scan at n = 13 509 / 20 epochs ("if measured and rejected,
# commit the kernel-stats CSV that shows it"): one workgroup per pair (the product library) and the persistent grid (tuning
# library, TL_FLAG_LK_SCAN_PERSIST = 1 << 17).   bash scripts/profile_lk_forms.sh r04
R=${1:-r04}
OUT=$PWD/gpurun_out/$R/lk_forms
mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/per_pair -- python3 $REPO/scripts/timing_lk_once.py > $OUT/per_pair.log 2>&1 \
&& export TEELINE_GPU_LIB=$REPO/teeline_amd/libteeline_gpu_tune.so TL_CREATE_FLAGS=131072 \
&& timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/persist -- python3 $REPO/scripts/timing_lk_once.py > $OUT/persist.log 2>&1 \
&& TL_LK_PERSIST_BLOCKS=9000 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/persist9000 -- python3 $REPO/scripts/timing_lk_once.py > $OUT/persist9000.log 2>&1
grep -h "us/round" $OUT/*.log
for d in per_pair persist persist9000; do f=$(find $OUT/$d -name '*kernel_stats.csv' | head -1); echo "== $d"; head -6 $f; done
