#!/bin/bash
# Secondary evidence: kernel stats of the bench extras (matrix build, matrix 2-opt, 3-opt / Or-opt scans, NN seed, best-sweep),
# LK kernel stats, and the HBM counters of the matrix build (the HBM-bound kernel of the path).
# Usage (repo root, GPU box): bash scripts/profile_extras.sh r02     — steps chained with &&, each with its own timeout
R=${1:-r02}
OUT=$PWD/gpurun_out/${R}_extras
mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/extras -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end > $OUT/extras.log 2>&1 \
&& timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/lk -- python3 $REPO/scripts/lk_profile.py > $OUT/lk.log 2>&1 \
&& timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dm_stats -- python3 $REPO/scripts/dm_build_once.py > $OUT/dm_stats.log 2>&1 \
&& timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/dm_write -- python3 $REPO/scripts/dm_build_once.py > $OUT/dm_write.log 2>&1 \
&& timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/dm_fetch -- python3 $REPO/scripts/dm_build_once.py > $OUT/dm_fetch.log 2>&1 \
&& python3 $REPO/scripts/lk_profile.py > $OUT/lk_plain.log 2>&1 && N=13509 python3 $REPO/scripts/timing_lk_large.py > $OUT/lk_large.log 2>&1; echo "rc=$?"; tail -n 3 $OUT/lk_large.log
