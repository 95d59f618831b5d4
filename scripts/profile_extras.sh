#!/bin/bash
# Secondary evidence: kernel stats of the bench extras (matrix build, matrix 2-opt, 3-opt / Or-opt scans, NN seed, best-sweep),
# LK kernel stats, and the HBM counters of the matrix build (the HBM-bound kernel of the path).
# Usage (repo root, GPU box): bash scripts/profile_extras.sh r01
R=${1:-r01}
OUT=$PWD/gpurun_out/${R}_extras
mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/extras -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/extras.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/lk -- python3 $REPO/scripts/lk_profile.py > $OUT/lk.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dm_stats -- python3 $REPO/scripts/dm_build_once.py > $OUT/dm_stats.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/dm_write -- python3 $REPO/scripts/dm_build_once.py > $OUT/dm_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/dm_fetch -- python3 $REPO/scripts/dm_build_once.py > $OUT/dm_fetch.log 2>&1
find $OUT -name '*.csv' | head -30
