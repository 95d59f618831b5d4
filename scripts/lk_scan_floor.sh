#!/bin/bash
# scan kernel averages for timing-experiment builds (wrong results): bash scripts/lk_scan_floor.sh v1 v2
for v in "$@"; do
  export TEELINE_GPU_LIB=$GRAFT_REPO_ROOT/build_variants/$v.so
  (cd /tmp && TMPDIR=/tmp timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/floor_$v -- python3 $GRAFT_REPO_ROOT/scripts/lk_profile.py > $GRAFT_REPO_ROOT/gpurun_out/floor_$v.log 2>&1)
  echo "== $v"; cat $GRAFT_REPO_ROOT/gpurun_out/floor_$v/*/*kernel_stats.csv | cut -c1-130 | sed -n 2,3p
done
