#!/bin/bash
# bash scripts/run_lk_variants.sh v1 v2 ...: LK wall time at n = 13 509 (20 epochs) for build_variants/<v>.so
for v in "$@"; do
  echo "== $v"
  TEELINE_GPU_LIB=$PWD/build_variants/$v.so N=13509 timeout -k 10 150 python scripts/timing_lk_large.py 2>/dev/null | tail -1
done
