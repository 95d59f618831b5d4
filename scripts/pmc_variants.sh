#!/bin/bash
# bash scripts/pmc_variants.sh v1 v2 ...: instruction-mix counters + kernel time of the headline kernel for build_variants/<v>.so
for v in "$@"; do
  echo "== $v"
  if [ "$v" = "default" ]; then unset TEELINE_GPU_LIB; else export TEELINE_GPU_LIB=$PWD/build_variants/$v.so; fi
  bash scripts/pmc_quick.sh $v | head -8
  timeout -k 10 150 python scripts/variant_timing.py 2>/dev/null | tail -1
done
