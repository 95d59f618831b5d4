set -e
mkdir -p gpurun_out
: > gpurun_out/variants.txt
for v in "$@"; do
  TEELINE_GPU_LIB=$PWD/build_variants/$v.so timeout -k 10 150 python scripts/variant_timing.py >> gpurun_out/variants.txt 2>&1
done
cat gpurun_out/variants.txt
