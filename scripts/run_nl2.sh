# timing of build variants (default flags), one line each
mkdir -p gpurun_out/r04b
OUT=gpurun_out/r04b/$1.txt; shift
: > $OUT
for v in "$@"; do
  TEELINE_GPU_LIB=$PWD/build_variants/$v.so FLAGS=0 timeout -k 10 150 python scripts/variant_timing.py >> $OUT 2>&1
done
grep -v amdgpu.ids $OUT
