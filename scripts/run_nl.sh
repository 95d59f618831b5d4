# NL variants against the tile-only instantiation of the same library on the same box
mkdir -p gpurun_out/r04b
OUT=gpurun_out/r04b/$1.txt; shift
: > $OUT
for v in "$@"; do
  export TEELINE_GPU_LIB=$PWD/build_variants/$v.so
  FLAGS=262144 timeout -k 10 150 python scripts/variant_timing.py >> $OUT 2>&1
  FLAGS=0 timeout -k 10 150 python scripts/variant_timing.py >> $OUT 2>&1
  FLAGS=0 timeout -k 10 150 python scripts/descent_balance.py >> $OUT 2>&1
done
grep -v amdgpu.ids $OUT
