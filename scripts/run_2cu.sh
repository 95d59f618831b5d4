mkdir -p gpurun_out/r04b
OUT=gpurun_out/r04b/$1.txt; shift
: > $OUT
for v in "$@"; do echo "== $v" >> $OUT; TEELINE_GPU_LIB=$PWD/build_variants/$v.so timeout -k 10 200 python scripts/two_per_cu_probe.py 2>&1 | grep "n=10000" >> $OUT; done
cat $OUT
