#!/bin/bash
# Round-end evidence run on the GPU box: tests, bench line, rocprofv3 kernel stats and the two PMC passes.
# Usage (from the repo root on the box): bash scripts/profile_round.sh r01
R=${1:-r01}
OUT=$PWD/gpurun_out/$R
mkdir -p $OUT
REPO=$PWD
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
python bench.py --steps 3 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err; tail -1 $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/pmc_sq1 -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $OUT/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_LDS_ATOMIC SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc_sq2 -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $OUT/pmc_sq2.log 2>&1
find $OUT -name '*.csv' | head -20
