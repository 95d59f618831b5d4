#!/bin/bash
# Round evidence run on the GPU box: tests, bench line, rocprofv3 kernel stats and the PMC passes (each in its own run).
# Usage (from the repo root on the box): bash scripts/profile_round.sh r02 [skip-tests]
# Steps are chained with && (a failed or timed-out GPU step ends the run) and each has its own timeout.
R=${1:-r02}
OUT=$PWD/gpurun_out/$R
mkdir -p $OUT
REPO=$PWD
BENCH_PMC="python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --no-work-count"
step() { echo "[profile_round] $(date +%T) $1"; }
{
  if [ "$2" != "skip-tests" ]; then
    step "pytest -m gpu" && timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log; [ $rc -eq 0 ]
  fi
} && step "bench" && timeout -k 10 300 python bench.py --steps 3 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err && cp bench_extras.json $OUT/bench_extras.json && tail -n 1 $OUT/bench.json | wc -c && tail -n 1 $OUT/bench.json && cd /tmp && export TMPDIR=/tmp \
&& step "kernel stats" && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-work-count > $OUT/stats.log 2>&1 \
&& step "pmc fetch" && timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $BENCH_PMC > $OUT/pmc_fetch.log 2>&1 \
&& step "pmc write" && timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $BENCH_PMC > $OUT/pmc_write.log 2>&1 \
&& step "pmc sq1" && timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/pmc_sq1 -- $BENCH_PMC > $OUT/pmc_sq1.log 2>&1 \
&& step "pmc sq2" && timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_LDS_ATOMIC SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc_sq2 -- $BENCH_PMC > $OUT/pmc_sq2.log 2>&1 \
&& step "done" && find $OUT -name '*.csv' | head -20
