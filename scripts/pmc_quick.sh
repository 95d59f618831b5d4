#!/bin/bash
# Instruction-mix counters of the headline kernel only (one launch): bash scripts/pmc_quick.sh <tag>
T=${1:-x}
OUT=$PWD/gpurun_out/pmcq_$T
mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $OUT/pmc -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --no-work-count > $OUT/log.txt 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
tot = collections.Counter()
for f in glob.glob(sys.argv[1] + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "two_opt_ref_lds" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(tot.items()): print(f"{k:24s} {v:.4g}")
PY
tail -c 400 $OUT/log.txt
