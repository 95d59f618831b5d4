#!/bin/bash
# Instruction-cache counters of the headline kernel (one launch): bash scripts/pmc_icache.sh <tag>
T=${1:-x}
OUT=$PWD/gpurun_out/pmci_$T
mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/pmc -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --no-work-count > $OUT/log.txt 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
tot = collections.Counter()
for f in glob.glob(sys.argv[1] + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "two_opt_ref_lds" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(tot.items()): print(f"{k:28s} {v:.5g}")
PY
tail -c 300 $OUT/log.txt
