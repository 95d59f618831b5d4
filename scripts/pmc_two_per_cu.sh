#!/bin/bash
# VALU / SALU instruction counts and kernel time of ONE 512-restart launch (two descents per CU, grid-coordinate form)
OUT=$PWD/gpurun_out/pmc512
mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $OUT/pmc -- python3 $REPO/bench.py --restarts 512 --steps 1 --warmup 0 --no-cpu-baseline --no-extras --no-work-count > $OUT/log.txt 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections, json
tot = collections.Counter(); name = set()
for f in glob.glob(sys.argv[1] + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "two_opt_ref_lds" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); name.add(r["Kernel_Name"][:60])
print(name)
for k, v in sorted(tot.items()): print(f"{k:24s} {v:.5g}")
line = [l for l in open(sys.argv[1] + "/log.txt") if l.startswith("{")][-1]
d = json.loads(line)
print("ms_per_step", d["ms_per_step"], "value", d["value"], "clock MHz", d["roofline"]["clock_mhz_live"], "kernel_ms", d["roofline"]["kernel_ms_avg"])
if "SQ_INSTS_VALU" in tot:
    print("VALU issue frac", tot["SQ_INSTS_VALU"] / (1024 * d["roofline"]["kernel_ms_avg"] * 1e-3 * d["roofline"]["clock_mhz_live"] * 1e6 / 2))
PY
