#!/bin/bash
# SQ counters of the scan kernels bench.py's extras report (3-opt scan n = 1002, Or-opt scan n = 5000, LK n = 13 509, a BEST_SWEEP descent and the NO_PRUNE batch at n = 10^4): one rocprofv3
# --pmc pass each (+ --kernel-trace only), summarised into gpurun_out/<round>/scans_pmc.json (copy it to profiles/<round>_scans_pmc.json).
#   bash scripts/pmc_scans.sh r03
R=${1:-r03}
OUT=$PWD/gpurun_out/$R/scans
mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
CNT="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CU_CYCLES SQ_WAVES"
timeout -k 10 200 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $OUT/scans -- python3 $REPO/scripts/scan_once.py scans > $OUT/scans.log 2>&1 \
&& timeout -k 10 300 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $OUT/lk -- python3 $REPO/scripts/scan_once.py lk > $OUT/lk.log 2>&1 \
&& timeout -k 10 200 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $OUT/best -- python3 $REPO/scripts/scan_once.py best > $OUT/best.log 2>&1 \
&& timeout -k 10 200 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $OUT/noprune -- python3 $REPO/scripts/scan_once.py noprune > $OUT/noprune.log 2>&1
python3 - $OUT $OUT/../scans_pmc.json <<'PY'
import csv, glob, json, sys, collections
out = {}
def load(sub):
    rows = []
    for f in glob.glob(sys.argv[1] + f"/{sub}/**/*counter_collection.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    return rows
def trace(sub):
    d = {}
    for f in glob.glob(sys.argv[1] + f"/{sub}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            d[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    return d
rows, tr = load("scans"), trace("scans")
for key, n in (("k_three_opt_scan", 1002), ("k_or_scan", 5000)):
    sel = [r for r in rows if key in r["Kernel_Name"]]
    if not sel: continue
    last = max(int(r["Dispatch_Id"]) for r in sel)          # the second (warm) launch
    e = {"n": n, "dispatch": last, "kernel_ms_profiled": tr.get(str(last))}
    for r in sel:
        if int(r["Dispatch_Id"]) == last: e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    out[key] = e
rows, tr = load("lk"), trace("lk")
for key in ("k_lk_scan_sub", "k_lk_control"):
    sel = [r for r in rows if key in r["Kernel_Name"]]
    if not sel: continue
    ids = sorted({int(r["Dispatch_Id"]) for r in sel})
    e = {"n": 13509, "launches": len(ids), "kernel_ms_profiled_mean": sum(tr.get(str(i), 0.0) for i in ids) / len(ids), "note": "per-launch means over the run"}
    tot = collections.Counter()
    for r in sel: tot[r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in tot.items(): e[k] = v / len(ids)
    out[key] = e
rows, tr = load("best"), trace("best")
for key in ("k_bs_scan", "k_bs_apply"):
    sel = [r for r in rows if key in r["Kernel_Name"]]
    if not sel: continue
    ids = sorted({int(r["Dispatch_Id"]) for r in sel})
    e = {"n": 10000, "launches": len(ids), "kernel_ms_profiled_mean": sum(tr.get(str(i), 0.0) for i in ids) / len(ids), "note": "per-launch means over the descent (one launch per sweep)"}
    tot = collections.Counter()
    for r in sel: tot[r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in tot.items(): e[k] = v / len(ids)
    out[key] = e
rows, tr = load("noprune"), trace("noprune")
sel = [r for r in rows if "k_two_opt_ref_lds" in r["Kernel_Name"]]
if sel:
    last = max(int(r["Dispatch_Id"]) for r in sel)
    e = {"n": 10000, "restarts": 256, "seed": 12345, "dispatch": last, "kernel_ms_profiled": tr.get(str(last)), "kernel_name": [r["Kernel_Name"] for r in sel if int(r["Dispatch_Id"]) == last][0]}
    for r in sel:
        if int(r["Dispatch_Id"]) == last: e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    out["k_two_opt_ref_lds_no_prune"] = e
out["source"] = "rocprofv3 --pmc SQ_* --kernel-trace on scripts/scan_once.py (scripts/pmc_scans.sh); summed over the dispatch; the scan kernels: second launch"
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])
PY
tail -n 3 $OUT/scans.log; tail -n 3 $OUT/lk.log; tail -n 2 $OUT/best.log; tail -n 2 $OUT/noprune.log
