"""Copy the judged summaries of a scripts/profile_round.sh run from gpurun_out/<round>/ into profiles/ (tracked)."""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
G = os.path.join(ROOT, "gpurun_out", rnd)
P = os.path.join(ROOT, "profiles")
KERNEL = "k_two_opt_ref_lds"


def one(pattern):
    f = glob.glob(os.path.join(G, pattern))
    return f[0] if f else None


def filt(src, dst):
    rows = list(csv.reader(open(src)))
    with open(dst, "w", newline="") as fh:
        w = csv.writer(fh, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(rows[0])
        for r in rows[1:]:
            if any(KERNEL in c for c in r):
                w.writerow(r)


def counters(src):
    d = {}
    for r in csv.DictReader(open(src)):
        if KERNEL in r["Kernel_Name"]:
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return d


ks = one("stats/*/*kernel_stats.csv")
if ks:
    open(os.path.join(P, f"{rnd}_bench_kernel_stats.csv"), "w").write(open(ks).read())
    filt(one("stats/*/*kernel_trace.csv"), os.path.join(P, f"{rnd}_bench_kernel_trace_two_opt.csv"))
fetch = write = None
for name in ("fetch", "write"):
    cc = one(f"pmc_{name}/*/*counter_collection.csv")
    if cc:
        filt(cc, os.path.join(P, f"{rnd}_pmc_{name}_size_two_opt.csv"))
        c = counters(cc)
        if name == "fetch":
            fetch = c.get("FETCH_SIZE")
        else:
            write = c.get("WRITE_SIZE")
sq = {}
for name in ("sq1", "sq2"):
    cc = one(f"pmc_{name}/*/*counter_collection.csv")
    if cc:
        sq.update(counters(cc))
if sq:
    with open(os.path.join(P, f"{rnd}_pmc_sq_two_opt.csv"), "w") as fh:
        fh.write("counter,sum_over_dispatch,fraction_of_SQ_WAVE_CYCLES\n")
        wc = sq.get("SQ_WAVE_CYCLES", 0.0)
        for k in sorted(sq):
            fh.write(f"{k},{sq[k]:.0f},{(sq[k] / wc if wc else 0):.4f}\n")
bench = os.path.join(G, "bench.json")
line = full = None
if os.path.exists(bench):
    line = json.loads(open(bench).read().strip().splitlines()[-1])   # the compact result line (<= 4096 bytes, the LAST stdout line)
    side = os.path.join(G, "bench_extras.json")                       # the whole record bench.py wrote beside it
    full = json.load(open(side)) if os.path.exists(side) else line
if fetch is not None and write is not None:
    tpath = os.path.join(P, f"{rnd}_hbm_traffic.json")
    t = json.load(open(tpath)) if os.path.exists(tpath) else {}
    t.update({"fetch_size_kb_raw": fetch, "write_size_kb_raw": write, "traffic_bytes_per_launch": (2 * fetch + write) * 1024})
    if line:
        t["algorithmic_bytes_per_launch"] = int(line["candidates_per_step_per_gpu"] * 8)
    json.dump(t, open(tpath, "w"), indent=1)
    if line:
        line["roofline"]["traffic"] = t["traffic_bytes_per_launch"]
        full["roofline"]["traffic"] = t["traffic_bytes_per_launch"]
if sq and line and "candidates_touched" in full:
    # what bench.py's roofline reads: the VALU instruction count of ONE launch of the deterministic workload, with the
    # in-kernel work counters of the same workload as its fingerprint
    v = {"n": line["config"]["n"], "restarts": line["config"]["restarts_per_gpu"], "seed": line["config"]["restart_seed"],
         "work": {k: full["candidates_touched"][k] for k in ("l0_tile_bounds", "l1_candidates", "l2_candidates", "l3_candidates")},
         "source": "rocprofv3 --pmc SQ_* --kernel-trace (two passes) on `bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras`, "
                   "k_two_opt_ref_lds, summed over the dispatch (scripts/profile_round.sh, scripts/harvest_profiles.py)",
         "kernel_ms_bench": line["roofline"]["kernel_ms_avg"], "clock_mhz_bench": line["roofline"]["clock_mhz_live"]}
    v.update({k: sq[k] for k in sorted(sq)})
    tpath = os.path.join(P, f"{rnd}_hbm_traffic.json")
    if os.path.exists(tpath):
        v["hbm_traffic_bytes_per_launch"] = json.load(open(tpath)).get("traffic_bytes_per_launch")
    json.dump(v, open(os.path.join(P, f"{rnd}_valu_roofline.json"), "w"), indent=1)
if line:
    open(os.path.join(P, f"{rnd}_bench.json"), "w").write(json.dumps(line, separators=(",", ":")) + "\n")
    if full is not line:
        json.dump(full, open(os.path.join(P, f"{rnd}_bench_extras.json"), "w"), indent=1)
print("harvested", rnd, "fetch", fetch, "write", write, "sq counters", len(sq))
