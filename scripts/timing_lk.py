"""LK cut-over measurement (GPU box): the LDS-resident single-workgroup form (64 / 256 / 1024 threads) against the chip-wide
scans, on the reference's LK benchmark instances and synthetic sizes, CLI options (epochs 10000, platoo 500, n_nearest 3,
depth 5) for the fixtures and the library defaults for the synthetic ones.  Needs the tuning build:
    python -m teeline_amd.build --tune && TEELINE_GPU_LIB=teeline_amd/libteeline_gpu_tune.so python scripts/timing_lk.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, teeline_amd as TA
import _tsplib as T

def run(xy, opts, seed, env):
    for k in ("TL_LK_SMALL_MAX_N", "TL_LK_SMALL_NT"):
        os.environ.pop(k, None)
    os.environ.update(env)
    with TA.Context(0) as ctx:
        p = TA.TspProblem(np.arange(len(xy)), xy)
        best = 1e9
        for _ in range(2):
            t = time.perf_counter()
            sol = TA.lin_kernighan.solve(p, opts, None, None, ctx=ctx, seed=seed)
            best = min(best, (time.perf_counter() - t) * 1e3)
    return best, float(sol.total), sol.stats["sweeps"]

cases = []
for name in ("berlin52", "a280", "att532"):
    xy = T.parse_tsplib(os.path.join(ROOT, "tests", "golden", "tsplib", f"{name}.tsp"))["xy"]
    cases.append((name, xy, TA.LKOptions(TA.HeuristicOptions(epochs=10000, platoo_epochs=500, n_nearest=3), 5)))
for n in (100, 200, 400, 800, 1500):
    cases.append((f"synth{n}", TA.synth.synth_xy(n), TA.LKOptions()))
for name, xy, opts in cases:
    row = [f"{name:10s} n={len(xy):5d}"]
    ref = None
    for label, env in (("chip", {"TL_LK_SMALL_MAX_N": "0"}), ("lds64", {"TL_LK_SMALL_MAX_N": "100000", "TL_LK_SMALL_NT": "64"}),
                       ("lds256", {"TL_LK_SMALL_MAX_N": "100000", "TL_LK_SMALL_NT": "256"}),
                       ("lds1024", {"TL_LK_SMALL_MAX_N": "100000", "TL_LK_SMALL_NT": "1024"})):
        ms, cost, scans = run(xy, opts, 1, env)
        if ref is None:
            ref = (cost, scans)
        assert (cost, scans) == ref, (name, label, cost, scans, ref)
        row.append(f"{label} {ms:8.2f} ms")
    print("  ".join(row) + f"  cost {ref[0]:.5f} scans {ref[1]}", flush=True)
