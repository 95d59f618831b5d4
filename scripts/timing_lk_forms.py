"""LK at n = 13 509 / 20 epochs (the bench extra's run) with one workgroup per pair (the product) and with the persistent scan grid
(TL_FLAG_LK_SCAN_PERSIST, tuning build: TEELINE_GPU_LIB=teeline_amd/libteeline_gpu_tune.so): kernel time, rounds, microseconds per round; the tours must agree.  Also berlin52 with the CLI's options."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import teeline_amd as TA
n = 13509
p13 = TA.TspProblem(np.arange(n), TA.synth.synth_xy(n))
opts = TA.LKOptions(TA.HeuristicOptions(epochs=20, platoo_epochs=10, n_nearest=5), 5)
res = {}
for name, flag in (("persistent", TA.TL_FLAG_LK_SCAN_PERSIST), ("per_pair", 0)):
    with TA.Context(0, flag) as ctx:
        best = None
        for _ in range(3):
            s = TA.lin_kernighan.solve(p13, opts, ctx=ctx, seed=1)
            if best is None or s.stats["kernel_ms"] < best.stats["kernel_ms"]:
                best = s
        res[name] = best
        print(f"{name:10s} n={n}: kernel {best.stats['kernel_ms']:.2f} ms, {best.stats['sweeps']} rounds, {best.stats['kernel_ms'] * 1e3 / best.stats['sweeps']:.2f} us/round, cost {float(best.total):.5f}", flush=True)
assert list(res["persistent"].route()) == list(res["per_pair"].route())
for nn in (52, 1000, 3000):
    pp = TA.TspProblem(np.arange(nn), TA.synth.synth_xy(nn, ) if nn != 52 else TA.synth.synth_xy(52))
    o2 = TA.LKOptions(TA.HeuristicOptions(epochs=200, platoo_epochs=50, n_nearest=5), 5)
    for name, flag in (("persistent", TA.TL_FLAG_LK_SCAN_PERSIST), ("per_pair", 0)):
        with TA.Context(0, flag) as ctx:
            s = min((TA.lin_kernighan.solve(pp, o2, ctx=ctx, seed=1) for _ in range(2)), key=lambda z: z.stats["kernel_ms"])
            print(f"{name:10s} n={nn}: kernel {s.stats['kernel_ms']:.2f} ms, {s.stats['sweeps']} rounds, {s.stats['kernel_ms'] * 1e3 / max(s.stats['sweeps'], 1):.2f} us/round", flush=True)
