"""LK at the config-5 size (n = 13 509, synthetic): wall / kernel time of tl_lk (20 epochs) and of tl_build_candidates (k = 5,
kd-tree walk vs brute force)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, teeline_amd as TA
n = int(os.environ.get("N", 13509))
xy = TA.synth.synth_xy(n)
p = TA.TspProblem(np.arange(n), xy)
for flag, name in ((0, "kd-tree walk (default)"), (TA.TL_FLAG_KNN_BRUTE, "brute force")):
    with TA.Context(0, flag) as ctx:
        best = 1e9
        for _ in range(4):
            t = time.perf_counter()
            TA.lin_kernighan.build_candidates(p, 5, ctx=ctx)
            best = min(best, (time.perf_counter() - t) * 1e3)
        print(f"build_candidates n={n} k=5, {name}: {best:.2f} ms per call (tree build + walk + download)")
with TA.Context(0) as ctx:
    for _ in range(2):
        t = time.perf_counter()
        s = TA.lin_kernighan.solve(p, TA.LKOptions(TA.HeuristicOptions(epochs=20, platoo_epochs=10, n_nearest=5), 5), ctx=ctx, seed=1)
        wall = (time.perf_counter() - t) * 1e3
    print(f"tl_lk n={n} 20 epochs: wall {wall:.1f} ms kernel {s.stats['kernel_ms']:.1f} ms  {s.stats} cost {float(s.total):.5f}")
