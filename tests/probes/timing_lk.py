"""Developer probe: LK timings on the GPU only (no oracle run)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, teeline_amd as TA
import _tsplib as T
with TA.Context(0) as ctx:
    b = T.parse_tsplib(os.path.join(ROOT, "tests/golden/tsplib/berlin52.tsp"))["xy"]
    for name, n in (("berlin52", 52), ("synth1000", 1000), ("synth13509", 13509)):
        xy = b if name == "berlin52" else TA.synth.synth_xy(n)
        kw = dict(epochs=10000, platoo_epochs=500, n_nearest=3) if name == "berlin52" else dict(epochs=20, platoo_epochs=10, n_nearest=5)
        s = TA.lin_kernighan.solve(TA.TspProblem(np.arange(len(xy)), xy), TA.LKOptions(TA.HeuristicOptions(**kw), 5), ctx=ctx, seed=1)
        print(f"LK {name}: total {s.stats['total_ms']:.1f} ms ({s.stats['moves']} moves, {s.stats['sweeps']} scans) cost {float(s.total):.5f}")
