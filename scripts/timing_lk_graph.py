"""LK wall times with and without the hipGraph replay of the round loop (TL_FLAG_LK_NO_GRAPH): berlin52 (CLI options), a280,
synthetic n = 1000 and n = 13 509."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, teeline_amd as TA
import _tsplib as T
for flag, label in ((0, "graph"), (TA.TL_FLAG_LK_NO_GRAPH, "launches")):
    with TA.Context(0, flag) as ctx:
        for name, n in (("berlin52", 52), ("a280", 280), ("synth1000", 1000), ("synth13509", 13509)):
            if name in ("berlin52", "a280"):
                xy = T.parse_tsplib(os.path.join(ROOT, f"tests/golden/tsplib/{name}.tsp"))["xy"]
            else:
                xy = TA.synth.synth_xy(n)
            kw = dict(epochs=10000, platoo_epochs=500, n_nearest=3) if name == "berlin52" else dict(epochs=20, platoo_epochs=10, n_nearest=5)
            best = None
            for _ in range(2):
                s = TA.lin_kernighan.solve(TA.TspProblem(np.arange(len(xy)), xy), TA.LKOptions(TA.HeuristicOptions(**kw), 5), ctx=ctx, seed=1)
                best = s if best is None or s.stats["total_ms"] < best.stats["total_ms"] else best
            s = best
            print(f"LK {label:8s} {name:10s}: total {s.stats['total_ms']:8.1f} ms kernel {s.stats['kernel_ms']:8.1f} ms ({s.stats['moves']} moves, {s.stats['sweeps']} scans) "
                  f"{s.stats['total_ms'] * 1e3 / max(s.stats['sweeps'], 1):.1f} us/scan cost {float(s.total):.5f}")
