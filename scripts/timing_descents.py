"""Full best-improvement descents (nn -> 3-opt at n = 1002 and berlin52; nn -> Or-opt at n = 5000 and n = 1002): wall and kernel ms, moves."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import teeline_amd as TA
import _tsplib as T
b = T.parse_tsplib(os.path.join(ROOT, "tests", "golden", "tsplib", "berlin52.tsp"))
with TA.Context(0) as ctx:
    for name, fn, xy in (("3opt berlin52", TA.three_opt, b["xy"]), ("3opt n=1002", TA.three_opt, TA.synth.synth_xy(1002)),
                         ("oropt n=1002", TA.or_opt, TA.synth.synth_xy(1002)), ("oropt n=5000", TA.or_opt, TA.synth.synth_xy(5000))):
        p = TA.TspProblem(np.arange(len(xy)), xy)
        nn = TA.nearest_neighbor.solve(p, ctx=ctx).route()
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            s = fn.solve(p, None, None, nn, ctx=ctx)
            w = (time.perf_counter() - t0) * 1e3
            if best is None or w < best[0]:
                best = (w, s)
        w, s = best
        print(f"{name:14s}: wall {w:8.2f} ms  kernel {s.stats['kernel_ms']:8.2f} ms  passes {s.stats['sweeps']}  moves {s.stats['moves']}  cost {float(s.total):.5f}", flush=True)
