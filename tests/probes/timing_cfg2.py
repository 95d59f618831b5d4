import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, _oracle as O, teeline_amd as TA
n = 1002
xy = O.synth_xy(n)
with TA.Context(0) as ctx:
    dm = TA.distance_matrix.build(np.arange(n), xy, ctx=ctx)
    pm = TA.TspProblem(np.arange(n), xy, TA.distance_matrix.DistanceMatrix(n, dm.items, np.arange(n), "explicit"))
    pc = TA.TspProblem(np.arange(n), xy)
    rc, nn, _ = O.nearest_neighbor(xy, None, n, 3)
    for name, init in (("nn", nn), ("identity", None), ("random", O.restart_perm(n, 5, 0))):
        for label, p in (("matrix-in-HBM", pm), ("on-the-fly", pc)):
            for rep in range(2):
                s = TA.two_opt.solve(p, None, None, None if init is None else [int(v) for v in init], ctx=ctx)
            st = s.stats
            print(f"n=1002 {name:8s} {label:14s}: cost {float(s.total):.5f} sweeps {st['sweeps']} moves {st['moves']} cand {st['candidates']:.3e} kernel {st['kernel_ms']:.3f} ms -> {st['candidates']/st['kernel_ms']/1e6:.2f} Gcand/s")
        t = time.perf_counter(); O.two_opt(xy, None, n, init=init); print(f"   oracle: {(time.perf_counter()-t)*1e3:.1f} ms")
