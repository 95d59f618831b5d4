import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, _oracle as O, teeline_amd as TA
n = 10000
xy = O.synth_xy(n); prob = TA.TspProblem(np.arange(n), xy)
with TA.Context(0) as ctx:
    rc, nn, _ = O.nearest_neighbor(xy, None, n, 3)
    for name, init in (("greedy", nn), ("random", O.restart_perm(n, 12345, 0))):
        sol = TA.two_opt.solve(prob, None, None, [int(v) for v in init], ctx=ctx, mode=1)
        s = sol.stats
        print(f"BEST_SWEEP {name}: cost={float(sol.total):.5f} sweeps={s['sweeps']} moves={s['moves']} cand={s['candidates']:.3e} kernel_ms={s['kernel_ms']:.1f} total_ms={s['total_ms']:.1f} -> {s['candidates']/s['kernel_ms']/1e6:.1f} Gcand/s, {1e3*s['kernel_ms']/s['sweeps']:.1f} us/sweep")
