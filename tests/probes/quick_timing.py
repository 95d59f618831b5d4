"""Developer timing probe (not a test, not the bench): kernel time of single descents and a batch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _oracle as O
import teeline_amd as TA

n = int(os.environ.get("N", 10000))
xy = O.synth_xy(n)
prob = TA.TspProblem(np.arange(n), xy)
flags = TA.TL_FLAG_NO_PRUNE if os.environ.get("NOPRUNE") else 0
with TA.Context(0, flags) as ctx:
    rc, nn, _ = O.nearest_neighbor(xy, None, n, 3)
    for name, init in (("greedy", nn), ("random", O.restart_perm(n, 12345, 0))):
        for rep in range(2):
            sol = TA.two_opt.solve(prob, None, None, [int(v) for v in init], ctx=ctx)
            s = sol.stats
            print(f"{name}: cost={float(sol.total):.5f} sweeps={s['sweeps']} moves={s['moves']} cand={s['candidates']:.3e} "
                  f"kernel_ms={s['kernel_ms']:.2f} -> {s['candidates']/s['kernel_ms']/1e6:.2f} Gcand/s  us/move={1e3*s['kernel_ms']/max(1,s['moves']):.2f}")
    for R in (1, 8, 64, 256, 512):
        sol, costs = TA.two_opt.multistart(prob, R, seed=12345, ctx=ctx, return_costs=True)
        s = sol.stats
        print(f"multistart R={R}: best={float(sol.total):.5f} cand={s['candidates']:.3e} kernel_ms={s['kernel_ms']:.2f} "
              f"-> {s['candidates']/s['kernel_ms']/1e6:.2f} Gcand/s moves={s['moves']}")
