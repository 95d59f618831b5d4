"""Wall time of every solver stage on synthetic instances (one MI355X): where does a pipeline spend its time?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, teeline_amd as TA
for n in (1002, 5000):
    xy = TA.synth.synth_xy(n)
    p = TA.TspProblem(np.arange(n), xy)
    with TA.Context(0) as ctx:
        nn = TA.nearest_neighbor.solve(p, ctx=ctx)
        route = nn.route()
        for name, fn in (("nn", lambda: TA.nearest_neighbor.solve(p, ctx=ctx)),
                         ("2opt", lambda: TA.two_opt.solve(p, None, None, route, ctx=ctx)),
                         ("or_opt", lambda: TA.or_opt.solve(p, None, None, route, ctx=ctx)),
                         ("3opt", (lambda: TA.three_opt.solve(p, None, None, route, ctx=ctx)) if n <= 1002 else None),
                         ("lk", lambda: TA.lin_kernighan.solve(p, TA.LKOptions(TA.HeuristicOptions(epochs=20, platoo_epochs=10, n_nearest=5), 5), None, route, ctx=ctx, seed=1))):
            if fn is None:
                continue
            fn()
            t = time.perf_counter(); s = fn(); dt = (time.perf_counter() - t) * 1e3
            st = getattr(s, "stats", {}) or {}
            print(f"n={n:5d} {name:7s}: {dt:9.2f} ms  cost {float(s.total):.2f}  moves {st.get('moves')}  kernel {st.get('kernel_ms')}")
