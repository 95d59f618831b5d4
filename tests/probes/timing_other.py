"""Developer probe: 3-opt / LK / NN / k-NN timings (kernel ms) vs the oracle on the same host."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _oracle as O, _tsplib as T
import teeline_amd as TA

def P(xy): return TA.TspProblem(np.arange(len(xy)), xy)
with TA.Context(0) as ctx:
    b = T.parse_tsplib(os.path.join(ROOT, "tests/golden/tsplib/berlin52.tsp"))["xy"]
    rc, nn, _ = O.nearest_neighbor(b, None, 52, 3)
    for rep in range(2):
        s = TA.three_opt.solve(P(b), None, None, [int(v) for v in nn], ctx=ctx)
    t = time.perf_counter(); O.three_opt(b, None, 52, init=nn); tc = time.perf_counter() - t
    print(f"3opt berlin52 nn-start: gpu kernel {s.stats['kernel_ms']:.2f} ms total {s.stats['total_ms']:.2f} ms ({s.stats['moves']} moves) | oracle {tc*1e3:.1f} ms")
    for n in (300, 1002):
        xy = O.synth_xy(n); rc, nn, _ = O.nearest_neighbor(xy, None, n, 3)
        TA.three_opt.find_best_move(P(xy), nn, ctx=ctx)
        TA.three_opt.find_best_move(P(xy), nn, ctx=ctx); ms = ctx.last_kernel_ms()
        t = time.perf_counter(); O.three_opt_find_best_move(xy, None, nn); tc = time.perf_counter() - t
        tri = n*(n-1)*(n-2)//6 - (n-2)
        print(f"3opt scan n={n}: gpu {ms:.3f} ms = {tri/ms/1e6:.2f} Gtriples/s | oracle {tc*1e3:.0f} ms = {tri/tc/1e9:.3f} Gtriples/s")
    xy = O.synth_xy(1002); rc, nn, _ = O.nearest_neighbor(xy, None, 1002, 3)
    s = TA.three_opt.solve(P(xy), None, None, [int(v) for v in nn], ctx=ctx)
    print(f"3opt solve n=1002 nn-start: {s.stats['moves']} moves, {s.stats['candidates']:.3e} triples, total {s.stats['total_ms']:.0f} ms -> {s.stats['candidates']/s.stats['total_ms']/1e6:.2f} Gtriples/s cost {float(s.total):.3f}")
    for name, n in (("berlin52", 52), ("synth1000", 1000), ("synth13509", 13509)):
        xy = b if name == "berlin52" else O.synth_xy(n)
        kw = dict(epochs=10000, platoo_epochs=500, n_nearest=3) if name == "berlin52" else dict(epochs=20, platoo_epochs=10, n_nearest=5)
        h = TA.HeuristicOptions(**kw)
        s = TA.lin_kernighan.solve(P(xy), TA.LKOptions(h, 5), ctx=ctx, seed=1)
        t = time.perf_counter(); o = O.lin_kernighan(xy, seed=1, **kw); tc = time.perf_counter() - t
        same = list(s.route()) == o[1].tolist()
        print(f"LK {name}: gpu total {s.stats['total_ms']:.1f} ms ({s.stats['moves']} moves, {s.stats['sweeps']} scans) cost {float(s.total):.5f} | oracle {tc*1e3:.1f} ms | identical={same}")
    xy = O.synth_xy(13509)
    s = TA.nearest_neighbor.solve(P(xy), ctx=ctx); t = time.perf_counter(); O.nearest_neighbor(xy, None, 13509, 3); tc = time.perf_counter() - t
    print(f"NN n=13509: gpu {s.stats['kernel_ms']:.1f} ms | oracle {tc*1e3:.0f} ms")
    t = time.perf_counter(); TA.lin_kernighan.build_candidates(P(xy), 5, ctx=ctx); tg = time.perf_counter() - t
    t = time.perf_counter(); O.build_candidates(xy, 5); tc = time.perf_counter() - t
    print(f"kNN k=5 n=13509: gpu call {tg*1e3:.1f} ms | oracle brute force {tc*1e3:.0f} ms")
