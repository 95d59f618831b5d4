"""Developer probe: Or-opt scan / solve timings (kernel ms)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _oracle as O
import teeline_amd as TA

def P(xy): return TA.TspProblem(np.arange(len(xy)), xy)
with TA.Context(0) as ctx:
    for n in (1002, 5000, 10000):
        xy = O.synth_xy(n); rc, nn, _ = O.nearest_neighbor(xy, None, n, 3)
        TA.or_opt.find_best_move(P(xy), nn, ctx=ctx)
        TA.or_opt.find_best_move(P(xy), nn, ctx=ctx); ms = ctx.last_kernel_ms()
        ev = 3 * n * n * 2
        print(f"or-opt scan n={n}: gpu {ms:.3f} ms = {ev/ms/1e6:.2f} G placements/s")
    for n in (1002, 5000):
        xy = O.synth_xy(n); rc, nn, _ = O.nearest_neighbor(xy, None, n, 3)
        t = time.perf_counter()
        s = TA.or_opt.solve(P(xy), None, None, [int(v) for v in nn], ctx=ctx)
        tw = time.perf_counter() - t
        print(f"or-opt solve n={n}: {s.stats['moves']} moves, kernel {s.stats['kernel_ms']:.1f} ms total {s.stats['total_ms']:.1f} ms wall {tw*1e3:.1f} cost {float(s.total):.3f}")
