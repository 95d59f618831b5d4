"""Developer probe: 3-opt scan / solve kernel times (kernel times only)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, teeline_amd as TA
with TA.Context(0) as ctx:
    for n in (300, 1002, 3000):
        p = TA.TspProblem(np.arange(n), TA.synth.synth_xy(n))
        nn = [int(v) for v in TA.nearest_neighbor.solve(p, ctx=ctx).route()]
        best = 1e9
        for rep in range(4):
            TA.three_opt.find_best_move(p, nn, ctx=ctx)
            best = min(best, ctx.last_kernel_ms())
        tri = n * (n - 1) * (n - 2) // 6 - (n - 2)
        print(f"3opt scan n={n}: {best:.3f} ms = {tri / best / 1e6:.1f} Gtriples/s")
    p = TA.TspProblem(np.arange(1002), TA.synth.synth_xy(1002))
    nn = [int(v) for v in TA.nearest_neighbor.solve(p, ctx=ctx).route()]
    s = TA.three_opt.solve(p, None, None, nn, ctx=ctx)
    print(f"3opt solve n=1002 nn-start: {s.stats['moves']} moves total {s.stats['total_ms']:.1f} ms kernel {s.stats['kernel_ms']:.1f} ms cost {float(s.total):.3f}")
