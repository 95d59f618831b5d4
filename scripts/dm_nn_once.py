"""One matrix-form 2-opt descent from the NN tour at n = 1002, five times (for rocprofv3 --kernel-trace --stats: which kernels the call's kernel_ms holds)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, teeline_amd as TA
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1002
xy = TA.synth.synth_xy(n)
with TA.Context(0) as ctx:
    dm = TA.distance_matrix.build(np.arange(n), xy, ctx=ctx)
    pm = TA.TspProblem(np.arange(n), xy, TA.distance_matrix.DistanceMatrix(n, dm.items, np.arange(n), "explicit"))
    nn = [int(v) for v in TA.nearest_neighbor.solve(TA.TspProblem(np.arange(n), xy), ctx=ctx).route()]
    for _ in range(5):
        s = TA.two_opt.solve(pm, None, None, nn, ctx=ctx)
    print(s.stats["kernel_ms"], ctx.two_opt_last_counters()[4:9])
    opt = list(s.route())
    for _ in range(3):
        s2 = TA.two_opt.solve(pm, None, None, [int(v) for v in np.argsort(np.asarray(opt))] if False else None, ctx=ctx) if False else None
