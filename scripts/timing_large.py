import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, teeline_amd as TA
with TA.Context(0) as ctx:
    for n in (20000, 50000):
        xy = TA.synth.synth_xy(n); prob = TA.TspProblem(np.arange(n), xy)
        nn = TA.nearest_neighbor.solve(prob, ctx=ctx)
        s = TA.two_opt.solve(prob, None, None, nn.route(), ctx=ctx)
        st = s.stats
        print(f"n={n} NN {float(nn.total):.2f} ({nn.stats['kernel_ms']:.0f} ms) -> 2opt {float(s.total):.2f}: sweeps {st['sweeps']} moves {st['moves']} cand {st['candidates']:.3e} kernel {st['kernel_ms']:.1f} ms -> {st['candidates']/st['kernel_ms']/1e6:.1f} Gcand/s")
