"""TL_MODE_BEST_SWEEP at n = 10^4 from the NN seed and from a random start: kernel ms, sweeps, microseconds per sweep, cost."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import teeline_amd as TA
n = 10000
p = TA.TspProblem(np.arange(n), TA.synth.synth_xy(n))
with TA.Context(0) as ctx:
    nn = TA.nearest_neighbor.solve(p, ctx=ctx).route()
    cases = (("nn", nn), ("random", [int(v) for v in TA.synth.restart_perm(n, 12345, 0)]))
    for name, init in (cases if len(sys.argv) < 2 else [c_ for c_ in cases if c_[0] == sys.argv[1]]):
        s = min((TA.two_opt.solve(p, None, None, init, ctx=ctx, mode=TA.TL_MODE_BEST_SWEEP) for _ in range(3)), key=lambda z: z.stats["kernel_ms"])
        print(f"best-sweep {name:6s}: kernel {s.stats['kernel_ms']:.2f} ms, {s.stats['sweeps']} sweeps, {s.stats['kernel_ms'] * 1e3 / s.stats['sweeps']:.2f} us/sweep, cost {float(s.total):.5f}", flush=True)
