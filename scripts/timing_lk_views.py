"""LK at n = 13 509 (20 epochs, seed 1) and n = 5000: the packed view (default) against the classic look-ups (TL_FLAG_LK_CLASSIC_VIEW); same tours asserted."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, teeline_amd as TA
for n in (13509, 5000, 2500):
    xy = TA.synth.synth_xy(n)
    p = TA.TspProblem(np.arange(n), xy)
    ref = None
    for name, fl in (("packed", 0), ("classic", TA.TL_FLAG_LK_CLASSIC_VIEW)):
        with TA.Context(0, fl) as ctx:
            best = 1e9
            for _ in range(3):
                s = TA.lin_kernighan.solve(p, TA.LKOptions(TA.HeuristicOptions(epochs=20, platoo_epochs=10, n_nearest=5), 5), ctx=ctx, seed=1)
                best = min(best, s.stats["kernel_ms"])
        key = (list(s.route()), s.stats["sweeps"], s.stats["moves"], s.stats["candidates"])
        if ref is None:
            ref = key
        print(f"n={n} {name:8s} kernel {best:8.2f} ms  rounds {s.stats['sweeps']}  {best * 1e3 / s.stats['sweeps']:.2f} us/round  cost {float(s.total):.5f}  same {key == ref}", flush=True)
        assert key == ref
