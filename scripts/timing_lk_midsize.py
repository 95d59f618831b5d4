"""LK with the CLI's defaults (10 000 epochs, plateau 500, k = 3, depth 5) at synthetic n = 1002 / 2000 / 3000: the form the library picks (LDS-resident ILS with
speculative epochs where it fits) against the chip-wide scans (TL_FLAG_LK_CHIP_WIDE); same tours asserted."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, teeline_amd as TA
ns = [int(v) for v in sys.argv[1:]] or [1002, 2000, 3000]
opts = TA.LKOptions(TA.HeuristicOptions(epochs=10_000, platoo_epochs=500, n_nearest=3), 5)
for n in ns:
    xy = TA.synth.synth_xy(n)
    p = TA.TspProblem(np.arange(n), xy)
    res = {}
    for name, fl in (("default", 0), ("chip-wide", TA.TL_FLAG_LK_CHIP_WIDE)):
        with TA.Context(0, fl) as ctx:
            t0 = time.perf_counter()
            s = TA.lin_kernighan.solve(p, opts, ctx=ctx, seed=1)
            wall = (time.perf_counter() - t0) * 1e3
            res[name] = (list(s.route()), float(s.total))
            print(f"n={n} {name:9s}: kernel {s.stats['kernel_ms']:9.2f} ms wall {wall:9.2f} ms rounds {s.stats['sweeps']} moves {s.stats['moves']} cost {float(s.total):.5f}", flush=True)
    assert res["default"] == res["chip-wide"], "the forms disagree"
